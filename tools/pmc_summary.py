"""Condense rocprofv3 counter_collection CSVs (one --pmc pass each) into a small per-kernel JSON for profiles/.

usage: python tools/pmc_summary.py out.json [--by-grid] [--meta key=value ...] pass1.csv [pass2.csv ...]
Every kernel whose name contains `k_` (this library's kernels) is kept; counters are averaged over its dispatches.
With --by-grid the key also carries the launch's Grid_Size (work-items), which separates workloads of one kernel.
"""
import csv
import json
import re
import sys
from collections import defaultdict


def short(name):
    m = re.search(r"(k_[A-Za-z0-9_]+(<[^>]*>)?)", name)
    return m.group(1) if m else None


def main():
    args = sys.argv[1:]
    out = args.pop(0)
    by_grid = "--by-grid" in args
    meta = {}
    files = []
    it = iter([a for a in args if a != "--by-grid"])
    for a in it:
        if a == "--meta":
            k, v = next(it).split("=", 1)
            meta[k] = v
        else:
            files.append(a)
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for f in files:
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                k = short(row["Kernel_Name"])
                if k is None:
                    continue
                if by_grid:
                    k = f"{k} grid={row['Grid_Size']}"
                a = acc[k][row["Counter_Name"]]
                a[0] += float(row["Counter_Value"])
                a[1] += 1
    res = {"_meta": meta} if meta else {}
    for k, cs in acc.items():
        res[k] = {c + "_avg": v[0] / v[1] for c, v in cs.items()}
        res[k]["dispatches"] = max(v[1] for v in cs.values())
    with open(out, "w") as fh:
        json.dump(res, fh, indent=1)
    print(f"{out}: {len(res)} kernels")


if __name__ == "__main__":
    main()
