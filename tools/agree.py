"""Does the live hipEvent timing in bench.py agree with rocprofv3?  Takes the kernel trace of a bench run made with
`--render-frames 0 --graph-leg-steps 0` (so the timed region is the tail of the process) and that run's JSON line:
averages the LAST n launches of the roofline kernel, n = the launch count bench.py reports for the timed region.

usage: python tools/agree.py kernel_trace.csv bench_stdout.log out.json"""
import csv
import json
import sys


def main():
    trace, log, out = sys.argv[1:4]
    line = [l for l in open(log) if l.startswith('{"metric"')][-1]
    d = json.loads(line)
    res = {"bench_ms_per_step": d["ms_per_step"]}
    for fam, needle in (("grid_fwd", "k_grid_fwd<"), ("grid_bwd", None)):
        if fam not in d.get("kernels", {}):
            continue
        n = d["kernels"][fam]["launches"]
        # bench.py times launch k of its timed region when k % every == 0: the same launches are picked here
        region = d["kernels"][fam].get("launches_in_region", n)
        every = d["kernels"][fam].get("timed_one_in", 1)
        rows = [r for r in csv.DictReader(open(trace))]
        rows.sort(key=lambda r: int(r["Start_Timestamp"]))
        if needle:
            durs = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows if needle in r["Kernel_Name"]]
            tail = durs[-region:][::every]
            res[fam] = {"launches": n, "launches_in_region": region, "timed_one_in": every,
                        "hipEvent_avg_ms": d["kernels"][fam]["avg_ms"],
                        "rocprofv3_avg_ms_same_launches": sum(tail) / len(tail) / 1e6,
                        "rocprofv3_avg_ms_all_region_launches": sum(durs[-region:]) / max(len(durs[-region:]), 1) / 1e6}
        else:
            # the backward is a family of kernels per call: atomic pass + bin + tile
            per = {}
            for key in ("k_grid_bwd<", "k_grid_bwd_bin<", "k_grid_bwd_tile<"):
                durs = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows if key in r["Kernel_Name"]]
                per[key] = sum(durs[-n:]) / max(len(durs[-n:]), 1) / 1e6
            res[fam] = {"launches": n, "hipEvent_avg_ms": d["kernels"][fam]["avg_ms"],
                        "rocprofv3_sum_of_kernel_avgs_ms": sum(per.values()), "kernels": per}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
