"""Scan a hipcc -S dump for MFMAs whose vDst range overlaps SrcA / SrcB (or partially overlaps SrcC): illegal for multi-pass
MFMAs -- the hardware reads A / B in later passes.  usage: python tools/dev/mfma_overlap.py file.s"""
import re, sys
def rng(s):
    m = re.match(r"([va])\[(\d+):(\d+)\]", s)
    return (m.group(1), int(m.group(2)), int(m.group(3))) if m else None
cur = None
bad = {}
tot = {}
for ln in open(sys.argv[1]):
    m = re.match(r"^(_Z\w+):", ln)
    if m:
        cur = m.group(1)
    s = ln.strip()
    if s.startswith("v_mfma"):
        ops = [o.strip() for o in s.split(None, 1)[1].split(",")]
        d, a, b = rng(ops[0]), rng(ops[1]), rng(ops[2])
        c = rng(ops[3]) if len(ops) > 3 else None
        tot[cur] = tot.get(cur, 0) + 1
        ov = lambda p, q: p and q and p[0] == q[0] and not (p[2] < q[1] or q[2] < p[1])
        if ov(d, a) or ov(d, b) or (c and ov(d, c) and c != d):
            bad.setdefault(cur, []).append(s)
for k, v in tot.items():
    print(f"{k[:90]:90s} mfma {v:4d} overlapping {len(bad.get(k, []))}")
    for s in bad.get(k, [])[:3]:
        print("    ", s)
