"""Opcode histogram of the largest basic block (the tile loop) of a kernel in a hipcc -S dump (development aid).
usage: python tools/isa_hist.py file.s <mangled-name-fragment>"""
import collections
import re
import sys

txt = open(sys.argv[1]).read().split("\n")
frag = sys.argv[2]
start = next(i for i, ln in enumerate(txt) if ln.startswith("_Z") and frag in ln.split(":")[0] and ":" in ln)
end = next(i for i in range(start, len(txt)) if txt[i].startswith(".Lfunc_end"))
blocks, cur = [], []
for ln in txt[start + 1:end]:
    if re.match(r"^\.LBB[0-9_]+:", ln):
        blocks.append(cur)
        cur = []
    s = ln.strip()
    if s and not s.startswith(";") and not s.startswith("."):
        cur.append(s.split()[0])
blocks.append(cur)
blocks.sort(key=len, reverse=True)
print("function", txt[start][:70], "| blocks:", len(blocks), "| largest:", [len(b) for b in blocks[:4]])
ops = collections.Counter(blocks[0])
groups = collections.Counter()
for k, v in ops.items():
    g = ("mfma" if "mfma" in k else "accvgpr" if "accvgpr" in k else "vmem" if k.startswith(("global_", "buffer_", "scratch_"))
         else "lds" if k.startswith("ds_") else "salu" if k.startswith("s_") else "valu")
    groups[g] += v
print("groups:", dict(groups))
for k, v in ops.most_common(int(sys.argv[3]) if len(sys.argv) > 3 else 30):
    print(f"  {k:32s} {v}")
