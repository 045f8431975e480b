"""For each MFMA: distance (in instructions) to the next instruction that WRITES one of its SrcA / SrcB registers."""
import re, sys, collections
def regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]$", tok)
    if m: return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()
cur = None; lines = []
funcs = collections.OrderedDict()
for ln in open(sys.argv[1]):
    m = re.match(r"^(_Z\w+):", ln)
    if m: cur = m.group(1); funcs[cur] = []
    s = ln.strip()
    if cur and s and not s.startswith((";", ".")) and not s.endswith(":"):
        funcs[cur].append(s)
want = sys.argv[2] if len(sys.argv) > 2 else ""
for name, ins in funcs.items():
    if want not in name: continue
    hist = collections.Counter(); ex = []
    for i, s in enumerate(ins):
        if not s.startswith("v_mfma"): continue
        ops = [o.strip() for o in s.split(None, 1)[1].split(",")]
        src = regs(ops[1]) | regs(ops[2])
        for d in range(1, 9):
            if i + d >= len(ins): break
            t = ins[i + d]
            if t.startswith(("s_", "global_store", "ds_write", "buffer_store")): 
                if t.startswith("s_nop"): pass
                continue
            parts = t.split(None, 1)
            if len(parts) < 2: continue
            dst = parts[1].split(",")[0].strip()
            if regs(dst) & src:
                hist[d] += 1
                if d <= 2: ex.append((s, ins[i+1:i+d+1]))
                break
    print(name[:80], dict(sorted(hist.items())))
    for e in ex[:6]: print("   ", e[0], "->", e[1])
