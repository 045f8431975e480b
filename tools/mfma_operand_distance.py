"""For every MFMA of a kernel: how many issue slots lie between the last instruction that WRITES one of its SrcA / SrcB
registers and the MFMA itself (s_nop N counts N + 1; the scan stops at branch targets).  The hazard of csrc/mlp32s_ops.h
(operand_ready) sits at distance <= 2 behind a v_cvt_pk_* / v_pk_add_f32: development aid and a CPU test's helper.

    hipcc -S --cuda-device-only ... file.hip -o file.s ; python tools/mfma_operand_distance.py file.s [kernel-name-fragment]
"""
import collections
import re
import sys


def regs(tok):
    tok = tok.strip()
    m = re.match(r"v\[(\d+):(\d+)\]$", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def functions(path):
    cur, out = None, collections.OrderedDict()
    for ln in open(path):
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            cur = m.group(1)
            out[cur] = []
            continue
        if ln.startswith(".Lfunc_end"):
            cur = None
        s = ln.strip()
        if cur and s and not s.startswith((";", ".")):
            out[cur].append(s)
    return out


def scan(ins):
    """-> list of (distance, writer mnemonic, mfma text) for MFMAs whose operands are written within 12 slots."""
    found = []
    for i, s in enumerate(ins):
        if not s.startswith("v_mfma"):
            continue
        ops = [o.strip() for o in s.split(None, 1)[1].split(",")]
        src = regs(ops[1]) | regs(ops[2])
        dist = 0
        for k in range(i - 1, max(i - 40, -1), -1):
            t = ins[k]
            if t.endswith(":"):                       # a label: another path joins here
                break
            if t.startswith("s_nop"):
                dist += int(t.split()[1]) + 1
                continue
            dist += 1
            if t.startswith(("s_", ";;")):
                continue
            parts = t.split(None, 1)
            if len(parts) < 2 or parts[0].startswith(("global_store", "ds_write", "buffer_store", "v_cmp")):
                continue
            if regs(parts[1].split(",")[0]) & src:
                found.append((dist - 1, parts[0], s))
                break
            if dist > 12:
                break
    return found


if __name__ == "__main__":
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    for name, ins in functions(sys.argv[1]).items():
        if want not in name:
            continue
        f = scan(ins)
        hist = collections.Counter((d, w) for d, w, _ in f if not w.startswith(("ds_read", "global_load", "v_mfma")))
        close = sorted(hist.items())[:6]
        print(name[:90], "| MFMAs:", sum(1 for s in ins if s.startswith("v_mfma")), "| nearest VALU writers (slots between, op): count", close)
