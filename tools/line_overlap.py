"""Share of a file's code lines that also occur (whitespace / quote-style normalised) in a reference file.
    python tools/line_overlap.py enerf_amd/renderer.py /root/reference/nerf/renderer.py"""
import re
import sys


def lines(path):
    out = []
    for ln in open(path, errors="replace"):
        ln = ln.split("#")[0] if not ln.lstrip().startswith("#") else ""
        ln = re.sub(r"\s+", "", ln).replace('"', "'")
        if len(ln) > 3 and not ln.startswith(("'''", "import", "from")):
            out.append(ln)
    return out


a, b = lines(sys.argv[1]), set(lines(sys.argv[2]))
same = sum(1 for ln in a if ln in b)
print(f"{sys.argv[1]}: {same} / {len(a)} code lines also in {sys.argv[2]} ({100.0 * same / max(len(a), 1):.0f} %)")
