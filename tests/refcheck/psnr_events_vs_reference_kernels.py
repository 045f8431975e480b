"""tools/psnr_ab_events.py with arm B on the reference's OWN ray-marching / compositing / SH / grid-encoder kernels
(oracle/_ref, built by oracle/build_ref.py): with its nn.Linear nets on torch's GEMMs and torch.optim.Adam, arm B is then the
reference's native code end to end under the event loss.  TEST INFRASTRUCTURE: it lives under tests/ because it runs the
checker's kernels.
    python -B tests/refcheck/psnr_events_vs_reference_kernels.py [steps] [seeds] [out.json] [first_seed] [arms]"""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import build_ref  # noqa: E402

backends = (build_ref.load("raymarching"), build_ref.load("shencoder"))
if os.environ.get("ENERF_PSNR_REF_GRID", "1") == "1":
    backends += (build_ref.load("gridencoder"),)
runpy.run_path(os.path.join(ROOT, "tools", "psnr_ab_events.py"), init_globals={"ROUTE_B_BACKENDS": backends},
               run_name="__main__")
