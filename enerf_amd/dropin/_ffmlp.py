"""Top-level `_ffmlp` module for the reference's untouched wrappers: put this directory (and the repo root) on
sys.path and `import _ffmlp as _backend` binds the MI355X HIP implementation.  See INTEGRATION.md."""
from enerf_amd.backends._ffmlp import *  # noqa: F401,F403
