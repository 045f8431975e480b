"""Top-level `_gridencoder` module for the reference's untouched wrappers: put this directory (and the repo root) on
sys.path and `import _gridencoder as _backend` binds the MI355X HIP implementation.  See INTEGRATION.md."""
from enerf_amd.backends._gridencoder import *  # noqa: F401,F403
