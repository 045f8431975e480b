"""Drop-in backend modules carrying the reference's native-extension names.

Putting this directory on sys.path makes `import _raymarching`, `import _gridencoder`, `import _shencoder` and
`import _ffmlp` resolve to the HIP implementations, which is all the reference's untouched wrappers need
(`try: import _X as _backend`, raymarching/raymarching.py:9-12 and siblings).  See INTEGRATION.md.
"""
from . import _raymarching, _gridencoder, _shencoder, _ffmlp  # noqa: F401
