"""enerf_amd -- MI355X (gfx950) implementation of the instant-ngp hot path under knelk/enerf.

Layout
  csrc/        hand-written HIP kernels + the C ABI (include/enerf_hip.h) -> lib/libenerf_hip.so
  backends/    ctypes shims with the reference's pybind module/function names (_raymarching, _gridencoder, ...)
  ext/         pybind11 modules of the same four names over the C ABI, importable as top-level `_raymarching` ... by the
               reference's untouched Python wrappers
  raymarching, gridencoder, shencoder, ffmlp      host-side mirror of the reference's autograd wrappers
  encoding, activation, renderer, network, network_ff, events   the callers re-stated for the bench harness
  optim, parallel                                  fused optimizer + ray-sharded data parallel step

There is no CPU fallback in this package: every hot-path call goes to libenerf_hip.so or raises.
"""
__version__ = "0.1.0"
