"""ctypes binding of libenerf_hip.so (C ABI in include/enerf_hip.h).

The product path has no CPU fallback: if the library is missing or a call fails, this raises.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libenerf_hip.so")
if os.environ.get("ENERF_LIB_PATH"):          # development aid: an A/B build of the same sources (tools/dev)
    LIB_PATH = os.environ["ENERF_LIB_PATH"]

_c = ctypes
_vp, _u32, _f32, _int, _sz = _c.c_void_p, _c.c_uint32, _c.c_float, _c.c_int, _c.c_size_t

# name -> argtypes (restype is int for all of these)
SIGNATURES = {
    "enerf_near_far_from_aabb": [_vp, _vp, _vp, _u32, _f32, _vp, _vp, _vp],
    "enerf_polar_from_ray": [_vp, _vp, _f32, _u32, _vp, _vp],
    "enerf_morton3D": [_vp, _u32, _vp, _vp],
    "enerf_morton3D_invert": [_vp, _u32, _vp, _vp],
    "enerf_packbits": [_vp, _u32, _f32, _vp, _vp],
    "enerf_march_rays_train": [_vp, _vp, _vp, _f32, _f32, _u32, _u32, _u32, _u32, _u32, _vp, _vp, _vp, _vp, _vp,
                               _vp, _vp, _u32, _vp],
    "enerf_march_fuse_near_far": [_vp, _f32],
    "enerf_march_mirror_count": [_vp],
    "enerf_march_rays_train_ex": [_vp, _vp, _vp, _f32, _f32, _u32, _u32, _u32, _u32, _u32, _vp, _vp, _vp, _vp, _vp,
                                  _vp, _vp, _u32, _u32, _vp],
    "enerf_march_rays_train_count": [_vp, _vp, _vp, _f32, _f32, _u32, _u32, _u32, _u32, _vp, _vp, _vp, _vp, _u32, _u32,
                                     _vp],
    "enerf_march_rays_train_write": [_vp, _vp, _vp, _f32, _f32, _u32, _u32, _u32, _u32, _u32, _vp, _vp, _vp, _vp, _vp,
                                     _vp, _vp, _u32, _u32, _vp],
    "enerf_composite_rays_train_forward": [_vp, _vp, _vp, _vp, _u32, _u32, _vp, _vp, _vp, _vp],
    "enerf_composite_rays_train_forward_blend": [_vp, _vp, _vp, _vp, _u32, _u32, _vp, _vp, _vp, _vp, _u32, _f32, _vp,
                                                 _vp],
    "enerf_composite_rays_train_backward_mse": [_vp, _vp, _f32, _vp, _u32, _f32, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                                _u32, _u32, _vp, _vp, _vp, _vp],
    "enerf_composite_rays_train_fwd_bwd_mse": [_vp, _vp, _vp, _vp, _u32, _u32, _vp, _vp, _vp, _u32, _f32, _vp, _vp, _f32,
                                               _vp, _vp, _vp, _vp, _vp],
    "enerf_composite_rays_train_backward": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _u32, _u32, _vp, _vp, _vp],
    "enerf_composite_rays_frame": [_vp, _vp, _vp, _vp, _u32, _u32, _vp, _vp, _vp, _u32, _f32, _vp, _vp, _vp, _vp, _vp],
    "enerf_march_rays": [_u32, _u32, _vp, _vp, _vp, _vp, _f32, _f32, _u32, _u32, _u32, _vp, _vp, _vp, _vp, _vp,
                         _vp, _u32, _vp],
    "enerf_march_rays_ex": [_u32, _u32, _vp, _vp, _vp, _vp, _f32, _f32, _u32, _u32, _u32, _vp, _vp, _vp, _vp, _vp,
                            _vp, _u32, _u32, _vp],
    "enerf_composite_rays": [_u32, _u32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "enerf_compact_rays": [_u32, _vp, _vp, _vp, _vp, _vp, _vp],
    "enerf_grid_encode_forward": [_vp, _vp, _vp, _vp, _u32, _u32, _u32, _u32, _f32, _u32, _int, _vp, _u32, _int,
                                  _int, _f32, _f32, _vp],
    "enerf_grid_encode_backward": [_vp, _vp, _vp, _vp, _vp, _u32, _u32, _u32, _u32, _f32, _u32, _int, _vp, _vp,
                                   _u32, _int, _int, _f32, _f32, _vp],
    "enerf_grid_encode_forward_sweep": [_vp, _vp, _vp, _u32, _u32, _f32, ctypes.c_uint64, _u32, _u32, _f32, _u32, _u32, _int, _f32,
                                        _f32, _vp],
    "enerf_grid_encode_backward_ex": [_vp, _vp, _vp, _vp, _vp, _u32, _u32, _u32, _u32, _f32, _u32, _int, _vp, _vp,
                                   _u32, _int, _int, _f32, _f32, _u32, _u32, _vp],
    "enerf_grid_records_discard": [_vp],
    "enerf_mlp32_signal_next_reduce": [_int],
    "enerf_stream_wait_mlp32_signal": [_vp],
    "enerf_debug_workspace_ordering": [_int],
    "enerf_mlp32_valid_rows": [_vp],
    "enerf_mlp32_valid_rows_ex": [_vp, _u32, _u32],
    "enerf_grid_adam_from_records_ex": [_vp, _vp, _vp, _vp, _vp, _u32, _u32, _f32, _f32, _f32, _f32, _u32, _u32, _vp, _vp,
                                        _vp, _vp, _vp, _vp, _vp, _vp],
    "enerf_grid_adam_from_records": [_vp, _vp, _vp, _vp, _vp, _u32, _u32, _f32, _f32, _f32, _f32, _u32, _vp],
    "enerf_sh_encode_forward": [_vp, _vp, _u32, _u32, _u32, _int, _vp, _int, _vp],
    "enerf_sh_encode_backward": [_vp, _vp, _u32, _u32, _u32, _vp, _vp, _int, _vp],
    "enerf_ffmlp_forward": [_vp, _vp, _u32, _u32, _u32, _u32, _u32, _u32, _u32, _vp, _vp, _int, _vp],
    "enerf_ffmlp_inference": [_vp, _vp, _u32, _u32, _u32, _u32, _u32, _u32, _u32, _vp, _vp, _int, _vp],
    "enerf_ffmlp_backward": [_vp, _vp, _vp, _vp, _u32, _u32, _u32, _u32, _u32, _u32, _u32, _int, _vp, _vp, _vp,
                             _int, _vp],
    "enerf_ffnerf_inference": [_vp, _vp, _vp, _vp, _u32, _int, _vp, _vp, _vp],
    "enerf_mlp32_forward": [_vp, _vp, _u32, _u32, _u32, _u32, _u32, _u32, _vp, _vp, _u32, _u32, _vp, _vp],
    "enerf_mlp32_forward_p": [_vp, _vp, _u32, _u32, _u32, _u32, _u32, _u32, _u32, _u32, _vp, _vp, _u32, _u32, _vp, _vp,
                              _vp],
    "enerf_mlp32_backward_p": [_vp, _vp, _vp, _vp, _u32, _u32, _u32, _vp, _u32, _u32, _u32, _u32, _u32, _vp, _vp, _u32,
                               _u32, _vp, _u32, _vp, _vp, _u32, _vp],
    "enerf_mlp32_forward_sh": [_vp, _vp, _u32, _u32, _u32, _u32, _u32, _u32, _vp, _vp, _u32, _u32, _vp, _vp, _vp],
    "enerf_mlp32_backward": [_vp, _vp, _vp, _vp, _u32, _u32, _u32, _u32, _u32, _vp, _vp, _vp, _u32, _u32, _vp, _u32, _vp, _vp,
                             _u32, _vp],
    "enerf_sh_encode_forward_strided": [_vp, _vp, _u32, _u32, _u32, _vp],
    "enerf_debug_grid_level_mask": [_u32],
    "enerf_debug_mlp32_wgrad_blocks": [_u32],
    "enerf_debug_mlp32_grid_caps": [_u32, _u32],
    "enerf_debug_march_wave_max_rays": [_u32],
    "enerf_debug_march_bg_blocks": [_u32],
    "enerf_debug_march_clip": [_int],
    "enerf_march_rays_use_box": [_int],
    "enerf_grid_owner_range": [ctypes.c_uint64, ctypes.c_uint64, _f32],
    "enerf_ffmlp_recompute": [_int],
    "enerf_amp_begin": [_vp, _vp, _vp, _vp],
    "enerf_amp_end": [_f32, _f32, _int, _vp],
    "enerf_amp_cancel": [],
    "enerf_debug_march_thread_min_rays": [_u32],
    "enerf_event_loss_fwd_bwd": [_vp, _vp, _vp, _u32, _u32, _u32, _f32, _f32, _f32, _vp, _vp, _vp, _vp, _vp],
    "enerf_event_pair_rays": [_vp, _vp, _vp, _vp, _u32, _vp, _vp, _u32, _u32, _vp, _vp, _vp, _vp, _u32, _f32, _f32, _f32,
                              _f32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "enerf_occupied_box_update": [_vp, _u32, _u32, _f32, _vp],
    "enerf_debug_mlp32_fused_backward": [_int],
    "enerf_mlp32_precision": [_int],
    "enerf_mlp32_recompute": [_int],
    "enerf_debug_grid_bwd_binned": [_u32, _u32],
    "enerf_density_grid_cells": [_vp, _u32, _u32, _f32, _u32, _c.c_uint64, _vp, _vp, _vp],
    "enerf_mark_untrained_grid": [_vp, _u32, _u32, _f32, _f32, _f32, _f32, _u32, _u32, _f32, _vp, _vp],
    "enerf_density_grid_update": [_vp, _vp, _u32, _u32, _u32, _f32, _f32, _f32, _vp, _vp, _vp, _u32, _vp, _vp],
    "enerf_adam_step": [_vp, _vp, _vp, _vp, _sz, _f32, _f32, _f32, _f32, _u32, _int, _vp],
    "enerf_allocate_splitk": [_sz],
    "enerf_free_splitk": [],
    "enerf_mlp32_defer_reduce": [_int],
    "enerf_march_train_samples": [_c.POINTER(_c.c_uint64), _int, _vp],
    "enerf_prof_enable": [_int],
    "enerf_prof_reset": [],
    "enerf_prof_read": [_int, _c.POINTER(_c.c_double), _c.POINTER(_c.c_uint64)],
    "enerf_adam_step_multi": [_u32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f32, _f32, _f32, _int, _vp],
    "enerf_prof_enable_mask": [_u32],
    "enerf_prof_sample_every": [_u32],
    "enerf_prof_read_units": [_int, _c.POINTER(_c.c_double), _c.POINTER(_c.c_uint64)],
    "enerf_train_step_mse": [_vp],
    "enerf_train_step_events": [_vp],
    "enerf_debug_step_timing": [_int, _c.POINTER(_c.c_double)],
    "enerf_abi_version": [],
    "enerf_nerf_mlp_available": [],
    "enerf_amp_armed": [],
    "enerf_debug_nerf_mlp_fused": [_int],
    "enerf_debug_nerf_frags_copy": [_vp, _vp],
    "enerf_debug_carry_frags": [_int],
    "enerf_debug_carry_count": [_int],
    "enerf_debug_march_carry_blocks": [_u32],
    "enerf_debug_fold_reduce": [_int],
    "enerf_nerf_mlp_forward": [_vp, _vp, _vp, _vp, _u32, _u32, _u32, _vp, _vp, _u32, _vp],
    "enerf_nerf_mlp_backward": [_vp, _vp, _f32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _u32, _u32, _u32, _u32, _vp, _u32, _vp],
}

F32, F16, BF16 = 0, 1, 2

HEADER_PATH = os.path.join(_HERE, "..", "include", "enerf_hip.h")


def header_abi_version():
    """ENERF_ABI_VERSION as include/enerf_hip.h states it (the one place the number is written)."""
    import re
    with open(HEADER_PATH) as f:
        m = re.search(r"^#define\s+ENERF_ABI_VERSION\s+(\d+)", f.read(), re.M)
    if not m:
        raise RuntimeError(f"enerf_amd: no ENERF_ABI_VERSION in {HEADER_PATH}")
    return int(m.group(1))

_lib = None


def lib():
    """Load libenerf_hip.so; raises (never falls back) when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"enerf_amd: {LIB_PATH} not found. Build it with `python -m enerf_amd.build` "
                "(hipcc --offload-arch=gfx950); there is no CPU/PyTorch fallback for the hot path.")
        # torch bundles its own libamdhip64.so (soname libamdhip64.so.7, requested by torch as "libamdhip64.so").
        # It must be in the process before this library is, so that both resolve to ONE HIP runtime (one set of
        # streams / queues): loaded the other way round the process ends up with two runtimes.
        import torch  # noqa: F401
        l = ctypes.CDLL(LIB_PATH)
        for name, sig in SIGNATURES.items():
            fn = getattr(l, name)
            fn.argtypes = sig
            fn.restype = _int
        l.enerf_last_error.restype = _c.c_char_p
        l.enerf_last_error.argtypes = []
        l.enerf_workspace_generation.restype = _c.c_uint64
        l.enerf_workspace_generation.argtypes = []
        want, got = header_abi_version(), l.enerf_abi_version()
        if got != want:
            raise RuntimeError(f"enerf_amd: {LIB_PATH} answers ABI {got}, include/enerf_hip.h says {want}: "
                               "stale build, run `python -m enerf_amd.build --force`")
        if os.environ.get("ENERF_TRACE_CALLS"):
            l = _TracedLib(l, os.environ["ENERF_TRACE_CALLS"])
        _lib = l
    return _lib


class _TracedLib:
    """Debugging aid (ENERF_TRACE_CALLS=<file>): every entry point writes its name to the file before it runs and the
    device is drained after it, so that after a memory access fault (which aborts the process) the file names the call
    whose kernels faulted."""

    def __init__(self, inner, path):
        self._inner = inner
        self._fd = os.open(path, os.O_CREAT | os.O_WRONLY | os.O_TRUNC, 0o644)
        self._cache = {}

    def __getattr__(self, name):
        fn = self._cache.get(name)
        if fn is None:
            raw = getattr(self._inner, name)
            if not name.startswith("enerf_") or name in ("enerf_last_error", "enerf_workspace_generation"):
                return raw
            tag = (name + " " * 80)[:80].encode()

            def fn(*a, _raw=raw, _tag=tag):
                import torch
                os.pwrite(self._fd, _tag, 0)
                rc = _raw(*a)
                torch.cuda.synchronize()
                os.pwrite(self._fd, (b"done " + _tag)[:80], 0)
                return rc
            self._cache[name] = fn
        return fn


def check(rc, what):
    if rc != 0:
        msg = lib().enerf_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"{what} failed (code {rc}): {msg}")


_raw_stream = None


def stream_handle():
    """hipStream_t of torch's current stream on the current device (the raw-handle query is ~20x cheaper than
    building a torch.cuda.Stream object, and this runs before every kernel launch)."""
    global _raw_stream
    import torch
    if _raw_stream is None:
        _raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", False)
    if _raw_stream:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def dtype_code(t):
    import torch
    try:
        return {torch.float32: F32, torch.float16: F16, torch.bfloat16: BF16}[t.dtype]
    except KeyError:
        raise RuntimeError(f"unsupported dtype {t.dtype}")


# ---- tensor argument checks mirroring the reference's CHECK_* macros (gridencoder.cu:15-18 etc.) ----
def check_cuda(t, name):
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA tensor")


def check_contiguous(t, name):
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be a contiguous tensor")


def check_floating(t, name):
    if not t.dtype.is_floating_point:
        raise RuntimeError(f"{name} must be a floating tensor")


def check_int(t, name):
    import torch
    if t.dtype != torch.int32:
        raise RuntimeError(f"{name} must be an int tensor")


def ptr(t):
    return t.data_ptr()


class prof:
    """Thin Python face of the enerf_prof_* hooks (used by bench.py)."""

    KERNELS = {"grid_fwd": 0, "grid_bwd": 1, "march_train": 2, "composite_fwd": 3, "composite_bwd": 4, "sh_fwd": 5,
               "ffmlp_fwd": 6, "ffmlp_bwd": 7, "march_infer": 8, "composite_infer": 9, "table_adam": 10, "mlp_reduce": 11}

    @staticmethod
    def enable(on=True, only=None):
        """Time every kernel family (on=True), none (False), or just the named ones (only=("grid_fwd", ...)): each
        timed call puts two event records on the stream, which is not free at ~60 launches per millisecond."""
        if on and only is not None:
            mask = 0
            for name in only:
                mask |= 1 << prof.KERNELS[name]
            lib().enerf_prof_enable_mask(mask)
        else:
            lib().enerf_prof_enable(1 if on else 0)

    @staticmethod
    def reset():
        lib().enerf_prof_reset()

    @staticmethod
    def read(name):
        ms = ctypes.c_double(0)
        n = ctypes.c_uint64(0)
        check(lib().enerf_prof_read(prof.KERNELS[name], ctypes.byref(ms), ctypes.byref(n)), "prof_read")
        return ms.value, n.value

    @staticmethod
    def sample_every(n):
        """Time one eligible call in n per family (a timed launch costs ~10 us of queue time)."""
        lib().enerf_prof_sample_every(int(n))

    @staticmethod
    def read_units(name):
        """(work units of the TIMED calls -- grid_encode: points, mlp32: samples --, eligible calls seen)."""
        u = ctypes.c_double(0)
        n = ctypes.c_uint64(0)
        check(lib().enerf_prof_read_units(prof.KERNELS[name], ctypes.byref(u), ctypes.byref(n)), "prof_read_units")
        return u.value, n.value
