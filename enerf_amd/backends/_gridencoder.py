"""`_gridencoder`: grid_encode_forward / grid_encode_backward with the reference's pybind signature
(gridencoder/src/bindings.cpp:5-8), plus keyword-only `layout` (0 = [L,B,C] as the reference, 1 = [B,L*C],
2 = [L,Bp,C] with Bp = B rounded up to 32) and `affine=(add, mul)`: the kernels read inputs as (x + add) * mul."""
from .. import _lib as L

# points processed per entry point since the last reset (bench.py: algorithmic bytes = 1164 B/point)
# (budget_rows: rows of the whole-step entry points' launches, which carry a sample BUDGET of which the kernels encode and bin
#  only the rows the marcher filled -- csrc/common.h grid_valid_rows; bench.py turns rows into real points with the device-side
#  sample total)
STATS = {"fwd_points": 0, "fwd_calls": 0, "bwd_points": 0, "bwd_calls": 0, "budget_rows": 0}
# the same forward counts from process start, never reset: the denominator of per-dispatch counter averages taken over a
# whole profiled process (tools/profile_round.sh)
LIFETIME = {"fwd_points": 0, "fwd_calls": 0}


def _chk(t, name, floating=True):
    L.check_cuda(t, name)
    L.check_contiguous(t, name)
    if floating:
        L.check_floating(t, name)
    else:
        L.check_int(t, name)
    return t.data_ptr()


def grid_encode_forward(inputs, embeddings, offsets, outputs, B, D, C, L_, S, H, calc_grad_inputs, dy_dx, gridtype,
                        *, layout=0, affine=(0.0, 1.0)):
    import torch
    if inputs.dtype != torch.float32:
        raise RuntimeError("inputs must be a float32 tensor (gridencoder.cu:437 reads inputs as float*)")
    STATS["fwd_points"] += int(B)
    STATS["fwd_calls"] += 1
    LIFETIME["fwd_points"] += int(B)
    LIFETIME["fwd_calls"] += 1
    dt = L.dtype_code(embeddings)
    if outputs.dtype != embeddings.dtype or dy_dx.dtype != embeddings.dtype:
        raise RuntimeError("outputs/dy_dx must have the dtype of embeddings")
    L.check(L.lib().enerf_grid_encode_forward(_chk(inputs, "inputs"), _chk(embeddings, "embeddings"),
                                              _chk(offsets, "offsets", False), _chk(outputs, "outputs"), int(B),
                                              int(D), int(C), int(L_), float(S), int(H), int(bool(calc_grad_inputs)),
                                              _chk(dy_dx, "dy_dx"), int(gridtype), dt, int(layout), affine[0],
                                              affine[1], L.stream_handle()), "grid_encode_forward")


def grid_encode_backward(grad, inputs, embeddings, offsets, grad_embeddings, B, D, C, L_, S, H, calc_grad_inputs,
                         dy_dx, grad_inputs, gridtype, *, layout=0, affine=(0.0, 1.0), defer=False, reserve=0):
    import torch
    if inputs.dtype != torch.float32:
        raise RuntimeError("inputs must be a float32 tensor")
    STATS["bwd_points"] += int(B)
    STATS["bwd_calls"] += 1
    dt = L.dtype_code(grad)
    for t, n in ((embeddings, "embeddings"), (grad_embeddings, "grad_embeddings"), (dy_dx, "dy_dx"),
                 (grad_inputs, "grad_inputs")):
        if t.dtype != grad.dtype:
            raise RuntimeError(f"{n} must have the dtype of grad")
    # defer: leave the binned levels' record lists to enerf_grid_adam_from_records (include/enerf_hip.h)
    L.check(L.lib().enerf_grid_encode_backward_ex(_chk(grad, "grad"), _chk(inputs, "inputs"),
                                                  _chk(embeddings, "embeddings"), _chk(offsets, "offsets", False),
                                                  _chk(grad_embeddings, "grad_embeddings"), int(B), int(D), int(C),
                                                  int(L_), float(S), int(H), int(bool(calc_grad_inputs)),
                                                  _chk(dy_dx, "dy_dx"), _chk(grad_inputs, "grad_inputs"),
                                                  int(gridtype), dt, int(layout), affine[0], affine[1],
                                                  1 if defer else 0, int(reserve), L.stream_handle()),
            "grid_encode_backward")


def grid_encode_forward_sweep(embeddings, offsets, outputs, n_cascades, grid_size, bound, seed, C, L_, S, H, gridtype,
                              layout, affine):
    """grid_encode_forward over a full density-grid sweep's query points, generated inside the kernel
    (include/enerf_hip.h: enerf_grid_encode_forward_sweep).  fp32 tables, D = 3."""
    import ctypes
    B = int(n_cascades) * int(grid_size) ** 3
    STATS["fwd_points"] += B
    STATS["fwd_calls"] += 1
    LIFETIME["fwd_points"] += B
    LIFETIME["fwd_calls"] += 1
    L.check(L.lib().enerf_grid_encode_forward_sweep(
        embeddings.data_ptr(), offsets.data_ptr(), outputs.data_ptr(), int(n_cascades), int(grid_size), float(bound),
        ctypes.c_uint64(int(seed)), int(C), int(L_), float(S), int(H), int(gridtype), int(layout), float(affine[0]),
        float(affine[1]), L.stream_handle()), "grid_encode_forward_sweep")
