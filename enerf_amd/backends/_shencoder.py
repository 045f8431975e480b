"""`_shencoder`: sh_encode_forward / sh_encode_backward (shencoder/src/bindings.cpp:5-8)."""
from .. import _lib as L


def _chk(t, name):
    L.check_cuda(t, name)
    L.check_contiguous(t, name)
    L.check_floating(t, name)
    return t.data_ptr()


def sh_encode_forward(inputs, outputs, B, D, C, calc_grad_inputs, dy_dx):
    L.check(L.lib().enerf_sh_encode_forward(_chk(inputs, "inputs"), _chk(outputs, "outputs"), int(B), int(D), int(C),
                                            int(bool(calc_grad_inputs)), _chk(dy_dx, "dy_dx"), L.dtype_code(inputs),
                                            L.stream_handle()), "sh_encode_forward")


def sh_encode_backward(grad, inputs, B, D, C, dy_dx, grad_inputs):
    L.check(L.lib().enerf_sh_encode_backward(_chk(grad, "grad"), _chk(inputs, "inputs"), int(B), int(D), int(C),
                                             _chk(dy_dx, "dy_dx"), _chk(grad_inputs, "grad_inputs"),
                                             L.dtype_code(grad), L.stream_handle()), "sh_encode_backward")
