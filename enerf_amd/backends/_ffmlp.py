"""`_ffmlp`: ffmlp_forward / ffmlp_inference / ffmlp_backward / allocate_splitk / free_splitk
(ffmlp/src/bindings.cpp:5-11).  The reference requires fp16 tensors (CHECK_IS_HALF, ffmlp.cu:636-642); this
backend accepts fp16 or bf16 (all tensors of one call must share the dtype)."""
from .. import _lib as L


def _chk16(t, name, dtype=None):
    import torch
    L.check_cuda(t, name)
    L.check_contiguous(t, name)
    if t.dtype not in (torch.float16, torch.bfloat16):
        raise RuntimeError(f"{name} must be a Half (or BFloat16) tensor")
    if dtype is not None and t.dtype != dtype:
        raise RuntimeError(f"{name} must have dtype {dtype}")
    return t.data_ptr()


def ffmlp_forward(inputs, weights, B, input_dim, output_dim, hidden_dim, num_layers, activation, output_activation,
                  forward_buffer, outputs):
    d = inputs.dtype
    L.check(L.lib().enerf_ffmlp_forward(_chk16(inputs, "inputs"), _chk16(weights, "weights", d), int(B),
                                        int(input_dim), int(output_dim), int(hidden_dim), int(num_layers),
                                        int(activation), int(output_activation),
                                        _chk16(forward_buffer, "forward_buffer", d), _chk16(outputs, "outputs", d),
                                        L.dtype_code(inputs), L.stream_handle()), "ffmlp_forward")


def ffmlp_inference(inputs, weights, B, input_dim, output_dim, hidden_dim, num_layers, activation, output_activation,
                    inference_buffer, outputs):
    d = inputs.dtype
    L.check(L.lib().enerf_ffmlp_inference(_chk16(inputs, "inputs"), _chk16(weights, "weights", d), int(B),
                                          int(input_dim), int(output_dim), int(hidden_dim), int(num_layers),
                                          int(activation), int(output_activation),
                                          _chk16(inference_buffer, "inference_buffer", d),
                                          _chk16(outputs, "outputs", d), L.dtype_code(inputs), L.stream_handle()),
            "ffmlp_inference")


def ffmlp_backward(grad, inputs, weights, forward_buffer, B, input_dim, output_dim, hidden_dim, num_layers, activation,
                   output_activation, calc_grad_inputs, backward_buffer, grad_inputs, grad_weights):
    d = inputs.dtype
    L.check(L.lib().enerf_ffmlp_backward(_chk16(grad, "grad", d), _chk16(inputs, "inputs"),
                                         _chk16(weights, "weights", d), _chk16(forward_buffer, "forward_buffer", d),
                                         int(B), int(input_dim), int(output_dim), int(hidden_dim), int(num_layers),
                                         int(activation), int(output_activation), int(bool(calc_grad_inputs)),
                                         _chk16(backward_buffer, "backward_buffer", d),
                                         _chk16(grad_inputs, "grad_inputs", d),
                                         _chk16(grad_weights, "grad_weights", d), L.dtype_code(inputs),
                                         L.stream_handle()), "ffmlp_backward")


def allocate_splitk(size):
    L.check(L.lib().enerf_allocate_splitk(int(size)), "allocate_splitk")


def free_splitk():
    L.check(L.lib().enerf_free_splitk(), "free_splitk")
