// CPython module `_shencoder`: sh_encode_forward / sh_encode_backward with the prototypes of
// shencoder/src/shencoder.h:9,12 (bound in shencoder/src/bindings.cpp:5-8), forwarding to libenerf_hip.so.
#include "shim_common.h"

void sh_encode_forward(at::Tensor inputs, at::Tensor outputs, const uint32_t B, const uint32_t D, const uint32_t C,
                       const bool calc_grad_inputs, at::Tensor dy_dx) {
    shim::need_floating(inputs, "inputs");
    shim::need_floating(outputs, "outputs");
    shim::need_floating(dy_dx, "dy_dx");
    const int dtype = shim::abi_dtype(inputs, "inputs");
    TORCH_CHECK(outputs.scalar_type() == inputs.scalar_type(), "outputs must have the inputs' dtype");
    shim::Launch l(inputs);
    shim::ok(enerf_sh_encode_forward(inputs.data_ptr(), outputs.data_ptr(), B, D, C, calc_grad_inputs ? 1 : 0,
                                     dy_dx.data_ptr(), dtype, l.stream), "sh_encode_forward");
}

void sh_encode_backward(at::Tensor grad, at::Tensor inputs, const uint32_t B, const uint32_t D, const uint32_t C,
                        at::Tensor dy_dx, at::Tensor grad_inputs) {
    shim::need_floating(grad, "grad");
    shim::need_floating(inputs, "inputs");
    shim::need_floating(dy_dx, "dy_dx");
    shim::need_floating(grad_inputs, "grad_inputs");
    const int dtype = shim::abi_dtype(inputs, "inputs");
    shim::Launch l(inputs);
    shim::ok(enerf_sh_encode_backward(grad.data_ptr(), inputs.data_ptr(), B, D, C, dy_dx.data_ptr(),
                                      grad_inputs.data_ptr(), dtype, l.stream), "sh_encode_backward");
}

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    m.def("sh_encode_forward", &sh_encode_forward, "SH encode forward (HIP, gfx950)");
    m.def("sh_encode_backward", &sh_encode_backward, "SH encode backward (HIP, gfx950)");
}
