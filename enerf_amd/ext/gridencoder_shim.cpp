// CPython module `_gridencoder`: grid_encode_forward / grid_encode_backward with the prototypes of
// gridencoder/src/gridencoder.h:12-13 (bound in gridencoder/src/bindings.cpp:5-8), forwarding to libenerf_hip.so.
// Layout 0 = the reference's [L, B, C] output / gradient; the table is fp32 or fp16 (as under the reference's autocast
// path), `inputs` always fp32 (gridencoder.cu:437).
#include "shim_common.h"

void grid_encode_forward(const at::Tensor inputs, const at::Tensor embeddings, const at::Tensor offsets,
                         at::Tensor outputs, const uint32_t B, const uint32_t D, const uint32_t C, const uint32_t L,
                         const float S, const uint32_t H, const bool calc_grad_inputs, at::Tensor dy_dx,
                         const uint32_t gridtype) {
    shim::need_f32(inputs, "inputs");
    shim::need_floating(embeddings, "embeddings");
    shim::need_i32(offsets, "offsets");
    shim::need_floating(outputs, "outputs");
    shim::need_floating(dy_dx, "dy_dx");
    const int dtype = shim::abi_dtype(embeddings, "embeddings");
    TORCH_CHECK(outputs.scalar_type() == embeddings.scalar_type(), "outputs must have the embeddings' dtype");
    shim::Launch l(inputs);
    shim::ok(enerf_grid_encode_forward(inputs.data_ptr<float>(), embeddings.data_ptr(), offsets.data_ptr<int32_t>(),
                                       outputs.data_ptr(), B, D, C, L, S, H, calc_grad_inputs ? 1 : 0, dy_dx.data_ptr(),
                                       gridtype, dtype, /*out_layout=*/0, /*in_add=*/0.0f, /*in_mul=*/1.0f, l.stream),
             "grid_encode_forward");
}

void grid_encode_backward(const at::Tensor grad, const at::Tensor inputs, const at::Tensor embeddings,
                          const at::Tensor offsets, at::Tensor grad_embeddings, const uint32_t B, const uint32_t D,
                          const uint32_t C, const uint32_t L, const float S, const uint32_t H,
                          const bool calc_grad_inputs, const at::Tensor dy_dx, at::Tensor grad_inputs,
                          const uint32_t gridtype) {
    shim::need_floating(grad, "grad");
    shim::need_f32(inputs, "inputs");
    shim::need_floating(embeddings, "embeddings");
    shim::need_i32(offsets, "offsets");
    shim::need_floating(grad_embeddings, "grad_embeddings");
    shim::need_floating(dy_dx, "dy_dx");
    shim::need_floating(grad_inputs, "grad_inputs");
    const int dtype = shim::abi_dtype(embeddings, "embeddings");
    TORCH_CHECK(grad.scalar_type() == embeddings.scalar_type() &&
                    grad_embeddings.scalar_type() == embeddings.scalar_type(),
                "grad and grad_embeddings must have the embeddings' dtype");
    shim::Launch l(inputs);
    shim::ok(enerf_grid_encode_backward(grad.data_ptr(), inputs.data_ptr<float>(), embeddings.data_ptr(),
                                        offsets.data_ptr<int32_t>(), grad_embeddings.data_ptr(), B, D, C, L, S, H,
                                        calc_grad_inputs ? 1 : 0, dy_dx.data_ptr(), grad_inputs.data_ptr(), gridtype,
                                        dtype, /*grad_layout=*/0, 0.0f, 1.0f, l.stream),
             "grid_encode_backward");
}

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    m.def("grid_encode_forward", &grid_encode_forward, "grid_encode_forward (HIP, gfx950)");
    m.def("grid_encode_backward", &grid_encode_backward, "grid_encode_backward (HIP, gfx950)");
}
