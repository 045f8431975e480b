// CPython module `_ffmlp`: the five functions of ffmlp/src/bindings.cpp:5-11 with the prototypes of
// ffmlp/src/ffmlp.h:8-14, forwarding to libenerf_hip.so.  The reference's kernels are half-only; here the storage type
// is whatever 16-bit type the tensors carry (half as the reference, or bfloat16), accumulation is fp32 on the MFMA units.
#include "shim_common.h"

static int mlp_dtype(const at::Tensor& inputs, const at::Tensor& weights) {
    shim::need_floating(inputs, "inputs");
    shim::need_floating(weights, "weights");
    const int dtype = shim::abi_dtype(inputs, "inputs", /*allow_bf16=*/true);
    TORCH_CHECK(dtype != ENERF_F32, "inputs must be a 16-bit tensor (half or bfloat16)");
    TORCH_CHECK(weights.scalar_type() == inputs.scalar_type(), "weights must have the inputs' dtype");
    return dtype;
}

void ffmlp_forward(const at::Tensor inputs, const at::Tensor weights, const uint32_t B, const uint32_t input_dim,
                   const uint32_t output_dim, const uint32_t hidden_dim, const uint32_t num_layers,
                   const uint32_t activation_, const uint32_t output_activation_, at::Tensor forward_buffer,
                   at::Tensor outputs) {
    const int dtype = mlp_dtype(inputs, weights);
    shim::need_floating(forward_buffer, "forward_buffer");
    shim::need_floating(outputs, "outputs");
    shim::Launch l(inputs);
    shim::ok(enerf_ffmlp_forward(inputs.data_ptr(), weights.data_ptr(), B, input_dim, output_dim, hidden_dim, num_layers,
                                 activation_, output_activation_, forward_buffer.data_ptr(), outputs.data_ptr(), dtype,
                                 l.stream), "ffmlp_forward");
}

void ffmlp_inference(const at::Tensor inputs, const at::Tensor weights, const uint32_t B, const uint32_t input_dim,
                     const uint32_t output_dim, const uint32_t hidden_dim, const uint32_t num_layers,
                     const uint32_t activation_, const uint32_t output_activation_, at::Tensor inference_buffer,
                     at::Tensor outputs) {
    const int dtype = mlp_dtype(inputs, weights);
    shim::need_floating(inference_buffer, "inference_buffer");
    shim::need_floating(outputs, "outputs");
    shim::Launch l(inputs);
    shim::ok(enerf_ffmlp_inference(inputs.data_ptr(), weights.data_ptr(), B, input_dim, output_dim, hidden_dim,
                                   num_layers, activation_, output_activation_, inference_buffer.data_ptr(),
                                   outputs.data_ptr(), dtype, l.stream), "ffmlp_inference");
}

void ffmlp_backward(const at::Tensor grad, const at::Tensor inputs, const at::Tensor weights,
                    const at::Tensor forward_buffer, const uint32_t B, const uint32_t input_dim,
                    const uint32_t output_dim, const uint32_t hidden_dim, const uint32_t num_layers,
                    const uint32_t activation, const uint32_t output_activation, const bool calc_grad_inputs,
                    at::Tensor backward_buffer, at::Tensor grad_inputs, at::Tensor grad_weights) {
    const int dtype = mlp_dtype(inputs, weights);
    shim::need_floating(grad, "grad");
    shim::need_floating(forward_buffer, "forward_buffer");
    shim::need_floating(backward_buffer, "backward_buffer");
    shim::need_floating(grad_inputs, "grad_inputs");
    shim::need_floating(grad_weights, "grad_weights");
    shim::Launch l(inputs);
    shim::ok(enerf_ffmlp_backward(grad.data_ptr(), inputs.data_ptr(), weights.data_ptr(), forward_buffer.data_ptr(), B,
                                  input_dim, output_dim, hidden_dim, num_layers, activation, output_activation,
                                  calc_grad_inputs ? 1 : 0, backward_buffer.data_ptr(), grad_inputs.data_ptr(),
                                  grad_weights.data_ptr(), dtype, l.stream), "ffmlp_backward");
}

void allocate_splitk(size_t size) { shim::ok(enerf_allocate_splitk(size), "allocate_splitk"); }
void free_splitk() { shim::ok(enerf_free_splitk(), "free_splitk"); }

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    m.def("ffmlp_forward", &ffmlp_forward, "ffmlp_forward (HIP, gfx950 MFMA)");
    m.def("ffmlp_inference", &ffmlp_inference, "ffmlp_inference (HIP, gfx950 MFMA)");
    m.def("ffmlp_backward", &ffmlp_backward, "ffmlp_backward (HIP, gfx950 MFMA)");
    m.def("allocate_splitk", &allocate_splitk, "allocate_splitk");
    m.def("free_splitk", &free_splitk, "free_splitk");
}
