// Shared by the four pybind11 extension modules (_raymarching, _gridencoder, _shencoder, _ffmlp): tensor checks in the
// reference's style (CHECK_CUDA / CHECK_CONTIGUOUS / CHECK_IS_*), the stream the kernels go to (torch's CURRENT stream
// of the tensor's device -- the reference uses the legacy default stream) and the C ABI's status -> c10::Error.
// Host code only: the kernels live in libenerf_hip.so (include/enerf_hip.h).
#pragma once
#include <torch/extension.h>
// torch on ROCm presents the GPU as device type `cuda`: its guard / stream classes for extension code are the
// "masquerading" ones (c10::hip::HIPGuard proper refuses a cuda-typed device)
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>

#include "enerf_hip.h"

namespace shim {

inline void need_device(const at::Tensor& t, const char* name) {
    TORCH_CHECK(t.device().is_cuda(), name, " must be a CUDA tensor");
}
inline void need_dense(const at::Tensor& t, const char* name) {
    need_device(t, name);
    TORCH_CHECK(t.is_contiguous(), name, " must be a contiguous tensor");
}
inline void need_f32(const at::Tensor& t, const char* name) {
    need_dense(t, name);
    TORCH_CHECK(t.scalar_type() == at::kFloat, name, " must be a float32 tensor");
}
inline void need_i32(const at::Tensor& t, const char* name) {
    need_dense(t, name);
    TORCH_CHECK(t.scalar_type() == at::kInt, name, " must be an int tensor");
}
inline void need_floating(const at::Tensor& t, const char* name) {
    need_dense(t, name);
    TORCH_CHECK(t.scalar_type() == at::kFloat || t.scalar_type() == at::kHalf || t.scalar_type() == at::kBFloat16 ||
                    t.scalar_type() == at::kDouble,
                name, " must be a floating tensor");
}

// element type code of the C ABI for tensors the reference dispatches on
inline int abi_dtype(const at::Tensor& t, const char* name, bool allow_bf16 = false) {
    switch (t.scalar_type()) {
        case at::kFloat: return ENERF_F32;
        case at::kHalf: return ENERF_F16;
        case at::kBFloat16:
            TORCH_CHECK(allow_bf16, name, ": bfloat16 is not supported here");
            return ENERF_BF16;
        default: TORCH_CHECK(false, name, ": unsupported dtype ", t.scalar_type());
    }
    return -1;
}

// RAII: make the tensor's device current and expose torch's current stream on it
struct Launch {
    c10::hip::HIPGuardMasqueradingAsCUDA guard;
    enerf_stream_t stream;
    explicit Launch(const at::Tensor& t)
        : guard((need_device(t, "the first tensor argument"), t.device())),
          stream((enerf_stream_t)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(t.device().index()).stream()) {}
};

inline void ok(int status, const char* what) {
    TORCH_CHECK(status == 0, what, " failed (", status, "): ", enerf_last_error());
}

}  // namespace shim
