// ffmlp.hip -- placeholder until the MFMA kernels land (filled in next milestone).
#include "common.h"
extern "C" {
int enerf_ffmlp_forward(const void*, const void*, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t,
                        void*, void*, int, enerf_stream_t) { enerf::set_error("ffmlp: not built"); return ENERF_E_UNSUPPORTED; }
int enerf_ffmlp_inference(const void*, const void*, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t,
                          void*, void*, int, enerf_stream_t) { enerf::set_error("ffmlp: not built"); return ENERF_E_UNSUPPORTED; }
int enerf_ffmlp_backward(const void*, const void*, const void*, const void*, uint32_t, uint32_t, uint32_t, uint32_t,
                         uint32_t, uint32_t, uint32_t, int, void*, void*, void*, int, enerf_stream_t) { enerf::set_error("ffmlp: not built"); return ENERF_E_UNSUPPORTED; }
int enerf_allocate_splitk(size_t) { return 0; }
int enerf_free_splitk(void) { return 0; }
}
