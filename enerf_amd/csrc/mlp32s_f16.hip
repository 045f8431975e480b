// mlp32s_f16.hip -- the kernels of mlp32s.hip once more, with IEEE half operands (v_mfma_f32_32x32x16_f16, one product per
// operand pair, fp32 accumulation; activations, activation gradients and weights rounded to fp16 where that file rounds them
// to bf16).  enerf_mlp32_precision(3): the arithmetic of the reference's `fp16 = True` regime (nerf/utils.py:964-975:
// autocast(float16) puts the nn.Linear GEMMs on half operands) for the closed-form training step; fp16's narrow exponent
// range is why that regime carries a loss scale (csrc/optim.hip: enerf_amp_*).
#define ENERF_MLP32S_F16 1
#include "mlp32s.hip"
