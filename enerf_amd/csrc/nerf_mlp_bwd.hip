// nerf_mlp_bwd.hip -- k_nerf_bwd of nerf_mlp.hip as a translation unit of its own, compiled WITHOUT the operand barrier of
// mlp32s_ops.h (operand_ready): the hazard it closes needs two or more wavefronts of these kernels on a SIMD, and this
// kernel runs one per SIMD by construction (149 KiB of LDS, the SIMD's whole register file claimed: nothing shares it).
// The barrier would only cost it its interleaving of conversions and MFMAs (49.9 -> 53.0 us on the 4096-ray batch).
// tools/nerf_fwd_residency.sh: the padded no-barrier build is clean at one workgroup per CU, 400 of 400 launches.
#define MLP32S_NO_OPERAND_BARRIER
#define NERF_MLP_BACKWARD_UNIT
#include "nerf_mlp.hip"
