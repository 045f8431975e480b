// mfma_guard.h -- keeps the result of a zero-initialised 32x32 MFMA off its own operands (translation units compiled with
// -mllvm -amdgpu-mfma-vgpr-form: ffmlp.hip, mlp32s.hip, mlp32s_f16.hip, nerf_mlp.hip).
//
// A 32x32 MFMA reads SrcA / SrcB over several passes and must not write its 16 result registers over them; the compiler
// knows (the AGPR form's result is early-clobber).  With -amdgpu-mfma-vgpr-form an MFMA whose accumulator input is the
// constant 0 is emitted in the untied three-address VGPR form WITHOUT that constraint, and the register allocator happily
// reuses the registers of an operand that dies at the instruction:  v_mfma_f32_32x32x16_bf16 v[24:39], v[22:25], v[44:47], 0.
// On the hardware that is right most of the time and wrong when the matrix pipe is contended: found in round 5 as colour
// outputs of samples 16..31 of a tile computed to bf16 accuracy only, in a few per cent of the tiles of workgroups that
// share a CU (tools/mfma_overlap.py lists such instructions in a hipcc -S dump; 170 of them in mlp32s.hip's kernels as
// shipped in rounds 3 and 4; tests/test_mfma_operand_overlap.py keeps the count at zero).
// The guard costs no instruction: an empty asm statement after the MFMA that reads the operands, so that A and B are still
// live where D is defined, and passes D through, which pins the statement behind the MFMA (without D nothing but source
// order does, and ffmlp.hip kept 16 overlaps).  One instance, k_mlp32s_bwd<3, 0, 1, false>, crashes this compiler's 'AMDGPU
// Rewrite AGPR-Copy-MFMA' pass with the guard in place and was retired (three hidden layers always recompute).
// Accumulating MFMAs (D = C, tied) cannot overlap their operands and are left alone -- their long-lived accumulators stay
// free to sit in AGPRs.  tests/test_mfma_operand_overlap.py scans the compiled device code: the guard is only as good as
// __builtin_constant_p's view of C.
#pragma once
#include <hip/hip_runtime.h>

// (build.py defines ENERF_MFMA_VGPR_FORM next to the -mllvm flag; in the default AGPR form the result cannot overlap a VGPR
//  operand and the guard would only force a copy)
#ifdef ENERF_MFMA_VGPR_FORM
#define ENERF_MFMA_GUARD(D, A, B, C)                                      \
    do {                                                                  \
        if (__builtin_constant_p((C)[0])) asm volatile("" : "+v"(D) : "v"(A), "v"(B)); \
    } while (0)
#else
#define ENERF_MFMA_GUARD(D, A, B, C) do {} while (0)
#endif
