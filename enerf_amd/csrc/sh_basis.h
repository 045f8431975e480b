// sh_basis.h -- the 16 real spherical harmonics of degree < 4 in registers (same recurrences and operation order as
// shencoder.hip's k_sh_fwd<T, 4, false>), for kernels that build the colour net's direction inputs themselves.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

namespace enerf {

struct ShNorm4 {
    float n[4][4];      // n[l][m], m <= l, includes sqrt(2) for m > 0 (shencoder.hip: fill_norm)
};

inline ShNorm4 make_sh_norm4() {
    ShNorm4 nrm;
    for (int l = 0; l < 4; l++)
        for (int m = 0; m < 4; m++) {
            double v = 0.0;
            if (m <= l) {
                double ratio = 1.0;
                for (int k = l - m + 1; k <= l + m; k++) ratio /= (double)k;
                v = sqrt((2.0 * l + 1.0) / (4.0 * M_PI) * ratio);
                if (m) v *= sqrt(2.0);
            }
            nrm.n[l][m] = (float)v;
        }
    return nrm;
}

__device__ __forceinline__ void sh4(float x, float y, float z, const ShNorm4& nrm, float (&Y)[16]) {
    float A[4], Bm[4];
    A[0] = 1.0f;
    Bm[0] = 0.0f;
#pragma unroll
    for (int m = 1; m < 4; m++) {
        A[m] = x * A[m - 1] - y * Bm[m - 1];
        Bm[m] = x * Bm[m - 1] + y * A[m - 1];
    }
#pragma unroll
    for (int m = 0; m < 4; m++) {
        float qmm = 1.0f;
#pragma unroll
        for (int k = 1; k <= m; k++) qmm *= -(2.0f * k - 1.0f);
        float Q[4];
        Q[m] = qmm;
        if (m + 1 < 4) Q[m + 1] = (2.0f * m + 1.0f) * z * qmm;
#pragma unroll
        for (int l = m + 2; l < 4; l++)
            Q[l] = ((2.0f * l - 1.0f) * z * Q[l - 1] - (float)(l + m - 1) * Q[l - 2]) * (1.0f / (float)(l - m));
#pragma unroll
        for (int l = m; l < 4; l++) {
            const float nq = nrm.n[l][m] * Q[l];
            Y[l * l + l + m] = nq * A[m];
            if (m) Y[l * l + l - m] = nq * Bm[m];
        }
    }
}

}  // namespace enerf
