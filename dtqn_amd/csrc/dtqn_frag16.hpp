// 16-grouped MFMA fragments shared by the row-block kernels (dtqn_tiled.hip) and the image encoder (dtqn_image.hip).
#pragma once
#include "dtqn_device.hpp"

namespace dtqn {

// Fragments of the row-block GEMMs with the contraction index grouped in 16s: at step s lane (i, kq) holds
// k = 16 s + 4 kq + {0..3} of weight row i, so the four kq lanes of a row read 64 contiguous bytes and a wave's load
// instruction touches 16 cache lines (frag_xwT_fetch's k = kq K/4 + 4 s + c layout touches 64: one per lane).  The A
// operand follows the same order out of LDS: 16-lane groups read rows 132 words apart -> 64 different banks.
template <int K>
__device__ __forceinline__ void frag16_fetch(float4 (&bf)[K / 16], const float* __restrict__ Wrow, const Thr& t) {
    const float4* wp = reinterpret_cast<const float4*>(Wrow + t.kq * 4);
#pragma unroll
    for (int s = 0; s < K / 16; ++s) bf[s] = wp[4 * s];
}
template <int K, int MG, int NB = K / 16>
__device__ __forceinline__ void frag16_mma(const float* Xs, int lda, const float4 (&bf)[NB], const Thr& t, f32x4 (&acc)[MG]) {
    constexpr int KS = K / 16;
    static_assert(NB >= KS, "fragment array too short");
    const float* xp = Xs + t.i * lda + t.kq * 4;
    float4 af[2][MG];
#pragma unroll
    for (int m = 0; m < MG; ++m) af[0][m] = ld4(xp + m * 16 * lda);
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        if (s + 1 < KS) {
#pragma unroll
            for (int m = 0; m < MG; ++m) af[(s + 1) & 1][m] = ld4(xp + m * 16 * lda + 16 * (s + 1));
        }
        const float b4[4] = {bf[s].x, bf[s].y, bf[s].z, bf[s].w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
#pragma unroll
            for (int m = 0; m < MG; ++m) {
                const float4 a = af[s & 1][m];
                const float av = c == 0 ? a.x : (c == 1 ? a.y : (c == 2 ? a.z : a.w));
                acc[m] = mfma16(av, b4[c], acc[m]);
            }
        }
    }
}

}  // namespace dtqn
