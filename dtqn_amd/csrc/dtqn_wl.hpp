// Weights through LDS (gfx950): GEMM stages whose B operand is an LDS-resident weight tile.
//
// The register-direct stages of dtqn_device.hpp (StageXwT / StageDyW) fetch every work item's weight fragment from L2
// one item ahead of its use.  At the small-batch shapes one wave has two or three items per stage and an item's MFMA
// chain (0.2 - 0.4 us) is shorter than an L2 round trip under load (~1 us): the stage is a chain of exposed round trips
// (stage clocks, profiles/r02_stage_profile_cfg1_before.txt: qkv 4.3 us against 1.3 us of MFMA issue).  Here the whole
// weight tile of stage s+1 is pulled global -> registers by ALL threads at the start of stage s (coalesced 16 B / lane,
// one read of every weight element per workgroup, as before), dropped into LDS once its region is free, and published
// by the barrier that ends stage s.  Stage s+1 then reads A and B fragments from LDS (ds_read_b128): no memory latency
// inside a stage, and no weight fragment double buffer in registers.
#pragma once
#include "dtqn_device.hpp"

namespace dtqn {

// acc[m] += X[m-th row tile][K] * W[col][K]^T.  Xs: LDS [rows][lda]; Wrow: LDS pointer to W[col of this lane][0] of a
// [N][K + 4] tile.  Same k-permutation as frag_xwT_mma (lane group kq owns k in [kq*K/4, (kq+1)*K/4)).
template <int K, int MG>
__device__ __forceinline__ void frag_xwl_mma(const float* Xs, int lda, const float* Wrow, const Thr& t, f32x4 (&acc)[MG]) {
    constexpr int KS = K / 16;
    const float* xp = Xs + t.i * lda + t.kq * (K / 4);
    const float* wp = Wrow + t.kq * (K / 4);
    float4 af[2][MG], bf[2];
#pragma unroll
    for (int m = 0; m < MG; ++m) af[0][m] = ld4(xp + m * 16 * lda);
    bf[0] = ld4(wp);
    f32x4 alt = zero4();
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        if (s + 1 < KS) {
#pragma unroll
            for (int m = 0; m < MG; ++m) af[(s + 1) & 1][m] = ld4(xp + m * 16 * lda + 4 * (s + 1));
            bf[(s + 1) & 1] = ld4(wp + 4 * (s + 1));
        }
        DTQN_SCHED_FENCE();      // the next step's LDS reads stay AHEAD of this step's MFMAs (hipcc otherwise sinks them behind)
        const float b4[4] = {bf[s & 1].x, bf[s & 1].y, bf[s & 1].z, bf[s & 1].w};
        if (MG == 1) {
            const float a4[4] = {af[s & 1][0].x, af[s & 1][0].y, af[s & 1][0].z, af[s & 1][0].w};
            acc[0] = mfma16(a4[0], b4[0], acc[0]);
            alt = mfma16(a4[1], b4[1], alt);
            acc[0] = mfma16(a4[2], b4[2], acc[0]);
            alt = mfma16(a4[3], b4[3], alt);
        } else {
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
    if (MG == 1) {
        acc[0][0] += alt[0]; acc[0][1] += alt[1]; acc[0][2] += alt[2]; acc[0][3] += alt[3];
    }
}

// Y[MT*16][NTILES*16] = X W^T (+ bias), W an LDS tile [NTILES*16][K + 4], bias an LDS vector (or nullptr).
// Work items = (16-column tile, group of MG row tiles) dealt round-robin to the NW waves, as StageXwT.
template <int K, int MT, int MG, int NW, int NTILES>
struct StageXwL {
    static constexpr int MGROUPS = MT / MG;
    static constexpr int ITEMS = NTILES * MGROUPS;
    static constexpr int PER_WAVE = (ITEMS + NW - 1) / NW;
    static constexpr int LDWL = K + 4;
    template <typename Epi>
    __device__ static __forceinline__ void run(const float* Xs, int lda, const float* Wl, const float* bl, const Thr& t, Epi epi) {
#pragma unroll
        for (int q = 0; q < PER_WAVE; ++q) {
            const int item = t.wave + q * NW;
            if (ITEMS >= (q + 1) * NW || item < ITEMS) {
                const int nt = item / MGROUPS, mg = item - nt * MGROUPS;
                const float bias = bl != nullptr ? bl[nt * 16 + t.i] : 0.f;
                f32x4 acc[MG];
#pragma unroll
                for (int m = 0; m < MG; ++m) acc[m] = zero4();
                frag_xwl_mma<K, MG>(Xs + mg * MG * 16 * lda, lda, Wl + (nt * 16 + t.i) * LDWL, t, acc);
#pragma unroll
                for (int m = 0; m < MG; ++m)
#pragma unroll
                    for (int r = 0; r < 4; ++r) epi((mg * MG + m) * 16 + t.kq * 4 + r, nt * 16 + t.i, acc[m][r] + bias);
            }
        }
    }
};

// LDS arena of the forward pass for a width-D network with FFN hidden chunks of 2D columns (floats):
//   region A = [0, 2D(D+4))            : W_in rows 0..2D-1 | FFN-1 chunk [2D][D+4]
//   region B = [2D(D+4), 4D(D+4))      : W_in rows 2D..3D-1 then W_out [D][D+4] | FFN-2 chunk [D][2D+4]
// plus two parameter blocks of 13 D floats (LayerNorm affines and biases of a layer, alternating by layer parity).
constexpr int wl_arena_floats(int D) { return 4 * D * (D + 4); }
constexpr int wl_small_floats(int D) { return 13 * D; }

}  // namespace dtqn
