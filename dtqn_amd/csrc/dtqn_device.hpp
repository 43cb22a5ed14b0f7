// Device building blocks shared by the DTQN forward / backward kernels (gfx950, wave64).
//
// One workgroup = NW waves (template parameter: 4, 8 or 16) owns ONE sequence: its [LP x D] context tile lives in LDS
// for the whole pass.  Dense projections run on the exact-f32 matrix core
// (v_mfma_f32_16x16x4_f32: lane l supplies A[i=l&15][k=l>>4], B[k=l>>4][j=l&15]; D reg r is
// D[(l>>4)*4+r][l&15]).  The contraction index is permuted so that each lane's k-slices are
// CONTIGUOUS in memory (lane group kq owns k in [kq*K/4, (kq+1)*K/4)): A-fragments come out of
// LDS as ds_read_b128 and weight fragments out of L2 as global_load_dwordx4, four MFMA steps per
// load.  Attention (head_dim 8..32, L <= 64) is too skinny for MFMA and runs on the VALU.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <type_traits>

#include "dtqn_hip.h"
#include "dtqn_limits.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Critical-path options of the whole-sequence TD kernels (round 5), one bit each so that tools/build_variant.py can build any subset
// for an A/B on the GPU (-DDTQN_OPT=<mask>).  Nine were built and traced one by one (profiles/r05_option_attribution_cfg1.txt); the four
// that paid are kept, the five that did not (LayerNorm gammas / ReLU ballots / head.2 weights prefetched into LDS: more live registers
// in a kernel that already spills; a loss record written by the forward: +0.6 us there for -0.2 us in the loss stage) are gone:
//   4   backward: log-sum-exp rows of a head group in flight with the group's tiles (-0.7 us)
//   32  backward: gradient records leave as write-through (sc1) stores (-0.5 us)
//   64  forward: activation records likewise (-2.5 us: the kernel no longer ends with 17 MB of dirty lines to write back)
//   128 forward: window-independent embedding operands in flight ahead of the window draw (-0.3 us)
// 16-row slices, products whose output is only D / 16 = 4 column tiles wide on a workgroup of 8 waves: the contraction is split over the two
// halves of the waves instead of leaving four of them idle (half the weight-fragment registers and half the fragment's fetch per wave):
//   1024 backward dh W_1 (-3.6 us: 42.9 -> 39.3 us per launch)      2048 forward FFN-2 (+0.5 us: the extra barrier costs more than the
//   16 MFMAs it saves; off)      4096 backward dqkv W_in      8192 backward head.1 and dO products
#ifndef DTQN_OPT
#define DTQN_OPT (4 | 32 | 64 | 128 | 1024 | 4096 | 8192 | 16384)
#endif
namespace dtqn {
constexpr bool kOptLse = (DTQN_OPT & 4) != 0, kOptBwdWT = (DTQN_OPT & 32) != 0, kOptFwdWT = (DTQN_OPT & 64) != 0, kOptHoist = (DTQN_OPT & 128) != 0;
constexpr bool kOptSplitK = (DTQN_OPT & 1024) != 0, kOptSplitKFwd = (DTQN_OPT & 2048) != 0, kOptSplitW = (DTQN_OPT & 4096) != 0;
constexpr bool kOptSplitS = (DTQN_OPT & 8192) != 0;
// 16384 backward dh W_1 on whole tiles (64-row workgroups: every wave is busy there) with HALF weight fragments in time instead of whole
//       ones: 32 fewer registers in a kernel that spills (BASELINE config 2: 3 077 -> 3 113 TD-updates/s)
constexpr bool kOptHalfW = (DTQN_OPT & 16384) != 0;
}

extern __shared__ __attribute__((aligned(16))) unsigned char dtqn_smem[];

extern "C" void* dtqn_debug_profile_buffer(void);

namespace dtqn {
// row-block tiled TD passes (dtqn_tiled.hip), dispatched to by dtqn_td_forward / dtqn_td_backward when net->tiled
int tiled_td_forward(const DtqnNet* net, const DtqnReplay* rp, const DtqnTd* td, hipStream_t stream);
int tiled_td_forward_part(const DtqnNet* net, const DtqnReplay* rp, const DtqnTd* td, int pass0, int npasses, hipStream_t stream);
int tiled_td_backward(const DtqnNet* net, const DtqnReplay* rp, const DtqnTd* td, hipStream_t stream);
int tiled_forward_actor(const DtqnNet* net, const float* theta, const float* obs, const uint8_t* actions, int batch, int n, int in_rows,
                        float* q_out, float* workspace, int train_mode, uint32_t drop_seed, uint32_t drop_step, hipStream_t stream, const int32_t* lens = nullptr);
// dtqn_forward with an optional pinned-host destination for Q of the last row of sequence 0 (dtqn_actor_forward)
int forward_infer(const DtqnNet* net, const float* theta, const float* obs, const uint8_t* actions, int batch, int n,
                  float* q_out, float* q_last_host, void* stream, float* xch, int32_t* xflags, const int32_t* last_rows = nullptr, int in_rows = 0,
                  uint32_t drop_seed = 0, uint32_t drop_step = 0, int train_mode = 0);

// Raise a kernel's dynamic-LDS limit once per (instantiation, device, size): the call is a driver round trip, and the
// attribute is per device, so a second GPU used by the same process needs its own call (`cache` = one function-static
// array per instantiation).
constexpr int kMaxDevices = 16;
inline void raise_lds_limit(const void* fn, size_t lds, size_t (&cache)[kMaxDevices]) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) {       // unknown device: always set
        (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        return;
    }
    if (lds > cache[dev]) {
        (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        cache[dev] = lds;
    }
}
// compute units of the current device (cached per device; 0 if the runtime cannot say)
inline int device_cu_count() {
    static int cache[kMaxDevices] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return 0;
    if (cache[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 0;
        cache[dev] = n > 0 ? n : -1;
    }
    return cache[dev] > 0 ? cache[dev] : 0;
}
}  // namespace dtqn

// Stage timestamps (debug): thread 0 of workgroups 0 and 1 (two row slices of sequence 0 in latency mode), 32 slots
// each.  Compiled in only with -DDTQN_ENABLE_PROF (DTQN_BUILD_PROF=1 python -m dtqn_amd.build): every mark is a
// conditional global store, and a conditional store between a prefetch and its use makes the compiler wait with
// vmcnt(0) there -- the product build must not carry them.
#ifndef DTQN_PROF_WG1
#define DTQN_PROF_WG1 1        /* the second profiled workgroup (-DDTQN_PROF_WG1=3: the top one of four row slices) */
#endif
#ifdef DTQN_ENABLE_PROF
#define DTQN_PROF(buf, slot) \
    do { if ((buf) != nullptr && (blockIdx.x == 0 || blockIdx.x == DTQN_PROF_WG1) && threadIdx.x == 0) (buf)[(blockIdx.x == 0 ? 0 : 32) + (slot)] = (long long)wall_clock64(); } while (0)
#else
#define DTQN_PROF(buf, slot) ((void)(slot))
#endif

#include <dtqn_gfx950.hpp>

namespace dtqn {

struct Thr {
    int tid, lane, wave, i, kq;
};
__device__ __forceinline__ Thr make_thr() {
    Thr t;
    t.tid = (int)threadIdx.x;
    t.lane = t.tid & 63;
    t.wave = t.tid >> 6;
    t.i = t.lane & 15;
    t.kq = t.lane >> 4;
    return t;
}

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 zero4() {
    f32x4 z = {0.f, 0.f, 0.f, 0.f};
    return z;
}
__device__ __forceinline__ void retire4(float4& v) {
    DTQN_ASM_KEEP(v.x); DTQN_ASM_KEEP(v.y); DTQN_ASM_KEEP(v.z); DTQN_ASM_KEEP(v.w);
}
// ReLU pattern of one accumulator register across the wave -> one 64-bit word of the mask record
// (index: (row tile, column tile, r)); must be called by every lane of the wave.
// Reductions over the four lanes {i, i+16, i+32, i+48} that hold one column of an MFMA 16x16 output tile.
__device__ __forceinline__ float kq_max(float v) {
    const auto a = DTQN_LANE_SWAP32(v);
    v = fmaxf(a[0], a[1]);
    const auto b = DTQN_LANE_SWAP16(v);
    return fmaxf(b[0], b[1]);
}
__device__ __forceinline__ float kq_sum(float v) {
    const auto a = DTQN_LANE_SWAP32(v);
    v = a[0] + a[1];
    const auto b = DTQN_LANE_SWAP16(v);
    return b[0] + b[1];
}
// All-reduce over the 64 lanes of a wave without touching LDS: halves and rows with the lane-swap instructions, the
// 16 lanes of a row with four DPP rotations.  Every lane ends up with the result.
#define DTQN_WAVE_ALLREDUCE(v, OP)                                   \
    do {                                                            \
        const auto a_ = DTQN_LANE_SWAP32(v); v = OP(a_[0], a_[1]);  \
        const auto b_ = DTQN_LANE_SWAP16(v); v = OP(b_[0], b_[1]);  \
        v = OP(v, DTQN_ROW_ROR(v, 8)); v = OP(v, DTQN_ROW_ROR(v, 4)); \
        v = OP(v, DTQN_ROW_ROR(v, 2)); v = OP(v, DTQN_ROW_ROR(v, 1)); \
    } while (0)
__device__ __forceinline__ float dtqn_addf(float a, float b) { return a + b; }
__device__ __forceinline__ float wave_sum(float v) { DTQN_WAVE_ALLREDUCE(v, dtqn_addf); return v; }
__device__ __forceinline__ float wave_max(float v) { DTQN_WAVE_ALLREDUCE(v, fmaxf); return v; }
__device__ __forceinline__ float wave_min(float v) { DTQN_WAVE_ALLREDUCE(v, fminf); return v; }
__device__ __forceinline__ void ballot_store(float* mask_rec, int ctiles, int row, int col, bool on, int lane) {
    const unsigned long long bits = __ballot(on ? 1 : 0);
    // every lane stores the same 8 bytes to the same address (one transaction): no exec-masked region around the store
    (void)lane;
    reinterpret_cast<unsigned long long*>(mask_rec)[((row >> 4) * ctiles + (col >> 4)) * 4 + (row & 3)] = bits;
}
__device__ __forceinline__ bool mask_bit(const float* mask_rec, int ctiles, int row, int col) {
    const unsigned long long w = reinterpret_cast<const unsigned long long*>(mask_rec)[((row >> 4) * ctiles + (col >> 4)) * 4 + (row & 3)];
    return (w >> ((((row >> 2) & 3) << 4) + (col & 15))) & 1ull;
}
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

// ------------------------------------------------------------------------------------------
// acc[m] += X[m-tile rows][K] * W[ncol][K]^T for one 16-column n-tile.
//   Xs : LDS, row-major, leading dim lda (multiple of 4), MT*16 rows
//   Wrow: global pointer to W[ncol][0] of THIS lane's column (or nullptr -> zero column)
// ------------------------------------------------------------------------------------------
template <int K, int MT>
__device__ __forceinline__ void mma_xwT_tile(const float* Xs, int lda, const float* __restrict__ Wrow, const Thr& t,
                                             f32x4 (&acc)[MT]) {
    static_assert(K % 16 == 0, "K must be a multiple of 16");
    constexpr int KS = K / 16;
    float4 bf[KS];
    if (Wrow != nullptr) {
        const float4* wp = reinterpret_cast<const float4*>(Wrow + t.kq * (K / 4));
#pragma unroll
        for (int s = 0; s < KS; ++s) bf[s] = wp[s];
    } else {
#pragma unroll
        for (int s = 0; s < KS; ++s) bf[s] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const float* xp = Xs + t.i * lda + t.kq * (K / 4);
#pragma unroll
    for (int s = 0; s < KS; ++s) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const float4 af = ld4(xp + m * 16 * lda + 4 * s);
            acc[m] = mfma16(af.x, bf[s].x, acc[m]);
            acc[m] = mfma16(af.y, bf[s].y, acc[m]);
            acc[m] = mfma16(af.z, bf[s].z, acc[m]);
            acc[m] = mfma16(af.w, bf[s].w, acc[m]);
        }
    }
}

// ------------------------------------------------------------------------------------------
// Split-phase fragments: the weight fragment of a work item is FETCHED (global loads issued) well
// before it is consumed, so the L2 round trip overlaps the barrier / the previous item's MFMAs.
// ------------------------------------------------------------------------------------------
template <int K>
__device__ __forceinline__ void frag_xwT_fetch(float4 (&bf)[K / 16], const float* __restrict__ Wrow, const Thr& t) {
    const float4* wp = reinterpret_cast<const float4*>(Wrow + t.kq * (K / 4));
#pragma unroll
    for (int s = 0; s < K / 16; ++s) bf[s] = wp[s];
}
// MFMA order matters: v_mfma_f32_16x16x4_f32 issues every 32 cycles per SIMD but a DEPENDENT
// accumulate needs 40, and a ds_read_b128 -> use is ~64+ cycles.  So the A fragments of step s+1 are
// read while step s multiplies, and consecutive MFMAs always hit different accumulators (across the
// MG row tiles; for MG == 1 the k-steps alternate between two accumulators that are summed at the end).
template <int K, int MG>
__device__ __forceinline__ void frag_xwT_mma(const float* Xs, int lda, const float4 (&bf)[K / 16], const Thr& t, f32x4 (&acc)[MG]) {
    constexpr int KS = K / 16;
    const float* xp = Xs + t.i * lda + t.kq * (K / 4);
    float4 af[2][MG];
#pragma unroll
    for (int m = 0; m < MG; ++m) af[0][m] = ld4(xp + m * 16 * lda);
    f32x4 alt = zero4();
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        if (s + 1 < KS) {
#pragma unroll
            for (int m = 0; m < MG; ++m) af[(s + 1) & 1][m] = ld4(xp + m * 16 * lda + 4 * (s + 1));
        }
        const float b4[4] = {bf[s].x, bf[s].y, bf[s].z, bf[s].w};
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
template <int NN>
__device__ __forceinline__ void frag_dyw_fetch(float (&bf)[NN / 4], const float* __restrict__ Wcol, int ldw, const Thr& t) {
    const float* wp = Wcol + (size_t)(t.kq * (NN / 4)) * ldw;
#pragma unroll
    for (int q = 0; q < NN / 4; ++q) bf[q] = wp[(size_t)q * ldw];
}
template <int NN, int MG>
__device__ __forceinline__ void frag_dyw_mma(const float* dYs, int lda, const float (&bf)[NN / 4], const Thr& t, f32x4 (&acc)[MG]) {
    constexpr int KS = NN / 16;
    const float* yp = dYs + t.i * lda + t.kq * (NN / 4);
    float4 af[2][MG];
#pragma unroll
    for (int m = 0; m < MG; ++m) af[0][m] = ld4(yp + m * 16 * lda);
    f32x4 alt = zero4();
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        if (s + 1 < KS) {
#pragma unroll
            for (int m = 0; m < MG; ++m) af[(s + 1) & 1][m] = ld4(yp + m * 16 * lda + 4 * (s + 1));
        }
        if (MG == 1) {
            const float4 a = af[s & 1][0];
            acc[0] = mfma16(a.x, bf[4 * s + 0], acc[0]);
            alt = mfma16(a.y, bf[4 * s + 1], alt);
            acc[0] = mfma16(a.z, bf[4 * s + 2], acc[0]);
            alt = mfma16(a.w, bf[4 * s + 3], alt);
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
#pragma unroll
                for (int m = 0; m < MG; ++m) {
                    const float4 a = af[s & 1][m];
                    const float av = c == 0 ? a.x : (c == 1 ? a.y : (c == 2 ? a.z : a.w));
                    acc[m] = mfma16(av, bf[4 * s + c], acc[m]);
                }
            }
        }
    }
    if (MG == 1) {
        acc[0][0] += alt[0]; acc[0][1] += alt[1]; acc[0][2] += alt[2]; acc[0][3] += alt[3];
    }
}

// How many of the MT row tiles one work item keeps (sharing one weight fragment): the largest
// power-of-two divisor of MT that still leaves at least NW items, so every wave has work.
constexpr int pick_mg(int ntiles, int MT, int NW) {
    // minimise the MFMA work of the busiest wave, ceil(items / NW) * MG; prefer the larger MG on ties
    // (more reuse of the weight fragment)
    int best = 1, best_cost = 1 << 30;
    for (int mg = 1; mg <= MT; mg <<= 1) {
        if (MT % mg != 0) continue;
        const int items = ntiles * (MT / mg);
        const int cost = ((items + NW - 1) / NW) * mg;
        if (cost <= best_cost) { best = mg; best_cost = cost; }
    }
    return best;
}

// Y[rows][N] = X * W^T, W global [N][ldw]; epi(row, col, value) is called for every element the lane owns.
// Work item = (16-column n-tile, group of MG row tiles), dealt round-robin to the NW waves.
template <int K, int MT, int MG, int NW, typename Epi>
__device__ __forceinline__ void gemm_xwT(const float* Xs, int lda, const float* __restrict__ W, int ldw, int N,
                                         const Thr& t, Epi epi) {
    constexpr int MGROUPS = MT / MG;
    const int ntiles = (N + 15) >> 4;
    for (int item = t.wave; item < ntiles * MGROUPS; item += NW) {
        const int nt = item / MGROUPS, mg = item - nt * MGROUPS;
        const int col = nt * 16 + t.i;
        f32x4 acc[MG];
#pragma unroll
        for (int m = 0; m < MG; ++m) acc[m] = zero4();
        mma_xwT_tile<K, MG>(Xs + mg * MG * 16 * lda, lda, col < N ? W + (size_t)col * ldw : nullptr, t, acc);
        if (col < N) {
#pragma unroll
            for (int m = 0; m < MG; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) epi((mg * MG + m) * 16 + t.kq * 4 + r, col, acc[m][r]);
        }
    }
}

// ------------------------------------------------------------------------------------------
// acc[m] += dY[m-tile rows][NN] * W[NN][col] for one 16-column tile of the OUTPUT (contraction
// over the ROWS of W).  Wcol: global pointer to W[0][col] of this lane's column (or nullptr).
// ------------------------------------------------------------------------------------------
template <int NN, int MT>
__device__ __forceinline__ void mma_dyw_tile(const float* dYs, int lda, const float* __restrict__ Wcol, int ldw,
                                             const Thr& t, f32x4 (&acc)[MT]) {
    static_assert(NN % 16 == 0, "contraction length must be a multiple of 16");
    constexpr int Q = NN / 4;
    float bf[Q];
    if (Wcol != nullptr) {
        const float* wp = Wcol + (size_t)(t.kq * Q) * ldw;
#pragma unroll
        for (int q = 0; q < Q; ++q) bf[q] = wp[(size_t)q * ldw];
    } else {
#pragma unroll
        for (int q = 0; q < Q; ++q) bf[q] = 0.f;
    }
    const float* yp = dYs + t.i * lda + t.kq * Q;
#pragma unroll
    for (int s = 0; s < Q / 4; ++s) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const float4 af = ld4(yp + m * 16 * lda + 4 * s);
            acc[m] = mfma16(af.x, bf[4 * s + 0], acc[m]);
            acc[m] = mfma16(af.y, bf[4 * s + 1], acc[m]);
            acc[m] = mfma16(af.z, bf[4 * s + 2], acc[m]);
            acc[m] = mfma16(af.w, bf[4 * s + 3], acc[m]);
        }
    }
}

// dX[rows][Kout] = dY * W, W global [NN][ldw]; epi(row, col, value).  Same item scheme as gemm_xwT.
template <int NN, int MT, int MG, int NW, typename Epi>
__device__ __forceinline__ void gemm_dyw(const float* dYs, int lda, const float* __restrict__ W, int ldw, int Kout,
                                         const Thr& t, Epi epi) {
    constexpr int MGROUPS = MT / MG;
    const int ktiles = (Kout + 15) >> 4;
    for (int item = t.wave; item < ktiles * MGROUPS; item += NW) {
        const int kt = item / MGROUPS, mg = item - kt * MGROUPS;
        const int col = kt * 16 + t.i;
        f32x4 acc[MG];
#pragma unroll
        for (int m = 0; m < MG; ++m) acc[m] = zero4();
        mma_dyw_tile<NN, MG>(dYs + mg * MG * 16 * lda, lda, col < Kout ? W + col : nullptr, ldw, t, acc);
        if (col < Kout) {
#pragma unroll
            for (int m = 0; m < MG; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) epi((mg * MG + m) * 16 + t.kq * 4 + r, col, acc[m][r]);
        }
    }
}

// Register-resident accumulation across several partial GEMMs with a FIXED output ownership:
// the D/16 x (MT/MG) output items of a [LP][D] tile are dealt to the NW waves once; item q of this
// wave is (nt, mg) = owned_item<..>(wave, q).
template <int D, int MT, int MG, int NW>
struct Owned {
    static constexpr int MGROUPS = MT / MG;
    static constexpr int ITEMS = (D / 16) * MGROUPS;
    static constexpr int PER_WAVE = (ITEMS + NW - 1) / NW;
    __device__ static __forceinline__ bool valid(int wave, int q) { return wave + q * NW < ITEMS; }
    // the same with the test folded away for full rounds of items (wave < NW always, so ITEMS >= (q+1)*NW needs none):
    // the guarded code is then unconditional and the compiler can count its loads / stores for later s_waitcnt's.
    // Used by the forward kernel only: the <D=128, 16-row slice> backward instantiation computed a wrong stream
    // gradient with it on the GPU (not on the emulation) -- unexplained, so the backward keeps the runtime test.
    __device__ static __forceinline__ bool valid_fast(int wave, int q) { return ITEMS >= (q + 1) * NW || wave + q * NW < ITEMS; }
    __device__ static __forceinline__ int nt(int wave, int q) { return (wave + q * NW) / MGROUPS; }
    __device__ static __forceinline__ int mg(int wave, int q) { return (wave + q * NW) % MGROUPS; }
};

// Pipelined GEMM stage  Y[LP][NTILES*16] = X W^T  (W global, [N][ldw]).  Usage:
//     StageXwT<...> g;  g.prefetch(W, ldw, t);  __syncthreads();  g.run(Xs, lda, t, epi);
// prefetch() issues the weight loads of this wave's first item BEFORE the barrier that publishes X;
// run() double-buffers the following items.
template <int K, int MT, int MG, int NW, int NTILES>
struct StageXwT {
    static constexpr int MGROUPS = MT / MG;
    static constexpr int ITEMS = NTILES * MGROUPS;
    static constexpr int PER_WAVE = (ITEMS + NW - 1) / NW;
    float4 bf[2][K / 16];
    float bv[2];                    // bias of the item's output column, fetched with its weight fragment
    const float* W;
    const float* bias;
    int ldw;
    __device__ __forceinline__ void fetch(int q, const Thr& t) {
        const int item = t.wave + q * NW;
        if (ITEMS >= (q + 1) * NW || item < ITEMS) {
            frag_xwT_fetch<K>(bf[q & 1], W + (size_t)((item / MGROUPS) * 16 + t.i) * ldw, t);
            bv[q & 1] = bias != nullptr ? bias[(item / MGROUPS) * 16 + t.i] : 0.f;
        }
    }
    // bias_ (optional, indexed by output column): added to every value handed to the epilogue; loading it here keeps
    // its L2 round trip off the tail of the item's MFMA chain
    __device__ __forceinline__ void prefetch(const float* __restrict__ W_, int ldw_, const Thr& t, const float* __restrict__ bias_ = nullptr) {
        W = W_;
        ldw = ldw_;
        bias = bias_;
        fetch(0, t);
    }
    // wait for the prefetched first fragment NOW (call before issuing stores: CDNA4's vmcnt also counts
    // stores, so a later wait for this fragment would sit behind their acknowledgements)
    __device__ __forceinline__ void retire() {
#pragma unroll
        for (int s = 0; s < K / 16; ++s) retire4(bf[0][s]);
        DTQN_ASM_KEEP(bv[0]);
    }
    template <typename Epi>
    __device__ __forceinline__ void run(const float* Xs, int lda, const Thr& t, Epi epi) {
#pragma unroll
        for (int q = 0; q < PER_WAVE; ++q) {
            if (q + 1 < PER_WAVE) fetch(q + 1, t);
            const int item = t.wave + q * NW;
            if (ITEMS >= (q + 1) * NW || item < ITEMS) {
                const int nt = item / MGROUPS, mg = item - nt * MGROUPS;
                f32x4 acc[MG];
#pragma unroll
                for (int m = 0; m < MG; ++m) acc[m] = zero4();
                frag_xwT_mma<K, MG>(Xs + mg * MG * 16 * lda, lda, bf[q & 1], t, acc);
#pragma unroll
                for (int m = 0; m < MG; ++m)
#pragma unroll
                    for (int r = 0; r < 4; ++r) epi((mg * MG + m) * 16 + t.kq * 4 + r, nt * 16 + t.i, acc[m][r] + bv[q & 1]);
            }
        }
    }
};

// Same for  dX[LP][KTILES*16] = dY W  (contraction over the NN rows of W).
// KS: the contraction of an item is cut into KS parts of NN / KS rows, each with its own (double-buffered) fragment: at
// NN = 128 a whole-contraction fragment pair is 64 registers, which (with everything else the D = 128 backward keeps live)
// ends in scratch; two half fragments are 32.
template <int NN, int MT, int MG, int NW, int KTILES, int KS = (NN > 64 ? 2 : 1)>
struct StageDyW {
    static constexpr int MGROUPS = MT / MG;
    static constexpr int ITEMS = KTILES * MGROUPS;
    static constexpr int PER_WAVE = (ITEMS + NW - 1) / NW;
    static constexpr int NP = NN / KS;              // contraction rows per part
    static constexpr int STEPS = PER_WAVE * KS;
    float bf[2][NP / 4];
    const float* W;
    int ldw;
    __device__ __forceinline__ void fetch(int st, const Thr& t) {
        const int q = st / KS, h = st - q * KS;
        const int item = t.wave + q * NW;
        if (ITEMS >= (q + 1) * NW || item < ITEMS)
            frag_dyw_fetch<NP>(bf[st & 1], W + (size_t)(h * NP) * ldw + (item / MGROUPS) * 16 + t.i, ldw, t);
    }
    __device__ __forceinline__ void prefetch(const float* __restrict__ W_, int ldw_, const Thr& t) {
        W = W_;
        ldw = ldw_;
        fetch(0, t);
    }
    __device__ __forceinline__ void retire() {
#pragma unroll
        for (int q = 0; q < NP / 4; ++q) DTQN_ASM_KEEP(bf[0][q]);
    }
    template <typename Epi>
    __device__ __forceinline__ void run(const float* dYs, int lda, const Thr& t, Epi epi) {
        run(dYs, lda, t, [](int, int) {}, epi);
    }
    // pre(kt, mg) runs before the MFMAs of each item: the place to issue the loads its epilogue needs
    template <typename Pre, typename Epi>
    __device__ __forceinline__ void run(const float* dYs, int lda, const Thr& t, Pre pre, Epi epi) {
#pragma unroll
        for (int q = 0; q < PER_WAVE; ++q) {
            const int item = t.wave + q * NW;
            const bool mine = ITEMS >= (q + 1) * NW || item < ITEMS;
            const int kt = item / MGROUPS, mg = item - kt * MGROUPS;
            f32x4 acc[MG];
#pragma unroll
            for (int m = 0; m < MG; ++m) acc[m] = zero4();
#pragma unroll
            for (int h = 0; h < KS; ++h) {
                const int st = q * KS + h;
                if (st + 1 < STEPS) fetch(st + 1, t);
                if (mine) {
                    if (h == 0) pre(kt, mg);
                    frag_dyw_mma<NP, MG>(dYs + mg * MG * 16 * lda + h * NP, lda, bf[st & 1], t, acc);
                }
            }
            if (mine) {
#pragma unroll
                for (int m = 0; m < MG; ++m)
#pragma unroll
                    for (int r = 0; r < 4; ++r) epi((mg * MG + m) * 16 + t.kq * 4 + r, kt * 16 + t.i, acc[m][r]);
            }
        }
    }
};

// StageDyW for ONE 16-row tile whose KTILES column tiles are half as many as the workgroup has waves: wave w takes tile w % KTILES and
// half w / KTILES of the NN contraction rows (half the fragment: NN / 8 registers, NN / 8 loads), the upper waves hand their sums to
// the lower ones through `scratch` (KTILES * 256 floats of LDS, free for the duration of run()), which then run the epilogue.
// run() contains one __syncthreads(): every wave calls it.
template <int NN, int NW, int KTILES>
struct StageDyWSplit {
    static_assert(2 * KTILES == NW && NN % 32 == 0, "two waves per column tile");
    float bf[NN / 8];
    __device__ __forceinline__ void prefetch(const float* __restrict__ W, int ldw, const Thr& t) {
        const int item = t.wave % KTILES, kh = t.wave / KTILES;
        frag_dyw_fetch<NN / 2>(bf, W + (size_t)(kh * (NN / 2)) * ldw + item * 16 + t.i, ldw, t);
    }
    __device__ __forceinline__ void retire() {
#pragma unroll
        for (int q = 0; q < NN / 8; ++q) DTQN_ASM_KEEP(bf[q]);
    }
    template <typename Pre, typename Epi>
    __device__ __forceinline__ void run(const float* dYs, int lda, const Thr& t, float* scratch, Pre pre, Epi epi) {
        const int item = t.wave % KTILES, kh = t.wave / KTILES;
        f32x4 acc[1] = {zero4()};
        if (kh == 0) pre(item, 0);
        frag_dyw_mma<NN / 2, 1>(dYs + kh * (NN / 2), lda, bf, t, acc);
        if (kh == 1) {
#pragma unroll
            for (int r = 0; r < 4; ++r) scratch[(item * 4 + r) * 64 + t.lane] = acc[0][r];
        }
        __syncthreads();
        if (kh == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) epi(t.kq * 4 + r, item * 16 + t.i, acc[0][r] + scratch[(item * 4 + r) * 64 + t.lane]);
        }
    }
    template <typename Epi>
    __device__ __forceinline__ void run(const float* dYs, int lda, const Thr& t, float* scratch, Epi epi) {
        run(dYs, lda, t, scratch, [](int, int) {}, epi);
    }
};

// ------------------------------------------------------------------------------------------
// LayerNorm over D of every row of a [LP][ld] LDS tile (eps 1e-5, biased variance;
// torch.nn.LayerNorm as used at dtqn/networks/transformer.py:28-29).  4 lanes per row.
// Optionally records (mean, rstd) per row to st_out[row*2..] (global).
// ------------------------------------------------------------------------------------------
// LPT: compile-time upper bound of LP (rows): with fewer rows more lanes share a row, every lane stays busy and the
// stores of the pass are unconditional (the compiler can then count them for later s_waitcnt's)
// SAVE = false: st_out / save_in / save_out are ignored at compile time (no conditional stores in the instruction stream)
// PAD (width-padded networks, DtqnNet.d_real): the statistics run over the first d_real columns; the columns behind them hold zeros
// (and get zeros back: their gamma / beta are zero).
// WTS: the save_in / save_out records leave as write-through (sc1) 16-byte stores (see rec_tile_store)
template <int D, int NW, int LPT = DTQN_MAX_LP, bool SAVE = true, bool PAD = false, bool WTS = false>
__device__ __forceinline__ void layernorm_rows(const float* src, float* dst, int ld, int LP,
                                               const float* __restrict__ gamma, const float* __restrict__ beta,
                                               float* __restrict__ st_out, const Thr& t,
                                               float* __restrict__ save_in = nullptr, float* __restrict__ save_out = nullptr,
                                               int ld_dst = 0, int d_real = D) {  // ld_dst: leading dim of dst when it differs from src's
    const float inv_d = PAD ? 1.0f / (float)d_real : (1.0f / D);
    constexpr int THREADS = NW * 64;
    if (ld_dst == 0) ld_dst = ld;
    constexpr int LPR = (THREADS / LPT) < (D / 4) ? (THREADS / LPT) : (D / 4);                  // lanes per row (4, 8 or 16)
    constexpr int NV = D / (4 * LPR);                                                           // float4 chunks per lane
    constexpr int ROWS = THREADS / LPR;
    for (int base = 0; base < LP; base += ROWS) {
        const int row = base + t.tid / LPR, part = t.tid % LPR;
        const bool valid = (ROWS <= LPT && LPT % ROWS == 0) ? true : row < LP;   // LPT rows exactly tiled by the lanes: no tail
        const float* sp = src + (valid ? row : 0) * ld + part * 4;
        float4 v[NV];
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            v[j] = ld4(sp + 4 * LPR * j);
            sum += (v[j].x + v[j].y) + (v[j].z + v[j].w);
        }
        if (SAVE && save_in != nullptr && valid) {   // dense [LP][D] record of the LN input
            if constexpr (WTS) {
                const DtqnRsrc rs = DTQN_XCH_RSRC(save_in, LP * D * 4);
#pragma unroll
                for (int j = 0; j < NV; ++j) dtqn_xch_store4(rs, (row * D + part * 4 + 4 * LPR * j) * 4, v[j]);
            } else {
#pragma unroll
                for (int j = 0; j < NV; ++j) st4(save_in + (size_t)row * D + part * 4 + 4 * LPR * j, v[j]);
            }
        }
#pragma unroll
        for (int m = 1; m < LPR; m <<= 1) sum += __shfl_xor(sum, m);
        const float mean = sum * inv_d;
        float sq = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            float a = v[j].x - mean, b = v[j].y - mean, c = v[j].z - mean, d = v[j].w - mean;
            if constexpr (PAD) {
                const int c0 = part * 4 + 4 * LPR * j;
                a = c0 < d_real ? a : 0.f; b = c0 + 1 < d_real ? b : 0.f; c = c0 + 2 < d_real ? c : 0.f; d = c0 + 3 < d_real ? d : 0.f;
            }
            sq += (a * a + b * b) + (c * c + d * d);
        }
#pragma unroll
        for (int m = 1; m < LPR; m <<= 1) sq += __shfl_xor(sq, m);
        const float rstd = 1.0f / sqrtf(sq * inv_d + 1e-5f);
        if (valid) {
            float* dp = dst + row * ld_dst + part * 4;
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                const float4 g = ld4(gamma + part * 4 + 4 * LPR * j), b = ld4(beta + part * 4 + 4 * LPR * j);
                float4 o;
                o.x = (v[j].x - mean) * rstd * g.x + b.x;
                o.y = (v[j].y - mean) * rstd * g.y + b.y;
                o.z = (v[j].z - mean) * rstd * g.z + b.z;
                o.w = (v[j].w - mean) * rstd * g.w + b.w;
                st4(dp + 4 * LPR * j, o);
                if (SAVE && save_out != nullptr) {
                    if constexpr (WTS) dtqn_xch_store4(DTQN_XCH_RSRC(save_out, LP * D * 4), (row * D + part * 4 + 4 * LPR * j) * 4, o);
                    else st4(save_out + (size_t)row * D + part * 4 + 4 * LPR * j, o);
                }
            }
            if (SAVE && st_out != nullptr && part == 0) {
                st_out[row * 2 + 0] = mean;
                st_out[row * 2 + 1] = rstd;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// Dropout (train-mode forwards only; dtqn/networks/dtqn.py:105,196 on embedding + position, transformer.py:34 on the
// attention probabilities, transformer.py:41 on the FFN output).  Keep masks are a counter-based hash of
// (seed, update step, pass, sequence, site, layer, element): nothing is stored, the backward recomputes them.
// torch's own Philox stream cannot be reproduced (it depends on launch geometry); the distribution is the same:
// keep with probability 1 - p, scale kept values by 1 / (1 - p).
// ------------------------------------------------------------------------------------------
struct Drop {
    uint32_t thresh;      // drop iff hash < thresh; 0 = dropout off
    float scale;          // 1 / (1 - p)
    uint32_t seed, step;
    uint32_t salt;        // (pass << 20) | sequence
};
enum { DROP_EMB = 0, DROP_ATTN = 1, DROP_FFN = 2, DROP_BAG = 3 };    // (the tag of a mask key is site + 4 * layer; DROP_BAG has no layer)
__device__ __forceinline__ Drop drop_off() { return Drop{0u, 1.0f, 0u, 0u, 0u}; }
__device__ __forceinline__ bool drop_keep(const Drop& d, int site, int layer, uint32_t idx) {
    unsigned long long z = ((unsigned long long)d.seed << 32) | (unsigned long long)d.step;
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= ((unsigned long long)d.salt << 40) | ((unsigned long long)(site + 4 * layer) << 32) | (unsigned long long)idx;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return (uint32_t)(z >> 32) >= d.thresh;
}
__device__ __forceinline__ float drop_apply(const Drop& d, int site, int layer, uint32_t idx, float v) {
    return d.thresh == 0u ? v : (drop_keep(d, site, layer, idx) ? v * d.scale : 0.f);
}
// element index of an attention probability: (head, query row, key row), rows of up to 256
__device__ __forceinline__ uint32_t drop_attn_idx(int h, int trow, int s) { return ((uint32_t)h << 16) | ((uint32_t)trow << 8) | (uint32_t)s; }

// Causal attention work is triangular: the 64-item block of query rows [8g, 8g+8) costs ~8(g+1) key steps
// (the reverse for the key-row pass of the backward).  Waves w and w + NW/2 share a SIMD (waves are dealt to
// SIMDs cyclically), so block b of the natural order is handed to the wave that pairs the most expensive
// block with the cheapest one: wave w < NW/2 takes the w-th most expensive block, wave w + NW/2 the w-th
// cheapest.  Only used when all items fit one round (blocks <= NW); returns the block this wave processes,
// or -1 for none.  `descending` = cost falls with the block index (key-row pass).
template <int NW>
__device__ __forceinline__ int balanced_block(int wave, int nblocks, int nlive, bool descending) {
    // live blocks 0..nlive-1 carry work (cost rising with index unless `descending`), blocks >= nlive are padding
    int rank;                                   // 0 = most expensive
    if (wave < NW / 2) rank = wave;
    else rank = NW - 1 - (wave - NW / 2);
    if (rank >= nblocks) return -1;
    if (rank >= nlive) return rank;             // padding block (still has to zero its outputs)
    return descending ? rank : nlive - 1 - rank;
}

// ------------------------------------------------------------------------------------------
// Causal multi-head self-attention on a [LP][ld] LDS tile laid out [q | k | v] (blocks D columns apart),
// on the f32 matrix core.  Work item = (head h, 16-row query tile ti), dealt to the waves head-fastest
// (H % NW == 0: every wave owns whole heads, so the causal triangle is balanced by construction).
//   scores are produced TRANSPOSED, S^T[s][t] = K[s,:] . Q[t,:] (A = K tile, B = Q tile): in the MFMA output
//   layout lane (i, kq) then holds keys s = kq*4 + r of query t = i, so that
//     * the softmax statistics of query t are a reduction over registers r and lane groups kq (2 shuffles),
//     * P^T is directly the B operand of O^T[c][t] += V^T[c][s] P^T[s][t] (no LDS round trip, no transpose),
//     * the running rescale exp(m_old - m_new) of query t applies to this lane's own accumulator column.
// Online softmax over the key tiles tj <= ti; the output o[t, h] OVERWRITES q[t, h] (only this item reads it);
// optional log-sum-exp per (h, t) for the backward pass.
//   torch.nn.MultiheadAttention as called at transformer.py:64-70: q scaled by hd^-0.5, float
//   additive mask = strictly-upper-triangular -inf  => keys s <= t only.
// ------------------------------------------------------------------------------------------
// One chunk of NT (<= 4) consecutive key tiles of one (head, query-tile) item; DIAG: the last tile of the chunk is
// the diagonal one (keys s > t masked).  Scores live in the log2 domain (q is pre-scaled by hd^-0.5 * log2 e), so
// every probability is one v_exp_f32.  The NT score tiles are independent MFMA chains and share ONE pair of
// max / sum reductions across the four lane groups.
template <int HD, int NT, bool DIAG>
__device__ __forceinline__ void attention_forward_chunk(const float* kbase, const float* vbase, int ld, int s_first, int trow,
                                                        const float (&qf)[HD / 4], float& m, float& l,
                                                        f32x4 (&acc)[(HD + 15) / 16][2], bool rescale, const Thr& t,
                                                        const Drop& dr, int layer, int h) {
    constexpr int KS = HD / 4, CT = (HD + 15) / 16;
    f32x4 st[NT];
#pragma unroll
    for (int u = 0; u < NT; ++u) {
        st[u] = zero4();
        const float* kp = kbase + (s_first + u * 16 + t.i) * ld + t.kq * KS;
#pragma unroll
        for (int s = 0; s < KS; ++s) st[u] = mfma16(kp[s], qf[s], st[u]);
    }
    // st[u][r] = S^T[s_first + u*16 + kq*4 + r][t]
    if (DIAG) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (s_first + (NT - 1) * 16 + t.kq * 4 + r > trow) st[NT - 1][r] = -INFINITY;
    }
    float tmax = st[0][0];
#pragma unroll
    for (int u = 0; u < NT; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) tmax = fmaxf(tmax, st[u][r]);
    tmax = kq_max(tmax);
    const float mn = fmaxf(m, tmax);             // finite: the first key of the chunk is never masked
    float ps = 0.f;
#pragma unroll
    for (int u = 0; u < NT; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) { st[u][r] = DTQN_EXP2(st[u][r] - mn); ps += st[u][r]; }
    ps = kq_sum(ps);
    if (rescale) {
        const float corr = DTQN_EXP2(m - mn);
        l = l * corr + ps;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int e = 0; e < 4; ++e) { acc[ct][0][e] *= corr; acc[ct][1][e] *= corr; }
    } else {
        l = ps;
    }
    m = mn;
    if (dr.thresh != 0u) {     // attention-probability dropout: the row sum above is of the undropped probabilities
#pragma unroll
        for (int u = 0; u < NT; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                st[u][r] = drop_keep(dr, DROP_ATTN, layer, drop_attn_idx(h, trow, s_first + u * 16 + t.kq * 4 + r)) ? st[u][r] * dr.scale : 0.f;
    }
    // O^T[c][t] += V^T[c][s] P^T[s][t]: A = V[s][ct*16 + i] (rows c >= HD of O^T are never stored: any in-range
    // column will do there), B = p
#pragma unroll
    for (int u = 0; u < NT; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float* vp = vbase + (s_first + u * 16 + t.kq * 4 + r) * ld;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                const int c = ct * 16 + t.i;
                acc[ct][r & 1] = mfma16(vp[c < HD ? c : 0], st[u][r], acc[ct][r & 1]);
            }
        }
}

template <int HD, int NW>
__device__ __forceinline__ void attention_forward_mfma(float* Ws, int ld, int D, int H, int LP, int n,
                                                       float* __restrict__ lse_out, const Thr& t, int row0 = 0, int lse_ld = 0,
                                                       const Drop& dr = Drop{0u, 1.0f, 0u, 0u, 0u}, int layer = 0, int head0 = 0,
                                                       float hd_eff = (float)HD) {     // hd_eff: see attention_forward
    if (lse_ld == 0) lse_ld = LP;               // query rows [row0, row0 + LP), row0 a multiple of 16
    constexpr int KS = HD / 4;                  // MFMA steps of the score contraction (4 columns of q/k per step)
    constexpr int CT = (HD + 15) / 16;          // 16-row tiles of O^T (rows = head columns c)
    const float scale = 1.4426950408889634f / sqrtf(hd_eff);           // hd^-0.5 * log2(e)
    const int MT = LP / 16;
    const int last_tile = (n - 1) / 16;         // query tiles beyond it hold only pad rows
    for (int item = t.wave; item < H * MT; item += NW) {
        const int h = item % H, ti = row0 / 16 + item / H;
        const int t0 = ti * 16, trow = t0 + t.i;
        float* qbase = Ws + h * HD;
        if (ti > last_tile) {                   // pad rows: o = 0, lse = 0
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                const int c = ct * 16 + t.kq * 4;
                if (c < HD) st4(qbase + trow * ld + c, make_float4(0.f, 0.f, 0.f, 0.f));
            }
            if (lse_out != nullptr && t.kq == 0) lse_out[h * lse_ld + trow] = 0.f;
            continue;
        }
        const float* kbase = Ws + D + h * HD;
        const float* vbase = kbase + D;
        float qf[KS];                           // B operand: Q[t0 + i][kq*KS + s], pre-scaled
#pragma unroll
        for (int s = 0; s < KS; ++s) qf[s] = qbase[trow * ld + t.kq * KS + s] * scale;
        f32x4 acc[CT][2];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) { acc[ct][0] = zero4(); acc[ct][1] = zero4(); }
        float m = -INFINITY, l = 0.f;
        int tc0 = 0;
        for (; tc0 + 4 <= ti; tc0 += 4)
            attention_forward_chunk<HD, 4, false>(kbase, vbase, ld, tc0 * 16, trow, qf, m, l, acc, tc0 > 0, t, dr, layer, head0 + h);
        switch (ti - tc0) {                     // the remaining 1..4 tiles end on the diagonal
            case 0: attention_forward_chunk<HD, 1, true>(kbase, vbase, ld, tc0 * 16, trow, qf, m, l, acc, tc0 > 0, t, dr, layer, head0 + h); break;
            case 1: attention_forward_chunk<HD, 2, true>(kbase, vbase, ld, tc0 * 16, trow, qf, m, l, acc, tc0 > 0, t, dr, layer, head0 + h); break;
            case 2: attention_forward_chunk<HD, 3, true>(kbase, vbase, ld, tc0 * 16, trow, qf, m, l, acc, tc0 > 0, t, dr, layer, head0 + h); break;
            default: attention_forward_chunk<HD, 4, true>(kbase, vbase, ld, tc0 * 16, trow, qf, m, l, acc, tc0 > 0, t, dr, layer, head0 + h); break;
        }
        // lane (i, kq) holds O^T[c = ct*16 + kq*4 + e][t = t0 + i]: four consecutive output columns of row t
        const bool live = trow < n;
        const float inv = live ? 1.0f / l : 0.f;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const int c = ct * 16 + t.kq * 4;
            if (c < HD)
                st4(qbase + trow * ld + c, make_float4((acc[ct][0][0] + acc[ct][1][0]) * inv, (acc[ct][0][1] + acc[ct][1][1]) * inv,
                                                       (acc[ct][0][2] + acc[ct][1][2]) * inv, (acc[ct][0][3] + acc[ct][1][3]) * inv));
        }
        if (lse_out != nullptr && t.kq == 0) lse_out[h * lse_ld + trow] = live ? m * 0.6931471805599453f + __logf(l) : 0.f;
    }
}

// ------------------------------------------------------------------------------------------
// The same attention on the VALU: work item = (query row t, head h), h fastest across lanes so a wave's
// 8 heads x 8 rows read each key row conflict-free and finish within 7 iterations of each other; blocked
// online softmax.  For head_dim 8 half of every 16-wide MFMA tile of O^T would be padding and the softmax
// bookkeeping of the 16x16 score tiles costs as many VALU slots as the dot products they replace: measured
// inside the whole-sequence kernels (cfg 1) this version is 1.5 % faster per update, so narrow heads use it;
// heads of 16 columns and more use the matrix-core version above (tools/microbench/attn_bench.hip).
// ------------------------------------------------------------------------------------------
template <int HD, int NW>
__device__ __forceinline__ void attention_forward_valu(float* Ws, int ld, int D, int H, int LP, int n,
                                                  float* __restrict__ lse_out, const Thr& t, int row0 = 0, int lse_ld = 0,
                                                  const Drop& dr = Drop{0u, 1.0f, 0u, 0u, 0u}, int layer = 0, int head0 = 0,
                                                  float hd_eff = (float)HD) {
    // query rows [row0, row0 + LP) of the tile (a row slice of the sequence when row0 > 0); keys 0 .. row
    if (lse_ld == 0) lse_ld = LP;
    const float scale = 1.0f / sqrtf(hd_eff);
    const int nblocks = (LP * H + 63) / 64;
    const bool one_round = nblocks <= NW && (64 % H) == 0;
    const int nloc = n - row0 < 0 ? 0 : (n - row0 > LP ? LP : n - row0);
    const int nlive = (nloc * H + 63) / 64;
    for (int it0 = t.tid; it0 < (one_round ? NW * 64 : LP * H); it0 += NW * 64) {
        int item = it0;
        if (one_round) {
            const int blk = balanced_block<NW>(t.wave, nblocks, nlive, false);
            if (blk < 0) continue;
            item = blk * 64 + t.lane;
            if (item >= LP * H) continue;
        }
        const int rl = item / H, h = item - rl * H;
        const int row = row0 + rl;
        float* qp = Ws + row * ld + h * HD;
        if (row >= n) {
#pragma unroll
            for (int c = 0; c < HD; ++c) qp[c] = 0.f;
            if (lse_out != nullptr) lse_out[h * lse_ld + row] = 0.f;
            continue;
        }
        float q[HD], acc[HD];
#pragma unroll
        for (int c = 0; c < HD; c += 4) {
            const float4 x = ld4(qp + c);
            q[c] = x.x * scale; q[c + 1] = x.y * scale; q[c + 2] = x.z * scale; q[c + 3] = x.w * scale;
            acc[c] = acc[c + 1] = acc[c + 2] = acc[c + 3] = 0.f;
        }
        // blocked online softmax: KB keys per step give KB independent dot products / exponentials (one
        // serial chain per key would leave the two waves of a SIMD nothing to overlap) and one rescale per block
        constexpr int KB = 4;
        float m = -INFINITY, l = 0.f;
        const float* kbase = Ws + D + h * HD;
        for (int s0 = 0; s0 <= row; s0 += KB) {
            float sc[KB];
#pragma unroll
            for (int j = 0; j < KB; ++j) {
                const int s = s0 + j <= row ? s0 + j : row;            // clamp: the duplicate is masked below
                const float* kp = kbase + s * ld;
                float a = 0.f;
#pragma unroll
                for (int c = 0; c < HD; c += 4) {
                    const float4 k = ld4(kp + c);
                    a = fmaf(q[c], k.x, a); a = fmaf(q[c + 1], k.y, a); a = fmaf(q[c + 2], k.z, a); a = fmaf(q[c + 3], k.w, a);
                }
                sc[j] = s0 + j <= row ? a : -INFINITY;
            }
            float mb = sc[0];
#pragma unroll
            for (int j = 1; j < KB; ++j) mb = fmaxf(mb, sc[j]);
            const float mn = fmaxf(m, mb);
            const float corr = __expf(m - mn);
            float p[KB], ps = 0.f;
#pragma unroll
            for (int j = 0; j < KB; ++j) { p[j] = __expf(sc[j] - mn); ps += p[j]; }
            l = l * corr + ps;
            if (dr.thresh != 0u) {     // attention-probability dropout (the row sum is of the undropped probabilities)
#pragma unroll
                for (int j = 0; j < KB; ++j)
                    if (s0 + j <= row) p[j] = drop_keep(dr, DROP_ATTN, layer, drop_attn_idx(head0 + h, row, s0 + j)) ? p[j] * dr.scale : 0.f;
            }
#pragma unroll
            for (int c = 0; c < HD; ++c) acc[c] *= corr;
#pragma unroll
            for (int j = 0; j < KB; ++j) {
                const int s = s0 + j <= row ? s0 + j : row;
                const float* vp = kbase + s * ld + D;
#pragma unroll
                for (int c = 0; c < HD; c += 4) {
                    const float4 v = ld4(vp + c);
                    acc[c] = fmaf(p[j], v.x, acc[c]); acc[c + 1] = fmaf(p[j], v.y, acc[c + 1]);
                    acc[c + 2] = fmaf(p[j], v.z, acc[c + 2]); acc[c + 3] = fmaf(p[j], v.w, acc[c + 3]);
                }
            }
            m = mn;
        }
        const float inv = 1.0f / l;
#pragma unroll
        for (int c = 0; c < HD; c += 4) st4(qp + c, make_float4(acc[c] * inv, acc[c + 1] * inv, acc[c + 2] * inv, acc[c + 3] * inv));
        if (lse_out != nullptr) lse_out[h * lse_ld + row] = m + __logf(l);
    }
}


#ifndef DTQN_ATTN_MFMA_MIN_HD
#define DTQN_ATTN_MFMA_MIN_HD 16
#endif
constexpr int kAttnMfmaMinHeadDim = DTQN_ATTN_MFMA_MIN_HD;
template <int HD, int NW, bool MFMA = (HD >= kAttnMfmaMinHeadDim)>
__device__ __forceinline__ void attention_forward(float* Ws, int ld, int D, int H, int LP, int n,
                                                  float* __restrict__ lse_out, const Thr& t, int row0 = 0, int lse_ld = 0,
                                                  const Drop& dr = Drop{0u, 1.0f, 0u, 0u, 0u}, int layer = 0, int head0 = 0,
                                                  float hd_eff = (float)HD) {
    // head0: global index of the tile's first head (the row-block kernels hold one head per workgroup): the keep masks of
    // the attention-probability dropout are keyed by the global head
    // hd_eff: the head width the softmax scale is taken from -- HD itself (a compile-time constant after inlining), or the caller's real
    // head width when the heads are zero-padded to HD columns (DtqnNet.hd_real)
    if constexpr (MFMA) attention_forward_mfma<HD, NW>(Ws, ld, D, H, LP, n, lse_out, t, row0, lse_ld, dr, layer, head0, hd_eff);
    else attention_forward_valu<HD, NW>(Ws, ld, D, H, LP, n, lse_out, t, row0, lse_ld, dr, layer, head0, hd_eff);
}

// Cooperative copy of a [rows][cols] LDS tile (leading dim ld) to / from a dense global array.
template <int NW>
__device__ __forceinline__ void tile_store(const float* s, int ld, float* __restrict__ g, int rows, int cols, const Thr& t, int gld = 0) {
    const int c4 = cols >> 2;
    if (gld == 0) gld = cols;
    for (int idx = t.tid; idx < rows * c4; idx += NW * 64) {
        const int r = idx / c4, c = (idx - r * c4) * 4;
        st4(g + (size_t)r * gld + c, ld4(s + r * ld + c));
    }
}
// ... the same with write-through (sc1) 16-byte stores: the tile is on its way to memory while the kernel still computes, instead of
// sitting dirty in this XCD's L2 until the kernel boundary writes it back (records another launch reads)
template <int NW, bool WT>
__device__ __forceinline__ void rec_tile_store(const float* s, int ld, float* __restrict__ g, int rows, int cols, const Thr& t, int gld = 0) {
    if constexpr (WT) {
        const int c4 = cols >> 2;
        if (gld == 0) gld = cols;
        const DtqnRsrc rs = DTQN_XCH_RSRC(g, (rows - 1) * gld * 4 + cols * 4);
        for (int idx = t.tid; idx < rows * c4; idx += NW * 64) {
            const int r = idx / c4, c = (idx - r * c4) * 4;
            dtqn_xch_store4(rs, (r * gld + c) * 4, ld4(s + r * ld + c));
        }
    } else {
        tile_store<NW>(s, ld, g, rows, cols, t, gld);
    }
}
template <int NW>
__device__ __forceinline__ void tile_load(float* s, int ld, const float* __restrict__ g, int rows, int cols, const Thr& t) {
    const int c4 = cols >> 2;
    for (int idx = t.tid; idx < rows * c4; idx += NW * 64) {
        const int r = idx / c4, c = (idx - r * c4) * 4;
        st4(s + r * ld + c, ld4(g + (size_t)r * cols + c));
    }
}

// Workgroup -> (sequence, position in the sequence's hand-over order) for launches with RS workgroups per sequence that WAIT for each
// other (row slices: a slice spins until the slices it depends on have published their tiles).  Position 0 waits for nobody, position
// k only for positions < k of the same sequence.  Blocks of a launch go to the eight XCDs round-robin, and every XCD places its share
// in order when it has room.  If the RS workgroups of a sequence sat on RS different XCDs (block = sequence * RS + position), an XCD
// could fill up with WAITING workgroups whose producers belong to another XCD -- harmless while one launch owns the chip (everything is
// resident), a deadlock as soon as a second process's launch of the same kind shares the GPU: found with two ranks on one MI355X once
// the forward was 2 B 4 = 256 workgroups (64 per position = two full XCDs; round 4).  So the RS workgroups of a sequence are blocks
// x, x + 8, ..., x + 8 (RS - 1) of a group of 8 sequences: the same XCD whatever the round-robin offset of the launch, in hand-over
// order -- every resident waiter has its producers resident or done, on every XCD, under any mix of launches.  (A last group of fewer
// than 8 sequences keeps the order, not the XCD.)
__device__ __forceinline__ void slice_block_map(int bid, int nseq, int RS, int& seq, int& pos) {
    if (RS == 1) { seq = bid; pos = 0; return; }
    const int g = bid / (8 * RS), u = bid - g * 8 * RS;
    const int rem = nseq - g * 8, w = rem < 8 ? rem : 8;
    pos = u / w;
    seq = g * 8 + (u - pos * w);
}

// splitmix64-style counter hash -> 32 random bits per (seed, step, element, draw)
__device__ __forceinline__ uint32_t hash_u32(uint32_t seed, uint32_t step, uint32_t elem, uint32_t draw) {
    unsigned long long z = ((unsigned long long)seed << 32) ^ ((unsigned long long)step * 0x9E3779B97F4A7C15ull) ^
                           ((unsigned long long)elem << 20) ^ draw;
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return (uint32_t)(z >> 32);
}
// The window draw of ReplayBuffer.sample (replay_buffer.py:141-158) for batch element b of update `step`:
// episode uniform over the finished slots [0, n_valid) minus `exclude`, start uniform on {0 .. max(0, len - L)}.
__device__ __forceinline__ void replay_draw(const int32_t* __restrict__ ep_len, int n_valid, int exclude, int ctx_len,
                                            uint32_t seed, uint32_t step, int b, int& ep, int& start) {
    const bool skip = exclude >= 0 && exclude < n_valid;
    const uint32_t choices = (uint32_t)(n_valid - (skip ? 1 : 0));
    uint32_t e = (uint32_t)(((unsigned long long)hash_u32(seed, step, (uint32_t)b, 0) * choices) >> 32);
    if (skip && (int)e >= exclude) e += 1;
    const int len = ep_len[e];
    const uint32_t span = (uint32_t)(len - ctx_len > 0 ? len - ctx_len : 0) + 1u;
    ep = (int)e;
    start = (int)(((unsigned long long)hash_u32(seed, step, (uint32_t)b, 1) * span) >> 32);
}

// Row-split hand-over of a [rows][cols] tile between the two workgroups of a sequence.  The producer publishes the
// LDS tile and raises the flag; the consumer waits for the flag, pulls the tile into its own LDS and lowers the
// flag again (so the next launch starts from 0).  ADD: accumulate into the destination instead of overwriting.
// The producer always has the LOWER blockIdx of the pair, so it is dispatched no later than its consumer.
template <int NW>
__device__ __forceinline__ void xch_send(const float* s, int ld, float* g, int rows, int cols, int32_t* flag, const Thr& t) {
    const DtqnRsrc rs = DTQN_XCH_RSRC(g, rows * cols * 4);
    const int c4 = cols >> 2;
    for (int idx = t.tid; idx < rows * c4; idx += NW * 64) {
        const int r = idx / c4, c = (idx - r * c4) * 4;
        dtqn_xch_store4(rs, idx * 16, ld4(s + r * ld + c));
    }
    DTQN_WAIT_VMEM();                              // every wave's stores are acknowledged at agent scope ...
    __syncthreads();                               // ... and every wave got here ...
    if (t.tid == 0) DTQN_AGENT_STORE(flag, (int32_t)1);   // ... before the flag is raised
}
template <int NW, bool ADD>
__device__ __forceinline__ void xch_recv(float* s, int ld, const float* g, int rows, int cols, int32_t* flag, const Thr& t) {
    if (t.tid == 0)
        while (DTQN_AGENT_LOAD(flag) == 0) DTQN_SPIN_PAUSE();
    __syncthreads();
    const DtqnRsrc rs = DTQN_XCH_RSRC(g, rows * cols * 4);
    const int c4 = cols >> 2;
    for (int idx = t.tid; idx < rows * c4; idx += NW * 64) {
        const int r = idx / c4, c = (idx - r * c4) * 4;
        const float4 v = dtqn_xch_load4(rs, idx * 16);
        float* d = s + r * ld + c;
        if (ADD) {
            const float4 o = ld4(d);
            st4(d, make_float4(o.x + v.x, o.y + v.y, o.z + v.z, o.w + v.w));
        } else {
            st4(d, v);
        }
    }
    __syncthreads();
    if (t.tid == 0) DTQN_AGENT_STORE(flag, (int32_t)0);
}

// Waves per workgroup for a given network.  More waves = more matrix-core issue slots per CU for the one
// sequence a workgroup owns (the small-batch regime is latency-bound), fewer = more registers per wave.
// DTQN_WAVES in the environment overrides the default (tuning / tests).
static inline int waves_for(const DtqnNet& net) {
    if (dtqn_ws_lite(net.tiled, net.d_model, net.head_dim, net.d_real)) return 8;      // eight-wave kernels only
    int nw = 8;                        // the default wave count of the network's instantiation (dtqn_limits.h)
    dtqn_ws_pick(net.d_model, net.head_dim, net.lp / 16, &nw);
    if (nw == 0) nw = 8;
    const char* e = getenv("DTQN_WAVES");
    if (e != nullptr) {
        const int v = atoi(e);
        if (v == 4 || v == 8 || v == 16) nw = v;
    }
    return nw;
}

// A dense [ROWS][COLS] global tile parked in registers: load() issues the (coalesced, 16 B/lane) global
// loads one or more stages before the data is needed; to_lds() drops it into an LDS tile later.
template <int NW, int ROWS, int COLS>
struct TileRegs {
    static constexpr int C4 = COLS / 4;
    static constexpr int N = (ROWS * C4 + NW * 64 - 1) / (NW * 64);
    float4 v[N];
    __device__ __forceinline__ void load(const float* __restrict__ g, int gld, const Thr& t) {
#pragma unroll
        for (int k = 0; k < N; ++k) {
            const int idx = t.tid + k * NW * 64;
            if (idx < ROWS * C4) {
                const int r = idx / C4, c = (idx - r * C4) * 4;
                v[k] = ld4(g + (size_t)r * gld + c);
            }
        }
    }
    __device__ __forceinline__ void to_lds(float* s, int ld, const Thr& t) {
#pragma unroll
        for (int k = 0; k < N; ++k) {
            const int idx = t.tid + k * NW * 64;
            if (idx < ROWS * C4) {
                const int r = idx / C4, c = (idx - r * C4) * 4;
                st4(s + r * ld + c, v[k]);
            }
        }
    }
};

__device__ __forceinline__ const float* layer_theta(const DtqnNet& net, const float* theta, int l) {
    return theta + net.off_layer0 + (size_t)l * net.layer_stride;
}

}  // namespace dtqn
