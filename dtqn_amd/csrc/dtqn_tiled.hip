// Row-block tiled DTQN path for contexts / widths that do not fit one workgroup's LDS
// (BASELINE configs 4 and 5: L = 128 / 256, D = 128 / 256): inference forward AND the TD training pass.
// Same arithmetic as dtqn_forward.hip / dtqn_backward.hip (DTQN.forward, dtqn/networks/dtqn.py:158-218;
// DtqnAgent.train, dtqn/agents/dtqn.py:199-256), different decomposition:
//   * every tensor lives in the per-sequence activation / gradient RECORDS of include/dtqn_hip.h (the same
//     fields the whole-sequence kernels save), token-major [LPB][cols], LPB = L rounded up to 64;
//   * every projection (forward X W^T, backward dY W) is a matrix-core GEMM over 64-row blocks: the block's
//     operand tile staged in LDS, 128 output columns per workgroup, the contraction walked in chunks with
//     register accumulation; residual adds, ReLU, the ReLU ballots and the gradient masks are GEMM epilogues;
//   * attention (forward and backward) runs per (sequence, head) with that head's tiles held in LDS;
//   * LayerNorm (forward and backward) are row-wise kernels.
// The weight gradients, the reduction and Adam are the SAME kernels as on the whole-sequence path
// (dtqn_wgrad.hip, dtqn_optim.hip): they only see the records.
// Covers post-LN and identity-reordered layers, residual and GRU gates (the GRU gate as two-operand GEMMs with the gate
// arithmetic in their epilogues); this is also where variants whose whole-sequence tile set exceeds LDS run
// (identity or GRU at D >= 128).
#include <cstdio>
#include "dtqn_device.hpp"
#include "dtqn_bwd_device.hpp"
#include "dtqn_gru.hpp"

#include "dtqn_frag16.hpp"
#include "dtqn_wpack.hpp"

namespace dtqn {

// Start skew of a launch that runs a round and a half (round 6).  Where the last round is half empty (BASELINE config 4 forward: 768
// 64-row workgroups on 2 x 256 slots), WHICH slots the hardware dispatcher gives the last 256 decides the kernel's time: one each on
// 256 compute units that have just run a pair (tl_ffn 114 us), or two each on the 128 units whose pair happened to finish first
// (141 us) -- the traces show both, at random, launch by launch, because the two workgroups of a unit start and finish together.
// The workgroups that fill the SECOND slot of every unit (dispatch order 256 .. 511) therefore start `ticks` 100-MHz ticks late (6 us):
// their partners have the unit's matrix pipe to themselves meanwhile (the delay is not lost), the two slots of a unit no longer free
// up together, and the last half round lands one per unit every time: 143 -> 129 us in tools/microbench/ffn_bench.hip, against 134 us
// for cutting the launch into its whole rounds and the rest.
__device__ __forceinline__ void tl_start_skew(int ticks) {
    if (ticks > 0 && blockIdx.x >= 256 && blockIdx.x < 512) {
        const long long t0 = wall_clock64();
        while (wall_clock64() - t0 < ticks) DTQN_SPIN_PAUSE_LONG();
    }
}
// skew of a launch of nblk workgroups on `slots` resident ones: only where the last round is between a quarter and three quarters full
// `ticks`: the kernel's own default -- the delay has to be a fair share of one workgroup's run time (measured at BASELINE config 4, forward
// stage: the 60-us projection kernel 600; the fused layer kernel, 160 us: 0 -> 626 us, 600 -> 574, 1200 -> 558, 2400 -> 551, 4000 -> 575)
static inline int tl_skew_ticks(int nblk, int slots, int ticks = 600, const char* own_env = nullptr) {
    const char* e = own_env != nullptr ? getenv(own_env) : nullptr;
    if (e == nullptr) e = getenv("DTQN_SKEW_TICKS");
    const int rest = nblk % slots;
    // (only between one and two rounds: at two and a half -- config 3's packed launches, 1312 workgroups -- the same delay costs 11 us per forward)
    if (nblk <= slots || nblk >= 2 * slots || slots != 512 || rest * 4 < slots || rest * 4 > 3 * slots) return 0;
    return e != nullptr ? atoi(e) : ticks;
}

constexpr int TNW = 8;                 // waves per workgroup of the GEMM / row-wise kernels
constexpr int TNT = TNW * 64;
constexpr int TROWS = 64;              // rows per block

// One tensor inside the per-sequence records: element (s, row, col) = base[s * stride + row * ld + col]
struct Fld {
    float* base;
    long long stride;
    int ld;
};
static inline Fld fld(float* rec, long long stride, int off, int ld) { return Fld{rec + off, stride, ld}; }
static inline Fld nofld() { return Fld{nullptr, 0, 0}; }
__host__ __device__ __forceinline__ Fld nofld_dev() { return Fld{nullptr, 0, 0}; }
__device__ __forceinline__ float* frow(const Fld& f, int s, int row) { return f.base + (size_t)s * f.stride + (size_t)row * f.ld; }

// ---- packed rows of the sequences nobody reads again (round 6) -------------------------------------------------------------------------
// A TD forward runs three passes; only the first (policy(o), sequences [0, batch)) is read by the backward.  The records stay per sequence
// ([LPB][cols], LPB = L rounded up to 64) because the attention kernels and the backward address them that way, but the row-local kernels
// (q | k | v projection, fused layer tail) need not walk (sequence, 64-row block of its record): for the other two passes their workgroups
// walk the LIVE rows of consecutive sequences, 64 at a time -- BASELINE config 3 (L = 50, LPB = 64): 400 workgroups per pass instead of 512;
// the 14 dead rows of a tile were never part of the loss (dtqn/agents/dtqn.py:215-243 runs over B * history positions).  A row's arithmetic
// does not depend on where in a workgroup it sits (MFMA sums run over k, LayerNorm rows over 8 fixed lanes), so Q is bit-identical to the
// unpacked walk.  Nothing is saved for these passes (no ReLU ballots, no h / s1 / s2 / statistics), which is what keeps this a matter of
// row addresses only.  Conditions (forward_records): 32 <= L < LPB, batch * L a multiple of 64 (a workgroup never mixes the parameter sets
// of two passes), 64-row workgroups, no dropout, no bag.
struct TlPack {
    int n0;                            // workgroups [0, n0) walk (sequence, row block) as ever
    int s0;                            // workgroup n0 + k: live rows [64 k, 64 k + 64) of sequences s0, s0 + 1, ...
    int L;                             // live rows per sequence; 0: no packing
};
struct TlBlk {
    int s, row0;                       // first row of the workgroup: sequence, row in its record
    int L;                             // 0: rows row0 .. row0 + MR - 1 of sequence s;  > 0: packed (rows run on into the next sequences)
};
__device__ __forceinline__ TlBlk tl_blk(int blk, int rpb, int MR, const TlPack& p) {
    TlBlk b;
    if (p.L == 0 || blk < p.n0) { b.s = blk / rpb; b.row0 = (blk - b.s * rpb) * MR; b.L = 0; }
    else { const int v0 = (blk - p.n0) * 64, q = v0 / p.L; b.s = p.s0 + q; b.row0 = v0 - q * p.L; b.L = p.L; }
    return b;
}
// row r of the workgroup: (sequence, row of its record); a packed workgroup of 64 rows spans at most three sequences (L >= 32)
__device__ __forceinline__ void tl_row(const TlBlk& b, int r, int& s, int& row) {
    row = b.row0 + r;
    s = b.s;
    if (b.L > 0) {
        const int k = (row >= b.L ? 1 : 0) + (row >= 2 * b.L ? 1 : 0);
        s += k;
        row -= k * b.L;
    }
}
__device__ __forceinline__ float* frow_b(const Fld& f, const TlBlk& b, int r) {
    int s, row;
    tl_row(b, r, s, row);
    return frow(f, s, row);
}

// Dropout on the row-block path (net.dropout > 0): the keep-mask hash of the whole-sequence kernels (dtqn_device.hpp drop_keep),
// keyed by (seed, step, pass, sequence, site, layer, element).  Sequence s of a launch belongs to pass s / batch; bit p of
// `passes` says whether pass p runs in train mode (TD update: policy(o) and policy(o') do, the target net does not).
struct TlDrop {
    uint32_t thresh;                   // 0: dropout off
    float scale;
    uint32_t seed, step;
    const int32_t* step_counter;       // TD update: step = step_counter[1]; else `step`
    int batch, passes;
    int dw;                            // width the element index of a [rows][D] mask is built with: the CALLER's d_model (DtqnNet.d_real) on a
                                       // width-padded network, so that the keep masks are those of the reference-shaped network the oracle evaluates
};
static inline TlDrop tl_drop_none() { TlDrop d = {}; d.scale = 1.0f; d.batch = 1; return d; }
static inline TlDrop tl_drop_make(const DtqnNet& net, uint32_t seed, uint32_t step, const int32_t* step_counter, int batch, int passes) {
    TlDrop d = tl_drop_none();
    if (net.dropout > 0.f && passes != 0) {
        d.thresh = (uint32_t)((double)net.dropout * 4294967296.0);
        d.scale = 1.0f / (1.0f - net.dropout);
        d.seed = seed; d.step = step; d.step_counter = step_counter; d.batch = batch > 0 ? batch : 1; d.passes = passes;
        d.dw = net.d_real > 0 ? net.d_real : net.d_model;
    }
    return d;
}
__device__ __forceinline__ Drop tl_drop(const TlDrop& d, int s) {
    if (d.thresh == 0u) return drop_off();
    const int which = s / d.batch;
    if (!((d.passes >> which) & 1)) return drop_off();
    return Drop{d.thresh, d.scale, d.seed, d.step_counter != nullptr ? (uint32_t)d.step_counter[1] : d.step,
                ((uint32_t)which << 20) | (uint32_t)(s - which * d.batch)};
}

// ---- embedding + position ---------------------------------------------------------------------------
struct TlEmbedArgs {
    DtqnNet net;
    const float* theta_a;              // sequences [0, split)
    const float* theta_b;              // sequences [split, S)
    int split;
    const float* obs;
    const uint8_t* actions;
    long long obs_ep_stride, act_ep_stride;
    const int32_t* ep_idx;             // replay mode (TD): sequence s = which * batch + b reads episode ep_idx[b]
    const int32_t* start;              //   from row start[b] + (which > 0); nullptr: sequence s reads "episode" s from row 0
    int batch;
    int seq0;                          // replay mode: the launch's sequence 0 is sequence seq0 of the update (a launch of some of the passes)
    int n, rpb;
    Fld x;                             // [LPB][D] embedded tokens + positions
    Fld ein;                           // [LPB][KEP] input of the embedding linear (training only; base may be null)
    int src_mod;                       // > 0: sequence s reads source sequence s % src_mod (one bag per window, three forwards)
    int bag;                           // 1: bag entries (dtqn.py:203-210): no position, the action embedding is not rolled
    const float* pre;                  // image nets: precomputed observation embeddings [S][pre_rows][D - a] (dtqn_img_encode); the linear is skipped
    int pre_rows;
    const int32_t* lens;               // ragged prefixes in one launch (dtqn_actor_forward_batch): live rows of sequence s, or nullptr = n.
                                       // Only the "a one-row sequence keeps its action embedding" rule (dtqn.py:187) looks at it
    TlDrop drop;                       // x0 = dropout(embedding + position) (dtqn.py:195-199)
    const float *ptab_a, *ptab_b;      // embedding product tables of the two parameter sets (dtqn_wpack.hpp), or nullptr: tl_embed_table_kernel
    int n_save;                        // tl_embed_table_kernel: e_in is written for sequences [0, n_save) only (the ones the backward reads)
    Fld qkv;                           // tl_embed_table_kernel<DQ > 0>: layer 0's q | k | v record and in_proj parameters (+ fragment-major copies)
    const float *Wina, *Winb, *bina, *binb, *Winpa, *Winpb;
};
// One workgroup per (sequence, 64-row block).  The gathered input rows e_in [64][KE] (observation floats, or the
// concatenated table rows of the observation tokens) and the embedding matrix go through LDS in K chunks of at most
// kEmbKC columns (padded with zeros to a multiple of 4); the product runs on the matrix cores: item = (16-column tile,
// 16-row tile), D / 4 items dealt round-robin to the 8 waves, one v_mfma_f32_16x16x4_f32 per 4 contraction columns, both
// operands read from LDS (leading dimensions of 4 mod 64 words: the 64 lanes of a fragment read hit 64 different banks).
// Round 3 (HBM-bound kernel: 67 MB written per launch at BASELINE config 3, it took 83 us): the tokens of the block, the
// embedding table and a column -> (token slot, table column) map are staged in LDS ONCE, so that the gather is LDS traffic
// without a division or a dependent global load per element; the output tile goes through LDS too and leaves as whole rows
// (16-byte stores, position rows added as 16-byte loads) instead of one dword per lane and 64-byte segments.
constexpr int kEmbKC = 64;
constexpr int kEmbLD = kEmbKC + 4;
constexpr int kEmbTabMax = 2048;       // table floats staged in LDS (larger tables are read from global memory)
static inline size_t tl_embed_lds(const DtqnNet& net) {
    const int D = net.d_model;
    size_t fl = ((size_t)TROWS + (size_t)D) * kEmbLD;                  // El | Wl, later the output tile [64][D + 4]
    const size_t out = (size_t)TROWS * (D + 4);
    fl = fl > out ? fl : out;
    if (net.discrete) fl += (size_t)TROWS * net.obs_dim + (size_t)net.ke + (net.vocab * net.embed_per_obs <= kEmbTabMax ? net.vocab * net.embed_per_obs : 0);
    return fl * sizeof(float);
}
__global__ __launch_bounds__(TNT) void tl_embed_kernel(TlEmbedArgs a) {
    const DtqnNet& net = a.net;
    const int D = net.d_model, O = net.obs_dim, adim = net.action_dim, KE = net.ke, KEP = net.kep, n = a.n;
    const int s = (int)blockIdx.x / a.rpb, rb = (int)blockIdx.x % a.rpb;
    const float* __restrict__ theta = s >= a.split ? a.theta_b : a.theta_a;
    int ep = a.src_mod > 0 ? s % a.src_mod : s, row_first = 0;
    if (a.ep_idx != nullptr) {
        const int sg = s + a.seq0;
        const int which = sg / a.batch, b = sg - which * a.batch;
        ep = a.ep_idx[b];
        row_first = a.start[b] + (which > 0 ? 1 : 0);
    }
    const float* obs_rows = a.obs + (size_t)ep * a.obs_ep_stride + (size_t)row_first * O;
    const uint8_t* act_rows = a.actions != nullptr ? a.actions + (size_t)ep * a.act_ep_stride + row_first : nullptr;
    const bool single = (a.lens != nullptr ? a.lens[s] : n) == 1;     // history_len == 1: no roll, row 0 keeps its action (dtqn.py:187-191)
    const Thr t = make_thr();
    const int tid = t.tid;
    const int LDO = D + 4;
    float* El = reinterpret_cast<float*>(dtqn_smem);                  // [64][kEmbLD]  e_in chunk
    float* Wl = El + TROWS * kEmbLD;                                   // [D][kEmbLD]   W_e chunk, row = OUTPUT column (rows < adim: zero)
    float* Xl = El;                                                    // [64][LDO]     output tile (after the last chunk)
    const size_t main_fl = ((size_t)TROWS + D) * kEmbLD > (size_t)TROWS * LDO ? ((size_t)TROWS + D) * kEmbLD : (size_t)TROWS * LDO;
    int* tokl = reinterpret_cast<int*>(El + main_fl);                  // [64][O] clamped tokens of the block's rows (discrete)
    int* kmap = tokl + TROWS * O;                                      // [KE] column k -> (token slot << 16) | table column
    float* tabl = reinterpret_cast<float*>(kmap + KE);                 // [V * e] the embedding table
    const int DO = D - adim;                                           // outputs of the embedding linear
    const int per_wave = D / 4 / TNW;                                  // items per wave: 2 / 4 / 8 at D = 64 / 128 / 256
    const int nrows = n - rb * TROWS < TROWS ? n - rb * TROWS : TROWS; // live rows of this block (may be <= 0)
    const bool use_pre = a.pre != nullptr;
    const bool tab_lds = net.discrete && net.vocab * net.embed_per_obs <= kEmbTabMax;
    if (net.discrete && !use_pre) {
        for (int idx = tid; idx < TROWS * O; idx += TNT) {
            const int rl = idx / O;
            int tok = rl < nrows ? (int)obs_rows[(size_t)rb * TROWS * O + idx] : 0;
            tokl[idx] = tok < 0 ? 0 : (tok >= net.vocab ? net.vocab - 1 : tok);
        }
        for (int k = tid; k < KE; k += TNT) {
            const int jj = k / net.embed_per_obs;
            kmap[k] = (jj << 16) | (k - jj * net.embed_per_obs);
        }
        if (tab_lds)
            for (int idx = tid; idx < net.vocab * net.embed_per_obs; idx += TNT) tabl[idx] = theta[net.off_obs_tab + idx];
    }
    f32x4 acc[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) acc[q] = zero4();
    float* eo = a.ein.base != nullptr ? frow(a.ein, s, rb * TROWS) : nullptr;
    const float* __restrict__ We = theta + net.off_obs_w;
    for (int k0 = 0; k0 < (use_pre ? 0 : KE); k0 += kEmbKC) {
        const int kc = KE - k0 < kEmbKC ? KE - k0 : kEmbKC, kc4 = (kc + 3) & ~3;
        __syncthreads();                                               // previous chunk consumed (first chunk: tokens / map / table staged)
        // a full chunk is 64 columns wide: shifts instead of divisions
        const int sh = kc4 == 64 ? 6 : -1;
        for (int idx = tid; idx < TROWS * kc4; idx += TNT) {
            const int rl = sh >= 0 ? idx >> 6 : idx / kc4, kl = idx - rl * kc4, k = k0 + kl;
            float v = 0.f;
            if (rl < nrows && kl < kc) {
                if (net.discrete) {
                    const int m = kmap[k], tok = tokl[rl * O + (m >> 16)], te = tok * net.embed_per_obs + (m & 0xffff);
                    v = tab_lds ? tabl[te] : theta[net.off_obs_tab + te];
                } else {
                    v = obs_rows[(size_t)(rb * TROWS + rl) * O + k];
                }
            }
            El[rl * kEmbLD + kl] = v;
            if (eo != nullptr && kl < kc) eo[(size_t)rl * KEP + k] = v;
        }
        for (int idx = tid; idx < D * kc4; idx += TNT) {
            const int d = sh >= 0 ? idx >> 6 : idx / kc4, kl = idx - d * kc4;
            Wl[d * kEmbLD + kl] = (d >= adim && kl < kc) ? We[(size_t)(d - adim) * KE + k0 + kl] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            if (q >= per_wave) break;
            const int item = t.wave + q * TNW, nt = item >> 2, mt = item & 3;
            const float* ap = El + (mt * 16 + t.i) * kEmbLD + t.kq;
            const float* bp = Wl + (nt * 16 + t.i) * kEmbLD + t.kq;
            for (int k = 0; k < kc4; k += 4) acc[q] = mfma16(ap[k], bp[k], acc[q]);
        }
    }
    if (eo != nullptr && !use_pre)                                     // zero the padding columns [KE, KEP) of the saved input
        for (int idx = tid; idx < TROWS * (KEP - KE); idx += TNT) {
            const int rl = idx / (KEP - KE), k = KE + idx - rl * (KEP - KE);
            eo[(size_t)rl * KEP + k] = 0.f;
        }
    __syncthreads();                                                   // El / Wl consumed: the output tile takes their place
    // accumulators (+ bias) -> Xl; the action-embedding columns are filled by their own small pass
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        if (q >= per_wave) break;
        const int item = t.wave + q * TNW, nt = item >> 2, mt = item & 3, d = nt * 16 + t.i;
        if (d >= adim) {
            const float bias = use_pre ? 0.f : theta[net.off_obs_b + d - adim];
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const int rl = mt * 16 + t.kq * 4 + r4;
                float v = acc[q][r4] + bias;
                if (use_pre && rl < nrows) v = a.pre[((size_t)s * a.pre_rows + rb * TROWS + rl) * DO + (d - adim)];   // bias already added by the encoder
                Xl[rl * LDO + d] = v;
            }
        }
    }
    for (int idx = tid; idx < TROWS * adim; idx += TNT) {
        const int rl = idx / adim, d = idx - rl * adim, r = rb * TROWS + rl;
        float v = 0.f;
        if (rl < nrows) {
            // previous action's embedding, rolled by one step, zero at t = 0 (dtqn.py:184-192); bag entries: their own action
            if (single || a.bag) v = theta[net.off_act_emb + (int)act_rows[a.bag ? r : 0] * adim + d];
            else if (r > 0) v = theta[net.off_act_emb + (int)act_rows[r - 1] * adim + d];
        }
        Xl[rl * LDO + d] = v;
    }
    __syncthreads();
    // whole rows out: + position (dtqn.py:193-199), dropout, zeros behind the live rows
    float* xo = frow(a.x, s, rb * TROWS);
    const Drop edr = tl_drop(a.drop, s);
    const int c4n = D >> 2;
    for (int idx = tid; idx < TROWS * c4n; idx += TNT) {
        const int rl = idx / c4n, d = (idx - rl * c4n) * 4, r = rb * TROWS + rl;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (rl < nrows) {
            v = ld4(Xl + rl * LDO + d);
            if (!a.bag) {
                const float4 pv = ld4(theta + net.off_pos + (size_t)r * D + d);
                v.x += pv.x; v.y += pv.y; v.z += pv.z; v.w += pv.w;
            }
            if (edr.thresh != 0u) {
                const uint32_t e0 = (uint32_t)(r * a.drop.dw + d);       // (columns behind dw are padding: zero with or without a mask)
                v.x = drop_apply(edr, DROP_EMB, 0, e0, v.x);
                v.y = drop_apply(edr, DROP_EMB, 0, e0 + 1, v.y);
                v.z = drop_apply(edr, DROP_EMB, 0, e0 + 2, v.z);
                v.w = drop_apply(edr, DROP_EMB, 0, e0 + 3, v.w);
            }
        }
        st4(xo + (size_t)rl * a.x.ld + d, v);
    }
}

// Discrete observations with the embedding product table of dtqn_wpack.hpp (TD forward of the covered row-block networks): the embedding of
// a row is b + sum_j P[j][tok_j] -- O gathered 16-byte pieces per output piece, no LDS tiles, no matrix product.  One workgroup per (sequence,
// 64-row block) like tl_embed_kernel, same outputs (x with positions and dropout, e_in for the sequences the backward reads).
__device__ __forceinline__ void wpack_fetch_f(float4 (&bf)[8], const float* __restrict__ Wp, int ntile, int kch, int kc, int lane);   // (below)
// DQ > 0 (= d_model, 128 or 256): the block's embedded rows also stay in an LDS tile and layer 0's packed q | k | v projection runs on it right
// here (qkv = x W_in^T + b_in, tl_wide_kernel's arithmetic): the projection launch and its read of x disappear.
template <int DQ, bool PK>
__global__ __launch_bounds__(TNT, DQ == 256 ? 2 : 4) void tl_embed_table_kernel(TlEmbedArgs a) {
    const DtqnNet& net = a.net;
    const int D = net.d_model, O = net.obs_dim, adim = net.action_dim, KE = net.ke, KEP = net.kep, n = a.n, V = net.vocab, E = net.embed_per_obs;
    const int DO = D - adim;
    const int s = (int)blockIdx.x / a.rpb, rb = (int)blockIdx.x % a.rpb;
    const float* __restrict__ theta = s >= a.split ? a.theta_b : a.theta_a;
    const float* __restrict__ P = s >= a.split ? a.ptab_b : a.ptab_a;
    int ep = a.src_mod > 0 ? s % a.src_mod : s, row_first = 0;
    if (a.ep_idx != nullptr) {
        const int sg = s + a.seq0;
        const int which = sg / a.batch, b = sg - which * a.batch;
        ep = a.ep_idx[b];
        row_first = a.start[b] + (which > 0 ? 1 : 0);
    }
    const float* obs_rows = a.obs + (size_t)ep * a.obs_ep_stride + (size_t)row_first * O;
    const uint8_t* act_rows = a.actions != nullptr ? a.actions + (size_t)ep * a.act_ep_stride + row_first : nullptr;
    const bool single = (a.lens != nullptr ? a.lens[s] : n) == 1;
    const int tid = (int)threadIdx.x;
    constexpr int LDXQ = DQ + 4, LDHQ = 128 + 4;
    float* Xt = reinterpret_cast<float*>(dtqn_smem);                   // DQ > 0: [64][LDXQ] embedded rows | [64][LDHQ] projection staging
    float* Hs = Xt + TROWS * LDXQ;
    int* tokl = reinterpret_cast<int*>(DQ > 0 ? Hs + TROWS * LDHQ : Xt);      // [64][O] clamped tokens of the block's rows
    const int nrows = n - rb * TROWS < TROWS ? n - rb * TROWS : TROWS;
    for (int idx = tid; idx < TROWS * O; idx += TNT) {
        const int rl = idx / O;
        int tok = rl < nrows ? (int)obs_rows[(size_t)rb * TROWS * O + idx] : 0;
        tokl[idx] = tok < 0 ? 0 : (tok >= V ? V - 1 : tok);
    }
    __syncthreads();
    float* xo = frow(a.x, s, rb * TROWS);
    const Drop edr = tl_drop(a.drop, s);
    const int c4n = D >> 2;
    for (int idx = tid; idx < TROWS * c4n; idx += TNT) {
        const int rl = idx / c4n, d = (idx - rl * c4n) * 4, r = rb * TROWS + rl;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (rl < nrows) {
            if (d >= adim) {
                const int* tk = tokl + rl * O;
                for (int j = 0; j < O; ++j) {
                    const float4 pv = ld4(P + ((size_t)j * V + tk[j]) * DO + (d - adim));
                    v.x += pv.x; v.y += pv.y; v.z += pv.z; v.w += pv.w;
                }
                const float4 bv = ld4(theta + net.off_obs_b + (d - adim));
                v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
            } else {
                // previous action's embedding, rolled by one step, zero at t = 0 (dtqn.py:184-192)
                if (single) v = ld4(theta + net.off_act_emb + (int)act_rows[0] * adim + d);
                else if (r > 0) v = ld4(theta + net.off_act_emb + (int)act_rows[r - 1] * adim + d);
            }
            const float4 pv = ld4(theta + net.off_pos + (size_t)r * D + d);
            v.x += pv.x; v.y += pv.y; v.z += pv.z; v.w += pv.w;
            if (edr.thresh != 0u) {
                const uint32_t e0 = (uint32_t)(r * a.drop.dw + d);
                v.x = drop_apply(edr, DROP_EMB, 0, e0, v.x);
                v.y = drop_apply(edr, DROP_EMB, 0, e0 + 1, v.y);
                v.z = drop_apply(edr, DROP_EMB, 0, e0 + 2, v.z);
                v.w = drop_apply(edr, DROP_EMB, 0, e0 + 3, v.w);
            }
        }
        st4(xo + (size_t)rl * a.x.ld + d, v);
        if constexpr (DQ > 0) st4(Xt + rl * LDXQ + d, v);
    }
    if (a.ein.base != nullptr && s < a.n_save) {                       // the embedding linear's input, for its weight gradient
        float* eo = frow(a.ein, s, rb * TROWS);
        for (int idx = tid; idx < TROWS * KEP; idx += TNT) {
            const int rl = idx / KEP, k = idx - rl * KEP;
            float v = 0.f;
            if (rl < nrows && k < KE) {
                const int j = k / E;
                v = theta[net.off_obs_tab + tokl[rl * O + j] * E + (k - j * E)];
            }
            eo[(size_t)rl * KEP + k] = v;
        }
    }
    if constexpr (DQ > 0) {
        // ---------------- layer 0's q | k | v projection on the tile ----------------
        constexpr int KA = 128, NKA = DQ / KA, NB = 3 * DQ / 128, MT = TROWS / 16;
        Thr t = make_thr();
        int wc = t.wave * 16 + t.i;
        const bool second = s >= a.split;
        const float* __restrict__ Win = second ? a.Winb : a.Wina;
        const float* __restrict__ Winp = second ? a.Winpb : a.Winpa;
        const float* __restrict__ bin = second ? a.binb : a.bina;
        auto fetchQ = [&](float4 (&bf)[8], int j, int kc) {
            if constexpr (PK) {
                wpack_fetch_f(bf, Winp, j * 8 + t.wave, NKA, kc, t.lane);
                return;
            }
            const float* wr = Win + (size_t)(j * 128 + wc) * DQ + kc * KA + t.kq * 4;
#pragma unroll
            for (int q = 0; q < KA / 16; ++q) bf[q] = ld4(wr + 16 * q);
        };
        float4 bf0[8], bf1[8];
        fetchQ(bf0, 0, 0);
        __syncthreads();                                               // the tile is complete
        auto blockq = [&](float4 (&cur)[8], float4 (&nxt)[8], int j) {
            f32x4 acc[MT];
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[m] = zero4();
            if (NKA == 1) {
                if (j + 1 < NB) fetchQ(nxt, j + 1, 0);
                frag16_mma<KA, 2, 8>(Xt, LDXQ, cur, t, reinterpret_cast<f32x4(&)[2]>(acc[0]));
                frag16_mma<KA, 2, 8>(Xt + 32 * LDXQ, LDXQ, cur, t, reinterpret_cast<f32x4(&)[2]>(acc[2]));
            } else {
                fetchQ(nxt, j, 1);
                frag16_mma<KA, MT, 8>(Xt, LDXQ, cur, t, acc);
                if (j + 1 < NB) fetchQ(cur, j + 1, 0);
                frag16_mma<KA, MT, 8>(Xt + KA, LDXQ, nxt, t, acc);
            }
            const float bv = bin[j * 128 + wc];
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) Hs[(m * 16 + t.kq * 4 + r4) * LDHQ + wc] = acc[m][r4] + bv;
            __syncthreads();
            for (int idx = t.tid; idx < TROWS * 32; idx += TNT) {
                const int rl = idx >> 5, c = (idx & 31) * 4;
                st4(frow(a.qkv, s, rb * TROWS + rl) + j * 128 + c, ld4(Hs + rl * LDHQ + c));
            }
            __syncthreads();                                           // staging tile free for the next block
        };
        for (int j = 0; j < NB; ++j) {
            { int tid_ = (int)threadIdx.x; DTQN_ASM_KEEP(tid_); t.tid = tid_; t.lane = tid_ & 63; t.wave = tid_ >> 6; t.i = t.lane & 15; t.kq = t.lane >> 4; wc = t.wave * 16 + t.i; }
            if (NKA == 2 || (j & 1) == 0) blockq(bf0, bf1, j);
            else blockq(bf1, bf0, j);
        }
    }
}

// frag_dyw_mma (dtqn_device.hpp) over the first NN / 4 entries of a 32-entry fragment array
template <int NN, int MG>
__device__ __forceinline__ void frag_dyw_mma_n(const float* dYs, int lda, const float (&bf)[32], const Thr& t, f32x4 (&acc)[MG]) {
    constexpr int KS = NN / 16;
    static_assert(NN <= 128 && MG >= 2, "fragment array holds 32 entries; MG == 1 uses frag_dyw_mma's two-accumulator form");
    const float* yp = dYs + t.i * lda + t.kq * (NN / 4);
    float4 af[2][MG];
#pragma unroll
    for (int m = 0; m < MG; ++m) af[0][m] = ld4(yp + m * 16 * lda);
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        if (s + 1 < KS) {
#pragma unroll
            for (int m = 0; m < MG; ++m) af[(s + 1) & 1][m] = ld4(yp + m * 16 * lda + 4 * (s + 1));
        }
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

// Weight fragments out of the fragment-major copies of dtqn_wpack.hpp: every load instruction of a wave reads 1 KB contiguous.
//   F copy of W[N][K]: fragment (16-column tile ntile, 128-wide contraction chunk kc of K / 128) -> the 8 float4 frag16_fetch<128> loads
__device__ __forceinline__ void wpack_fetch_f(float4 (&bf)[8], const float* __restrict__ Wp, int ntile, int kch, int kc, int lane) {
    const float* p = Wp + ((size_t)(ntile * kch + kc) * 8) * 256 + lane * 4;
#pragma unroll
    for (int q = 0; q < 8; ++q) bf[q] = ld4(p + 256 * q);
}
//   B copy of W[N][K] (contraction over N): fragment (16-column tile ktile of K, chunk nc of N / 128) -> the 32 floats frag_dyw_fetch<128> loads
__device__ __forceinline__ void wpack_fetch_b(float (&bf)[32], const float* __restrict__ Wp, int ktile, int nch, int nc, int lane) {
    const float* p = Wp + ((size_t)(ktile * nch + nc) * 8) * 256 + lane * 4;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const float4 v = ld4(p + 256 * q);
        bf[4 * q] = v.x; bf[4 * q + 1] = v.y; bf[4 * q + 2] = v.z; bf[4 * q + 3] = v.w;
    }
}

// ---- linear: OUT[rows][N] = f(IN[rows][K] * W[N][K]^T [+ IN2[rows][K2] * W2[N][K2]^T] + b) ----------------------
//   mode 0: OUT = y      mode 1: OUT = relu(y)      mode 2: OUT = RES + relu(y)   (residual gate)
//   modes 1 / 2 optionally save the ReLU pattern as wave ballots (same word layout as the whole-sequence kernels)
//   GRU gate (gates.py:26-31; RES = the stream x):
//   mode 3: OUT = sigmoid(y);  OUT2 = OUT * RES (r * x) and OUT3 = RES (copy of x for the weight gradients) when given
//   mode 4: OUT = tanh(y);     OUT2 = (1 - AUX) * RES + AUX * OUT          (AUX = z: the gate output)
struct TlLinearArgs {
    Fld in, out, res, mask;            // mask.base == nullptr: no ballots
    const float *Wa, *Wb, *ba, *bb;    // sequences >= split use Wb / bb; ba == nullptr: no bias
    int split;
    int K, N, rpb, mode;
    Fld in2;                           // second operand pair (K2 == 0: none)
    const float *W2a, *W2b;
    int K2;
    Fld aux, out2, out3;
    const float *Wpa, *Wpb;            // fragment-major F copies of Wa / Wb (dtqn_wpack.hpp), or nullptr
};
// (second launch bound = waves per SIMD the register budget must allow: two resident workgroups up to D = 128)
// MR rows per workgroup: 64, or 32 when the launch would be only a few rounds of resident workgroups (launch_linear)
template <int D, int MR, bool PK>
__global__ __launch_bounds__(TNT, D <= 128 ? 4 : 2) void tl_linear_kernel(TlLinearArgs a) {
    constexpr int MT = MR / 16;
    constexpr int LDT = D + 4;
    float* Xt = reinterpret_cast<float*>(dtqn_smem);                   // [64][LDT] input tile of the current K chunk
    const Thr t = make_thr();
    const int s = (int)blockIdx.x / a.rpb, row0 = ((int)blockIdx.x % a.rpb) * MR;       // a.rpb: MR-row blocks per sequence
    const int ntile = (int)blockIdx.y * TNW + t.wave;                 // this wave's 16-column output tile
    const int col = ntile * 16 + t.i;
    const bool live = col < a.N;                                      // N is a multiple of 16: wave-uniform
    const float* __restrict__ W = s >= a.split ? a.Wb : a.Wa;
    const float* __restrict__ W2 = s >= a.split ? a.W2b : a.W2a;
    const float* __restrict__ bias = s >= a.split ? a.bb : a.ba;
    f32x4 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m] = zero4();
    // two weight fragments that alternate by NAME: indexing them with the chunk counter would put the array into
    // scratch memory (dynamic register indexing does not exist) -- 272 / 528 bytes per lane before this was unrolled
    float4 bf0[D / 16], bf1[D / 16];
    const float* wrow = W + (size_t)(live ? col : 0) * a.K;
    const float* wrow2 = a.K2 > 0 ? W2 + (size_t)(live ? col : 0) * a.K2 : nullptr;
    const float* __restrict__ Wp = s >= a.split ? a.Wpb : a.Wpa;
    // chunk kc of the (first) operand's weights: D columns = D / 128 packed chunks of 8 float4
    auto wfetch = [&](float4 (&bf)[D / 16], int kc) {
        if constexpr (PK) {
            if (kc < a.K / D) {
                const float* p = Wp + ((size_t)((live ? ntile : 0) * (a.K / 128) + kc * (D / 128)) * 8) * 256 + t.lane * 4;
#pragma unroll
                for (int q = 0; q < D / 16; ++q) bf[q] = ld4(p + 256 * q);
                return;
            }
        }
        frag16_fetch<D>(bf, kc < a.K / D ? wrow + (size_t)kc * D : wrow2 + (size_t)(kc - a.K / D) * D, t);
    };
    wfetch(bf0, 0);
    const float* in0 = frow(a.in, s, row0);
    const float* in20 = a.K2 > 0 ? frow(a.in2, s, row0) : nullptr;
    const int nch1 = a.K / D, nchunks = nch1 + a.K2 / D;
    auto stage = [&](int kc) {                                        // chunk kc of the operand(s) -> LDS
        const float* src = kc < nch1 ? in0 + (size_t)kc * D : in20 + (size_t)(kc - nch1) * D;
        const int ld = kc < nch1 ? a.in.ld : a.in2.ld;
        for (int idx = t.tid; idx < MR * (D / 4); idx += TNT) {
            const int r = idx / (D / 4), c = (idx - r * (D / 4)) * 4;
            st4(Xt + r * LDT + c, ld4(src + (size_t)r * ld + c));
        }
    };
    // D = 256 (one workgroup per CU: 250 registers): the operand tile of chunk kc + 1 is pulled global -> registers while chunk
    // kc multiplies and dropped into the LDS tile once chunk kc's readers are through.  At D <= 128 two workgroups share a
    // CU and hide each other's staging; the 16 extra registers would cost that (measured: cfg 4 603 -> 581 updates/s).
    constexpr bool PREFETCH = D >= 256;
    constexpr int XN = MR * (D / 4) / TNT;                         // float4 per thread of a [64][D] tile
    float4 xr[PREFETCH ? XN : 1];
    auto stage_load = [&](int kc) {
        const float* src = kc < nch1 ? in0 + (size_t)kc * D : in20 + (size_t)(kc - nch1) * D;
        const int ld = kc < nch1 ? a.in.ld : a.in2.ld;
#pragma unroll
        for (int k = 0; k < (PREFETCH ? XN : 0); ++k) {
            const int idx = t.tid + k * TNT;
            const int r = idx / (D / 4), c = (idx - r * (D / 4)) * 4;
            xr[k] = ld4(src + (size_t)r * ld + c);
        }
    };
    auto stage_store = [&]() {
#pragma unroll
        for (int k = 0; k < (PREFETCH ? XN : 0); ++k) {
            const int idx = t.tid + k * TNT;
            const int r = idx / (D / 4), c = (idx - r * (D / 4)) * 4;
            st4(Xt + r * LDT + c, xr[k]);
        }
    };
    if (PREFETCH) stage_load(0);
    for (int kc = 0; kc < nchunks; kc += 2) {
        if (kc > 0) __syncthreads();                                  // previous chunk's tile fully consumed
        if (PREFETCH) stage_store(); else stage(kc);
        if (kc + 1 < nchunks) {
            if (PREFETCH) stage_load(kc + 1);
            wfetch(bf1, kc + 1);
        }
        __syncthreads();
        frag16_mma<D, MT>(Xt, LDT, bf0, t, acc);
        if (kc + 1 < nchunks) {
            __syncthreads();
            if (PREFETCH) stage_store(); else stage(kc + 1);
            if (kc + 2 < nchunks) {
                if (PREFETCH) stage_load(kc + 2);
                wfetch(bf0, kc + 2);
            }
            __syncthreads();
            frag16_mma<D, MT>(Xt, LDT, bf1, t, acc);
        }
    }
    // epilogue through LDS: the accumulators (MFMA layout: lane = column, 4 rows) go into the operand tile as [row][column of
    // this 128-column block], then every lane handles float4 pieces of whole rows: 16-byte loads of RES / AUX and 16-byte
    // stores (a quarter of the memory instructions of per-lane dword accesses).  ReLU ballots are taken in the MFMA layout.
    constexpr int LDE = 16 * TNW + 4;                                 // epilogue tile [64][128 + 4]
    __syncthreads();                                                  // the last chunk's tile is consumed
    if (live) {
        const float b = bias != nullptr ? bias[col] : 0.f;
        float* mrec = a.mask.base != nullptr ? a.mask.base + (size_t)s * a.mask.stride : nullptr;
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const int rl = m * 16 + t.kq * 4 + r4;
                const float v = acc[m][r4] + b;
                if (mrec != nullptr && (a.mode == 1 || a.mode == 2)) ballot_store(mrec, a.N / 16, row0 + rl, col, v > 0.f, t.lane);
                Xt[rl * LDE + t.wave * 16 + t.i] = v;
            }
    }
    __syncthreads();
    const int cb = (int)blockIdx.y * TNW * 16;                        // first column of this block
    for (int idx = t.tid; idx < MR * (TNW * 4); idx += TNT) {
        const int rl = idx / (TNW * 4), c = (idx - rl * (TNW * 4)) * 4, cg = cb + c, row = row0 + rl;
        if (cg >= a.N) continue;
        const float4 v = ld4(Xt + rl * LDE + c);
        float* op = frow(a.out, s, row) + cg;
        if (a.mode == 0) {
            st4(op, v);
        } else if (a.mode == 3) {
            const float4 g = make_float4(sigmoidf_(v.x), sigmoidf_(v.y), sigmoidf_(v.z), sigmoidf_(v.w));
            st4(op, g);
            if (a.out2.base != nullptr) {
                const float4 x = ld4(frow(a.res, s, row) + cg);
                st4(frow(a.out2, s, row) + cg, make_float4(g.x * x.x, g.y * x.y, g.z * x.z, g.w * x.w));
                if (a.out3.base != nullptr) st4(frow(a.out3, s, row) + cg, x);
            }
        } else if (a.mode == 4) {
            const float4 hc = make_float4(tanhf(v.x), tanhf(v.y), tanhf(v.z), tanhf(v.w));
            const float4 z = ld4(frow(a.aux, s, row) + cg), x = ld4(frow(a.res, s, row) + cg);
            st4(op, hc);
            st4(frow(a.out2, s, row) + cg, make_float4((1.0f - z.x) * x.x + z.x * hc.x, (1.0f - z.y) * x.y + z.y * hc.y,
                                                       (1.0f - z.z) * x.z + z.z * hc.z, (1.0f - z.w) * x.w + z.w * hc.w));
        } else {
            const float4 y = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
            if (a.mode == 1) {
                st4(op, y);
            } else {
                const float4 r = ld4(frow(a.res, s, row) + cg);
                st4(op, make_float4(r.x + y.x, r.y + y.y, r.z + y.z, r.w + y.w));
            }
        }
    }
}

// ---- wide linear: OUT[rows][N] = IN[rows][D] W[N][D]^T + b for N a multiple of 128 (the packed q | k | v projection) -----------
// One workgroup per (sequence, MR-row block) walks ALL column blocks of 128: the input tile is staged once (tl_linear_kernel stages
// it once per column block), weight fragments are fetched one step ahead of the MFMAs that use them, and each block of
// accumulators leaves through a second LDS tile as 16-byte row pieces.  Same structure as the first product of tl_ffn_kernel.
struct TlWideArgs {
    Fld in, out;
    const float *Wa, *Wb, *ba, *bb;    // sequences >= split use Wb / bb
    int split, rpb, N;
    // LN instantiation (N == D): OUT = RES + relu(y) followed by the LayerNorm that closes the attention half of a post-LN layer
    // (transformer.py:70-72): ln_out = LN(OUT).  OUT, its ReLU ballots and the row statistics are stored only for the sequences
    // the backward pass reads (s < n_save)
    Fld res, mask, ln_out, ln_st;
    const float *lga, *lgb, *lba, *lbb;
    int n_save;
    const float *Wpa, *Wpb;            // fragment-major F copies of Wa / Wb (dtqn_wpack.hpp), or nullptr
    int skew;                          // start skew of the second-slot workgroups, 100-MHz ticks (tl_start_skew; 0: none)
    TlPack pack;                       // plain instantiation, 64-row workgroups: packed rows of the sequences that are not saved (L == 0: none)
};
template <int D, int MR, bool LN, bool PK>
__global__ __launch_bounds__(TNT, D <= 128 ? 4 : 2) void tl_wide_kernel(TlWideArgs a) {
    constexpr int MT = MR / 16, KA = D < 128 ? D : 128, NKA = D / KA, LDX = D + 4, LDH = 128 + 4;
    static_assert(NKA == 1 || NKA == 2, "step sequence written for D in {64, 128, 256}");
    float* Xt = reinterpret_cast<float*>(dtqn_smem);                   // [MR][LDX] input rows
    float* Hs = Xt + MR * LDX;                                         // [MR][LDH] output staging (plain) | [MR][LDX] all columns (LN)
    Thr t = make_thr();
    const int blk = (int)blockIdx.x;
    const TlBlk rb = tl_blk(blk, a.rpb, MR, LN ? TlPack{0, 0, 0} : a.pack);
    const int s = rb.s, row0 = rb.row0;                                // (packed workgroups: of their first row; every row goes through frow_b)
    const bool second = s >= a.split, save = s < a.n_save;
    const float* __restrict__ W = second ? a.Wb : a.Wa;
    const float* __restrict__ bias = second ? a.bb : a.ba;
    const int wc = t.wave * 16 + t.i, NB = LN ? (D + 127) / 128 : a.N / 128;
    const float* __restrict__ Wp = second ? a.Wpb : a.Wpa;
    auto fetch = [&](float4 (&bf)[8], int j, int kc) {                 // W [N][D]: output column j * 128 + wc, contraction chunk kc
        const int col = j * 128 + wc;
        if constexpr (PK) {
            wpack_fetch_f(bf, Wp, (!LN || col < D) ? j * 8 + t.wave : 0, NKA, kc, t.lane);
            return;
        }
        const float* wr = W + (size_t)((!LN || col < D) ? col : 0) * D + kc * KA + t.kq * 4;
#pragma unroll
        for (int q = 0; q < KA / 16; ++q) bf[q] = ld4(wr + 16 * q);
    };
    float4 bf0[8], bf1[8];
    fetch(bf0, 0, 0);
    {
        for (int idx = t.tid; idx < MR * (D / 4); idx += TNT) {
            const int r = idx / (D / 4), c = (idx - r * (D / 4)) * 4;
            st4(Xt + r * LDX + c, ld4(frow_b(a.in, rb, r) + c));
        }
    }
    float* mrec = LN && save && a.mask.base != nullptr ? a.mask.base + (size_t)s * a.mask.stride : nullptr;
    tl_start_skew(a.skew);
    __syncthreads();
    // one column block: `cur` holds its first fragment on entry; on exit the first fragment of block j + 1 sits in `nxt`
    // (NKA == 1: the buffers swap roles from block to block) or in `cur` again (NKA == 2)
    auto block = [&](float4 (&cur)[8], float4 (&nxt)[8], int j) {
        f32x4 acc[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[m] = zero4();
        const bool live = !LN || j * 128 + wc < D;                     // D = 64: waves 4..7 have no column
        if (NKA == 1) {
            if (j + 1 < NB) fetch(nxt, j + 1, 0);
            if constexpr (!LN && KA == 128 && MT == 4) {
                // two weight fragments (64 registers) + four accumulators + the double-buffered A fragments of four row tiles do not
                // fit the 128 registers of four workgroups per CU (round 2: 72 bytes of scratch per lane): the row tiles go through
                // in two pairs against the same weight fragment
                frag16_mma<KA, 2, 8>(Xt, LDX, cur, t, reinterpret_cast<f32x4(&)[2]>(acc[0]));
                frag16_mma<KA, 2, 8>(Xt + 32 * LDX, LDX, cur, t, reinterpret_cast<f32x4(&)[2]>(acc[2]));
            } else if (live) frag16_mma<KA, MT, 8>(Xt, LDX, cur, t, acc);
        } else {
            fetch(nxt, j, 1);
            frag16_mma<KA, MT, 8>(Xt, LDX, cur, t, acc);
            if (j + 1 < NB) fetch(cur, j + 1, 0);
            frag16_mma<KA, MT, 8>(Xt + KA, LDX, nxt, t, acc);
        }
        const int col = j * 128 + wc;
        const float bv = bias != nullptr && live ? bias[col] : 0.f;
        if (LN) {
            if (live) {
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) {
                        const int rl = m * 16 + t.kq * 4 + r4;
                        const float v = acc[m][r4] + bv;
                        if (mrec != nullptr) ballot_store(mrec, D / 16, row0 + rl, col, v > 0.f, t.lane);
                        Hs[rl * LDX + col] = fmaxf(v, 0.f);
                    }
            }
            return;
        }
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) Hs[(m * 16 + t.kq * 4 + r4) * LDH + wc] = acc[m][r4] + bv;
        __syncthreads();
        for (int idx = t.tid; idx < MR * 32; idx += TNT) {
            const int rl = idx >> 5, c = (idx & 31) * 4;
            st4(frow_b(a.out, rb, rl) + j * 128 + c, ld4(Hs + rl * LDH + c));
        }
        __syncthreads();                                               // staging tile free for the next block
    };
    for (int j = 0; j < NB; ++j) {
        // thread coordinates opaque per column block: the per-thread LDS / global addresses are recomputed instead of being hoisted
        // out of the loop and kept live next to two weight fragments and MT accumulators (72 bytes of scratch per lane at
        // <128, 64> under the 128-register bound of four workgroups per CU)
        DTQN_ASM_KEEP(t.tid); DTQN_ASM_KEEP(t.lane); DTQN_ASM_KEEP(t.wave); DTQN_ASM_KEEP(t.i); DTQN_ASM_KEEP(t.kq);
        if (NKA == 2 || (j & 1) == 0) block(bf0, bf1, j);
        else block(bf1, bf0, j);
    }
    if (LN) {
        // rows: LPR lanes per row hold the row in registers, add the residual, take mean / variance in a butterfly (the two-pass
        // arithmetic of layernorm_rows) and write LN(OUT)
        __syncthreads();
        // 8 lanes per row whatever MR is: see tl_ffn_kernel
        constexpr int LPR = 8, NV = D / (4 * LPR);
        if (t.tid >= MR * LPR) return;
        const int rl = t.tid / LPR, part = t.tid % LPR, row = row0 + rl;
        const float* gamma = second ? a.lgb : a.lga;
        const float* beta = second ? a.lbb : a.lba;
        float4 y[NV];
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int c = part * 4 + 4 * LPR * j;
            const float4 v = ld4(Hs + rl * LDX + c), r = ld4(frow(a.res, s, row) + c);
            y[j] = make_float4(r.x + v.x, r.y + v.y, r.z + v.z, r.w + v.w);
            sum += (y[j].x + y[j].y) + (y[j].z + y[j].w);
            if (save) st4(frow(a.out, s, row) + c, y[j]);
        }
#pragma unroll
        for (int m = 1; m < LPR; m <<= 1) sum += __shfl_xor(sum, m);
        const float mean = sum * (1.0f / D);
        float sq = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const float p = y[j].x - mean, q = y[j].y - mean, u = y[j].z - mean, w = y[j].w - mean;
            sq += (p * p + q * q) + (u * u + w * w);
        }
#pragma unroll
        for (int m = 1; m < LPR; m <<= 1) sq += __shfl_xor(sq, m);
        const float rstd = 1.0f / sqrtf(sq * (1.0f / D) + 1e-5f);
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int c = part * 4 + 4 * LPR * j;
            const float4 g = ld4(gamma + c), bb = ld4(beta + c);
            st4(frow(a.ln_out, s, row) + c, make_float4((y[j].x - mean) * rstd * g.x + bb.x, (y[j].y - mean) * rstd * g.y + bb.y,
                                                        (y[j].z - mean) * rstd * g.z + bb.z, (y[j].w - mean) * rstd * g.w + bb.w));
        }
        if (save && a.ln_st.base != nullptr && part == 0) {
            float* stp = a.ln_st.base + (size_t)s * a.ln_st.stride + (size_t)row * 2;
            stp[0] = mean;
            stp[1] = rstd;
        }
    }
}

// ---- fused feed-forward block: OUT = f(relu(IN W1^T + b1) W2^T + b2), hidden width 4 D (transformer.py:55-61, 76-77) -------------
//   mode 1: OUT = relu(y) (GRU gate input)      mode 2: OUT = RES + relu(y) (residual gate)
// One workgroup per (sequence, 64-row block).  The input tile stays in LDS; the hidden layer is produced in chunks of 128
// units (wave w: units 16 w .. 16 w + 15 of the chunk, all 64 rows), dropped into a second LDS tile after bias + ReLU and
// consumed right there as the contraction chunk of the second product, whose [64][D] accumulators stay in registers.  The
// hidden activations never make the trip through memory that two separate GEMMs need ([rows][4 D] floats out and in again --
// the row-block path is bound by exactly that traffic); they are WRITTEN (with their ReLU ballots) only
// for the sequences the backward pass will read (s < n_save: the training third of a TD update).
// Weight fragments alternate between two register arrays by name; the fetch of the next step is issued before the MFMAs of
// the current one.
struct TlFfnArgs {
    Fld in, out, res;
    Fld h, mh, m2;                     // hidden record [LPB][4D], its ballots, the output's ballots (bases may be null)
    const float *W1a, *W1b, *b1a, *b1b, *W2a, *W2b, *b2a, *b2b;
    int split, rpb, mode, n_save;
    // mode 2 with the LayerNorm that follows the block in a post-LN layer (transformer.py:78) folded in: ln_out = LN(OUT), the
    // row statistics to ln_st and OUT itself are stored only for the sequences the backward pass reads (s < n_save)
    Fld ln_out, ln_st;                 // ln_out.base == nullptr: no LayerNorm
    const float *lga, *lgb, *lba, *lbb;
    TlDrop drop;                       // dropout on the block's output, before the gate's ReLU (transformer.py:38-42)
    int layer;
    const float *W1pa, *W1pb, *W2pa, *W2pb;    // fragment-major F copies of W1 / W2 (dtqn_wpack.hpp), or nullptr
    int skew;                          // start skew of the second-slot workgroups, 100-MHz ticks (tl_start_skew; 0: none)
};
// MR rows per workgroup (64, or 32 when the launch would otherwise be a round and a half of workgroups: launch_ffn)
template <int D, int MR, bool PK>
__global__ __launch_bounds__(TNT, D <= 128 ? 4 : 2) void tl_ffn_kernel(TlFfnArgs a) {
    constexpr int MT = MR / 16;
    constexpr int KA = D < 128 ? D : 128, NKA = D / KA, NOT = (D + 127) / 128, HID = 4 * D, NJ = HID / 128;
    constexpr int LDX = D + 4, LDH = 128 + 4;
    static_assert((NKA == 1 && NOT == 1) || (NKA == 2 && NOT == 2), "step sequence written for D in {64, 128, 256}");
    float* Xt = reinterpret_cast<float*>(dtqn_smem);                   // [MR][LDX] input rows
    float* Hs = Xt + MR * LDX;                                         // [MR][LDH] hidden chunk / output staging
    const Thr t = make_thr();
    const int blk = (int)blockIdx.x;
    const int s = blk / a.rpb, row0 = (blk % a.rpb) * MR;              // a.rpb: MR-row blocks per sequence
    const bool second = s >= a.split, save = s < a.n_save;
    const float* __restrict__ W1 = second ? a.W1b : a.W1a;
    const float* __restrict__ W2 = second ? a.W2b : a.W2a;
    const float* __restrict__ b1 = second ? a.b1b : a.b1a;
    const float* __restrict__ b2 = second ? a.b2b : a.b2a;
    const int wc = t.wave * 16 + t.i;                                  // this lane's column inside a 128-column block
    const float* __restrict__ W1p = second ? a.W1pb : a.W1pa;
    const float* __restrict__ W2p = second ? a.W2pb : a.W2pa;
    auto fetchA = [&](float4 (&bf)[8], int j, int kc) {                // W1 [4D][D]: hidden unit j * 128 + wc, contraction chunk kc
        if constexpr (PK) {
            wpack_fetch_f(bf, W1p, j * 8 + t.wave, NKA, kc, t.lane);
            return;
        }
        const float* wr = W1 + (size_t)(j * 128 + wc) * D + kc * KA + t.kq * 4;
#pragma unroll
        for (int q = 0; q < KA / 16; ++q) bf[q] = ld4(wr + 16 * q);
    };
    auto fetchB = [&](float4 (&bf)[8], int j, int ot) {                // W2 [D][4D]: output column ot * 128 + wc, hidden chunk j
        const int col = ot * 128 + wc;
        if constexpr (PK) {
            wpack_fetch_f(bf, W2p, col < D ? ot * 8 + t.wave : 0, NJ, j, t.lane);
            return;
        }
        const float* wr = W2 + (size_t)(col < D ? col : 0) * HID + j * 128 + t.kq * 4;
#pragma unroll
        for (int q = 0; q < 8; ++q) bf[q] = ld4(wr + 16 * q);
    };
    float4 bf0[8], bf1[8];
    fetchA(bf0, 0, 0);
    {
        const float* in0 = frow(a.in, s, row0);
        for (int idx = t.tid; idx < MR * (D / 4); idx += TNT) {
            const int r = idx / (D / 4), c = (idx - r * (D / 4)) * 4;
            st4(Xt + r * LDX + c, ld4(in0 + (size_t)r * a.in.ld + c));
        }
    }
    f32x4 accO[NOT][MT];
#pragma unroll
    for (int o = 0; o < NOT; ++o)
#pragma unroll
        for (int m = 0; m < MT; ++m) accO[o][m] = zero4();
    float* mrec_h = save && a.mh.base != nullptr ? a.mh.base + (size_t)s * a.mh.stride : nullptr;
    tl_start_skew(a.skew);
    __syncthreads();
    for (int j = 0; j < NJ; ++j) {
        f32x4 accA[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) accA[m] = zero4();
        if (NKA == 1) {
            fetchB(bf1, j, 0);
            frag16_mma<KA, MT, 8>(Xt, LDX, bf0, t, accA);
        } else {
            fetchA(bf1, j, 1);
            frag16_mma<KA, MT, 8>(Xt, LDX, bf0, t, accA);
            fetchB(bf0, j, 0);
            frag16_mma<KA, MT, 8>(Xt + KA, LDX, bf1, t, accA);
        }
        {
            const int hc = j * 128 + wc;
            const float bv = b1[hc];
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const int rl = m * 16 + t.kq * 4 + r4;
                    const float v = accA[m][r4] + bv;
                    if (mrec_h != nullptr) ballot_store(mrec_h, HID / 16, row0 + rl, hc, v > 0.f, t.lane);
                    Hs[rl * LDH + wc] = fmaxf(v, 0.f);
                }
        }
        __syncthreads();                                               // the hidden chunk is complete
        if (save && a.h.base != nullptr) {
            for (int idx = t.tid; idx < MR * 32; idx += TNT) {
                const int rl = idx >> 5, c = (idx & 31) * 4;
                st4(frow(a.h, s, row0 + rl) + j * 128 + c, ld4(Hs + rl * LDH + c));
            }
        }
        if (NOT == 1) {
            if (j + 1 < NJ) fetchA(bf0, j + 1, 0);
            if (wc < D) frag16_mma<128, MT, 8>(Hs, LDH, bf1, t, accO[0]);
        } else {
            fetchB(bf1, j, 1);
            frag16_mma<128, MT, 8>(Hs, LDH, bf0, t, accO[0]);
            if (j + 1 < NJ) fetchA(bf0, j + 1, 0);
            frag16_mma<128, MT, 8>(Hs, LDH, bf1, t, accO[NOT - 1]);
        }
        __syncthreads();                                               // ... and consumed
    }
    float* mrec_o = save && a.m2.base != nullptr ? a.m2.base + (size_t)s * a.m2.stride : nullptr;
    const Drop fdr = tl_drop(a.drop, s);
    if (a.ln_out.base != nullptr) {
        // relu(y) of all D columns into the (spent) input tile, then rows: LPR lanes per row hold the row in registers, add the
        // residual, take mean / variance in a butterfly (same two-pass arithmetic as layernorm_rows) and write LN(OUT)
#pragma unroll
        for (int o = 0; o < NOT; ++o) {
            const int col = o * 128 + wc;
            if (col < D) {
                const float bv = b2[col];
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) {
                        const int rl = m * 16 + t.kq * 4 + r4;
                        const float v = drop_apply(fdr, DROP_FFN, a.layer, (uint32_t)((row0 + rl) * a.drop.dw + col), accO[o][m][r4] + bv);
                        if (mrec_o != nullptr) ballot_store(mrec_o, D / 16, row0 + rl, col, v > 0.f, t.lane);
                        Xt[rl * LDX + col] = fmaxf(v, 0.f);
                    }
            }
        }
        __syncthreads();
        // 8 lanes per row whatever MR is (at MR = 32 the upper four waves sit this out): the reduction order of a row must not
        // depend on the rows-per-workgroup variant the launcher picked from the batch size, or Q of a sequence would
        // depend on how many sequences share its batch
        constexpr int LPR = 8, NV = D / (4 * LPR);
        if (t.tid >= MR * LPR) return;                                 // wave-uniform (MR * 8 is a multiple of 64)
        const int rl = t.tid / LPR, part = t.tid % LPR, row = row0 + rl;
        const float* gamma = second ? a.lgb : a.lga;
        const float* beta = second ? a.lbb : a.lba;
        float4 y[NV];
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int c = part * 4 + 4 * LPR * j;
            const float4 v = ld4(Xt + rl * LDX + c), r = ld4(frow(a.res, s, row) + c);
            y[j] = make_float4(r.x + v.x, r.y + v.y, r.z + v.z, r.w + v.w);
            sum += (y[j].x + y[j].y) + (y[j].z + y[j].w);
            if (save) st4(frow(a.out, s, row) + c, y[j]);
        }
#pragma unroll
        for (int m = 1; m < LPR; m <<= 1) sum += __shfl_xor(sum, m);
        const float mean = sum * (1.0f / D);
        float sq = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const float p = y[j].x - mean, q = y[j].y - mean, u = y[j].z - mean, w = y[j].w - mean;
            sq += (p * p + q * q) + (u * u + w * w);
        }
#pragma unroll
        for (int m = 1; m < LPR; m <<= 1) sq += __shfl_xor(sq, m);
        const float rstd = 1.0f / sqrtf(sq * (1.0f / D) + 1e-5f);
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int c = part * 4 + 4 * LPR * j;
            const float4 g = ld4(gamma + c), bb = ld4(beta + c);
            st4(frow(a.ln_out, s, row) + c, make_float4((y[j].x - mean) * rstd * g.x + bb.x, (y[j].y - mean) * rstd * g.y + bb.y,
                                                        (y[j].z - mean) * rstd * g.z + bb.z, (y[j].w - mean) * rstd * g.w + bb.w));
        }
        if (save && a.ln_st.base != nullptr && part == 0) {
            float* stp = a.ln_st.base + (size_t)s * a.ln_st.stride + (size_t)row * 2;
            stp[0] = mean;
            stp[1] = rstd;
        }
        return;
    }
#pragma unroll
    for (int o = 0; o < NOT; ++o) {
        const int col = o * 128 + wc;
        if (col < D) {
            const float bv = b2[col];
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const int rl = m * 16 + t.kq * 4 + r4;
                    const float v = drop_apply(fdr, DROP_FFN, a.layer, (uint32_t)((row0 + rl) * a.drop.dw + col), accO[o][m][r4] + bv);
                    if (mrec_o != nullptr) ballot_store(mrec_o, D / 16, row0 + rl, col, v > 0.f, t.lane);
                    Hs[rl * LDH + wc] = fmaxf(v, 0.f);
                }
        }
        __syncthreads();
        for (int idx = t.tid; idx < MR * 32; idx += TNT) {
            const int rl = idx >> 5, c = (idx & 31) * 4, cg = o * 128 + c;
            if (cg >= D) continue;
            float4 y = ld4(Hs + rl * LDH + c);
            if (a.mode == 2) {
                const float4 r = ld4(frow(a.res, s, row0 + rl) + cg);
                y = make_float4(r.x + y.x, r.y + y.y, r.z + y.z, r.w + y.w);
            }
            st4(frow(a.out, s, row0 + rl) + cg, y);
        }
        if (o + 1 < NOT) __syncthreads();
    }
}

// ---- fused post-LN residual layer tail (round 6): everything of a layer behind its attention, in one launch ---------------------------
//   s1 = RES + relu(O W_o^T + b_o);  u2 = LN1(s1);  s2 = u2 + relu(relu(u2 W_1^T + b_1) W_2^T + b_2);  ln_out = LN2(s2)
//   (transformer.py:70-78) and, on the last layer of a network without a bag (HEAD): hh = relu(ln_out W_h1^T + b_h1), Q = hh W_h2^T + b_h2
//   (dtqn.py:149-153, 216).
// tl_wide_kernel<LN> + tl_ffn_kernel [+ tl_linear_kernel + tl_qhead_kernel] were two to four launches per layer whose short members ran
// at a third of the matrix rate of the long one (BASELINE config 4: out-projection + LayerNorm 32 us for 1.6 GFLOP, head 31 + 17 us,
// against 120 us for the 12.9 GFLOP of the feed-forward block) because a K = D product is over before its prologue is paid for, and
// u2 / xf / hh made a round trip through memory between them.  Here the row block's tile stays in LDS from the attention output to the
// Q rows: phase 0 leaves u2 in the input tile of the feed-forward loop, the closing LayerNorm leaves xf there for the head.  The
// records are written exactly where the separate kernels wrote them, and only for the sequences the backward pass reads (s < n_save:
// s1, u2, h, s2, hh, the ballots, the row statistics); ln_out always (the next layer's projection and attention read it).
// Arithmetic: the same products in the same order as the separate kernels (same fragments, same 8-lane row butterflies, the 16-lane dot
// products of tl_qhead_kernel), so Q does not depend on which way a layer went.
struct TlLayerArgs {
    Fld o, res, s1, m1, u2, st1;       // phase 0: O [LPB][D] in; RES = the stream entering the layer; s1 | m1 | u2 | st1 records out
    const float *Woa, *Wob, *boa, *bob, *g1a, *g1b, *be1a, *be1b;
    const float *Wopa, *Wopb;          // fragment-major F copies of W_o, or nullptr
    TlFfnArgs f;                       // the feed-forward block and LN2 (f.in / f.res unused: the tile is u2; f.mode == 2, f.ln_out given)
    // HEAD instantiation
    Fld hh;
    const float *Wh1a, *Wh1b, *bh1a, *bh1b, *Wh1pa, *Wh1pb;
    const float *Wqa, *Wqb, *bqa, *bqb;
    float* q;
    long long q_seq_stride;
    int q_row_stride, A, n;
    TlPack pack;                       // 64-row workgroups: packed rows of the sequences that are not saved (L == 0: none)
    // TAIL == 2: the NEXT layer's packed q | k | v projection on the tile LayerNorm 2 leaves (its input u1): qkv = ln_out W_in^T + b_in
    Fld qkv;
    const float *Wina, *Winb, *bina, *binb, *Winpa, *Winpb;
};
// TAIL: 0 nothing behind LayerNorm 2;  1 the Q head (last layer);  2 the next layer's q | k | v projection (d_model 128 / 256) -- the launch
// of tl_wide_kernel it replaces read the same rows back from memory and ran at 0.55 of the matrix rate for want of length
template <int D, int MR, bool PK, int TAIL>
__global__ __launch_bounds__(TNT, D <= 128 ? 4 : 2) void tl_layer_kernel(TlLayerArgs a) {
    constexpr bool HEAD = TAIL == 1, QKV = TAIL == 2 && (3 * D) % 128 == 0;
    constexpr int MT = MR / 16;
    constexpr int KA = D < 128 ? D : 128, NKA = D / KA, NOT = (D + 127) / 128, HID = 4 * D, NJ = HID / 128;
    constexpr int LDX = D + 4, LDH = 128 + 4;
    static_assert((NKA == 1 && NOT == 1) || (NKA == 2 && NOT == 2), "step sequence written for D in {64, 128, 256}");
    float* Xt = reinterpret_cast<float*>(dtqn_smem);                   // [MR][LDX] o -> u2 -> s2 -> xf
    float* Hs = Xt + MR * LDX;                                         // [MR][max(LDX, LDH)] relu(y1) | hidden chunk | hh
    Thr t = make_thr();
    const int blk = (int)blockIdx.x;
    const TlBlk rb = tl_blk(blk, a.f.rpb, MR, a.pack);
    const int s = rb.s, row0 = rb.row0;                                // (packed workgroups: of their first row; they save nothing, and every
    const bool second = s >= a.f.split, save = s < a.f.n_save;         //  row they touch goes through tl_row)
    int wc = t.wave * 16 + t.i;
    float4 bf0[8], bf1[8];
    // thread coordinates made opaque phase by phase: per-thread addresses of a later phase are recomputed there instead of living through
    // the feed-forward loop next to two weight fragments and the accumulators (184 - 208 bytes of scratch per lane at <128, 64> otherwise)
#define TL_LAYER_RELAUNDER() do { int tid_ = (int)threadIdx.x; DTQN_ASM_KEEP(tid_); t.tid = tid_; t.lane = tid_ & 63; t.wave = tid_ >> 6; t.i = t.lane & 15; t.kq = t.lane >> 4; wc = t.wave * 16 + t.i; } while (0)
    // a [D][D] matrix (W_o, W_h1): output column j * 128 + wc, contraction chunk kc
    auto fetchDD = [&](float4 (&bf)[8], const float* __restrict__ W, const float* __restrict__ Wp, int j, int kc) {
        const int col = j * 128 + wc;
        if constexpr (PK) {
            wpack_fetch_f(bf, Wp, col < D ? j * 8 + t.wave : 0, NKA, kc, t.lane);
            return;
        }
        const float* wr = W + (size_t)(col < D ? col : 0) * D + kc * KA + t.kq * 4;
#pragma unroll
        for (int q = 0; q < KA / 16; ++q) bf[q] = ld4(wr + 16 * q);
    };
    // [MR][D] = Xt [MR][D] * W^T, one 128-column block after the other; sink(j, col, rl, v) takes the accumulators (MFMA layout)
    auto gemmDD = [&](const float* __restrict__ W, const float* __restrict__ Wp, auto&& sink) {
#pragma unroll
        for (int j = 0; j < NOT; ++j) {
            f32x4 acc[MT];
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[m] = zero4();
            const int col = j * 128 + wc;
            const bool live = col < D;
            if (NKA == 1) {
                if (live) frag16_mma<KA, MT, 8>(Xt, LDX, bf0, t, acc);
            } else {
                fetchDD(bf1, W, Wp, j, 1);
                frag16_mma<KA, MT, 8>(Xt, LDX, bf0, t, acc);
                if (j + 1 < NOT) fetchDD(bf0, W, Wp, j + 1, 0);
                frag16_mma<KA, MT, 8>(Xt + KA, LDX, bf1, t, acc);
            }
            if (live) {
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) sink(col, m * 16 + t.kq * 4 + r4, acc[m][r4]);
            }
        }
    };
    // rows of Xt (+ `add` when given): 8 lanes per row whatever MR is (see tl_ffn_kernel), two-pass statistics, LN(row) to `dst` (global,
    // may be null) and back into Xt when `keep`; the un-normalised row to `raw` and (mean, rstd) to `st` for the sequences the backward reads
    auto ln_rows = [&](const float* add_tile, const Fld& res, const Fld& raw, const Fld& st, const Fld& dst, const float* gamma, const float* beta, bool keep) {
        constexpr int LPR = 8, NV = D / (4 * LPR);
        if (t.tid >= MR * LPR) return;
        const int rl = t.tid / LPR, part = t.tid % LPR;
        int s, row;                                                    // this row's sequence and record row (shadow the workgroup's)
        tl_row(rb, rl, s, row);
        float4 y[NV];
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int c = part * 4 + 4 * LPR * j;
            if (add_tile != nullptr) {
                const float4 v = ld4(add_tile + rl * LDX + c), r = ld4(frow(res, s, row) + c);
                y[j] = make_float4(r.x + v.x, r.y + v.y, r.z + v.z, r.w + v.w);
            } else {
                y[j] = ld4(Xt + rl * LDX + c);
            }
            sum += (y[j].x + y[j].y) + (y[j].z + y[j].w);
            if (save) st4(frow(raw, s, row) + c, y[j]);
        }
#pragma unroll
        for (int m = 1; m < LPR; m <<= 1) sum += __shfl_xor(sum, m);
        const float mean = sum * (1.0f / D);
        float sq = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            // (every multiply-add written out: the instantiations of this kernel must not differ in which products the compiler fuses --
            //  with the plain expressions TAIL = 0 and TAIL = 2 came out one ulp apart in rstd on a few rows, 1e-5 apart in Q)
            const float p = y[j].x - mean, q = y[j].y - mean, u = y[j].z - mean, w = y[j].w - mean;
            sq += fmaf(p, p, q * q) + fmaf(u, u, w * w);
        }
#pragma unroll
        for (int m = 1; m < LPR; m <<= 1) sq += __shfl_xor(sq, m);
        const float rstd = 1.0f / sqrtf(fmaf(sq, 1.0f / D, 1e-5f));
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int c = part * 4 + 4 * LPR * j;
            const float4 g = ld4(gamma + c), bb = ld4(beta + c);
            const float4 o = make_float4(fmaf((y[j].x - mean) * rstd, g.x, bb.x), fmaf((y[j].y - mean) * rstd, g.y, bb.y),
                                         fmaf((y[j].z - mean) * rstd, g.z, bb.z), fmaf((y[j].w - mean) * rstd, g.w, bb.w));
            if (dst.base != nullptr) st4(frow(dst, s, row) + c, o);
            if (keep) st4(Xt + rl * LDX + c, o);
        }
        if (save && st.base != nullptr && part == 0) {
            float* stp = st.base + (size_t)s * st.stride + (size_t)row * 2;
            stp[0] = mean;
            stp[1] = rstd;
        }
    };

    // ---------------- phase 0: out-projection, residual gate, LayerNorm 1 ----------------
    const float* __restrict__ Wo = second ? a.Wob : a.Woa;
    const float* __restrict__ Wop = second ? a.Wopb : a.Wopa;
    fetchDD(bf0, Wo, Wop, 0, 0);
    {
        for (int idx = t.tid; idx < MR * (D / 4); idx += TNT) {
            const int r = idx / (D / 4), c = (idx - r * (D / 4)) * 4;
            st4(Xt + r * LDX + c, ld4(frow_b(a.o, rb, r) + c));
        }
    }
    tl_start_skew(a.f.skew);
    __syncthreads();
    {
        const float* __restrict__ bo = second ? a.bob : a.boa;
        float* mrec = save && a.m1.base != nullptr ? a.m1.base + (size_t)s * a.m1.stride : nullptr;
        gemmDD(Wo, Wop, [&](int col, int rl, float acc) {
            const float v = acc + bo[col];
            if (mrec != nullptr) ballot_store(mrec, D / 16, row0 + rl, col, v > 0.f, t.lane);
            Hs[rl * LDX + col] = fmaxf(v, 0.f);
        });
    }
    const float* __restrict__ W1 = second ? a.f.W1b : a.f.W1a;
    const float* __restrict__ W2 = second ? a.f.W2b : a.f.W2a;
    const float* __restrict__ b1 = second ? a.f.b1b : a.f.b1a;
    const float* __restrict__ b2 = second ? a.f.b2b : a.f.b2a;
    const float* __restrict__ W1p = second ? a.f.W1pb : a.f.W1pa;
    const float* __restrict__ W2p = second ? a.f.W2pb : a.f.W2pa;
    auto fetchA = [&](float4 (&bf)[8], int j, int kc) {                // W1 [4D][D]: hidden unit j * 128 + wc, contraction chunk kc
        if constexpr (PK) {
            wpack_fetch_f(bf, W1p, j * 8 + t.wave, NKA, kc, t.lane);
            return;
        }
        const float* wr = W1 + (size_t)(j * 128 + wc) * D + kc * KA + t.kq * 4;
#pragma unroll
        for (int q = 0; q < KA / 16; ++q) bf[q] = ld4(wr + 16 * q);
    };
    auto fetchB = [&](float4 (&bf)[8], int j, int ot) {                // W2 [D][4D]: output column ot * 128 + wc, hidden chunk j
        const int col = ot * 128 + wc;
        if constexpr (PK) {
            wpack_fetch_f(bf, W2p, col < D ? ot * 8 + t.wave : 0, NJ, j, t.lane);
            return;
        }
        const float* wr = W2 + (size_t)(col < D ? col : 0) * HID + j * 128 + t.kq * 4;
#pragma unroll
        for (int q = 0; q < 8; ++q) bf[q] = ld4(wr + 16 * q);
    };
    fetchA(bf0, 0, 0);                                                 // in flight across the LayerNorm rows
    __syncthreads();                                                   // relu(y1) complete; every reader of the o tile is through
    ln_rows(Hs, a.res, a.s1, a.st1, save ? a.u2 : nofld_dev(), second ? a.g1b : a.g1a, second ? a.be1b : a.be1a, true);
    __syncthreads();                                                   // the tile is u2

    // ---------------- phase 1: feed-forward block (tl_ffn_kernel's loop) ----------------
    TL_LAYER_RELAUNDER();
    f32x4 accO[NOT][MT];
#pragma unroll
    for (int o = 0; o < NOT; ++o)
#pragma unroll
        for (int m = 0; m < MT; ++m) accO[o][m] = zero4();
    float* mrec_h = save && a.f.mh.base != nullptr ? a.f.mh.base + (size_t)s * a.f.mh.stride : nullptr;
    for (int j = 0; j < NJ; ++j) {
        TL_LAYER_RELAUNDER();
        f32x4 accA[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) accA[m] = zero4();
        if (NKA == 1) {
            fetchB(bf1, j, 0);
            frag16_mma<KA, MT, 8>(Xt, LDX, bf0, t, accA);
        } else {
            fetchA(bf1, j, 1);
            frag16_mma<KA, MT, 8>(Xt, LDX, bf0, t, accA);
            fetchB(bf0, j, 0);
            frag16_mma<KA, MT, 8>(Xt + KA, LDX, bf1, t, accA);
        }
        {
            const int hc = j * 128 + wc;
            const float bv = b1[hc];
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const int rl = m * 16 + t.kq * 4 + r4;
                    const float v = accA[m][r4] + bv;
                    if (mrec_h != nullptr) ballot_store(mrec_h, HID / 16, row0 + rl, hc, v > 0.f, t.lane);
                    Hs[rl * LDH + wc] = fmaxf(v, 0.f);
                }
        }
        __syncthreads();                                               // the hidden chunk is complete
        if (save && a.f.h.base != nullptr) {
            for (int idx = t.tid; idx < MR * 32; idx += TNT) {
                const int rl = idx >> 5, c = (idx & 31) * 4;
                st4(frow(a.f.h, s, row0 + rl) + j * 128 + c, ld4(Hs + rl * LDH + c));
            }
        }
        if (NOT == 1) {
            if (j + 1 < NJ) fetchA(bf0, j + 1, 0);
            if (wc < D) frag16_mma<128, MT, 8>(Hs, LDH, bf1, t, accO[0]);
        } else {
            fetchB(bf1, j, 1);
            frag16_mma<128, MT, 8>(Hs, LDH, bf0, t, accO[0]);
            if (j + 1 < NJ) fetchA(bf0, j + 1, 0);
            frag16_mma<128, MT, 8>(Hs, LDH, bf1, t, accO[NOT - 1]);
        }
        __syncthreads();                                               // ... and consumed
    }
    TL_LAYER_RELAUNDER();
    // ---------------- phase 2: residual gate (u2 is the tile: s2 = tile + relu(y2), in place) and LayerNorm 2 ----------------
    const float* __restrict__ Wh1 = second ? a.Wh1b : a.Wh1a;
    const float* __restrict__ Wh1p = second ? a.Wh1pb : a.Wh1pa;
    if constexpr (HEAD) fetchDD(bf0, Wh1, Wh1p, 0, 0);                // in flight across the LayerNorm rows
    const float* __restrict__ Win = second ? a.Winb : a.Wina;
    const float* __restrict__ Winp = second ? a.Winpb : a.Winpa;
    auto fetchQ = [&](float4 (&bf)[8], int j, int kc) {               // W_in [3D][D]: output column j * 128 + wc, contraction chunk kc
        if constexpr (PK) {
            wpack_fetch_f(bf, Winp, j * 8 + t.wave, NKA, kc, t.lane);
            return;
        }
        const float* wr = Win + (size_t)(j * 128 + wc) * D + kc * KA + t.kq * 4;
#pragma unroll
        for (int q = 0; q < KA / 16; ++q) bf[q] = ld4(wr + 16 * q);
    };
    if constexpr (QKV) fetchQ(bf0, 0, 0);
    {
        float* mrec_o = save && a.f.m2.base != nullptr ? a.f.m2.base + (size_t)s * a.f.m2.stride : nullptr;
        const Drop fdr = tl_drop(a.f.drop, s);
#pragma unroll
        for (int o = 0; o < NOT; ++o) {
            const int col = o * 128 + wc;
            if (col < D) {
                const float bv = b2[col];
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) {
                        const int rl = m * 16 + t.kq * 4 + r4;
                        const float v = drop_apply(fdr, DROP_FFN, a.f.layer, (uint32_t)((row0 + rl) * a.f.drop.dw + col), accO[o][m][r4] + bv);
                        if (mrec_o != nullptr) ballot_store(mrec_o, D / 16, row0 + rl, col, v > 0.f, t.lane);
                        Xt[rl * LDX + col] += fmaxf(v, 0.f);
                    }
            }
        }
    }
    __syncthreads();
    ln_rows(nullptr, a.res, a.f.out, a.f.ln_st, a.f.ln_out, second ? a.f.lgb : a.f.lga, second ? a.f.lbb : a.f.lba, HEAD || QKV);
    if constexpr (QKV) {
        // ---------------- phase 3': the next layer's q | k | v projection on the tile (its u1), one 128-column block after the other ----------------
        __syncthreads();
        TL_LAYER_RELAUNDER();
        constexpr int NB = 3 * D / 128;
        const float* __restrict__ bin = second ? a.binb : a.bina;
        auto blockq = [&](float4 (&cur)[8], float4 (&nxt)[8], int j) {
            f32x4 acc[MT];
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[m] = zero4();
            if (NKA == 1) {
                if (j + 1 < NB) fetchQ(nxt, j + 1, 0);
                if constexpr (KA == 128 && MT == 4) {                  // (two fragments + four accumulators + double-buffered A of four row tiles: see tl_wide_kernel)
                    frag16_mma<KA, 2, 8>(Xt, LDX, cur, t, reinterpret_cast<f32x4(&)[2]>(acc[0]));
                    frag16_mma<KA, 2, 8>(Xt + 32 * LDX, LDX, cur, t, reinterpret_cast<f32x4(&)[2]>(acc[2]));
                } else frag16_mma<KA, MT, 8>(Xt, LDX, cur, t, acc);
            } else {
                fetchQ(nxt, j, 1);
                frag16_mma<KA, MT, 8>(Xt, LDX, cur, t, acc);
                if (j + 1 < NB) fetchQ(cur, j + 1, 0);
                frag16_mma<KA, MT, 8>(Xt + KA, LDX, nxt, t, acc);
            }
            const float bv = bin[j * 128 + wc];
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) Hs[(m * 16 + t.kq * 4 + r4) * LDH + wc] = acc[m][r4] + bv;
            __syncthreads();
            for (int idx = t.tid; idx < MR * 32; idx += TNT) {
                const int rl = idx >> 5, c = (idx & 31) * 4;
                st4(frow_b(a.qkv, rb, rl) + j * 128 + c, ld4(Hs + rl * LDH + c));
            }
            __syncthreads();                                           // staging tile free for the next block
        };
        for (int j = 0; j < NB; ++j) {
            TL_LAYER_RELAUNDER();
            if (NKA == 2 || (j & 1) == 0) blockq(bf0, bf1, j);
            else blockq(bf1, bf0, j);
        }
    }
    if constexpr (HEAD) {
        // ---------------- phase 3: Q head on the tile (xf) ----------------
        __syncthreads();
        TL_LAYER_RELAUNDER();
        const float* __restrict__ bh = second ? a.bh1b : a.bh1a;
        gemmDD(Wh1, Wh1p, [&](int col, int rl, float acc) { Hs[rl * LDX + col] = fmaxf(acc + bh[col], 0.f); });
        __syncthreads();
        if (save && a.hh.base != nullptr) {
            for (int idx = t.tid; idx < MR * (D / 4); idx += TNT) {
                const int r = idx / (D / 4), c = (idx - r * (D / 4)) * 4;
                st4(frow(a.hh, s, row0 + r) + c, ld4(Hs + r * LDX + c));
            }
        }
        // 16 lanes per row (tl_qhead_kernel's arithmetic, the hidden row out of LDS)
        const int c = t.tid & 15;
        constexpr int NJQ = D / 64;
        const float* Wq = (second ? a.Wqb : a.Wqa) + 4 * c;
        const float* bq = second ? a.bqb : a.bqa;
        for (int rr = t.tid >> 4; rr < MR; rr += TNT / 16) {
            int sq, row;
            tl_row(rb, rr, sq, row);
            float4 hv[NJQ];
#pragma unroll
            for (int j = 0; j < NJQ; ++j) hv[j] = ld4(Hs + rr * LDX + 4 * c + 64 * j);
            float* qrow = a.q + (size_t)sq * a.q_seq_stride + (size_t)row * a.q_row_stride;
            for (int ac = 0; ac < a.A; ++ac) {
                float p = 0.f;
#pragma unroll
                for (int j = 0; j < NJQ; ++j) {
                    const float4 wv = ld4(Wq + (size_t)ac * D + 64 * j);
                    p = fmaf(hv[j].x, wv.x, p); p = fmaf(hv[j].y, wv.y, p); p = fmaf(hv[j].z, wv.z, p); p = fmaf(hv[j].w, wv.w, p);
                }
#pragma unroll
                for (int m = 8; m >= 1; m >>= 1) p += __shfl_xor(p, m);
                if (row < a.n && c == (ac & 15)) qrow[ac] = p + bq[ac];
            }
        }
    }
}
#undef TL_LAYER_RELAUNDER

// ---- backward linear: dX[rows][KOUT] (op)= sum_p dY_p[rows][N] * W_p[N][KOUT],  p < nsrc <= 3 ------------------
//   mode 0: OUT = v      mode 1: OUT = v where the saved ReLU ballot of OUT's forward twin is set, else 0
//   mode 2: OUT += v
//   (nsrc > 1: the GRU gate, whose dx and dy each collect several matrix products -- gates.py:26-31)
struct TlDxArgs {
    Fld dy, out, mask;
    const float* W;
    int N, KOUT, rpb, mode;
    int nsrc;
    Fld dy2, dy3;
    const float *W2, *W3;
    const float* Wp;                   // fragment-major B copy of W (dtqn_wpack.hpp; single-operand launches), or nullptr
    // Q-head mode (hh.base != nullptr, round 6): dY is not read but MADE in the staging -- dhh = (dq W_q) * [hh > 0], tl_head_bwd_kernel's
    // arithmetic -- and written to `dy` (the dhh record the weight gradients read) by the first column block: one launch instead of two
    Fld hh, dq;
    const float* Wq;
    int A;
};
template <int KC, int MR, bool PK, bool HEADM = false>
__global__ __launch_bounds__(TNT) void tl_dx_kernel(TlDxArgs a) {
    constexpr int MT = MR / 16;
    constexpr int LDT = KC + 4;
    float* Yt = reinterpret_cast<float*>(dtqn_smem);                   // [64][LDT] dY tile of the current chunk
    const Thr t = make_thr();
    const int s = (int)blockIdx.x / a.rpb, row0 = ((int)blockIdx.x % a.rpb) * MR;       // a.rpb: MR-row blocks per sequence
    const int ntile = (int)blockIdx.y * TNW + t.wave;
    const int col = ntile * 16 + t.i;
    const bool live = col < a.KOUT;
    f32x4 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m] = zero4();
    float bf0[KC / 4], bf1[KC / 4];                                    // alternate by name (see tl_linear_kernel)
    const int cl = live ? col : 0;
    const int per = a.N / KC, nchunks = per * a.nsrc;
    // chunk kc = (operand pair p, contraction chunk j)
    auto wchunk = [&](int kc) {
        const int p = kc / per, j = kc - p * per;
        return (p == 0 ? a.W : p == 1 ? a.W2 : a.W3) + cl + (size_t)j * KC * a.KOUT;
    };
    auto stage = [&](int kc) {
        const int p = kc / per, j = kc - p * per;
        const Fld& dyf = p == 0 ? a.dy : p == 1 ? a.dy2 : a.dy3;
        if constexpr (HEADM) {
            for (int idx = t.tid; idx < MR * (KC / 4); idx += TNT) {
                const int r = idx / (KC / 4), c0 = j * KC + (idx - r * (KC / 4)) * 4;
                const float4 hv4 = ld4(frow(a.hh, s, row0 + r) + c0);
                const float hv[4] = {hv4.x, hv4.y, hv4.z, hv4.w};
                const float* dqr = frow(a.dq, s, row0 + r);
                float g4[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float g = 0.f;
                    if (hv[e] > 0.f)
                        for (int c = 0; c < a.A; ++c) g = fmaf(dqr[c], a.Wq[(size_t)c * a.N + c0 + e], g);
                    g4[e] = g;
                }
                const float4 gv = make_float4(g4[0], g4[1], g4[2], g4[3]);
                st4(Yt + r * LDT + (c0 - j * KC), gv);
                if (blockIdx.y == 0) st4(frow(a.dy, s, row0 + r) + c0, gv);
            }
            return;
        }
        const float* dy0 = frow(dyf, s, row0) + (size_t)j * KC;
        for (int idx = t.tid; idx < MR * (KC / 4); idx += TNT) {
            const int r = idx / (KC / 4), c = (idx - r * (KC / 4)) * 4;
            st4(Yt + r * LDT + c, ld4(dy0 + (size_t)r * dyf.ld + c));
        }
    };
    auto wfetch = [&](float (&bf)[KC / 4], int kc) {
        if constexpr (PK) {
            wpack_fetch_b(bf, a.Wp, live ? ntile : 0, per, kc, t.lane);
            return;
        }
        frag_dyw_fetch<KC>(bf, wchunk(kc), a.KOUT, t);
    };
    wfetch(bf0, 0);
    for (int kc = 0; kc < nchunks; kc += 2) {
        if (kc > 0) __syncthreads();                                  // previous chunk's tile fully consumed
        stage(kc);
        if (kc + 1 < nchunks) wfetch(bf1, kc + 1);
        __syncthreads();
        frag_dyw_mma<KC, MT>(Yt, LDT, bf0, t, acc);
        if (kc + 1 < nchunks) {
            __syncthreads();
            stage(kc + 1);
            if (kc + 2 < nchunks) wfetch(bf0, kc + 2);
            __syncthreads();
            frag_dyw_mma<KC, MT>(Yt, LDT, bf1, t, acc);
        }
    }
    // epilogue through LDS (see tl_linear_kernel): accumulators -> [row][column] tile -> 16-byte row pieces
    constexpr int LDE = 16 * TNW + 4;
    __syncthreads();                                                  // the last chunk's tile is consumed
    if (live) {
        const unsigned long long* mrec =
            a.mode == 1 ? reinterpret_cast<const unsigned long long*>(a.mask.base + (size_t)s * a.mask.stride) : nullptr;
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const int rl = m * 16 + t.kq * 4 + r4, row = row0 + rl;
                float v = acc[m][r4];
                if (a.mode == 1) {
                    const unsigned long long w = mrec[((row >> 4) * (a.KOUT / 16) + (col >> 4)) * 4 + r4];
                    v = ((w >> t.lane) & 1ull) ? v : 0.f;
                }
                Yt[rl * LDE + t.wave * 16 + t.i] = v;
            }
    }
    __syncthreads();
    const int cb = (int)blockIdx.y * TNW * 16;
    for (int idx = t.tid; idx < MR * (TNW * 4); idx += TNT) {
        const int rl = idx / (TNW * 4), c = (idx - rl * (TNW * 4)) * 4, cg = cb + c;
        if (cg >= a.KOUT) continue;
        float4 v = ld4(Yt + rl * LDE + c);
        float* op = frow(a.out, s, row0 + rl) + cg;
        if (a.mode == 2) {
            const float4 p = ld4(op);
            v = make_float4(p.x + v.x, p.y + v.y, p.z + v.z, p.w + v.w);
        }
        st4(op, v);
    }
}

// ---- fused feed-forward backward: dh' = (df W2) * [h > 0] -> du = dh' W1 (the mirror of tl_ffn_kernel) ------------------------
//   df = dL/d(output of the block before its gate's ReLU mask is applied) * [f > 0] when m2 is given (residual gate: the
//   mask stage of the gate backward rides in this kernel's staging and df is stored for the weight gradients), else dy is df.
//   OUT (op)= du:  out_mode 2: OUT += du (post-LN: the stream IS the block's input)   out_mode 0: OUT = du (identity layers)
// One workgroup per (sequence, MR-row block): the df tile stays in LDS; the hidden gradient is produced in chunks of 128 units
// (contraction over the D outputs of the block), masked with the saved ballots of the hidden ReLU, written out once (the
// weight gradients need it) and consumed from a second LDS tile as the contraction chunk of the product with W1, whose
// [MR][D] accumulators stay in registers.  Weight fragments (dword columns of W2 / W1) are fetched one step ahead.
struct TlFfnBwdArgs {
    Fld dy, m2, df;                    // m2.base == nullptr: dy is df itself (no mask, df not stored)
    Fld dhp, mh;                       // dh' record [LPB][4D] (out), ballots of the hidden ReLU
    Fld out;
    const float *W1, *W2;              // W1 [4D][D], W2 [D][4D]
    int rpb, out_mode;
    TlDrop drop;                       // the forward's dropout on the block's output: df also takes its keep mask (and is stored
    int layer;                         // to `df` when that is given, masked or not)
    const float *W1p, *W2p;            // fragment-major B copies of W1 / W2 (dtqn_wpack.hpp), or nullptr
};
template <int D, int MR, bool PK>
__global__ __launch_bounds__(TNT, D <= 128 ? 4 : 2) void tl_ffn_bwd_kernel(TlFfnBwdArgs a) {
    constexpr int MT = MR / 16, KA = D < 128 ? D : 128, NKA = D / KA, NOT = (D + 127) / 128, HID = 4 * D, NJ = HID / 128;
    constexpr int LDX = D + 4, LDH = 128 + 4;
    static_assert((NKA == 1 && NOT == 1) || (NKA == 2 && NOT == 2), "step sequence written for D in {64, 128, 256}");
    float* Yt = reinterpret_cast<float*>(dtqn_smem);                   // [MR][LDX] df rows
    float* Hs = Yt + MR * LDX;                                         // [MR][LDH] dh' chunk / output staging
    const Thr t = make_thr();
    const int s = (int)blockIdx.x / a.rpb, row0 = ((int)blockIdx.x % a.rpb) * MR;
    const int wc = t.wave * 16 + t.i;
    // phase A fragment: W2[n][j * 128 + wc], n over contraction chunk kc (KA rows); phase B: W1[j * 128 + k][ot * 128 + wc], k < 128
    auto fetchA = [&](float (&bf)[32], int j, int kc) {
        if constexpr (PK) {                                            // W2 [D][4D]: contraction over its D rows, output = hidden columns
            wpack_fetch_b(bf, a.W2p, j * 8 + t.wave, NKA, kc, t.lane);
            return;
        }
        const float* wp = a.W2 + (size_t)(kc * KA + t.kq * (KA / 4)) * HID + j * 128 + wc;
#pragma unroll
        for (int q = 0; q < KA / 4; ++q) bf[q] = wp[(size_t)q * HID];
    };
    auto fetchB = [&](float (&bf)[32], int j, int ot) {
        const int col = ot * 128 + wc;
        if constexpr (PK) {                                            // W1 [4D][D]: contraction over its 4D rows (chunk j), output = D columns
            wpack_fetch_b(bf, a.W1p, col < D ? ot * 8 + t.wave : 0, NJ, j, t.lane);
            return;
        }
        const float* wp = a.W1 + (size_t)(j * 128 + t.kq * 32) * D + (col < D ? col : 0);
#pragma unroll
        for (int q = 0; q < 32; ++q) bf[q] = wp[(size_t)q * D];
    };
    float bf0[32], bf1[32];
    fetchA(bf0, 0, 0);
    {
        const unsigned long long* mrec =
            a.m2.base != nullptr ? reinterpret_cast<const unsigned long long*>(a.m2.base + (size_t)s * a.m2.stride) : nullptr;
        const Drop bdr = tl_drop(a.drop, s);
        for (int idx = t.tid; idx < MR * (D / 4); idx += TNT) {
            const int rl = idx / (D / 4), c = (idx - rl * (D / 4)) * 4, row = row0 + rl;
            float4 v = ld4(frow(a.dy, s, row) + c);
            if (mrec != nullptr) {
                const unsigned long long w = mrec[((row >> 4) * (D / 16) + (c >> 4)) * 4 + (row & 3)];
                const int b0 = (((row >> 2) & 3) << 4) + (c & 15);
                v.x = ((w >> b0) & 1ull) ? v.x : 0.f;
                v.y = ((w >> (b0 + 1)) & 1ull) ? v.y : 0.f;
                v.z = ((w >> (b0 + 2)) & 1ull) ? v.z : 0.f;
                v.w = ((w >> (b0 + 3)) & 1ull) ? v.w : 0.f;
            }
            if (bdr.thresh != 0u) {
                const uint32_t e0 = (uint32_t)(row * a.drop.dw + c);
                v.x = drop_apply(bdr, DROP_FFN, a.layer, e0, v.x);
                v.y = drop_apply(bdr, DROP_FFN, a.layer, e0 + 1, v.y);
                v.z = drop_apply(bdr, DROP_FFN, a.layer, e0 + 2, v.z);
                v.w = drop_apply(bdr, DROP_FFN, a.layer, e0 + 3, v.w);
            }
            if (a.df.base != nullptr) st4(frow(a.df, s, row) + c, v);
            st4(Yt + rl * LDX + c, v);
        }
    }
    f32x4 accO[NOT][MT];
#pragma unroll
    for (int o = 0; o < NOT; ++o)
#pragma unroll
        for (int m = 0; m < MT; ++m) accO[o][m] = zero4();
    const unsigned long long* mrec_h = reinterpret_cast<const unsigned long long*>(a.mh.base + (size_t)s * a.mh.stride);
    __syncthreads();
    for (int j = 0; j < NJ; ++j) {
        f32x4 accA[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) accA[m] = zero4();
        if (NKA == 1) {
            fetchB(bf1, j, 0);
            frag_dyw_mma_n<KA, MT>(Yt, LDX, bf0, t, accA);
        } else {
            fetchA(bf1, j, 1);
            frag_dyw_mma_n<KA, MT>(Yt, LDX, bf0, t, accA);
            fetchB(bf0, j, 0);
            frag_dyw_mma_n<KA, MT>(Yt + KA, LDX, bf1, t, accA);
        }
        {
            const int hc = j * 128 + wc;
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const int rl = m * 16 + t.kq * 4 + r4, row = row0 + rl;
                    const unsigned long long w = mrec_h[((row >> 4) * (HID / 16) + (hc >> 4)) * 4 + r4];
                    Hs[rl * LDH + wc] = ((w >> t.lane) & 1ull) ? accA[m][r4] : 0.f;
                }
        }
        __syncthreads();                                               // the dh' chunk is complete
        for (int idx = t.tid; idx < MR * 32; idx += TNT) {
            const int rl = idx >> 5, c = (idx & 31) * 4;
            st4(frow(a.dhp, s, row0 + rl) + j * 128 + c, ld4(Hs + rl * LDH + c));
        }
        if (NOT == 1) {
            if (j + 1 < NJ) fetchA(bf0, j + 1, 0);
            if (wc < D) frag_dyw_mma_n<128, MT>(Hs, LDH, bf1, t, accO[0]);
        } else {
            fetchB(bf1, j, 1);
            frag_dyw_mma_n<128, MT>(Hs, LDH, bf0, t, accO[0]);
            if (j + 1 < NJ) fetchA(bf0, j + 1, 0);
            frag_dyw_mma_n<128, MT>(Hs, LDH, bf1, t, accO[NOT - 1]);
        }
        __syncthreads();                                               // ... and consumed
    }
#pragma unroll
    for (int o = 0; o < NOT; ++o) {
        if (o * 128 + wc < D) {
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) Hs[(m * 16 + t.kq * 4 + r4) * LDH + wc] = accO[o][m][r4];
        }
        __syncthreads();
        for (int idx = t.tid; idx < MR * 32; idx += TNT) {
            const int rl = idx >> 5, c = (idx & 31) * 4, cg = o * 128 + c;
            if (cg >= D) continue;
            float4 v = ld4(Hs + rl * LDH + c);
            float* op = frow(a.out, s, row0 + rl) + cg;
            if (a.out_mode == 2) {
                const float4 p = ld4(op);
                v = make_float4(p.x + v.x, p.y + v.y, p.z + v.z, p.w + v.w);
            }
            st4(op, v);
        }
        if (o + 1 < NOT) __syncthreads();
    }
}

// ---- fused backward of a post-LN residual layer's row-local half (round 6) -----------------------------------------------------------
//   ds2 = LN2'(G);  df = ds2 * [f > 0];  dh' = (df W2) * [h > 0];  t = ds2 + dh' W1;  ds1 = LN1'(t);  da = ds1 * [a > 0];  dO = da W_o
//   (the mirror of tl_layer_kernel; loss.backward() through transformer.py:70-78)
// tl_layernorm_bwd + tl_ffn_bwd + tl_layernorm_bwd + tl_mask + tl_dx were five launches per layer, four of them 6 - 15 us long for a
// few MB of traffic each (BASELINE config 4: 70 us of a 332 us chain).  One workgroup per (sequence, 64-row block) -- the granularity of
// the LayerNorm column partials (DtqnNet.sp_parts) -- keeps the block's gradient tile in LDS from the stream gradient entering the layer
// to the attention-output gradient: the two LayerNorm backwards run on the tile with tl_layernorm_bwd_kernel's lane map (8 lanes per
// row, row butterfly over the 8 lanes, column sums over the 8 rows of a wave and then over the waves through LDS, same order of
// additions), the stream gradient G is rewritten in place (ds2 after the first, ds1 after the second: what flows on to the layer's
// input), and df | dh' | da | dO and the d gamma | d beta partials go to the records the weight gradients read.
struct TlChainBwdArgs {
    TlFfnBwdArgs f;                    // f.dy = f.out = G (in place), f.m2, f.df, f.dhp, f.mh, W1 / W2 (+ packed), drop, layer; f.rpb = 64-row blocks per sequence
    Fld s2, st2, s1, st1;              // LayerNorm inputs and (mean, rstd) of the forward
    const float *gamma2, *gamma1;
    float* small;                      // per-(sequence, row block) partial records
    long long small_stride;
    int dgb2_off, dgb1_off;
    Fld m1, da, dO;
    const float *Wo, *Wop;             // W_o [D][D]; its fragment-major B copy or nullptr
};
// LayerNorm backward of the 8 rows of a wave (lane = 8 * row-in-wave + part): y = dL/d(LN output), xp = this lane's piece of the LN input
// row; out: o = dL/d(LN input), and the wave's column sums of (y * xhat, y) in red[wave][2][D]
template <int D>
__device__ __forceinline__ void tl_ln_bwd_rows8(const float4 (&y)[D / 32], const float* xp, float mean, float rstd, const float* gamma, int part,
                                                float* red, const Thr& t, float4 (&o)[D / 32]) {
    constexpr int NV = D / 32;
    float4 xh[NV];
    float c1 = 0.f, c2 = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const float4 x = ld4(xp + 32 * j), gm = ld4(gamma + part * 4 + 32 * j);
        xh[j] = make_float4((x.x - mean) * rstd, (x.y - mean) * rstd, (x.z - mean) * rstd, (x.w - mean) * rstd);
        const float4 g = make_float4(y[j].x * gm.x, y[j].y * gm.y, y[j].z * gm.z, y[j].w * gm.w);
        c1 += (g.x + g.y) + (g.z + g.w);
        c2 += (g.x * xh[j].x + g.y * xh[j].y) + (g.z * xh[j].z + g.w * xh[j].w);
    }
#pragma unroll
    for (int m = 1; m < 8; m <<= 1) { c1 += __shfl_xor(c1, m); c2 += __shfl_xor(c2, m); }
    c1 *= (1.0f / D);
    c2 *= (1.0f / D);
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        float sg[4] = {y[j].x * xh[j].x, y[j].y * xh[j].y, y[j].z * xh[j].z, y[j].w * xh[j].w};
        float sb[4] = {y[j].x, y[j].y, y[j].z, y[j].w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
#pragma unroll
            for (int m = 8; m < 64; m <<= 1) { sg[c] += __shfl_xor(sg[c], m); sb[c] += __shfl_xor(sb[c], m); }
        }
        if (t.lane < 8) {
            st4(red + (t.wave * 2 + 0) * D + part * 4 + 32 * j, make_float4(sg[0], sg[1], sg[2], sg[3]));
            st4(red + (t.wave * 2 + 1) * D + part * 4 + 32 * j, make_float4(sb[0], sb[1], sb[2], sb[3]));
        }
        const float4 gm = ld4(gamma + part * 4 + 32 * j);
        o[j].x = rstd * (y[j].x * gm.x - c1 - xh[j].x * c2);
        o[j].y = rstd * (y[j].y * gm.y - c1 - xh[j].y * c2);
        o[j].z = rstd * (y[j].z * gm.z - c1 - xh[j].z * c2);
        o[j].w = rstd * (y[j].w * gm.w - c1 - xh[j].w * c2);
    }
}
// v where the saved ReLU ballots of row `row`, columns c .. c + 3 are set (word layout of ballot_store), else 0
__device__ __forceinline__ float4 tl_mask4(const unsigned long long* mrec, int ctiles, int row, int c, float4 v) {
    const unsigned long long w = mrec[((row >> 4) * ctiles + (c >> 4)) * 4 + (row & 3)];
    const int b0 = (((row >> 2) & 3) << 4) + (c & 15);
    v.x = ((w >> b0) & 1ull) ? v.x : 0.f;
    v.y = ((w >> (b0 + 1)) & 1ull) ? v.y : 0.f;
    v.z = ((w >> (b0 + 2)) & 1ull) ? v.z : 0.f;
    v.w = ((w >> (b0 + 3)) & 1ull) ? v.w : 0.f;
    return v;
}
template <int D, bool PK>
__global__ __launch_bounds__(TNT, D <= 128 ? 4 : 2) void tl_chain_bwd_kernel(TlChainBwdArgs ca) {
    constexpr int MR = 64, MT = MR / 16, KA = D < 128 ? D : 128, NKA = D / KA, NOT = (D + 127) / 128, HID = 4 * D, NJ = HID / 128, NV = D / 32;
    constexpr int LDX = D + 4, LDH = 128 + 4;
    static_assert((NKA == 1 && NOT == 1) || (NKA == 2 && NOT == 2), "step sequence written for D in {64, 128, 256}");
    const TlFfnBwdArgs& a = ca.f;
    float* Yt = reinterpret_cast<float*>(dtqn_smem);                   // [MR][LDX] df rows, then du -> da rows
    float* Hs = Yt + MR * LDX;                                         // [MR][LDH] dh' chunk / dO staging
    float* red = Hs + MR * LDH;                                        // [TNW][2][D] LayerNorm column sums
    Thr t = make_thr();
    const int s = (int)blockIdx.x / a.rpb, rb = (int)blockIdx.x % a.rpb, row0 = rb * MR;
    int wc = t.wave * 16 + t.i;
    // (thread coordinates re-derived from ONE opaque register at the phase boundaries: nothing per-thread computed before a phase lives through it)
#define TL_CHAIN_RELAUNDER() do { int tid_ = (int)threadIdx.x; DTQN_ASM_KEEP(tid_); t.tid = tid_; t.lane = tid_ & 63; t.wave = tid_ >> 6; t.i = t.lane & 15; t.kq = t.lane >> 4; wc = t.wave * 16 + t.i; } while (0)
    // phase A fragment: W2[n][j * 128 + wc], n over contraction chunk kc (KA rows); phase B: W1[j * 128 + k][ot * 128 + wc], k < 128
    auto fetchA = [&](float (&bf)[32], int j, int kc) {
        if constexpr (PK) {                                            // W2 [D][4D]: contraction over its D rows, output = hidden columns
            wpack_fetch_b(bf, a.W2p, j * 8 + t.wave, NKA, kc, t.lane);
            return;
        }
        const float* wp = a.W2 + (size_t)(kc * KA + t.kq * (KA / 4)) * HID + j * 128 + wc;
#pragma unroll
        for (int q = 0; q < KA / 4; ++q) bf[q] = wp[(size_t)q * HID];
    };
    auto fetchB = [&](float (&bf)[32], int j, int ot) {
        const int col = ot * 128 + wc;
        if constexpr (PK) {                                            // W1 [4D][D]: contraction over its 4D rows (chunk j), output = D columns
            wpack_fetch_b(bf, a.W1p, col < D ? ot * 8 + t.wave : 0, NJ, j, t.lane);
            return;
        }
        const float* wp = a.W1 + (size_t)(j * 128 + t.kq * 32) * D + (col < D ? col : 0);
#pragma unroll
        for (int q = 0; q < 32; ++q) bf[q] = wp[(size_t)q * D];
    };
    // W_o [D][D], dO = da W_o: contraction over its rows (chunk kc of KA), output column ot * 128 + wc
    auto fetchO = [&](float (&bf)[32], int ot, int kc) {
        const int col = ot * 128 + wc;
        if constexpr (PK) {
            wpack_fetch_b(bf, ca.Wop, col < D ? ot * 8 + t.wave : 0, NKA, kc, t.lane);
            return;
        }
        const float* wp = ca.Wo + (size_t)(kc * KA + t.kq * (KA / 4)) * D + (col < D ? col : 0);
#pragma unroll
        for (int q = 0; q < KA / 4; ++q) bf[q] = wp[(size_t)q * D];
    };
    float bf0[32], bf1[32];
    fetchA(bf0, 0, 0);
    float* dgb = ca.small + ((size_t)s * a.rpb + rb) * ca.small_stride;
    // ---------------- LayerNorm-2 backward on the rows of G, the gate's ReLU mask, df ----------------
    {
        const int rl = t.tid >> 3, part = t.tid & 7, row = row0 + rl;
        const float* stp = ca.st2.base + (size_t)s * ca.st2.stride + (size_t)row * 2;
        const float mean = stp[0], rstd = stp[1];
        float* gp = frow(a.dy, s, row) + part * 4;
        float4 y[NV], o[NV];
#pragma unroll
        for (int j = 0; j < NV; ++j) y[j] = ld4(gp + 32 * j);
        tl_ln_bwd_rows8<D>(y, frow(ca.s2, s, row) + part * 4, mean, rstd, ca.gamma2, part, red, t, o);
        const unsigned long long* mrec = reinterpret_cast<const unsigned long long*>(a.m2.base + (size_t)s * a.m2.stride);
        const Drop bdr = tl_drop(a.drop, s);
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int c = part * 4 + 32 * j;
            st4(gp + 32 * j, o[j]);                                    // ds2: the skip path of the residual gate (read back in the epilogue)
            float4 v = tl_mask4(mrec, D / 16, row, c, o[j]);
            if (bdr.thresh != 0u) {
                const uint32_t e0 = (uint32_t)(row * a.drop.dw + c);
                v.x = drop_apply(bdr, DROP_FFN, a.layer, e0, v.x);
                v.y = drop_apply(bdr, DROP_FFN, a.layer, e0 + 1, v.y);
                v.z = drop_apply(bdr, DROP_FFN, a.layer, e0 + 2, v.z);
                v.w = drop_apply(bdr, DROP_FFN, a.layer, e0 + 3, v.w);
            }
            st4(frow(a.df, s, row) + c, v);
            st4(Yt + rl * LDX + c, v);
        }
    }
    __syncthreads();
    for (int idx = t.tid; idx < 2 * D; idx += TNT) {
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < TNW; ++w) v += red[w * 2 * D + idx];
        dgb[ca.dgb2_off + idx] = v;
    }
    // ---------------- feed-forward backward (tl_ffn_bwd_kernel's loop) ----------------
    f32x4 accO[NOT][MT];
#pragma unroll
    for (int o = 0; o < NOT; ++o)
#pragma unroll
        for (int m = 0; m < MT; ++m) accO[o][m] = zero4();
    const unsigned long long* mrec_h = reinterpret_cast<const unsigned long long*>(a.mh.base + (size_t)s * a.mh.stride);
    for (int j = 0; j < NJ; ++j) {
        TL_CHAIN_RELAUNDER();
        f32x4 accA[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) accA[m] = zero4();
        if (NKA == 1) {
            fetchB(bf1, j, 0);
            frag_dyw_mma_n<KA, MT>(Yt, LDX, bf0, t, accA);
        } else {
            fetchA(bf1, j, 1);
            frag_dyw_mma_n<KA, MT>(Yt, LDX, bf0, t, accA);
            fetchB(bf0, j, 0);
            frag_dyw_mma_n<KA, MT>(Yt + KA, LDX, bf1, t, accA);
        }
        {
            const int hc = j * 128 + wc;
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const int rl = m * 16 + t.kq * 4 + r4, row = row0 + rl;
                    const unsigned long long w = mrec_h[((row >> 4) * (HID / 16) + (hc >> 4)) * 4 + r4];
                    Hs[rl * LDH + wc] = ((w >> t.lane) & 1ull) ? accA[m][r4] : 0.f;
                }
        }
        __syncthreads();                                               // the dh' chunk is complete
        for (int idx = t.tid; idx < MR * 32; idx += TNT) {
            const int rl = idx >> 5, c = (idx & 31) * 4;
            st4(frow(a.dhp, s, row0 + rl) + j * 128 + c, ld4(Hs + rl * LDH + c));
        }
        if (NOT == 1) {
            if (j + 1 < NJ) fetchA(bf0, j + 1, 0);
            if (wc < D) frag_dyw_mma_n<128, MT>(Hs, LDH, bf1, t, accO[0]);
        } else {
            fetchB(bf1, j, 1);
            frag_dyw_mma_n<128, MT>(Hs, LDH, bf0, t, accO[0]);
            if (j + 1 < NJ) fetchA(bf0, j + 1, 0);
            frag_dyw_mma_n<128, MT>(Hs, LDH, bf1, t, accO[NOT - 1]);
        }
        __syncthreads();                                               // ... and consumed
    }
    TL_CHAIN_RELAUNDER();
    fetchO(bf0, 0, 0);                                                 // first W_o fragment of the last product: in flight across the LayerNorm rows
    // du into the (spent) df tile
#pragma unroll
    for (int o = 0; o < NOT; ++o) {
        const int col = o * 128 + wc;
        if (col < D) {
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) Yt[(m * 16 + t.kq * 4 + r4) * LDX + col] = accO[o][m][r4];
        }
    }
    __syncthreads();
    // ---------------- dL/du2 = ds2 + du;  LayerNorm-1 backward;  G = ds1;  da = ds1 * [a > 0] (into the tile, in place) ----------------
    TL_CHAIN_RELAUNDER();
    {
        const int rl = t.tid >> 3, part = t.tid & 7, row = row0 + rl;
        const float* stp = ca.st1.base + (size_t)s * ca.st1.stride + (size_t)row * 2;
        const float mean = stp[0], rstd = stp[1];
        float* gp = frow(a.out, s, row) + part * 4;
        float4 y[NV], o[NV];
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const float4 p = ld4(gp + 32 * j), v = ld4(Yt + rl * LDX + part * 4 + 32 * j);
            y[j] = make_float4(p.x + v.x, p.y + v.y, p.z + v.z, p.w + v.w);
        }
        tl_ln_bwd_rows8<D>(y, frow(ca.s1, s, row) + part * 4, mean, rstd, ca.gamma1, part, red, t, o);
        const unsigned long long* mrec = reinterpret_cast<const unsigned long long*>(ca.m1.base + (size_t)s * ca.m1.stride);
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int c = part * 4 + 32 * j;
            st4(gp + 32 * j, o[j]);
            const float4 v = tl_mask4(mrec, D / 16, row, c, o[j]);
            st4(frow(ca.da, s, row) + c, v);
            st4(Yt + rl * LDX + c, v);
        }
    }
    __syncthreads();
    for (int idx = t.tid; idx < 2 * D; idx += TNT) {
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < TNW; ++w) v += red[w * 2 * D + idx];
        dgb[ca.dgb1_off + idx] = v;
    }
    // ---------------- dO = da W_o, one 128-column block after the other ----------------
    TL_CHAIN_RELAUNDER();
#pragma unroll
    for (int ot = 0; ot < NOT; ++ot) {
        f32x4 acc[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[m] = zero4();
        const bool live = ot * 128 + wc < D;
        if (NKA == 1) {
            if (live) frag_dyw_mma_n<KA, MT>(Yt, LDX, bf0, t, acc);
        } else {
            fetchO(bf1, ot, 1);
            frag_dyw_mma_n<KA, MT>(Yt, LDX, bf0, t, acc);
            if (ot + 1 < NOT) fetchO(bf0, ot + 1, 0);
            frag_dyw_mma_n<KA, MT>(Yt + KA, LDX, bf1, t, acc);
        }
        if (ot > 0) __syncthreads();                                   // the staging tile's previous block has left
        if (live) {
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) Hs[(m * 16 + t.kq * 4 + r4) * LDH + wc] = acc[m][r4];
        }
        __syncthreads();
        for (int idx = t.tid; idx < MR * 32; idx += TNT) {
            const int rl = idx >> 5, c = (idx & 31) * 4, cg = ot * 128 + c;
            if (cg >= D) continue;
            st4(frow(ca.dO, s, row0 + rl) + cg, ld4(Hs + rl * LDH + c));
        }
    }
#undef TL_CHAIN_RELAUNDER
}

// ---- attention per (sequence, head) -----------------------------------------------------------------------------
struct TlAttnArgs {
    Fld qkv;                           // [LPB][3D]
    Fld o;                             // [LPB][D]
    Fld lse;                           // [H][LPB] (base may be null)
    int D, lpb, n;
    TlDrop drop;                       // dropout on the attention probabilities (MultiheadAttention(dropout=p), transformer.py:34)
    int layer;
    float hd_eff;                      // head width of the softmax scale: HD, or DtqnNet.hd_real when the heads are zero-padded to HD columns
};
// DROP: compile-time, so that the default build of the kernel carries no keep-mask code
template <int HD, bool DROP>
__global__ __launch_bounds__(256) void tl_attn_kernel(TlAttnArgs a) {
    constexpr int LDH = 3 * HD + 4;
    float* T = reinterpret_cast<float*>(dtqn_smem);                    // [lpb][q | k | v] of this head
    const Thr t = make_thr();
    const int s = (int)blockIdx.x, h = (int)blockIdx.y;
    const float* src = frow(a.qkv, s, 0);
    for (int idx = t.tid; idx < a.lpb * 3 * (HD / 4); idx += 256) {
        const int r = idx / (3 * (HD / 4)), rem = idx - r * (3 * (HD / 4));
        const int which = rem / (HD / 4), c = (rem - which * (HD / 4)) * 4;
        st4(T + r * LDH + which * HD + c, ld4(src + (size_t)r * a.qkv.ld + which * a.D + h * HD + c));
    }
    __syncthreads();
    float* lse = a.lse.base != nullptr ? a.lse.base + (size_t)s * a.lse.stride + (size_t)h * a.lpb : nullptr;
    Drop dr = drop_off();
    if constexpr (DROP) dr = tl_drop(a.drop, s);
    attention_forward<HD, 4>(T, LDH, HD, 1, a.lpb, a.n, lse, t, 0, 0, dr, a.layer, h, a.hd_eff);       // one head: "D" = HD, H = 1
    __syncthreads();
    float* dst = frow(a.o, s, 0) + h * HD;
    for (int idx = t.tid; idx < a.lpb * (HD / 4); idx += 256) {
        const int r = idx / (HD / 4), c = (idx - r * (HD / 4)) * 4;
        st4(dst + (size_t)r * a.o.ld + c, ld4(T + r * LDH + c));
    }
}

// backward: LDS tile [q | k | v | dO] of one head; dq goes straight to the record, dk / dv replace k / v in LDS
struct TlAttnBwdArgs {
    Fld qkv, o, lse;                   // saved by the forward
    Fld dO;                            // [LPB][D] dL/d(attention output)
    Fld dqkv;                          // [LPB][3D] out
    int D, lpb, n;
    TlDrop drop;                       // the forward's attention-probability dropout, recomputed
    int layer;
    float hd_eff;                      // as in TlAttnArgs
};
template <int HD, bool DROP>
__global__ __launch_bounds__(256) void tl_attn_bwd_kernel(TlAttnBwdArgs a) {
    constexpr int LDH = 4 * HD + 4;
    float* T = reinterpret_cast<float*>(dtqn_smem);
    float* delta_s = T + (size_t)a.lpb * LDH;
    float* lse_s = delta_s + a.lpb;
    const Thr t = make_thr();
    const int s = (int)blockIdx.x, h = (int)blockIdx.y;
    const float* src = frow(a.qkv, s, 0);
    const float* dsrc = frow(a.dO, s, 0);
    for (int idx = t.tid; idx < a.lpb * 4 * (HD / 4); idx += 256) {
        const int r = idx / (4 * (HD / 4)), rem = idx - r * (4 * (HD / 4));
        const int which = rem / (HD / 4), c = (rem - which * (HD / 4)) * 4;
        const float* p = which < 3 ? src + (size_t)r * a.qkv.ld + which * a.D + h * HD + c : dsrc + (size_t)r * a.dO.ld + h * HD + c;
        st4(T + r * LDH + which * HD + c, ld4(p));
    }
    __syncthreads();
    {
        const float* og = frow(a.o, s, 0) + h * HD;
        const float* lg = a.lse.base + (size_t)s * a.lse.stride + (size_t)h * a.lpb;
        for (int r = t.tid; r < a.lpb; r += 256) {
            float p = 0.f;
#pragma unroll
            for (int c = 0; c < HD; c += 4) {
                const float4 ov = ld4(og + (size_t)r * a.o.ld + c), dv = ld4(T + r * LDH + 3 * HD + c);
                p = fmaf(ov.x, dv.x, p); p = fmaf(ov.y, dv.y, p); p = fmaf(ov.z, dv.z, p); p = fmaf(ov.w, dv.w, p);
            }
            delta_s[r] = p;
            lse_s[r] = lg[r];
        }
    }
    __syncthreads();
    float* dq = frow(a.dqkv, s, 0) + h * HD;
    Drop dr = drop_off();
    if constexpr (DROP) dr = tl_drop(a.drop, s);
    attention_backward_group<HD, 4>(T, LDH, HD, a.lpb, a.n, delta_s, lse_s, t, dq, a.dqkv.ld, 0, 0, dr, a.layer, h, a.hd_eff);
    __syncthreads();
    for (int idx = t.tid; idx < a.lpb * 2 * (HD / 4); idx += 256) {
        const int r = idx / (2 * (HD / 4)), rem = idx - r * (2 * (HD / 4));
        const int which = 1 + rem / (HD / 4), c = (rem % (HD / 4)) * 4;
        st4(dq + (size_t)r * a.dqkv.ld + which * a.D + c, ld4(T + r * LDH + which * HD + c));
    }
}

// ---- LayerNorm over 64-row blocks -----------------------------------------------------------------------------------
struct TlLnArgs {
    Fld src, dst, st;                  // st: (mean, rstd) per row, base may be null
    const float *ga, *gb, *ba, *bb;    // gamma / beta, sequences >= split use the second set
    int split, rpb;
    int d_real;                        // width-padded network (DtqnNet.d_real): statistics over the first d_real columns; 0 = all D
};
template <int D, bool PAD>
__global__ __launch_bounds__(TNT) void tl_layernorm_kernel(TlLnArgs a) {
    const Thr t = make_thr();
    const int s = (int)blockIdx.x / a.rpb, row0 = ((int)blockIdx.x % a.rpb) * TROWS;
    float* st = a.st.base != nullptr ? a.st.base + (size_t)s * a.st.stride + (size_t)row0 * 2 : nullptr;
    layernorm_rows<D, TNW, DTQN_MAX_LP, true, PAD>(frow(a.src, s, row0), frow(a.dst, s, row0), a.src.ld, TROWS, s >= a.split ? a.gb : a.ga,
                                                   s >= a.split ? a.bb : a.ba, st, t, nullptr, nullptr, a.dst.ld, PAD ? a.d_real : D);
}

// backward, one workgroup per (sequence, 64-row block): 8 lanes per row, each with D / 32 float4 of the row in registers.
// The row statistics meet in a butterfly over the 8 lanes; the column sums (d gamma = sum_r dy * xhat, d beta = sum_r dy)
// over the 8 rows of a wave in a butterfly over the lane bits 3..5, then over the 8 waves through LDS, and go to the
// small-partial record of THIS row block (DtqnNet.sp_parts records per sequence; the reducers of dtqn_wgrad.hip add them
// up in a fixed order).  dst may alias dy (every element is read and written by the same lane); accumulate: dst +=
// dL/d(LN input) (identity-reordered layers, where the LayerNorm sits on the branch and the stream gradient passes by it).
struct TlLnBwdArgs {
    Fld dy, xin, st, dst;
    const float* gamma;
    float* small;                      // per-(sequence, row block) partial records
    long long small_stride;
    int dgb_off;
    int rpb;
    int accumulate;
    int d_real;                        // DtqnNet.d_real (0 = not padded)
};
// PAD: width-padded network -- the two row means run over the d_real real columns, and the padded columns of dx are 0 (their dy and gamma
// are zero, but -c1 - xhat c2 is not: left in, it would reach the padded rows of every weight gradient below)
template <int D, bool PAD>
__global__ __launch_bounds__(TNT) void tl_layernorm_bwd_kernel(TlLnBwdArgs a) {
    constexpr int NV = D / 32;                                         // float4 per lane: columns part * 4 + 32 * j
    float* red = reinterpret_cast<float*>(dtqn_smem);                  // [TNW][2][D]
    const Thr t = make_thr();
    const int s = (int)blockIdx.x / a.rpb, rb = (int)blockIdx.x % a.rpb, row = rb * TROWS + (t.tid >> 3), part = t.tid & 7;
    const float* stp = a.st.base + (size_t)s * a.st.stride + (size_t)row * 2;
    const float mean = stp[0], rstd = stp[1];
    const float* yp = frow(a.dy, s, row) + part * 4;
    const float* xp = frow(a.xin, s, row) + part * 4;
    float4 y[NV], xh[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) { y[j] = ld4(yp + 32 * j); xh[j] = ld4(xp + 32 * j); }
    float c1 = 0.f, c2 = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const float4 gm = ld4(a.gamma + part * 4 + 32 * j);
        xh[j] = make_float4((xh[j].x - mean) * rstd, (xh[j].y - mean) * rstd, (xh[j].z - mean) * rstd, (xh[j].w - mean) * rstd);
        const float4 g = make_float4(y[j].x * gm.x, y[j].y * gm.y, y[j].z * gm.z, y[j].w * gm.w);
        c1 += (g.x + g.y) + (g.z + g.w);
        c2 += (g.x * xh[j].x + g.y * xh[j].y) + (g.z * xh[j].z + g.w * xh[j].w);
    }
#pragma unroll
    for (int m = 1; m < 8; m <<= 1) { c1 += __shfl_xor(c1, m); c2 += __shfl_xor(c2, m); }
    const float inv_d = PAD ? 1.0f / (float)a.d_real : (1.0f / D);
    c1 *= inv_d;
    c2 *= inv_d;
    // column sums over the 8 rows of this wave
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        float sg[4] = {y[j].x * xh[j].x, y[j].y * xh[j].y, y[j].z * xh[j].z, y[j].w * xh[j].w};
        float sb[4] = {y[j].x, y[j].y, y[j].z, y[j].w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
#pragma unroll
            for (int m = 8; m < 64; m <<= 1) { sg[c] += __shfl_xor(sg[c], m); sb[c] += __shfl_xor(sb[c], m); }
        }
        if (t.lane < 8) {
            st4(red + (t.wave * 2 + 0) * D + part * 4 + 32 * j, make_float4(sg[0], sg[1], sg[2], sg[3]));
            st4(red + (t.wave * 2 + 1) * D + part * 4 + 32 * j, make_float4(sb[0], sb[1], sb[2], sb[3]));
        }
    }
    float* dp = frow(a.dst, s, row) + part * 4;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const float4 gm = ld4(a.gamma + part * 4 + 32 * j);
        float4 o;
        o.x = rstd * (y[j].x * gm.x - c1 - xh[j].x * c2);
        o.y = rstd * (y[j].y * gm.y - c1 - xh[j].y * c2);
        o.z = rstd * (y[j].z * gm.z - c1 - xh[j].z * c2);
        o.w = rstd * (y[j].w * gm.w - c1 - xh[j].w * c2);
        if constexpr (PAD) {
            const int c0 = part * 4 + 32 * j;
            o.x = c0 < a.d_real ? o.x : 0.f; o.y = c0 + 1 < a.d_real ? o.y : 0.f; o.z = c0 + 2 < a.d_real ? o.z : 0.f; o.w = c0 + 3 < a.d_real ? o.w : 0.f;
        }
        if (a.accumulate) {
            const float4 p = ld4(dp + 32 * j);
            o = make_float4(p.x + o.x, p.y + o.y, p.z + o.z, p.w + o.w);
        }
        st4(dp + 32 * j, o);
    }
    __syncthreads();
    float* dgb = a.small + ((size_t)s * a.rpb + rb) * a.small_stride + a.dgb_off;
    for (int idx = t.tid; idx < 2 * D; idx += TNT) {
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < TNW; ++w) v += red[w * 2 * D + idx];
        dgb[idx] = v;
    }
}

// ---- dst = src where the saved ReLU ballot is set, else 0 (gate backward) ---------------------------------------------
struct TlMaskArgs {
    Fld src, dst, mask;
    int D, rpb;
};
__global__ __launch_bounds__(TNT) void tl_mask_kernel(TlMaskArgs a) {
    const int s = (int)blockIdx.x / a.rpb, row0 = ((int)blockIdx.x % a.rpb) * TROWS;
    const float* mrec = a.mask.base + (size_t)s * a.mask.stride;
    const int c4 = a.D / 4;
    for (int idx = (int)threadIdx.x; idx < TROWS * c4; idx += TNT) {
        const int r = row0 + idx / c4, c = (idx % c4) * 4;
        const float4 v = ld4(frow(a.src, s, r) + c);
        float4 o;
        o.x = mask_bit(mrec, a.D / 16, r, c) ? v.x : 0.f;
        o.y = mask_bit(mrec, a.D / 16, r, c + 1) ? v.y : 0.f;
        o.z = mask_bit(mrec, a.D / 16, r, c + 2) ? v.z : 0.f;
        o.w = mask_bit(mrec, a.D / 16, r, c + 3) ? v.w : 0.f;
        st4(frow(a.dst, s, r) + c, o);
    }
}

// ---- x *= keep mask of a dropout site, in place (backward: gradients of x0 / of the feed-forward output where no GEMM kernel
//      carries the mask in its staging) -------------------------------------------------------------------------------------
struct TlDropRowsArgs {
    Fld x;
    int D, rpb, site, layer;
    TlDrop drop;
};
__global__ __launch_bounds__(TNT) void tl_drop_rows_kernel(TlDropRowsArgs a) {
    const int s = (int)blockIdx.x / a.rpb, row0 = ((int)blockIdx.x % a.rpb) * TROWS;
    const Drop dr = tl_drop(a.drop, s);
    if (dr.thresh == 0u) return;
    const int c4 = a.D / 4;
    for (int idx = (int)threadIdx.x; idx < TROWS * c4; idx += TNT) {
        const int r = row0 + idx / c4, c = (idx % c4) * 4;
        float* p = frow(a.x, s, r) + c;
        float4 v = ld4(p);
        const uint32_t e0 = (uint32_t)(r * a.drop.dw + c);
        v.x = drop_apply(dr, a.site, a.layer, e0, v.x);
        v.y = drop_apply(dr, a.site, a.layer, e0 + 1, v.y);
        v.z = drop_apply(dr, a.site, a.layer, e0 + 2, v.z);
        v.w = drop_apply(dr, a.site, a.layer, e0 + 3, v.w);
        st4(p, v);
    }
}

// ---- GRU gate backward, elementwise parts (the matrix products run through tl_dx_kernel) ------------------------------------
//   step 1: dz_pre = g (h - x) z (1 - z);  dh_pre = g z (1 - h^2);  g <- g (1 - z)              (g = dL/d(gate output), in place)
//   step 2: t = dh_pre U_g (given);  dr_pre = t x r (1 - r);  g += t r
struct TlGateBwdArgs {
    Fld g, t;                          // stream gradient; step 2: d(r * x)
    Fld z, r, h, x;                    // the gate's saved record
    Fld dz, dr, dh;                    // its gradient record (pre-activations)
    int D, rpb, step;
};
__global__ __launch_bounds__(TNT) void tl_gate_bwd_kernel(TlGateBwdArgs a) {
    const int s = (int)blockIdx.x / a.rpb, row0 = ((int)blockIdx.x % a.rpb) * TROWS;
    const int c4 = a.D / 4;
    for (int idx = (int)threadIdx.x; idx < TROWS * c4; idx += TNT) {
        const int r = row0 + idx / c4, c = (idx % c4) * 4;
        float* gp = frow(a.g, s, r) + c;
        const float4 g4 = ld4(gp), x4 = ld4(frow(a.x, s, r) + c);
        const float g[4] = {g4.x, g4.y, g4.z, g4.w}, x[4] = {x4.x, x4.y, x4.z, x4.w};
        float o0[4], o1[4], o2[4];
        if (a.step == 1) {
            const float4 z4 = ld4(frow(a.z, s, r) + c), h4 = ld4(frow(a.h, s, r) + c);
            const float z[4] = {z4.x, z4.y, z4.z, z4.w}, h[4] = {h4.x, h4.y, h4.z, h4.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                o0[q] = g[q] * (h[q] - x[q]) * z[q] * (1.0f - z[q]);
                o1[q] = g[q] * z[q] * (1.0f - h[q] * h[q]);
                o2[q] = g[q] * (1.0f - z[q]);
            }
            st4(frow(a.dz, s, r) + c, make_float4(o0[0], o0[1], o0[2], o0[3]));
            st4(frow(a.dh, s, r) + c, make_float4(o1[0], o1[1], o1[2], o1[3]));
            st4(gp, make_float4(o2[0], o2[1], o2[2], o2[3]));
        } else {
            const float4 t4 = ld4(frow(a.t, s, r) + c), r4 = ld4(frow(a.r, s, r) + c);
            const float tv[4] = {t4.x, t4.y, t4.z, t4.w}, rr[4] = {r4.x, r4.y, r4.z, r4.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                o0[q] = tv[q] * x[q] * rr[q] * (1.0f - rr[q]);
                o2[q] = g[q] + tv[q] * rr[q];
            }
            st4(frow(a.dr, s, r) + c, make_float4(o0[0], o0[1], o0[2], o0[3]));
            st4(gp, make_float4(o2[0], o2[1], o2[2], o2[3]));
        }
    }
}

// ---- Q = HH W2^T + b2 ---------------------------------------------------------------------------------------------------
struct TlQArgs {
    Fld hh;
    const float *W2a, *W2b, *b2a, *b2b;
    int split;
    float* q;
    long long q_seq_stride;
    int q_row_stride;
    int D, A, n, S;
};
// 16 lanes per token row: lane c holds columns 4c + 64j of the hidden row (one coalesced read of the row per 16 lanes), multiplies
// them with the same columns of every action's weight row (a few KB, cache resident) and the 16 partials meet in a butterfly.
__global__ __launch_bounds__(256) void tl_qhead_kernel(TlQArgs a) {
    const int rows = a.S * a.n, c = (int)threadIdx.x & 15, nj = a.D / 64;
    for (int g = (int)(blockIdx.x * 16 + (threadIdx.x >> 4)); g < (rows + 15) / 16 * 16; g += (int)gridDim.x * 16) {
        const bool live = g < rows;                                   // dead row groups still take part in the shuffles
        const int gs = live ? g : 0, s = gs / a.n, r = gs - s * a.n;
        const float* hrow = frow(a.hh, s, r) + 4 * c;
        float4 hv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) hv[j] = j < nj ? ld4(hrow + 64 * j) : make_float4(0.f, 0.f, 0.f, 0.f);
        const float* W = (s >= a.split ? a.W2b : a.W2a) + 4 * c;
        const float* bias = s >= a.split ? a.b2b : a.b2a;
        float* qrow = a.q + (size_t)s * a.q_seq_stride + (size_t)r * a.q_row_stride;
        for (int ac = 0; ac < a.A; ++ac) {
            float p = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (j < nj) {
                    const float4 wv = ld4(W + (size_t)ac * a.D + 64 * j);
                    p = fmaf(hv[j].x, wv.x, p); p = fmaf(hv[j].y, wv.y, p); p = fmaf(hv[j].z, wv.z, p); p = fmaf(hv[j].w, wv.w, p);
                }
#pragma unroll
            for (int m = 8; m >= 1; m >>= 1) p += __shfl_xor(p, m);
            if (live && c == (ac & 15)) qrow[ac] = p + bias[ac];
        }
    }
}

// ---- TD loss (one workgroup per sampled sequence) and the VALU half of the Q-head backward ----------------------
struct TlLossArgs {
    DtqnNet net;
    const float* q3;
    float* grd;
    float* stats_partial;
    const uint8_t* actions;
    const float* rewards;
    const uint8_t* dones;
    long long act_ep_stride, rew_ep_stride;
    const int32_t* ep_idx;
    const int32_t* start;
    int batch, history;
    float gamma;
};
__global__ __launch_bounds__(256) void tl_loss_kernel(TlLossArgs a) {
    const DtqnNet& net = a.net;
    const int b = (int)blockIdx.x, LP = net.lp, AP = net.ap;
    float* dq = a.grd + (size_t)b * net.grd_stride + net.go_dq;
    for (int idx = (int)threadIdx.x; idx < LP * AP; idx += 256) dq[idx] = 0.f;
    __syncthreads();
    if (threadIdx.x < 64) {
        const int ep = a.ep_idx[b], st0 = a.start[b];
        const size_t qs = (size_t)LP * AP;
        td_loss_wave(a.q3 + ((size_t)0 * a.batch + b) * qs, a.q3 + ((size_t)1 * a.batch + b) * qs, a.q3 + ((size_t)2 * a.batch + b) * qs,
                     AP, net.num_actions, net.ctx_len, LP, a.history, a.gamma, 1.0f / ((float)a.batch * (float)a.history),
                     a.actions + (size_t)ep * a.act_ep_stride + st0, a.rewards + (size_t)ep * a.rew_ep_stride + st0,
                     a.dones + (size_t)ep * a.rew_ep_stride + st0, dq, a.stats_partial + (size_t)b * 8, (int)threadIdx.x);
    }
}
// dhh = (dq W2) * [hh > 0]
struct TlHeadBwdArgs {
    Fld hh, dq, dhh;
    const float* W2;
    int D, A, rpb;
};
__global__ __launch_bounds__(TNT) void tl_head_bwd_kernel(TlHeadBwdArgs a) {
    const int s = (int)blockIdx.x / a.rpb, row0 = ((int)blockIdx.x % a.rpb) * TROWS;
    const int c4 = a.D / 4;
    for (int idx = (int)threadIdx.x; idx < TROWS * c4; idx += TNT) {
        const int r = row0 + idx / c4, c0 = (idx % c4) * 4;
        const float4 hv4 = ld4(frow(a.hh, s, r) + c0);
        const float hv[4] = {hv4.x, hv4.y, hv4.z, hv4.w};
        const float* dqr = frow(a.dq, s, r);
        float g4[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float g = 0.f;
            if (hv[e] > 0.f)
                for (int c = 0; c < a.A; ++c) g = fmaf(dqr[c], a.W2[(size_t)c * a.D + c0 + e], g);
            g4[e] = g;
        }
        st4(frow(a.dhh, s, r) + c0, make_float4(g4[0], g4[1], g4[2], g4[3]));
    }
}

// ---- embedding backward: table / action-embedding partials of one sequence ------------------------------------------
struct TlEmbedBwdArgs {
    DtqnNet net;
    const float* theta;
    const float* grd;
    float* small;
    const float* obs;
    const uint8_t* actions;
    long long obs_ep_stride, act_ep_stride;
    const int32_t* ep_idx;
    const int32_t* start;
    // bag entries (second pass over the same partials): gradient at grd + dx_off instead of go_dx0, `rows` live rows read from
    // source sequence b of (obs, actions) directly, action embedding not rolled, partials ADDED to the context's
    int bag, dx_off, rows;
};
// One workgroup per (sequence, 64-row block); its partials go to the small-partial record of that row block.
// Discrete observations: d(e_in) [64][KE] = dx0[:, a:] [64][DO] * W_e [DO][KE] on the matrix cores (contraction in chunks of 64
// columns through LDS, zero padded to multiples of 4 / 16), then the scatter onto table rows: thread (k = j * e + c, row
// group) adds its rows into a private [token][c] column of an LDS table (no two threads share an address), and the
// O * (row groups) private tables are summed in a fixed order at the end -- deterministic, no atomics.  The tokens and
// actions of the block are staged in LDS once (no global load inside the scatter loops).
// row groups of a 64-row block walked by different threads of the scatter: 4 if the thread count allows, else 2 or 1
static __host__ __device__ inline int emb_row_groups(int ke) { return ke * 4 <= TNT ? 4 : ke * 2 <= TNT ? 2 : 1; }
constexpr int kEbKC = 64, kEbLDT = kEbKC + 4;
static __host__ __device__ inline int emb_bwd_ldk(int ke) {            // leading dim of the W_e chunk / d(e_in): >= KE16, == 16 mod 64
    const int ke16 = (ke + 15) & ~15;
    return ((ke16 - 16 + 63) / 64) * 64 + 16;
}
static inline size_t tl_embed_bwd_lds(const DtqnNet& net) {
    const int ldk = emb_bwd_ldk(net.ke);
    size_t f = (size_t)TROWS * (net.action_dim + 1) + TROWS + 4;                           // action columns of dx0, actions
    if (net.discrete)
        f += (size_t)TROWS * kEbLDT + (size_t)kEbKC * ldk + (size_t)TROWS * ldk + (size_t)TROWS * net.obs_dim +
             (size_t)emb_row_groups(net.ke) * net.obs_dim * net.vocab * net.embed_per_obs;
    return f * sizeof(float);
}
__global__ __launch_bounds__(TNT) void tl_embed_bwd_kernel(TlEmbedBwdArgs a) {
    const DtqnNet& net = a.net;
    const Thr t = make_thr();
    const int tid = t.tid;
    const int rpb = net.lp / TROWS, b = (int)blockIdx.x / rpb, rb = (int)blockIdx.x % rpb, row0 = rb * TROWS;
    const int D = net.d_model, L = a.bag ? a.rows : net.ctx_len, A = net.num_actions, adim = net.action_dim;
    if (a.bag && row0 >= L) return;                                    // bag pass: nothing of the bag in this row block
    const float* DX = a.grd + (size_t)b * net.grd_stride + (a.bag ? a.dx_off : net.go_dx0) + (size_t)row0 * D;
    float* srec = a.small + ((size_t)b * rpb + rb) * net.sp_stride;
    const int ep = a.bag ? b : a.ep_idx[b], st0 = a.bag ? 0 : a.start[b];
    const float* obs_rows = a.obs + (size_t)ep * a.obs_ep_stride + (size_t)(st0 + row0) * net.obs_dim;
    const uint8_t* act_rows = a.actions + (size_t)ep * a.act_ep_stride + st0;
    const int nrows = L - row0 < TROWS ? L - row0 : TROWS;             // live rows of this block
    float* Al = reinterpret_cast<float*>(dtqn_smem);                   // [64][adim + 1]  action columns of dx0
    int* actl = reinterpret_cast<int*>(Al + TROWS * (adim + 1));       // [64 + 1] the action that pairs with row r (see below)
    float* rest = reinterpret_cast<float*>(actl + TROWS + 4);
    if (adim > 0) {
        for (int idx = tid; idx < TROWS * adim; idx += TNT) {
            const int rl = idx / adim, c = idx - rl * adim;
            Al[rl * (adim + 1) + c] = rl < nrows ? DX[(size_t)rl * D + c] : 0.f;
        }
        // row r carries the embedding of: its own action (bag entries), the action of row 0 (a single-row forward), else the
        // action of row r - 1 with nothing at r = 0 (the roll of dtqn.py:184-192)
        for (int rl = tid; rl < TROWS; rl += TNT) {
            const int r = row0 + rl;
            int v = -1;
            if (rl < nrows) {
                if (a.bag) v = (int)act_rows[r];
                else if (L == 1) v = (int)act_rows[0];
                else if (r > 0) v = (int)act_rows[r - 1];
            }
            actl[rl] = v;
        }
    }
    if (net.discrete) {
        const int KE = net.ke, e = net.embed_per_obs, V = net.vocab, O = net.obs_dim, DO = D - adim;
        const int LDK = emb_bwd_ldk(KE), kEmbRQ = emb_row_groups(KE), NT16 = (KE + 15) / 16;
        float* Tl = rest;                                      // [64][kEbLDT]  chunk of dx0[:, a:]
        float* Wl = Tl + TROWS * kEbLDT;                       // [64][LDK]     chunk of W_e (row = contraction index)
        float* dein = Wl + (size_t)kEbKC * LDK;                // [64][LDK]     dL/d(gathered table rows)
        int* tokl = reinterpret_cast<int*>(dein + (size_t)TROWS * LDK);   // [64][O]
        float* part = reinterpret_cast<float*>(tokl + TROWS * O);         // [kEmbRQ][O][V][e] private scatter tables
        const float* __restrict__ We = a.theta + net.off_obs_w;
        for (int idx = tid; idx < kEmbRQ * O * V * e; idx += TNT) part[idx] = 0.f;
        for (int idx = tid; idx < TROWS * O; idx += TNT) {
            const int rl = idx / O;
            int tok = rl < nrows ? (int)obs_rows[idx] : 0;
            tokl[idx] = tok < 0 ? 0 : (tok >= V ? V - 1 : tok);
        }
        // items (16-row tile mt, 16-column tile nt) dealt to the waves; at most 2 per wave for KE <= 64, else serial loops
        const int items = 4 * NT16;
        f32x4 acc[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = zero4();
        for (int k0 = 0; k0 < DO; k0 += kEbKC) {
            const int kc = DO - k0 < kEbKC ? DO - k0 : kEbKC, kc4 = (kc + 3) & ~3;
            __syncthreads();                                   // previous chunk consumed
            for (int idx = tid; idx < TROWS * kc4; idx += TNT) {
                const int rl = idx / kc4, kl = idx - rl * kc4;
                Tl[rl * kEbLDT + kl] = (rl < nrows && kl < kc) ? DX[(size_t)rl * D + adim + k0 + kl] : 0.f;
            }
            for (int idx = tid; idx < kc4 * NT16 * 16; idx += TNT) {
                const int kl = idx / (NT16 * 16), k = idx - kl * (NT16 * 16);
                Wl[kl * LDK + k] = (kl < kc && k < KE) ? We[(size_t)(k0 + kl) * KE + k] : 0.f;
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int item = t.wave + q * TNW;
                if (item < items) {
                    const int nt = item >> 2, mt = item & 3;
                    const float* ap = Tl + (mt * 16 + t.i) * kEbLDT + t.kq;
                    const float* bp = Wl + t.kq * LDK + nt * 16 + t.i;
                    for (int k = 0; k < kc4; k += 4) acc[q] = mfma16(ap[k], bp[(size_t)k * LDK], acc[q]);
                }
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int item = t.wave + q * TNW;
            if (item < items) {
                const int nt = item >> 2, mt = item & 3;
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) dein[(mt * 16 + t.kq * 4 + r4) * LDK + nt * 16 + t.i] = acc[q][r4];
            }
        }
        __syncthreads();
        if (tid < KE * kEmbRQ) {
            const int k = tid % KE, rq = tid / KE, jj = k / e, c = k - jj * e;
            float* mine = part + ((size_t)(rq * O + jj) * V) * e + c;
            for (int rl = rq * (TROWS / kEmbRQ); rl < (rq + 1) * (TROWS / kEmbRQ); ++rl)
                if (rl < nrows) mine[tokl[rl * O + jj] * e] += dein[rl * LDK + k];
        }
        __syncthreads();
        for (int idx = tid; idx < V * e; idx += TNT) {
            float g = 0.f;
            for (int q = 0; q < kEmbRQ * O; ++q) g += part[(size_t)q * V * e + idx];
            srec[net.so_tab + idx] = a.bag ? srec[net.so_tab + idx] + g : g;
        }
    } else {
        __syncthreads();
    }
    if (adim > 0) {
        for (int idx = tid; idx < A * adim; idx += TNT) {
            const int v = idx / adim, c = idx - v * adim;
            float g = 0.f;
            for (int rl = 0; rl < nrows; ++rl)
                if (actl[rl] == v) g += Al[rl * (adim + 1) + c];
            srec[net.so_act + idx] = a.bag ? srec[net.so_act + idx] + g : g;
        }
    }
}

// ---- bag attention (dtqn.py:211-213): nn.MultiheadAttention(query = working memory, key = value = bag embeddings), no mask ----
// One workgroup per (sequence, head); the bag is at most a few dozen entries, so this is plain VALU code out of LDS.
struct TlBagAttnArgs {
    Fld q;                             // [LPB][D]   W_q xf + b_q
    Fld kv;                            // [LPB][2D]  k | v of the bag entries (rows < bag)
    Fld p;                             // [H][LPB][bag_ld] attention weights (training; base may be null)
    Fld o;                             // [LPB][D]   attention output (before the out-projection)
    int D, HD, n, bag, bag_ld, lpb;
    TlDrop drop;                       // dropout on the attention weights (nn.MultiheadAttention(dropout=p), dtqn.py:136-141): site DROP_BAG
};
__global__ __launch_bounds__(256) void tl_bag_attn_kernel(TlBagAttnArgs a) {
    float* kl = reinterpret_cast<float*>(dtqn_smem);                   // [bag][HD] keys, then [bag][HD] values
    const int s = (int)blockIdx.x, h = (int)blockIdx.y, tid = (int)threadIdx.x, HD = a.HD;
    float* vl = kl + a.bag * HD;
    for (int idx = tid; idx < a.bag * HD; idx += 256) {
        const int j = idx / HD, c = idx - j * HD;
        const float* row = frow(a.kv, s, j);
        kl[idx] = row[h * HD + c];
        vl[idx] = row[a.D + h * HD + c];
    }
    __syncthreads();
    const float scale = 1.0f / sqrtf((float)HD);
    const Drop dr = tl_drop(a.drop, s);
    for (int t = tid; t < a.n; t += 256) {
        const float* qr = frow(a.q, s, t) + h * HD;
        float m = -INFINITY;
        for (int j = 0; j < a.bag; ++j) {
            float sc = 0.f;
            for (int c = 0; c < HD; ++c) sc = fmaf(qr[c] * scale, kl[j * HD + c], sc);
            m = fmaxf(m, sc);
        }
        float l = 0.f;
        for (int j = 0; j < a.bag; ++j) {
            float sc = 0.f;
            for (int c = 0; c < HD; ++c) sc = fmaf(qr[c] * scale, kl[j * HD + c], sc);
            l += __expf(sc - m);
        }
        float* orow = frow(a.o, s, t) + h * HD;
        for (int c = 0; c < HD; ++c) orow[c] = 0.f;
        float* prow = a.p.base != nullptr ? a.p.base + (size_t)s * a.p.stride + ((size_t)h * a.lpb + t) * a.bag_ld : nullptr;
        for (int j = 0; j < a.bag; ++j) {
            float sc = 0.f;
            for (int c = 0; c < HD; ++c) sc = fmaf(qr[c] * scale, kl[j * HD + c], sc);
            const float pj = __expf(sc - m) / l;
            if (prow != nullptr) prow[j] = pj;                         // the backward wants the weights BEFORE dropout
            const float pd = drop_apply(dr, DROP_BAG, 0, drop_attn_idx(h, t, j), pj);
            for (int c = 0; c < HD; ++c) orow[c] = fmaf(pd, vl[j * HD + c], orow[c]);
        }
    }
}
// backward: dP = dO v^T, dS = P (dP - sum_j P dP), dq = scale dS k, dk = scale dS^T q, dv = P^T dO
struct TlBagAttnBwdArgs {
    Fld q, kv, p, dO;                  // dO: [LPB][D] gradient of the attention output
    Fld dq;                            // [LPB][D]
    Fld dkv;                           // [LPB][2D] (rows < bag written)
    int D, HD, n, bag, bag_ld, lpb;
    TlDrop drop;
};
__global__ __launch_bounds__(256) void tl_bag_attn_bwd_kernel(TlBagAttnBwdArgs a) {
    float* kl = reinterpret_cast<float*>(dtqn_smem);                   // [bag][HD] k, [bag][HD] v, then [n][bag] dS
    const int s = (int)blockIdx.x, h = (int)blockIdx.y, tid = (int)threadIdx.x, HD = a.HD;
    float* vl = kl + a.bag * HD;
    float* dsl = vl + a.bag * HD;
    for (int idx = tid; idx < a.bag * HD; idx += 256) {
        const int j = idx / HD, c = idx - j * HD;
        const float* row = frow(a.kv, s, j);
        kl[idx] = row[h * HD + c];
        vl[idx] = row[a.D + h * HD + c];
    }
    __syncthreads();
    const float scale = 1.0f / sqrtf((float)HD);
    const Drop dr = tl_drop(a.drop, s);
    for (int t = tid; t < a.n; t += 256) {
        const float* dor = frow(a.dO, s, t) + h * HD;
        const float* prow = a.p.base + (size_t)s * a.p.stride + ((size_t)h * a.lpb + t) * a.bag_ld;
        float dsum = 0.f;
        for (int j = 0; j < a.bag; ++j) {
            float dp = 0.f;
            for (int c = 0; c < HD; ++c) dp = fmaf(dor[c], vl[j * HD + c], dp);
            dp = drop_apply(dr, DROP_BAG, 0, drop_attn_idx(h, t, j), dp);     // d(weights) = keep mask * d(dropped weights)
            dsl[t * a.bag + j] = dp;
            dsum = fmaf(prow[j], dp, dsum);
        }
        float* dqr = frow(a.dq, s, t) + h * HD;
        for (int c = 0; c < HD; ++c) dqr[c] = 0.f;
        for (int j = 0; j < a.bag; ++j) {
            const float ds = prow[j] * (dsl[t * a.bag + j] - dsum);
            dsl[t * a.bag + j] = ds;
            for (int c = 0; c < HD; ++c) dqr[c] = fmaf(ds * scale, kl[j * HD + c], dqr[c]);
        }
    }
    __syncthreads();
    for (int idx = tid; idx < a.bag * HD; idx += 256) {
        const int j = idx / HD, c = idx - j * HD;
        float dk = 0.f, dv = 0.f;
        for (int t = 0; t < a.n; ++t) {
            const float pj = drop_apply(dr, DROP_BAG, 0, drop_attn_idx(h, t, j),
                                        a.p.base[(size_t)s * a.p.stride + ((size_t)h * a.lpb + t) * a.bag_ld + j]);    // dv takes the dropped weights
            dk = fmaf(dsl[t * a.bag + j] * scale, frow(a.q, s, t)[h * HD + c], dk);
            dv = fmaf(pj, frow(a.dO, s, t)[h * HD + c], dv);
        }
        float* row = frow(a.dkv, s, j);
        row[h * HD + c] = dk;
        row[a.D + h * HD + c] = dv;
    }
}
// dst[rows][cols] = src[rows][cols] (any leading dims): the working-memory half of d xcat -> the stream gradient
struct TlCopyArgs {
    Fld src, dst;
    int cols, rpb;
};
__global__ __launch_bounds__(TNT) void tl_copy_kernel(TlCopyArgs a) {
    const int s = (int)blockIdx.x / a.rpb, row0 = ((int)blockIdx.x % a.rpb) * TROWS;
    const int c4 = a.cols / 4;
    for (int idx = (int)threadIdx.x; idx < TROWS * c4; idx += TNT) {
        const int r = idx / c4, c = (idx - r * c4) * 4;
        st4(frow(a.dst, s, row0 + r) + c, ld4(frow(a.src, s, row0 + r) + c));
    }
}

// ---- launch helpers --------------------------------------------------------------------------------------------------------
// DTQN_TL_TRACE=1 (tests): one line per launch on stderr, so a test can tell WHICH kernels an update went through
static inline bool tl_trace_on() {
    const char* e = getenv("DTQN_TL_TRACE");                            // (read per launch: a test switches it on for one case)
    return e != nullptr && e[0] == '1';
}
#define TL_LAUNCH(kernel, grid, block, lds, stream, args)                                                            \
    do {                                                                                                             \
        if (tl_trace_on()) fprintf(stderr, "tl_launch %s\n", #kernel);                                               \
        if ((lds) > 48 * 1024) {            /* beyond the default dynamic-LDS limit: raise it once per kernel and device */   \
            static size_t tl_attr_lds_[kMaxDevices] = {};                                                            \
            raise_lds_limit(reinterpret_cast<const void*>(&kernel), (lds), tl_attr_lds_);                               \
        }                                                                                                            \
        (void)hipGetLastError();                                                                                     \
        hipLaunchKernelGGL(kernel, grid, block, lds, stream, args);                                                  \
        if (hipGetLastError() != hipSuccess) return DTQN_ERR_LAUNCH;                                                 \
    } while (0)

// Rows per workgroup of the linear / dY W kernels: 64.  The 32-row instantiations (DTQN_GEMM_ROWS=32) even out short launches
// (768 workgroups on 512 slots; 256 for the backward products into a D-wide output) but fetch every weight fragment for half
// the MFMAs: measured cfg 4 787 -> 748, cfg 5 445 -> 408 updates/s, so they stay an experiment switch.  (The fused feed-forward
// kernel, whose workgroups run 8-16 weight fragments deep, gains from 32 rows: launch_ffn.)
static bool tl_half_rows(int blocks64, int slots) {
    const char* e = getenv("DTQN_GEMM_ROWS");
    if (e == nullptr) return false;
    if (e[0] == 'a') {                 // "auto" (experiment, round 4: cfg 5 485.5 -> 476.3, cfg 4 847.5 -> 840.0, cfg 3 508.2 -> 507.4: stays off):
                                       // 32 rows only for launches that would leave more than 40 % of their slots idle
        const int rounds = (blocks64 + slots - 1) / slots;
        return (rounds * slots - blocks64) * 100 > 40 * rounds * slots;
    }
    return atoi(e) == 32;
}
template <int D>
static int launch_linear(TlLinearArgs a, int S, hipStream_t stream) {
    const int cb = (a.N + 16 * TNW - 1) / (16 * TNW);
    if (tl_half_rows(S * a.rpb * cb, 256 * (D <= 128 ? 2 : 1))) {
        a.rpb *= 2;
        const size_t lds = (size_t)32 * ((D > 16 * TNW ? D : 16 * TNW) + 4) * sizeof(float);   // operand tile, reused by the epilogue tile
        if (D % 128 == 0 && a.Wpa != nullptr && a.Wpb != nullptr) TL_LAUNCH((tl_linear_kernel<D, 32, D % 128 == 0>), dim3(S * a.rpb, cb), dim3(TNT), lds, stream, a);
        else TL_LAUNCH((tl_linear_kernel<D, 32, false>), dim3(S * a.rpb, cb), dim3(TNT), lds, stream, a);
    } else {
        const size_t lds = (size_t)64 * ((D > 16 * TNW ? D : 16 * TNW) + 4) * sizeof(float);
        if (D % 128 == 0 && a.Wpa != nullptr && a.Wpb != nullptr) TL_LAUNCH((tl_linear_kernel<D, 64, D % 128 == 0>), dim3(S * a.rpb, cb), dim3(TNT), lds, stream, a);
        else TL_LAUNCH((tl_linear_kernel<D, 64, false>), dim3(S * a.rpb, cb), dim3(TNT), lds, stream, a);
    }
    return DTQN_OK;
}
template <int D>
static int launch_ffn_bwd(TlFfnBwdArgs a, int S, hipStream_t stream);
// a.rpb on entry: 64-row blocks per sequence; rows per workgroup chosen like launch_ffn's (DTQN_FFN_ROWS forces)
// `which`: the kernel's own switch (DTQN_ROWS_FFN / _FFNB / _WIDE = 32 | 64), else DTQN_FFN_ROWS for all three, else the rule
static bool tl_rows32(int blocks64, int slots, int D, const char* which) {
    const char* e = getenv(which);
    if (e == nullptr) e = getenv("DTQN_FFN_ROWS");
    if (e != nullptr) return atoi(e) == 32;
    const int rounds = (blocks64 + slots - 1) / slots;
    // D = 256 (one workgroup per CU): 32 rows when the last round of 64-row workgroups would leave more than 15 % of the slots idle.
    // D <= 128 (two per CU), round 6: tools/microbench/mfma_probe.hip shows these loops bound by the weight fragments every workgroup
    // pulls out of L2 for its rows (the same loop with the fetches removed: 0.65 -> 0.81 of the matrix peak at 32 rows), so a 64-row
    // workgroup -- half the fetches per MFMA -- wins even at a round and a half (config 4 forward, 768 on 512 slots: tl_ffn 155 ->
    // 138 us); 32 rows only where 64-row workgroups would not even give every compute unit one (config 4 backward, 256 blocks: 64
    // rows 334 us per chain against 342).
    if (D <= 128) return blocks64 * 2 < slots;
    return (rounds * slots - blocks64) * 100 > 15 * rounds * slots;
}
template <int D>
static int launch_ffn_bwd(TlFfnBwdArgs a, int S, hipStream_t stream) {
    if (tl_rows32(S * a.rpb, 256 * (D <= 128 ? 2 : 1), D, "DTQN_ROWS_FFNB")) {
        a.rpb *= 2;
        const size_t lds = (size_t)32 * ((D + 4) + (128 + 4)) * sizeof(float);
        if (D % 128 == 0 && a.W1p != nullptr && a.W2p != nullptr) TL_LAUNCH((tl_ffn_bwd_kernel<D, 32, D % 128 == 0>), dim3(S * a.rpb), dim3(TNT), lds, stream, a);
        else TL_LAUNCH((tl_ffn_bwd_kernel<D, 32, false>), dim3(S * a.rpb), dim3(TNT), lds, stream, a);
    } else {
        const size_t lds = (size_t)64 * ((D + 4) + (128 + 4)) * sizeof(float);
        if (D % 128 == 0 && a.W1p != nullptr && a.W2p != nullptr) TL_LAUNCH((tl_ffn_bwd_kernel<D, 64, D % 128 == 0>), dim3(S * a.rpb), dim3(TNT), lds, stream, a);
        else TL_LAUNCH((tl_ffn_bwd_kernel<D, 64, false>), dim3(S * a.rpb), dim3(TNT), lds, stream, a);
    }
    return DTQN_OK;
}
template <int D>
static int launch_chain_bwd(const TlChainBwdArgs& a, int S, hipStream_t stream) {
    {
        const size_t lds = ((size_t)64 * ((D + 4) + (128 + 4)) + (size_t)TNW * 2 * D) * sizeof(float);
        if (D % 128 == 0 && a.f.W1p != nullptr && a.f.W2p != nullptr && a.Wop != nullptr) TL_LAUNCH((tl_chain_bwd_kernel<D, D % 128 == 0>), dim3(S * a.f.rpb), dim3(TNT), lds, stream, a);
        else TL_LAUNCH((tl_chain_bwd_kernel<D, false>), dim3(S * a.f.rpb), dim3(TNT), lds, stream, a);
        return DTQN_OK;
    }
}
template <int D, int MR>
static int launch_wide_rows(TlWideArgs a, int nblk, bool ln, bool pk, size_t lds, hipStream_t stream) {
    constexpr bool CAN = D % 128 == 0;                                 // fragment-major weights exist for these widths
    a.skew = tl_skew_ticks(nblk, 256 * (D <= 128 ? 2 : 1), 600, "DTQN_SKEW_WIDE");
    if (ln && pk) TL_LAUNCH((tl_wide_kernel<D, MR, true, CAN>), dim3(nblk), dim3(TNT), lds, stream, a);
    else if (ln) TL_LAUNCH((tl_wide_kernel<D, MR, true, false>), dim3(nblk), dim3(TNT), lds, stream, a);
    else if (pk) TL_LAUNCH((tl_wide_kernel<D, MR, false, CAN>), dim3(nblk), dim3(TNT), lds, stream, a);
    else TL_LAUNCH((tl_wide_kernel<D, MR, false, false>), dim3(nblk), dim3(TNT), lds, stream, a);
    return DTQN_OK;
}
// nblk_rows > 0: a packed launch (TlPack) of that many 64-row workgroups
template <int D>
static int launch_wide(TlWideArgs a, int S, hipStream_t stream, int nblk_rows = 0) {
    const bool ln = a.ln_out.base != nullptr;
    const size_t cols = ln ? (size_t)2 * (D + 4) : (size_t)(D + 4) + (128 + 4);
    const bool pk = D % 128 == 0 && a.Wpa != nullptr && a.Wpb != nullptr;
    if (tl_rows32(S * a.rpb, 256 * (D <= 128 ? 2 : 1), D, "DTQN_ROWS_WIDE")) {
        a.rpb *= 2;
        a.pack = TlPack{0, 0, 0};
        return launch_wide_rows<D, 32>(a, S * a.rpb, ln, pk, 32 * cols * sizeof(float), stream);
    }
    if (a.pack.L > 0 && !ln && nblk_rows > 0) return launch_wide_rows<D, 64>(a, nblk_rows, ln, pk, 64 * cols * sizeof(float), stream);
    a.pack = TlPack{0, 0, 0};
    return launch_wide_rows<D, 64>(a, S * a.rpb, ln, pk, 64 * cols * sizeof(float), stream);
}
// a.rpb on entry: 64-row blocks per sequence.  64-row workgroups by default; when the last round of 64-row workgroups would
// leave more than 15 % of the launch's slots idle (resident workgroups: two per CU at D <= 128, one at D = 256), 32-row
// workgroups even the rounds out (cfg 4: 768 workgroups on 512 slots = 1.5 rounds -> 1536 = 3, 764 -> 788 updates/s; cfg 5: 384
// on 256 -> 768 = 3, 437 -> 445).  DTQN_FFN_ROWS=32|64 forces one.
template <int D, int MR>
static int launch_ffn_rows(TlFfnArgs a, int nblk, bool pk, hipStream_t stream) {
    const size_t lds = (size_t)MR * ((D + 4) + (128 + 4)) * sizeof(float);
    a.skew = tl_skew_ticks(nblk, 256 * (D <= 128 ? 2 : 1));
    if (pk) TL_LAUNCH((tl_ffn_kernel<D, MR, D % 128 == 0>), dim3(nblk), dim3(TNT), lds, stream, a);
    else TL_LAUNCH((tl_ffn_kernel<D, MR, false>), dim3(nblk), dim3(TNT), lds, stream, a);
    return DTQN_OK;
}
template <int D>
static int launch_ffn(TlFfnArgs a, int S, hipStream_t stream) {
    const int blocks64 = S * a.rpb, slots = 256 * (D <= 128 ? 2 : 1);
    const bool pk = D % 128 == 0 && a.W1pa != nullptr && a.W1pb != nullptr && a.W2pa != nullptr && a.W2pb != nullptr;
    if (tl_rows32(blocks64, slots, D, "DTQN_ROWS_FFN")) {
        a.rpb *= 2;
        return launch_ffn_rows<D, 32>(a, S * a.rpb, pk, stream);
    }
    return launch_ffn_rows<D, 64>(a, S * a.rpb, pk, stream);
}
static int g_last_packed_blocks = 0;    // grid of the last fused-layer launch if it was a packed one, else 0 (tests: dtqn_debug_last_packed_blocks)
// the fused layer tail: rows per workgroup by launch_ffn's rule (its long phase is the feed-forward loop)
template <int D, int MR>
static int launch_layer_rows(TlLayerArgs a, int nblk, bool pk, int tail, hipStream_t stream) {
    constexpr int LDX = D + 4, LDH = 128 + 4;
    const size_t lds = (size_t)MR * (LDX + (LDX > LDH ? LDX : LDH)) * sizeof(float);
    a.f.skew = tl_skew_ticks(nblk, 256 * (D <= 128 ? 2 : 1), 2000, "DTQN_SKEW_LAYER");
    constexpr bool CAN = D % 128 == 0;
    if (tail == 2) {
        if constexpr (CAN && MR == 64) {                               // (the projection tail exists for 64-row workgroups of d_model 128 / 256)
            if (pk) TL_LAUNCH((tl_layer_kernel<D, MR, true, 2>), dim3(nblk), dim3(TNT), lds, stream, a);
            else TL_LAUNCH((tl_layer_kernel<D, MR, false, 2>), dim3(nblk), dim3(TNT), lds, stream, a);
            return DTQN_OK;
        }
        return DTQN_ERR_CONFIG;
    }
    const bool head = tail == 1;
    if (pk && head) TL_LAUNCH((tl_layer_kernel<D, MR, CAN, 1>), dim3(nblk), dim3(TNT), lds, stream, a);
    else if (pk) TL_LAUNCH((tl_layer_kernel<D, MR, CAN, 0>), dim3(nblk), dim3(TNT), lds, stream, a);
    else if (head) TL_LAUNCH((tl_layer_kernel<D, MR, false, 1>), dim3(nblk), dim3(TNT), lds, stream, a);
    else TL_LAUNCH((tl_layer_kernel<D, MR, false, 0>), dim3(nblk), dim3(TNT), lds, stream, a);
    return DTQN_OK;
}
// can this launch carry the next layer's q | k | v projection? (64-row workgroups, d_model 128 / 256: launch_layer's row rule, asked in advance)
template <int D>
static bool layer_qkv_tail_ok(int blocks64) {
    return D % 128 == 0 && !tl_rows32(blocks64, 256 * (D <= 128 ? 2 : 1), D, "DTQN_ROWS_FFN");
}
template <int D>
static int launch_layer(TlLayerArgs a, int S, int tail, hipStream_t stream, int nblk_rows = 0) {
    const int blocks64 = S * a.f.rpb, slots = 256 * (D <= 128 ? 2 : 1);
    const bool head = tail == 1;
    const bool pk = D % 128 == 0 && a.f.W1pa != nullptr && a.f.W1pb != nullptr && a.f.W2pa != nullptr && a.f.W2pb != nullptr &&
                    a.Wopa != nullptr && a.Wopb != nullptr && (!head || (a.Wh1pa != nullptr && a.Wh1pb != nullptr)) &&
                    (tail != 2 || (a.Winpa != nullptr && a.Winpb != nullptr));
    if (tl_rows32(blocks64, slots, D, "DTQN_ROWS_FFN")) {
        a.f.rpb *= 2;
        a.pack = TlPack{0, 0, 0};
        return launch_layer_rows<D, 32>(a, S * a.f.rpb, pk, tail, stream);
    }
    g_last_packed_blocks = a.pack.L > 0 && nblk_rows > 0 ? nblk_rows : 0;
    if (a.pack.L > 0 && nblk_rows > 0) return launch_layer_rows<D, 64>(a, nblk_rows, pk, tail, stream);
    a.pack = TlPack{0, 0, 0};
    return launch_layer_rows<D, 64>(a, S * a.f.rpb, pk, tail, stream);
}
template <int KC>
static int launch_dx(TlDxArgs a, int S, hipStream_t stream) {
    const int cb = (a.KOUT + 16 * TNW - 1) / (16 * TNW);
    if (a.hh.base == nullptr && tl_half_rows(S * a.rpb * cb, 512)) {
        a.rpb *= 2;
        const size_t lds = (size_t)32 * ((KC > 16 * TNW ? KC : 16 * TNW) + 4) * sizeof(float);   // operand tile, reused by the epilogue tile
        if (KC == 128 && a.Wp != nullptr && a.nsrc == 1) TL_LAUNCH((tl_dx_kernel<KC, 32, KC == 128>), dim3(S * a.rpb, cb), dim3(TNT), lds, stream, a);
        else TL_LAUNCH((tl_dx_kernel<KC, 32, false>), dim3(S * a.rpb, cb), dim3(TNT), lds, stream, a);
    } else {
        const size_t lds = (size_t)64 * ((KC > 16 * TNW ? KC : 16 * TNW) + 4) * sizeof(float);
        if (a.hh.base != nullptr) {                                    // Q-head mode (64-row workgroups)
            if (KC == 128 && a.Wp != nullptr) TL_LAUNCH((tl_dx_kernel<KC, 64, KC == 128, true>), dim3(S * a.rpb, cb), dim3(TNT), lds, stream, a);
            else TL_LAUNCH((tl_dx_kernel<KC, 64, false, true>), dim3(S * a.rpb, cb), dim3(TNT), lds, stream, a);
            return DTQN_OK;
        }
        if (KC == 128 && a.Wp != nullptr && a.nsrc == 1) TL_LAUNCH((tl_dx_kernel<KC, 64, KC == 128>), dim3(S * a.rpb, cb), dim3(TNT), lds, stream, a);
        else TL_LAUNCH((tl_dx_kernel<KC, 64, false>), dim3(S * a.rpb, cb), dim3(TNT), lds, stream, a);
    }
    return DTQN_OK;
}
template <int D>
static int launch_ln(const TlLnArgs& a, int S, hipStream_t stream) {
    if (a.d_real > 0) TL_LAUNCH((tl_layernorm_kernel<D, true>), dim3(S * a.rpb), dim3(TNT), 0, stream, a);
    else TL_LAUNCH((tl_layernorm_kernel<D, false>), dim3(S * a.rpb), dim3(TNT), 0, stream, a);
    return DTQN_OK;
}
template <int D>
static int launch_ln_bwd(const TlLnBwdArgs& a, int S, hipStream_t stream) {
    const size_t lds = (size_t)TNW * 2 * D * sizeof(float);
    if (a.d_real > 0) TL_LAUNCH((tl_layernorm_bwd_kernel<D, true>), dim3(S * a.rpb), dim3(TNT), lds, stream, a);
    else TL_LAUNCH((tl_layernorm_bwd_kernel<D, false>), dim3(S * a.rpb), dim3(TNT), lds, stream, a);
    return DTQN_OK;
}
// head_dim instantiations of the attention kernels (dtqn_net_init admits exactly these on the row-block path)
#define TL_ATTN_HEAD_DIMS(X) X(4) X(8) X(16) X(32) X(64) X(128)
static int launch_attn(const TlAttnArgs& a, int S, int H, int HD, hipStream_t stream) {
    const size_t lds = (size_t)a.lpb * (3 * HD + 4) * sizeof(float);
#define TL_ATTN_CASE(hd)                                                                                             \
    if (HD == hd) {                                                                                                  \
        if (a.drop.thresh != 0u) TL_LAUNCH((tl_attn_kernel<hd, true>), dim3(S, H), dim3(256), lds, stream, a);       \
        else TL_LAUNCH((tl_attn_kernel<hd, false>), dim3(S, H), dim3(256), lds, stream, a);                          \
        return DTQN_OK;                                                                                              \
    }
    TL_ATTN_HEAD_DIMS(TL_ATTN_CASE)
#undef TL_ATTN_CASE
    return DTQN_ERR_CONFIG;
}
static int launch_attn_bwd(const TlAttnBwdArgs& a, int S, int H, int HD, hipStream_t stream) {
    const size_t lds = ((size_t)a.lpb * (4 * HD + 4) + 2 * (size_t)a.lpb) * sizeof(float);
#define TL_ATTN_CASE(hd)                                                                                             \
    if (HD == hd) {                                                                                                  \
        if (a.drop.thresh != 0u) TL_LAUNCH((tl_attn_bwd_kernel<hd, true>), dim3(S, H), dim3(256), lds, stream, a);   \
        else TL_LAUNCH((tl_attn_bwd_kernel<hd, false>), dim3(S, H), dim3(256), lds, stream, a);                      \
        return DTQN_OK;                                                                                              \
    }
    TL_ATTN_HEAD_DIMS(TL_ATTN_CASE)
#undef TL_ATTN_CASE
    return DTQN_ERR_CONFIG;
}

// Where the forward keeps its tensors.  Training: the DtqnNet activation record (every layer saved).
// Inference: the same per-layer fields but ONE layer region reused by every layer, followed by xf | hh.
struct RecMap {
    long long stride;
    int layer_stride;
    int xf, hh;
    int bag_e = -1, bag_kv = -1, bag_q = -1, bag_o = -1, xcat = -1;
};
static RecMap rec_map(const DtqnNet& net, bool training) {
    RecMap m;
    if (training) {
        m.stride = net.act_stride; m.layer_stride = net.act_layer_stride; m.xf = net.ao_xf; m.hh = net.ao_hh;
    } else {
        m.layer_stride = 0;
        m.xf = net.ao_layer0 + net.act_layer_stride;
        m.hh = m.xf + net.lp * net.d_model;
        m.stride = m.hh + net.lp * net.d_model;
        if (net.bag_size > 0) {            // bag branch scratch: e | kv | q | o | xcat (no e_in, no attention weights saved)
            m.bag_e = (int)m.stride;
            m.bag_kv = m.bag_e + net.lp * net.d_model;
            m.bag_q = m.bag_kv + 2 * net.lp * net.d_model;
            m.bag_o = m.bag_q + net.lp * net.d_model;
            m.xcat = m.bag_o + net.lp * net.d_model;
            m.stride = m.xcat + 2 * net.lp * net.d_model;
        }
    }
    if (training && net.bag_size > 0) { m.bag_e = net.ao_bag_e; m.bag_kv = net.ao_bag_kv; m.bag_q = net.ao_bag_q; m.bag_o = net.ao_bag_o; m.xcat = net.ao_xcat; }
    return m;
}

struct EmbedSrc {
    const float* obs;
    const uint8_t* actions;
    long long obs_ep_stride, act_ep_stride;
    const int32_t* ep_idx;
    const int32_t* start;
    int batch;
    int seq0 = 0;
    // bag_size > 0: [bag_batch][bag_size][O] observations / [bag_batch][bag_size] actions; sequence s uses bag s % bag_batch
    const float* bag_obs = nullptr;
    const uint8_t* bag_actions = nullptr;
    int bag_batch = 0;
    const int32_t* lens = nullptr;     // ragged prefixes (device array [S]); nullptr: every sequence has n live rows
    const float* pre = nullptr;        // image nets: observation embeddings [S][pre_rows][D - a]
    int pre_rows = 0;
};

// All S sequences through the network.  theta_a serves sequences [0, split), theta_b the rest.
template <int D>
static int forward_records(const DtqnNet& net, const float* theta_a, const float* theta_b, int split, const EmbedSrc& src,
                           int S, int n, float* rec, bool training, float* q_out, long long q_seq_stride, int q_row_stride,
                           hipStream_t stream, const TlDrop& drop, const float* pk_a = nullptr, const float* pk_b = nullptr) {
    // fragment-major weight copies of the two parameter sets (dtqn_wpack.hpp; nullptr: the kernels read the parameter layout)
    const WPackPlan wplan = (pk_a != nullptr && pk_b != nullptr) ? wpack_plan(net) : WPackPlan{};
    const int lpb = net.lp, H = net.num_heads, HD = net.head_dim, rpb = lpb / TROWS;
    const RecMap rm = rec_map(net, training);
    bool qkv0_done = false;            // the embedding launch also ran layer 0's q | k | v projection
    const bool ident = net.identity != 0;
    int rc;
    auto F = [&](int off, int ld) { return fld(rec, rm.stride, off, ld); };
    auto L0 = [&](int l) { return net.ao_layer0 + l * rm.layer_stride; };
    {
        TlEmbedArgs e;
        e.net = net; e.theta_a = theta_a; e.theta_b = theta_b; e.split = split;
        e.obs = src.obs; e.actions = src.actions; e.obs_ep_stride = src.obs_ep_stride; e.act_ep_stride = src.act_ep_stride;
        e.ep_idx = src.ep_idx; e.start = src.start; e.batch = src.batch; e.seq0 = src.seq0;
        e.n = n; e.rpb = rpb;
        e.x = ident ? F(net.ao_x0, D) : F(L0(0) + net.al_u1, D);
        e.ein = training ? F(net.ao_ein, net.kep) : nofld();
        e.src_mod = 0; e.bag = 0; e.drop = drop; e.lens = src.lens;
        e.pre = src.pre; e.pre_rows = src.pre_rows;
        if (net.img_c > 0 && e.pre == nullptr) return DTQN_ERR_ARG;     // image nets come through dtqn_img_encode
        e.ein = e.pre != nullptr ? nofld() : e.ein;
        e.ptab_a = wpack_etab(wplan, pk_a); e.ptab_b = wpack_etab(wplan, pk_b);
        e.n_save = training ? (src.batch - src.seq0 > 0 ? src.batch - src.seq0 : 0) : 0;
        if (e.ptab_a != nullptr && e.ptab_b != nullptr && net.discrete && e.pre == nullptr) {
            // ... with layer 0's q | k | v projection in the same launch (d_model 128 / 256, post-LN: the layer reads its own input; DTQN_EMBED_QKV=0: apart)
            const char* eqe = getenv("DTQN_EMBED_QKV");
            const int tb0 = net.off_layer0;
            e.qkv = F(L0(0) + net.al_qkv, 3 * D);
            e.Wina = theta_a + tb0 + net.lo_in_w; e.Winb = theta_b + tb0 + net.lo_in_w; e.bina = theta_a + tb0 + net.lo_in_b; e.binb = theta_b + tb0 + net.lo_in_b;
            e.Winpa = wpack_f(wplan, pk_a, tb0 + net.lo_in_w); e.Winpb = wpack_f(wplan, pk_b, tb0 + net.lo_in_w);
            bool with_qkv = false;
            if constexpr (D % 128 == 0) {
                // Measured (TD-updates/s, fused | apart): config 3 (1536 workgroups, three rounds) 588 | 585; config 4 (768: a round and a half,
                // where the projection launch has its start skew) 993 | 998; config 5 (d_model 256, 384 workgroups on 256 slots) 564 | 568.
                // So: d_model 128 and launches of two rounds and more; DTQN_EMBED_QKV=1 forces it wherever it exists, =0 never.
                const bool pays = D == 128 && S * rpb >= 2 * 512;
                if (!ident && net.num_layers > 0 && getenv("DTQN_NO_WIDE") == nullptr && (eqe != nullptr ? atoi(eqe) != 0 : pays)) {
                    const size_t lds = ((size_t)TROWS * ((D + 4) + (128 + 4))) * sizeof(float) + (size_t)TROWS * net.obs_dim * sizeof(int);
                    if (e.Winpa != nullptr && e.Winpb != nullptr) TL_LAUNCH((tl_embed_table_kernel<D, true>), dim3(S * rpb), dim3(TNT), lds, stream, e);
                    else TL_LAUNCH((tl_embed_table_kernel<D, false>), dim3(S * rpb), dim3(TNT), lds, stream, e);
                    with_qkv = true;
                    qkv0_done = true;
                }
            }
            if (!with_qkv) TL_LAUNCH((tl_embed_table_kernel<0, false>), dim3(S * rpb), dim3(TNT), (size_t)TROWS * net.obs_dim * sizeof(int), stream, e);
        } else {
        const size_t elds = tl_embed_lds(net);
        TL_LAUNCH(tl_embed_kernel, dim3(S * rpb), dim3(TNT), elds, stream, e);
        }
    }
    auto linear = [&](Fld in, int K, int N, int w_off, int b_off, Fld out, int mode, Fld res, Fld mask) {
        TlLinearArgs a = {};
        a.in = in; a.out = out; a.res = res; a.mask = mask;
        a.Wa = theta_a + w_off; a.Wb = theta_b + w_off; a.ba = theta_a + b_off; a.bb = theta_b + b_off;
        a.split = split; a.K = K; a.N = N; a.rpb = rpb; a.mode = mode;
        a.Wpa = wpack_f(wplan, pk_a, w_off); a.Wpb = wpack_f(wplan, pk_b, w_off);
        return launch_linear<D>(a, S, stream);
    };
    // out = GRUGate(x, y) (gates.py:26-31): three two-operand GEMMs with the gate arithmetic as their epilogues.
    // grec: the gate's record (z, r, h~, r*x, x, y), y already in place.
    const bool gru = net.gate == DTQN_GATE_GRU;
    const int LPD = lpb * D;
    auto gate = [&](Fld x, int grec, int gw, Fld out) -> int {
        const Fld z = F(grec, D), r = F(grec + LPD, D), h = F(grec + 2 * LPD, D), rx = F(grec + 3 * LPD, D), y = F(grec + 5 * LPD, D);
        auto pair = [&](Fld in2, int w_off, int u_off, int b_off) {
            TlLinearArgs a = {};
            a.in = y; a.K = D; a.Wa = theta_a + gw + w_off; a.Wb = theta_b + gw + w_off;
            a.in2 = in2; a.K2 = D; a.W2a = theta_a + gw + u_off; a.W2b = theta_b + gw + u_off;
            if (b_off >= 0) { a.ba = theta_a + gw + b_off; a.bb = theta_b + gw + b_off; }
            a.split = split; a.N = D; a.rpb = rpb; a.res = x;
            return a;
        };
        TlLinearArgs a = pair(x, net.go_w_r, net.go_u_r, -1);
        a.mode = 3; a.out = r; a.out2 = rx; a.out3 = training ? F(grec + 4 * LPD, D) : nofld();
        int rc2 = launch_linear<D>(a, S, stream);
        if (rc2 != DTQN_OK) return rc2;
        a = pair(x, net.go_w_z, net.go_u_z, net.go_b_z);
        a.mode = 3; a.out = z;
        if ((rc2 = launch_linear<D>(a, S, stream)) != DTQN_OK) return rc2;
        a = pair(rx, net.go_w_g, net.go_u_g, -1);
        a.mode = 4; a.out = h; a.aux = z; a.out2 = out;
        return launch_linear<D>(a, S, stream);
    };
    auto lnorm = [&](Fld s_, Fld d_, Fld st, int w_off, int b_off) {
        TlLnArgs a;
        a.src = s_; a.dst = d_; a.st = st;
        a.ga = theta_a + w_off; a.gb = theta_b + w_off; a.ba = theta_a + b_off; a.bb = theta_b + b_off;
        a.split = split; a.rpb = rpb; a.d_real = net.d_real;
        return launch_ln<D>(a, S, stream);
    };
    // width-padded networks keep their LayerNorms in launches of their own (the fused epilogues below take the statistics over all D columns)
    const bool padded = net.d_real > 0;
    // DTQN_LAYER_FUSE=0: the separate launches (out-projection + LayerNorm | feed-forward + LayerNorm | head | Q) for A/B timing and tests
    const char* lfe = getenv("DTQN_LAYER_FUSE");
    const bool fuse_tail = !gru && !ident && !padded && net.bag_size == 0 && getenv("DTQN_NO_WIDE") == nullptr && (lfe == nullptr || atoi(lfe) != 0);
    bool head_done = false;
    int qkv_done_for = qkv0_done ? 0 : -1;
    // sequences the backward reads: the training third of a TD update ([0, batch) of the update = [0, batch - seq0) of this launch)
    const int n_save = training ? (src.batch - src.seq0 > 0 ? src.batch - src.seq0 : 0) : 0;
    // packed rows for the rest (TlPack): DTQN_PACK_ROWS=0 keeps every workgroup on (sequence, row block)
    TlPack pack = {0, 0, 0};
    {
        const char* pe = getenv("DTQN_PACK_ROWS");
        const bool rows64 = !tl_rows32(S * rpb, 256 * (D <= 128 ? 2 : 1), D, "DTQN_ROWS_FFN") && !tl_rows32(S * rpb, 256 * (D <= 128 ? 2 : 1), D, "DTQN_ROWS_WIDE");
        if (training && fuse_tail && S > n_save && drop.thresh == 0u && src.lens == nullptr && n >= 32 && n < lpb && ((long long)src.batch * n) % 64 == 0 &&
            rows64 && (3 * D) % 128 == 0 && (pe == nullptr || atoi(pe) != 0)) {
            pack.n0 = n_save * rpb; pack.s0 = n_save; pack.L = n;
        }
    }
    const int nblk_rows = pack.L > 0 ? pack.n0 + (S - n_save) * n / 64 : 0;      // workgroups of a packed launch
    for (int l = 0; l < net.num_layers; ++l) {
        const int tb = net.off_layer0 + l * net.layer_stride, ab = L0(l);
        const bool last = l + 1 == net.num_layers;
        const Fld u1 = F(ab + net.al_u1, D), s1 = F(ab + net.al_s1, D), u2 = F(ab + net.al_u2, D), s2 = F(ab + net.al_s2, D);
        const Fld st1 = training ? F(ab + net.al_st1, 2) : nofld(), st2 = training ? F(ab + net.al_st2, 2) : nofld();
        // the residual stream entering the layer: post-LN keeps it in u1 itself; identity in x0 / the previous s2
        const Fld stream_in = !ident ? u1 : (l == 0 ? F(net.ao_x0, D) : F(L0(l - 1) + net.al_s2, D));
        if (ident && (rc = lnorm(stream_in, u1, st1, tb + net.lo_ln1_w, tb + net.lo_ln1_b)) != DTQN_OK) return rc;
        if (qkv_done_for == l) {
            rc = DTQN_OK;                                                      // the previous layer's fused launch wrote this layer's q | k | v
        } else if ((3 * D) % 128 == 0 && getenv("DTQN_NO_WIDE") == nullptr) {  // packed q | k | v projection: one workgroup per row block walks the column blocks
            TlWideArgs wa = {};
            wa.in = u1; wa.out = F(ab + net.al_qkv, 3 * D);
            wa.Wa = theta_a + tb + net.lo_in_w; wa.Wb = theta_b + tb + net.lo_in_w; wa.ba = theta_a + tb + net.lo_in_b; wa.bb = theta_b + tb + net.lo_in_b;
            wa.split = split; wa.rpb = rpb; wa.N = 3 * D;
            wa.Wpa = wpack_f(wplan, pk_a, tb + net.lo_in_w); wa.Wpb = wpack_f(wplan, pk_b, tb + net.lo_in_w);
            wa.pack = pack;
            rc = launch_wide<D>(wa, S, stream, nblk_rows);
        } else {
            rc = linear(u1, D, 3 * D, tb + net.lo_in_w, tb + net.lo_in_b, F(ab + net.al_qkv, 3 * D), 0, nofld(), nofld());
        }
        if (rc != DTQN_OK) return rc;
        {
            TlAttnArgs at;
            at.qkv = F(ab + net.al_qkv, 3 * D); at.o = F(ab + net.al_o, D);
            at.lse = training ? F(ab + net.al_lse, lpb) : nofld();
            at.D = D; at.lpb = lpb; at.n = n; at.drop = drop; at.layer = l;
            at.hd_eff = (float)(net.hd_real > 0 ? net.hd_real : HD);
            if ((rc = launch_attn(at, S, H, HD, stream)) != DTQN_OK) return rc;
        }
        // post-LN residual layer: everything behind the attention in ONE launch (tl_layer_kernel), with the Q head on the last layer
        if (fuse_tail) {
            TlLayerArgs la = {};
            la.o = F(ab + net.al_o, D); la.res = stream_in; la.s1 = s1; la.u2 = u2;
            la.m1 = training ? F(ab + net.al_m1, 0) : nofld(); la.st1 = st1;
            la.Woa = theta_a + tb + net.lo_out_w; la.Wob = theta_b + tb + net.lo_out_w; la.boa = theta_a + tb + net.lo_out_b; la.bob = theta_b + tb + net.lo_out_b;
            la.Wopa = wpack_f(wplan, pk_a, tb + net.lo_out_w); la.Wopb = wpack_f(wplan, pk_b, tb + net.lo_out_w);
            la.g1a = theta_a + tb + net.lo_ln1_w; la.g1b = theta_b + tb + net.lo_ln1_w; la.be1a = theta_a + tb + net.lo_ln1_b; la.be1b = theta_b + tb + net.lo_ln1_b;
            TlFfnArgs& fa = la.f;
            fa.W1a = theta_a + tb + net.lo_f1_w; fa.W1b = theta_b + tb + net.lo_f1_w; fa.b1a = theta_a + tb + net.lo_f1_b; fa.b1b = theta_b + tb + net.lo_f1_b;
            fa.W2a = theta_a + tb + net.lo_f2_w; fa.W2b = theta_b + tb + net.lo_f2_w; fa.b2a = theta_a + tb + net.lo_f2_b; fa.b2b = theta_b + tb + net.lo_f2_b;
            fa.split = split; fa.rpb = rpb; fa.drop = drop; fa.layer = l;
            fa.W1pa = wpack_f(wplan, pk_a, tb + net.lo_f1_w); fa.W1pb = wpack_f(wplan, pk_b, tb + net.lo_f1_w);
            fa.W2pa = wpack_f(wplan, pk_a, tb + net.lo_f2_w); fa.W2pb = wpack_f(wplan, pk_b, tb + net.lo_f2_w);
            fa.n_save = n_save;
            la.pack = pack;
            fa.h = training ? F(ab + net.al_h, 4 * D) : nofld();
            fa.mh = training ? F(ab + net.al_mh, 0) : nofld();
            fa.m2 = training ? F(ab + net.al_m2, 0) : nofld();
            fa.mode = 2; fa.out = s2;
            fa.ln_out = last ? F(rm.xf, D) : F(L0(l + 1) + net.al_u1, D);
            fa.ln_st = st2;
            fa.lga = theta_a + tb + net.lo_ln2_w; fa.lgb = theta_b + tb + net.lo_ln2_w;
            fa.lba = theta_a + tb + net.lo_ln2_b; fa.lbb = theta_b + tb + net.lo_ln2_b;
            const bool head = last;
            if (head) {
                la.hh = F(rm.hh, D);
                la.Wh1a = theta_a + net.off_head1_w; la.Wh1b = theta_b + net.off_head1_w; la.bh1a = theta_a + net.off_head1_b; la.bh1b = theta_b + net.off_head1_b;
                la.Wh1pa = wpack_f(wplan, pk_a, net.off_head1_w); la.Wh1pb = wpack_f(wplan, pk_b, net.off_head1_w);
                la.Wqa = theta_a + net.off_head2_w; la.Wqb = theta_b + net.off_head2_w; la.bqa = theta_a + net.off_head2_b; la.bqb = theta_b + net.off_head2_b;
                la.q = q_out; la.q_seq_stride = q_seq_stride; la.q_row_stride = q_row_stride; la.A = net.num_actions; la.n = n;
                head_done = true;
            }
            // ... or, on the other layers of d_model 128 / 256, the NEXT layer's q | k | v projection (DTQN_QKV_FUSE=0: its own launch)
            const char* qfe = getenv("DTQN_QKV_FUSE");
            const bool qkv_next = !last && layer_qkv_tail_ok<D>(S * rpb) && getenv("DTQN_NO_WIDE") == nullptr && (qfe == nullptr || atoi(qfe) != 0);
            if (qkv_next) {
                const int tbn = net.off_layer0 + (l + 1) * net.layer_stride;
                la.qkv = F(L0(l + 1) + net.al_qkv, 3 * D);
                la.Wina = theta_a + tbn + net.lo_in_w; la.Winb = theta_b + tbn + net.lo_in_w; la.bina = theta_a + tbn + net.lo_in_b; la.binb = theta_b + tbn + net.lo_in_b;
                la.Winpa = wpack_f(wplan, pk_a, tbn + net.lo_in_w); la.Winpb = wpack_f(wplan, pk_b, tbn + net.lo_in_w);
                qkv_done_for = l + 1;
            }
            if ((rc = launch_layer<D>(la, S, head ? 1 : (qkv_next ? 2 : 0), stream, nblk_rows)) != DTQN_OK) return rc;
            continue;
        }
        // s1 = gate(stream, relu(o W_o^T + b)), then the LayerNorm behind it
        if (!gru && !ident && !padded && getenv("DTQN_NO_WIDE") == nullptr) {
            // post-LN residual layer: out-projection, residual add and LayerNorm-1 in one launch (s1 kept for the backward only)
            TlWideArgs wa = {};
            wa.in = F(ab + net.al_o, D); wa.out = s1; wa.N = D;
            wa.Wa = theta_a + tb + net.lo_out_w; wa.Wb = theta_b + tb + net.lo_out_w; wa.ba = theta_a + tb + net.lo_out_b; wa.bb = theta_b + tb + net.lo_out_b;
            wa.split = split; wa.rpb = rpb;
            wa.Wpa = wpack_f(wplan, pk_a, tb + net.lo_out_w); wa.Wpb = wpack_f(wplan, pk_b, tb + net.lo_out_w);
            wa.res = stream_in; wa.mask = training ? F(ab + net.al_m1, 0) : nofld();
            wa.ln_out = u2; wa.ln_st = st1;
            wa.lga = theta_a + tb + net.lo_ln1_w; wa.lgb = theta_b + tb + net.lo_ln1_w; wa.lba = theta_a + tb + net.lo_ln1_b; wa.lbb = theta_b + tb + net.lo_ln1_b;
            wa.n_save = n_save;
            if ((rc = launch_wide<D>(wa, S, stream)) != DTQN_OK) return rc;
        } else {
            if (!gru) {
                rc = linear(F(ab + net.al_o, D), D, D, tb + net.lo_out_w, tb + net.lo_out_b, s1, 2, stream_in,
                            training ? F(ab + net.al_m1, 0) : nofld());
            } else {
                rc = linear(F(ab + net.al_o, D), D, D, tb + net.lo_out_w, tb + net.lo_out_b, F(ab + net.al_gate1 + 5 * LPD, D), 1, nofld(),
                            training ? F(ab + net.al_m1, 0) : nofld());
                if (rc == DTQN_OK) rc = gate(stream_in, ab + net.al_gate1, net.off_gate_attn, s1);
            }
            if (rc != DTQN_OK) return rc;
            if (!ident) rc = lnorm(s1, u2, st1, tb + net.lo_ln1_w, tb + net.lo_ln1_b);
            else rc = lnorm(s1, u2, st2, tb + net.lo_ln2_w, tb + net.lo_ln2_b);
            if (rc != DTQN_OK) return rc;
        }
        // s2 = gate(post-LN: u2 | identity: s1, relu(relu(u2 W_1^T + b) W_2^T + b)): one fused launch, the hidden layer stays in LDS
        bool ln2_folded = false;
        {
            TlFfnArgs fa = {};
            fa.in = u2;
            fa.W1a = theta_a + tb + net.lo_f1_w; fa.W1b = theta_b + tb + net.lo_f1_w; fa.b1a = theta_a + tb + net.lo_f1_b; fa.b1b = theta_b + tb + net.lo_f1_b;
            fa.W2a = theta_a + tb + net.lo_f2_w; fa.W2b = theta_b + tb + net.lo_f2_w; fa.b2a = theta_a + tb + net.lo_f2_b; fa.b2b = theta_b + tb + net.lo_f2_b;
            fa.split = split; fa.rpb = rpb; fa.drop = drop; fa.layer = l;
            fa.W1pa = wpack_f(wplan, pk_a, tb + net.lo_f1_w); fa.W1pb = wpack_f(wplan, pk_b, tb + net.lo_f1_w);
            fa.W2pa = wpack_f(wplan, pk_a, tb + net.lo_f2_w); fa.W2pb = wpack_f(wplan, pk_b, tb + net.lo_f2_w);
            fa.n_save = n_save;                                       // only the training third of a TD update is read again
            fa.h = training ? F(ab + net.al_h, 4 * D) : nofld();
            fa.mh = training ? F(ab + net.al_mh, 0) : nofld();
            fa.m2 = training ? F(ab + net.al_m2, 0) : nofld();
            if (!gru) { fa.mode = 2; fa.out = s2; fa.res = ident ? s1 : u2; }
            else { fa.mode = 1; fa.out = F(ab + net.al_gate2 + 5 * LPD, D); fa.res = nofld(); }
            ln2_folded = !gru && !ident && !padded;
            if (ln2_folded) {            // post-LN residual layer: the LayerNorm that closes it rides in the kernel's epilogue
                fa.ln_out = last ? (net.bag_size > 0 ? F(rm.xcat, 2 * D) : F(rm.xf, D)) : F(L0(l + 1) + net.al_u1, D);
                fa.ln_st = st2;
                fa.lga = theta_a + tb + net.lo_ln2_w; fa.lgb = theta_b + tb + net.lo_ln2_w;
                fa.lba = theta_a + tb + net.lo_ln2_b; fa.lbb = theta_b + tb + net.lo_ln2_b;
            }
            rc = launch_ffn<D>(fa, S, stream);
            if (rc == DTQN_OK && gru) rc = gate(ident ? s1 : u2, ab + net.al_gate2, net.off_gate_mlp, s2);
        }
        if (rc != DTQN_OK) return rc;
        if (!ident && !ln2_folded) {
            // the working memory goes straight into the left half of xcat when there is a bag
            const Fld nxt = last ? (net.bag_size > 0 ? F(rm.xcat, 2 * D) : F(rm.xf, D)) : F(L0(l + 1) + net.al_u1, D);
            if ((rc = lnorm(s2, nxt, st2, tb + net.lo_ln2_w, tb + net.lo_ln2_b)) != DTQN_OK) return rc;
        }
    }
    const Fld xf = ident ? F(L0(net.num_layers - 1) + net.al_s2, D) : F(rm.xf, D);
    if (net.bag_size > 0) {
        // ---- persistent memory (dtqn.py:201-214): embed the bag entries, attend from the working memory over them, and feed
        //      [working memory | persistent memory] to the head
        if (src.bag_obs == nullptr || (net.action_dim > 0 && src.bag_actions == nullptr) || src.bag_batch < 1) return DTQN_ERR_ARG;
        const int bag = net.bag_size;
        if (ident) {        // no closing LayerNorm wrote the working memory into xcat: identity layers end in s2 of the last layer
            TlCopyArgs c;
            c.src = xf; c.dst = F(rm.xcat, 2 * D); c.cols = D; c.rpb = rpb;
            TL_LAUNCH(tl_copy_kernel, dim3(S * rpb), dim3(TNT), 0, stream, c);
        }
        {
            TlEmbedArgs e;
            e.net = net; e.theta_a = theta_a; e.theta_b = theta_b; e.split = split;
            e.obs = src.bag_obs; e.actions = src.bag_actions;
            e.obs_ep_stride = (long long)bag * net.obs_dim; e.act_ep_stride = bag;
            e.ep_idx = nullptr; e.start = nullptr; e.batch = src.bag_batch; e.seq0 = 0;
            e.n = bag; e.rpb = rpb;
            e.x = F(rm.bag_e, D);
            e.ein = training ? F(net.ao_bag_ein, net.kep) : nofld();
            e.src_mod = src.bag_batch; e.bag = 1; e.drop = tl_drop_none(); e.lens = nullptr; e.pre = nullptr; e.pre_rows = 0;
            e.ptab_a = nullptr; e.ptab_b = nullptr;
            const size_t elds = tl_embed_lds(net);
            TL_LAUNCH(tl_embed_kernel, dim3(S * rpb), dim3(TNT), elds, stream, e);
        }
        const Fld xw = F(rm.xcat, 2 * D);                      // working memory = left half of xcat
        if ((rc = linear(F(rm.bag_e, D), D, 2 * D, net.off_bag_in_w + D * D, net.off_bag_in_b + D, F(rm.bag_kv, 2 * D), 0, nofld(), nofld())) != DTQN_OK) return rc;
        if ((rc = linear(xw, D, D, net.off_bag_in_w, net.off_bag_in_b, F(rm.bag_q, D), 0, nofld(), nofld())) != DTQN_OK) return rc;
        {
            TlBagAttnArgs at;
            at.q = F(rm.bag_q, D); at.kv = F(rm.bag_kv, 2 * D); at.o = F(rm.bag_o, D);
            at.p = training ? F(net.ao_bag_p, net.bag_ld) : nofld();
            at.D = D; at.HD = HD; at.n = n; at.bag = bag; at.bag_ld = net.bag_ld; at.lpb = lpb; at.drop = drop;
            TL_LAUNCH(tl_bag_attn_kernel, dim3(S, H), dim3(256), (size_t)2 * bag * HD * sizeof(float), stream, at);
        }
        if ((rc = linear(F(rm.bag_o, D), D, D, net.off_bag_out_w, net.off_bag_out_b, F(rm.xcat + D, 2 * D), 0, nofld(), nofld())) != DTQN_OK) return rc;
        if ((rc = linear(xw, 2 * D, D, net.off_head1_w, net.off_head1_b, F(rm.hh, D), 1, nofld(), nofld())) != DTQN_OK) return rc;
    } else
    if (!head_done && (rc = linear(xf, D, D, net.off_head1_w, net.off_head1_b, F(rm.hh, D), 1, nofld(), nofld())) != DTQN_OK) return rc;
    if (head_done) return DTQN_OK;
    TlQArgs qa;
    qa.hh = F(rm.hh, D);
    qa.W2a = theta_a + net.off_head2_w; qa.W2b = theta_b + net.off_head2_w;
    qa.b2a = theta_a + net.off_head2_b; qa.b2b = theta_b + net.off_head2_b;
    qa.split = split; qa.q = q_out; qa.q_seq_stride = q_seq_stride; qa.q_row_stride = q_row_stride;
    qa.D = D; qa.A = net.num_actions; qa.n = n; qa.S = S;
    const int qgroups = (S * n + 15) / 16;
    TL_LAUNCH(tl_qhead_kernel, dim3(qgroups < 4096 ? qgroups : 4096), dim3(256), 0, stream, qa);
    return DTQN_OK;
}

// Data-gradient chain of the B TRAIN sequences (records [0, B) of td->act), residual gate; post-LN (transformer.py:63-78)
// or identity-reordered (transformer.py:86-101) layers.
// The gradient of the residual stream lives in grd.go_dx0 throughout (it IS dL/dx0 at the end).
template <int D>
static int backward_records(const DtqnNet& net, const DtqnReplay& rp, const DtqnTd& td, hipStream_t stream) {
    constexpr int KC = D < 128 ? D : 128;
    const int lpb = net.lp, H = net.num_heads, HD = net.head_dim, rpb = lpb / TROWS, B = td.batch, L = net.ctx_len;
    const float* theta = td.theta_pol;
    float* act = td.act;
    float* grd = td.grd;
    int rc;
    auto FA = [&](int off, int ld) { return fld(act, net.act_stride, off, ld); };
    auto FG = [&](int off, int ld) { return fld(grd, net.grd_stride, off, ld); };
    const long long obs_ep_stride = (long long)(rp.max_steps + 1) * rp.obs_dim, act_ep_stride = rp.max_steps + 1;
    // fragment-major B copies of the policy weights (dtqn_wpack.hpp), written by this update's forward
    const float* pk = td.wpack_tgt != nullptr ? td.wpack_pol : nullptr;
    const WPackPlan wplan = pk != nullptr ? wpack_plan(net) : WPackPlan{};
    {
        TlLossArgs a;
        a.net = net; a.q3 = td.q3; a.grd = grd; a.stats_partial = td.stats_partial;
        a.actions = rp.actions; a.rewards = rp.rewards; a.dones = rp.dones;
        a.act_ep_stride = act_ep_stride; a.rew_ep_stride = rp.max_steps;
        a.ep_idx = td.ep_idx; a.start = td.start; a.batch = B; a.history = td.history; a.gamma = td.gamma;
        TL_LAUNCH(tl_loss_kernel, dim3(B), dim3(256), 0, stream, a);
    }
    // the VALU half of the Q-head backward rides in the staging of the dL/dxf product below (TlDxArgs, Q-head mode); with a bag the head's
    // input is [working memory | persistent memory] and the separate launch stays.  DTQN_HEAD_FUSE=0: the separate launch (A/B, tests)
    const char* hfe = getenv("DTQN_HEAD_FUSE");
    const bool head_fused = net.bag_size == 0 && (hfe == nullptr || atoi(hfe) != 0);
    if (!head_fused) {
        TlHeadBwdArgs a;
        a.hh = FA(net.ao_hh, D); a.dq = FG(net.go_dq, net.ap); a.dhh = FG(net.go_dhh, D);
        a.W2 = theta + net.off_head2_w; a.D = D; a.A = net.num_actions; a.rpb = rpb;
        TL_LAUNCH(tl_head_bwd_kernel, dim3(B * rpb), dim3(TNT), 0, stream, a);
    }
    const Fld G = FG(net.go_dx0, D);
    auto dx = [&](Fld dy, int N, int w_off, int KOUT, Fld out, int mode, Fld mask) {
        TlDxArgs a = {};
        a.dy = dy; a.out = out; a.mask = mask; a.W = theta + w_off; a.N = N; a.KOUT = KOUT; a.rpb = rpb; a.mode = mode; a.nsrc = 1;
        a.Wp = wpack_b(wplan, pk, w_off);
        return launch_dx<KC>(a, B, stream);
    };
    auto ln_bwd = [&](Fld dy, Fld xin, Fld st, int gamma_off, int dgb_off, bool accumulate) {
        TlLnBwdArgs a;
        a.dy = dy; a.xin = xin; a.st = st; a.dst = G; a.gamma = theta + gamma_off;
        a.small = td.small; a.small_stride = net.sp_stride; a.dgb_off = dgb_off; a.rpb = rpb; a.accumulate = accumulate ? 1 : 0;
        a.d_real = net.d_real;
        return launch_ln_bwd<D>(a, B, stream);
    };
    auto mask = [&](Fld m, Fld dst) -> int {
        TlMaskArgs a;
        a.src = G; a.dst = dst; a.mask = m; a.D = D; a.rpb = rpb;
        TL_LAUNCH(tl_mask_kernel, dim3(B * rpb), dim3(TNT), 0, stream, a);
        return DTQN_OK;
    };
    const bool ident = net.identity != 0, gru = net.gate == DTQN_GATE_GRU;
    // the training forward's dropout (pass 0 of the TD update), recomputed: step = step_counter[1], not yet advanced
    const TlDrop drop = tl_drop_make(net, td.dropout_seed, 0u, td.step_counter, B, 0x1);
    auto drop_rows = [&](Fld x, int site, int layer) -> int {
        TlDropRowsArgs a;
        a.x = x; a.D = D; a.rpb = rpb; a.site = site; a.layer = layer; a.drop = drop;
        TL_LAUNCH(tl_drop_rows_kernel, dim3(B * rpb), dim3(TNT), 0, stream, a);
        return DTQN_OK;
    };
    const Fld T = FG(net.go_do, D);                       // branch-gradient scratch (dO of the attention; LN inputs' gradients)
    const int LPD = lpb * D;
    // dL/d(gate output) in G  ->  G = the part that flows on along the stream (dL/dx), dst = dL/d(sub-layer output):
    // ResGate: G stays, dst = G * [y > 0].  GRUGate (gates.py:26-31): the chain of dtqn_gru.hpp as GEMMs over the records.
    auto gate_bwd = [&](int grec, int ggrd, int gw, Fld m, Fld dst) -> int {
        if (!gru) return mask(m, dst);
        const Fld dz = FG(ggrd, D), dr = FG(ggrd + LPD, D), dh = FG(ggrd + 2 * LPD, D);
        TlGateBwdArgs e = {};
        e.g = G; e.t = T; e.z = FA(grec, D); e.r = FA(grec + LPD, D); e.h = FA(grec + 2 * LPD, D); e.x = FA(grec + 4 * LPD, D);
        e.dz = dz; e.dr = dr; e.dh = dh; e.D = D; e.rpb = rpb; e.step = 1;
        TL_LAUNCH(tl_gate_bwd_kernel, dim3(B * rpb), dim3(TNT), 0, stream, e);
        int rc2 = dx(dh, D, gw + net.go_u_g, D, T, 0, nofld());                      // d(r * x) = dh_pre U_g
        if (rc2 != DTQN_OK) return rc2;
        e.step = 2;
        TL_LAUNCH(tl_gate_bwd_kernel, dim3(B * rpb), dim3(TNT), 0, stream, e);
        TlDxArgs a = {};
        a.N = D; a.KOUT = D; a.rpb = rpb;
        a.dy = dz; a.W = theta + gw + net.go_u_z; a.dy2 = dr; a.W2 = theta + gw + net.go_u_r; a.nsrc = 2;
        a.out = G; a.mode = 2;                                                        // dx += dz_pre U_z + dr_pre U_r
        if ((rc2 = launch_dx<KC>(a, B, stream)) != DTQN_OK) return rc2;
        a.dy = dh; a.W = theta + gw + net.go_w_g; a.dy2 = dz; a.W2 = theta + gw + net.go_w_z; a.dy3 = dr; a.W3 = theta + gw + net.go_w_r;
        a.nsrc = 3; a.out = dst; a.mode = 1; a.mask = m;                              // dy, through the ReLU of the sub-layer output
        return launch_dx<KC>(a, B, stream);
    };
    if (net.bag_size > 0) {
        // d xcat = dhh W_1 ([D][2D]); its left half is dL/d(working memory) so far, its right half dL/d(persistent memory)
        const int bag = net.bag_size;
        if (!td.bag_obs || (net.action_dim > 0 && !td.bag_actions)) return DTQN_ERR_ARG;
        if ((rc = dx(FG(net.go_dhh, D), D, net.off_head1_w, 2 * D, FG(net.go_dcat, 2 * D), 0, nofld())) != DTQN_OK) return rc;
        {
            TlCopyArgs c;
            c.src = FG(net.go_dcat, 2 * D); c.dst = G; c.cols = D; c.rpb = rpb;
            TL_LAUNCH(tl_copy_kernel, dim3(B * rpb), dim3(TNT), 0, stream, c);
        }
        if ((rc = dx(FG(net.go_dcat + D, 2 * D), D, net.off_bag_out_w, D, FG(net.go_bag_do, D), 0, nofld())) != DTQN_OK) return rc;
        {
            TlBagAttnBwdArgs at;
            at.q = FA(net.ao_bag_q, D); at.kv = FA(net.ao_bag_kv, 2 * D); at.p = FA(net.ao_bag_p, net.bag_ld); at.dO = FG(net.go_bag_do, D);
            at.dq = FG(net.go_bag_dq, D); at.dkv = FG(net.go_bag_dkv, 2 * D);
            at.D = D; at.HD = HD; at.n = L; at.bag = bag; at.bag_ld = net.bag_ld; at.lpb = lpb; at.drop = drop;
            TL_LAUNCH(tl_bag_attn_bwd_kernel, dim3(B, H), dim3(256), ((size_t)2 * bag * HD + (size_t)L * bag) * sizeof(float), stream, at);
        }
        if ((rc = dx(FG(net.go_bag_dq, D), D, net.off_bag_in_w, D, G, 2, nofld())) != DTQN_OK) return rc;                  // + dq W_q
        if ((rc = dx(FG(net.go_bag_dkv, 2 * D), 2 * D, net.off_bag_in_w + D * D, D, FG(net.go_bag_de, D), 0, nofld())) != DTQN_OK) return rc;   // d E_bag
    } else
    if (head_fused) {
        TlDxArgs a = {};
        a.dy = FG(net.go_dhh, D); a.out = G; a.W = theta + net.off_head1_w; a.N = D; a.KOUT = D; a.rpb = rpb; a.mode = 0; a.nsrc = 1;
        a.Wp = wpack_b(wplan, pk, net.off_head1_w);
        a.hh = FA(net.ao_hh, D); a.dq = FG(net.go_dq, net.ap); a.Wq = theta + net.off_head2_w; a.A = net.num_actions;
        if ((rc = launch_dx<KC>(a, B, stream)) != DTQN_OK) return rc;                                    // dhh and dL/dxf
    } else
    if ((rc = dx(FG(net.go_dhh, D), D, net.off_head1_w, D, G, 0, nofld())) != DTQN_OK) return rc;       // dL/dxf
    // DTQN_BWD_CHAIN=0: the separate launches (A/B timing, tests).  64-row workgroups only: the LayerNorm column partials are per 64-row block
    const char* bce = getenv("DTQN_BWD_CHAIN");
    const char* ffb0 = getenv("DTQN_FFN_BWD");
    // d_model 256 (one workgroup per compute unit): only beside a second stream (DtqnTd.side_stream; DTQN_BWD_CHAIN256=0|1 forces).  Measured at
    // BASELINE config 5 (32 sequences = 128 64-row workgroups on 256 compute units): alone the chain is slower than the 32-row separate launches
    // (backward stage 633 against 583 us), with the next update's target pass running on the other half of the chip it wins (522 -> 552 updates/s)
    const char* bc256 = getenv("DTQN_BWD_CHAIN256");
    const bool chain = !gru && !ident && net.d_real == 0 && (bce == nullptr || atoi(bce) != 0) &&
                       (D <= 128 ? (ffb0 == nullptr || atoi(ffb0) != 0) && !tl_rows32(B * rpb, 256 * 2, D, "DTQN_ROWS_FFNB")
                                 : (bc256 != nullptr ? atoi(bc256) != 0 : td.side_stream != 0));
    for (int l = net.num_layers - 1; l >= 0; --l) {
        const int tb = net.off_layer0 + l * net.layer_stride;
        const int ab = net.ao_layer0 + l * net.act_layer_stride, gb = net.go_layer0 + l * net.grd_layer_stride;
        const int sm = net.so_ln + l * 4 * D;
        if (chain) {
            // post-LN residual layer: LN2', the feed-forward block, LN1', the gate mask and dO = da W_o in one launch (tl_chain_bwd_kernel)
            TlChainBwdArgs c = {};
            c.f.dy = G; c.f.out = G; c.f.out_mode = 2; c.f.m2 = FA(ab + net.al_m2, 0); c.f.df = FG(gb + net.gl_df, D);
            c.f.dhp = FG(gb + net.gl_dhp, 4 * D); c.f.mh = FA(ab + net.al_mh, 0);
            c.f.W1 = theta + tb + net.lo_f1_w; c.f.W2 = theta + tb + net.lo_f2_w; c.f.rpb = rpb; c.f.drop = drop; c.f.layer = l;
            c.f.W1p = wpack_b(wplan, pk, tb + net.lo_f1_w); c.f.W2p = wpack_b(wplan, pk, tb + net.lo_f2_w);
            c.s2 = FA(ab + net.al_s2, D); c.st2 = FA(ab + net.al_st2, 2); c.s1 = FA(ab + net.al_s1, D); c.st1 = FA(ab + net.al_st1, 2);
            c.gamma2 = theta + tb + net.lo_ln2_w; c.gamma1 = theta + tb + net.lo_ln1_w;
            c.small = td.small; c.small_stride = net.sp_stride; c.dgb2_off = sm + 2 * D; c.dgb1_off = sm;
            c.m1 = FA(ab + net.al_m1, 0); c.da = FG(gb + net.gl_da, D); c.dO = FG(net.go_do, D);
            c.Wo = theta + tb + net.lo_out_w; c.Wop = wpack_b(wplan, pk, tb + net.lo_out_w);
            if ((rc = launch_chain_bwd<D>(c, B, stream)) != DTQN_OK) return rc;
        } else {
        // post-LN: x_out = LN2(s2)
        if (!ident && (rc = ln_bwd(G, FA(ab + net.al_s2, D), FA(ab + net.al_st2, 2), tb + net.lo_ln2_w, sm + 2 * D, false)) != DTQN_OK) return rc;
        // s2 = (u2 | s1) + relu(f):  df = ds2 * [f > 0];  dh' = (df W2) * [h > 0];  du2 = dh' W1
        // one fused launch: the gate's ReLU mask (residual gate) rides in its staging, dh' is written once and never read back.
        // D <= 128 only: measured cfg 4 825 -> 838 updates/s, but cfg 5 (D = 256: 172 registers, one workgroup per CU) 461 -> 458
        const char* ffb = getenv("DTQN_FFN_BWD");
        if (ffb != nullptr ? atoi(ffb) != 0 : D <= 128) {
            TlFfnBwdArgs fb = {};
            if (!gru) { fb.dy = G; fb.m2 = FA(ab + net.al_m2, 0); fb.df = FG(gb + net.gl_df, D); }
            else {
                if ((rc = gate_bwd(ab + net.al_gate2, gb + net.gl_gate2, net.off_gate_mlp, FA(ab + net.al_m2, 0), FG(gb + net.gl_df, D))) != DTQN_OK) return rc;
                fb.dy = FG(gb + net.gl_df, D); fb.m2 = nofld();
                fb.df = drop.thresh != 0u ? FG(gb + net.gl_df, D) : nofld();          // dropout: the keep mask goes into df in place
            }
            fb.dhp = FG(gb + net.gl_dhp, 4 * D); fb.mh = FA(ab + net.al_mh, 0);
            fb.W1 = theta + tb + net.lo_f1_w; fb.W2 = theta + tb + net.lo_f2_w; fb.rpb = rpb; fb.drop = drop; fb.layer = l;
            fb.W1p = wpack_b(wplan, pk, tb + net.lo_f1_w); fb.W2p = wpack_b(wplan, pk, tb + net.lo_f2_w);
            if (!ident) { fb.out = G; fb.out_mode = 2; } else { fb.out = T; fb.out_mode = 0; }
            if ((rc = launch_ffn_bwd<D>(fb, B, stream)) != DTQN_OK) return rc;
            if (!ident) {
                if ((rc = ln_bwd(G, FA(ab + net.al_s1, D), FA(ab + net.al_st1, 2), tb + net.lo_ln1_w, sm, false)) != DTQN_OK) return rc;
            } else {
                if ((rc = ln_bwd(T, FA(ab + net.al_s1, D), FA(ab + net.al_st2, 2), tb + net.lo_ln2_w, sm + 2 * D, true)) != DTQN_OK) return rc;
            }
        } else {                                                       // the separate launches (A/B timing)
            if ((rc = gate_bwd(ab + net.al_gate2, gb + net.gl_gate2, net.off_gate_mlp, FA(ab + net.al_m2, 0), FG(gb + net.gl_df, D))) != DTQN_OK) return rc;
            if (drop.thresh != 0u && (rc = drop_rows(FG(gb + net.gl_df, D), DROP_FFN, l)) != DTQN_OK) return rc;
            if ((rc = dx(FG(gb + net.gl_df, D), D, tb + net.lo_f2_w, 4 * D, FG(gb + net.gl_dhp, 4 * D), 1, FA(ab + net.al_mh, 0))) != DTQN_OK) return rc;
            if (!ident) {
                // the stream IS u2: ds2 + du2, then u2 = LN1(s1)
                if ((rc = dx(FG(gb + net.gl_dhp, 4 * D), 4 * D, tb + net.lo_f1_w, D, G, 2, nofld())) != DTQN_OK) return rc;
                if ((rc = ln_bwd(G, FA(ab + net.al_s1, D), FA(ab + net.al_st1, 2), tb + net.lo_ln1_w, sm, false)) != DTQN_OK) return rc;
            } else {
                // u2 = LN2(s1) sits on the branch: ds1 = ds2 + LN2'(du2)
                if ((rc = dx(FG(gb + net.gl_dhp, 4 * D), 4 * D, tb + net.lo_f1_w, D, T, 0, nofld())) != DTQN_OK) return rc;
                if ((rc = ln_bwd(T, FA(ab + net.al_s1, D), FA(ab + net.al_st2, 2), tb + net.lo_ln2_w, sm + 2 * D, true)) != DTQN_OK) return rc;
            }
        }
        // s1 = x + relu(a):  da = ds1 * [a > 0];  dO = da W_o;  attention backward;  du1 = dqkv W_in
        if ((rc = gate_bwd(ab + net.al_gate1, gb + net.gl_gate1, net.off_gate_attn, FA(ab + net.al_m1, 0), FG(gb + net.gl_da, D))) != DTQN_OK) return rc;
        if ((rc = dx(FG(gb + net.gl_da, D), D, tb + net.lo_out_w, D, FG(net.go_do, D), 0, nofld())) != DTQN_OK) return rc;
        }
        {
            TlAttnBwdArgs a;
            a.qkv = FA(ab + net.al_qkv, 3 * D); a.o = FA(ab + net.al_o, D); a.lse = FA(ab + net.al_lse, lpb);
            a.dO = FG(net.go_do, D); a.dqkv = FG(gb + net.gl_dqkv, 3 * D);
            a.D = D; a.lpb = lpb; a.n = L; a.drop = drop; a.layer = l;
            a.hd_eff = (float)(net.hd_real > 0 ? net.hd_real : HD);
            if ((rc = launch_attn_bwd(a, B, H, HD, stream)) != DTQN_OK) return rc;
        }
        if (!ident) {
            if ((rc = dx(FG(gb + net.gl_dqkv, 3 * D), 3 * D, tb + net.lo_in_w, D, G, 2, nofld())) != DTQN_OK) return rc;
        } else {
            // u1 = LN1(x): dx = ds1 + LN1'(du1)
            const Fld stream_in = l == 0 ? FA(net.ao_x0, D) : FA(net.ao_layer0 + (l - 1) * net.act_layer_stride + net.al_s2, D);
            if ((rc = dx(FG(gb + net.gl_dqkv, 3 * D), 3 * D, tb + net.lo_in_w, D, T, 0, nofld())) != DTQN_OK) return rc;
            if ((rc = ln_bwd(T, stream_in, FA(ab + net.al_st1, 2), tb + net.lo_ln1_w, sm, true)) != DTQN_OK) return rc;
        }
    }
    // x0 = dropout(embedding + position): dL/d(embedding) and the position gradient take the keep mask
    if (drop.thresh != 0u && (rc = drop_rows(G, DROP_EMB, 0)) != DTQN_OK) return rc;
    if (net.discrete || net.action_dim > 0) {
        TlEmbedBwdArgs a;
        a.net = net; a.theta = theta; a.grd = grd; a.small = td.small;
        a.obs = rp.obs; a.actions = rp.actions; a.obs_ep_stride = obs_ep_stride; a.act_ep_stride = act_ep_stride;
        a.ep_idx = td.ep_idx; a.start = td.start;
        const size_t lds = tl_embed_bwd_lds(net);
        // (KE <= 256: at most 4 x 16 column tiles x 4 row tiles = the 4 items a wave keeps accumulators for)
        if (lds > 150 * 1024 || (net.discrete && net.ke > 128)) return DTQN_ERR_CONFIG;
        a.bag = 0; a.dx_off = 0; a.rows = 0;
        TL_LAUNCH(tl_embed_bwd_kernel, dim3(B * rpb), dim3(TNT), lds, stream, a);
        if (net.bag_size > 0) {          // the bag entries went through the same tables: their partials are added
            a.obs = td.bag_obs; a.actions = td.bag_actions;
            a.obs_ep_stride = (long long)net.bag_size * net.obs_dim; a.act_ep_stride = net.bag_size;
            a.bag = 1; a.dx_off = net.go_bag_de; a.rows = net.bag_size;
            TL_LAUNCH(tl_embed_bwd_kernel, dim3(B * rpb), dim3(TNT), lds, stream, a);
        }
    }
    return DTQN_OK;
}

int tiled_td_forward(const DtqnNet* net, const DtqnReplay* rp, const DtqnTd* td, hipStream_t stream) {
    return tiled_td_forward_part(net, rp, td, 0, 3, stream);
}

// Passes [pass0, pass0 + npasses) of {policy(o), policy(o'), target(o')} (dtqn_td_forward_part): the same kernels over the sequences
// of those passes only -- records, Q rows and the parameter split are addressed from the first sequence of the launch, the window
// lookup of the embedding kernel knows where in the update it sits (EmbedSrc::seq0).
int tiled_td_forward_part(const DtqnNet* net, const DtqnReplay* rp, const DtqnTd* td, int pass0, int npasses, hipStream_t stream) {
    const bool whole = pass0 == 0 && npasses == 3;
    if (!whole && (net->bag_size > 0 || net->img_c > 0 || net->dropout > 0.f)) return DTQN_ERR_CONFIG;
    EmbedSrc src;
    src.obs = rp->obs; src.actions = rp->actions;
    src.obs_ep_stride = (long long)(rp->max_steps + 1) * rp->obs_dim; src.act_ep_stride = rp->max_steps + 1;
    src.ep_idx = td->ep_idx; src.start = td->start; src.batch = td->batch;
    src.bag_obs = td->bag_obs; src.bag_actions = td->bag_actions; src.bag_batch = td->batch;
    src.pre = td->xemb; src.pre_rows = net->lp;
    src.seq0 = pass0 * td->batch;
    const int S = npasses * td->batch;
    const long long qs = (long long)net->lp * net->ap;
    const size_t rstride = rec_map(*net, true).stride;
    float* rec0 = td->act + (size_t)src.seq0 * rstride;
    float* q0 = td->q3 + (size_t)src.seq0 * qs;
    const int split = 2 * td->batch - src.seq0 > 0 ? 2 * td->batch - src.seq0 : 0;     // sequences of passes 0 - 1 use theta_pol
    // policy(o) and policy(o') run in train mode, the target net in eval mode (dtqn.py:215-230); step = optimizer steps so far
    const TlDrop drop = tl_drop_make(*net, td->dropout_seed, 0u, td->step_counter, td->batch, 0x3);
    // the first forward launch of an update rewrites the fragment-major weight copies from the parameters as they are NOW
    if (pass0 == 0) {
        const int rc = dtqn_td_wpack(net, td, stream);
        if (rc != DTQN_OK) return rc;
    }
    const bool packed = td->wpack_pol != nullptr && td->wpack_tgt != nullptr;
    const float* pk_pol = packed ? td->wpack_pol : nullptr;
    const float* pk_tgt = packed ? td->wpack_tgt : nullptr;
#define DTQN_TL_CASE(d) \
    case d: return forward_records<d>(*net, td->theta_pol, td->theta_tgt, split, src, S, net->ctx_len, rec0, true, q0, qs, net->ap, stream, drop, pk_pol, pk_tgt);
    switch (net->d_model) {
        DTQN_TL_CASE(64)
        DTQN_TL_CASE(128)
        DTQN_TL_CASE(256)
        default: return DTQN_ERR_CONFIG;
    }
#undef DTQN_TL_CASE
}

int tiled_td_backward(const DtqnNet* net, const DtqnReplay* rp, const DtqnTd* td, hipStream_t stream) {
    switch (net->d_model) {
        case 64: return backward_records<64>(*net, *rp, *td, stream);
        case 128: return backward_records<128>(*net, *rp, *td, stream);
        case 256: return backward_records<256>(*net, *rp, *td, stream);
        default: return DTQN_ERR_CONFIG;
    }
}

}  // namespace dtqn

using namespace dtqn;

extern "C" int dtqn_debug_last_packed_blocks(void) { return g_last_packed_blocks; }

extern "C" int dtqn_forward_workspace_floats(const DtqnNet* net, int batch) {
    if (!net || batch < 1) return 0;
    if (!net->tiled)       // whole-sequence kernels: only dtqn_actor_forward's latency mode wants a (ZEROED) workspace
        return dtqn_td_row_split(net, batch) >= 2 ? dtqn_td_xch_floats(net, batch) + dtqn_td_xch_flags(net, batch) : 0;
    const long long fl = (long long)batch * rec_map(*net, false).stride;
    return fl < 0x7fffffffLL ? (int)fl : 0;
}

static int forward_tiled_impl(const DtqnNet* net, const float* theta, const float* obs, const uint8_t* actions, const float* bag_obs,
                              const uint8_t* bag_actions, int batch, int n, int in_rows, float* q_out, float* workspace, void* stream,
                              int train_mode = 0, uint32_t drop_seed = 0u, uint32_t drop_step = 0u, const int32_t* lens = nullptr,
                              const float* pre = nullptr) {
    if (!net || !theta || (!obs && !pre) || !q_out || !workspace || batch < 1) return DTQN_ERR_ARG;
    if (n < 1 || n > net->ctx_len || in_rows < n) return DTQN_ERR_ARG;  // dtqn.py:170-173
    if (net->action_dim > 0 && !actions) return DTQN_ERR_ARG;
    if (!net->tiled) return DTQN_ERR_CONFIG;
    if (net->bag_size > 0 && (!bag_obs || (net->action_dim > 0 && !bag_actions))) return DTQN_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    EmbedSrc src;
    src.obs = obs; src.actions = actions;
    src.obs_ep_stride = (long long)in_rows * net->obs_dim; src.act_ep_stride = in_rows;
    src.ep_idx = nullptr; src.start = nullptr; src.batch = batch;
    src.bag_obs = bag_obs; src.bag_actions = bag_actions; src.bag_batch = batch;
    src.lens = lens;
    src.pre = pre; src.pre_rows = n;
    const long long qs = (long long)n * net->num_actions;
    // a train-mode forward of the actor (the reference's policy network stays in train mode during rollouts): one pass, sequence = salt
    const TlDrop drop = tl_drop_make(*net, drop_seed, drop_step, nullptr, batch, train_mode ? 0x1 : 0);
    switch (net->d_model) {
        case 64: return forward_records<64>(*net, theta, theta, batch, src, batch, n, workspace, false, q_out, qs, net->num_actions, s, drop);
        case 128: return forward_records<128>(*net, theta, theta, batch, src, batch, n, workspace, false, q_out, qs, net->num_actions, s, drop);
        case 256: return forward_records<256>(*net, theta, theta, batch, src, batch, n, workspace, false, q_out, qs, net->num_actions, s, drop);
        default: return DTQN_ERR_CONFIG;
    }
}

namespace dtqn {
// dtqn_actor_forward / dtqn_actor_forward_batch on a row-block net: the strided forward with the actor's train-mode dropout
int tiled_forward_actor(const DtqnNet* net, const float* theta, const float* obs, const uint8_t* actions, int batch, int n, int in_rows,
                        float* q_out, float* workspace, int train_mode, uint32_t drop_seed, uint32_t drop_step, hipStream_t stream,
                        const int32_t* lens) {
    if (net && net->bag_size > 0) return DTQN_ERR_ARG;
    return forward_tiled_impl(net, theta, obs, actions, nullptr, nullptr, batch, n, in_rows, q_out, workspace, stream, train_mode, drop_seed,
                              drop_step, lens);
}
}  // namespace dtqn

// in_rows: rows per sequence in the obs / actions arrays (>= n; the batched actor packs whole contexts)
extern "C" int dtqn_forward_tiled_strided(const DtqnNet* net, const float* theta, const float* obs, const uint8_t* actions,
                                          int batch, int n, int in_rows, float* q_out, float* workspace, void* stream) {
    if (net && net->bag_size > 0) return DTQN_ERR_ARG;                  // bag networks: dtqn_forward_bag
    return forward_tiled_impl(net, theta, obs, actions, nullptr, nullptr, batch, n, in_rows, q_out, workspace, stream);
}

extern "C" int dtqn_forward_tiled_pre(const DtqnNet* net, const float* theta, const float* xemb, const uint8_t* actions, int batch, int n,
                                      float* q_out, float* workspace, int train_mode, uint32_t dropout_seed, uint32_t dropout_step, void* stream) {
    if (!net || net->img_c <= 0 || !xemb) return DTQN_ERR_ARG;
    return forward_tiled_impl(net, theta, nullptr, actions, nullptr, nullptr, batch, n, n, q_out, workspace, stream, train_mode, dropout_seed,
                              dropout_step, nullptr, xemb);
}

extern "C" int dtqn_forward_tiled(const DtqnNet* net, const float* theta, const float* obs, const uint8_t* actions,
                                  int batch, int n, float* q_out, float* workspace, void* stream) {
    return dtqn_forward_tiled_strided(net, theta, obs, actions, batch, n, n, q_out, workspace, stream);
}

extern "C" int dtqn_forward_bag(const DtqnNet* net, const float* theta, const float* obs, const uint8_t* actions, const float* bag_obs,
                                const uint8_t* bag_actions, int batch, int n, float* q_out, float* workspace, int train_mode,
                                uint32_t dropout_seed, uint32_t dropout_step, void* stream) {
    if (!net || net->bag_size < 1) return DTQN_ERR_ARG;
    return forward_tiled_impl(net, theta, obs, actions, bag_obs, bag_actions, batch, n, n, q_out, workspace, stream, train_mode, dropout_seed,
                              dropout_step);
}
