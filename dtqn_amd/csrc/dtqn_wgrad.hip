// Weight gradients of the DTQN policy network: dW[N][K] = sum over all sampled tokens dY[tok][N]^T X[tok][K].
//
// Replaces the parameter-gradient half of loss.backward() (dtqn/agents/dtqn.py:256).  The contraction
// runs over the TOKEN axis (B*LP rows of the per-sequence act / grd records), so it is a real
// GEMM with K_contract = B*LP: it goes on the f32 matrix core.  Two kernels:
//   * dtqn_wgrad_kernel (large batches): split-K over the batch, summed by dtqn_td_reduce;
//   * dtqn_wgrad_direct_kernel (B*LP <= 2048 tokens, further down): small output tiles, no splits, writes grad itself.
// dtqn_wgrad_kernel: a workgroup owns one 64x64 block of
// one dW and one split of the batch; a 64-wide block is FOUR INTERLEAVED 16-row MFMA tiles
// (tile c holds rows 4*i + c), so one lane's A operands for the four tiles are 4 consecutive floats
// of one dY row and its B operands 4 consecutive floats of one X row: every global access is a
// coalesced 16 B/lane load, 16 MFMAs per pair of loads.  The four waves take different sequences
// and are summed through LDS; splits are summed by dtqn_td_reduce (deterministic, no atomics).
#include <mutex>
#include "dtqn_device.hpp"
#include "dtqn_wgrad_direct.hpp"

namespace dtqn {

constexpr int kMaxWJobs = 48;          // 1 + 4*NL + 12 (gru) + 2 <= 47 for NL <= 8; the kernarg block stays under 4 KB
struct WgradArgs {
    DtqnNet net;
    DtqnWJob jobs[kMaxWJobs];           // by value: the job lookup is a scalar-cache walk, not L2 round trips
    const float* act;
    const float* grd;
    float* gsplit;          // [n_split][n_trainable]
    const float* small;     // [B][sp_stride] per-sequence partials of the backward kernel
    int batch, n_split, n_jobs;
    int n_small;            // elements of the small-partials index space (LN affine, tables, learned pos table)
    int row_split;          // small-partial records per sequence (one per workgroup of the backward kernel)
    int small_blocks;       // blocks per split that sum the small partials
    int xcd_map;            // 1: (split, tile) pairs dealt to the XCDs in contiguous runs (see the kernel); 0: pair = block (A/B timing)
    int n_wtiles;           // 64 x 64 tiles of the jobs in `jobs` (net.n_wtiles, or fewer when dtqn_wgrad_lds_kernel takes the large matrices)
};

// Extra blocks of the same launch: sum the backward kernel's per-sequence partials (LayerNorm gamma/beta,
// embedding tables, learned position table = dL/dx0 summed over sequences) over this split's sequences and
// drop them into gsplit at their parameter offsets, so that dtqn_td_reduce is one uniform sum over splits.
__device__ __forceinline__ void small_partials_block(const WgradArgs& a, int sb, int split) {
    const DtqnNet& net = a.net;
    const int D = net.d_model;
    const int n_ln = net.num_layers * 4 * D;
    const int n_tab = net.discrete ? net.vocab * net.embed_per_obs : 0;
    const int n_act = net.action_dim > 0 ? net.num_actions * net.action_dim : 0;
    const int b_lo = (int)((long long)a.batch * split / a.n_split);
    const int b_hi = (int)((long long)a.batch * (split + 1) / a.n_split);
    float* out = a.gsplit + (size_t)split * net.n_trainable;
    for (int e = sb * 1024 + (int)threadIdx.x; e < min(a.n_small, (sb + 1) * 1024); e += DTQN_THREADS) {
        int dst, src;
        const float* base = a.small;
        size_t stride = (size_t)net.sp_stride;
        if (e < n_ln) {
            const int l = e / (4 * D);
            dst = net.off_layer0 + l * net.layer_stride + (e - l * 4 * D);
            src = net.so_ln + e;
        } else if (e < n_ln + n_tab) {
            dst = net.off_obs_tab + (e - n_ln);
            src = net.so_tab + (e - n_ln);
        } else if (e < n_ln + n_tab + n_act) {
            dst = net.off_act_emb + (e - n_ln - n_tab);
            src = net.so_act + (e - n_ln - n_tab);
        } else {
            dst = net.off_pos + (e - n_ln - n_tab - n_act);
            src = net.go_dx0 + (e - n_ln - n_tab - n_act);
            base = a.grd;
            stride = (size_t)net.grd_stride;
        }
        float v = 0.f;
        const int per = base == a.small ? a.row_split * net.sp_parts : 1;    // the grd record is per sequence, the small records per workgroup
        for (int b = b_lo * per; b < b_hi * per; ++b) v += base[(size_t)b * stride + src];
        out[dst] = v;
    }
}

__global__ __launch_bounds__(DTQN_THREADS, 2) void dtqn_wgrad_kernel(WgradArgs a) {
    const Thr t = make_thr();
    const DtqnNet& net = a.net;
    const int LP = net.lp;
    // (split, tile) pairs in split-major order, P of them.  Workgroup b runs on XCD b % 8 (round-robin dispatch); it takes
    // pair (b % 8) * ceil(P / 8) + b / 8, so every XCD works through a contiguous run of pairs: the tiles of one weight
    // matrix and one token range one after the other, which share their dY / X columns -- re-reads of the records then hit
    // that XCD's L2 instead of each of the eight L2s pulling its own copy out of MALL / HBM.
    const int P = a.n_wtiles * a.n_split, per = (P + 7) / 8, tile_blocks = per * 8, id = (int)blockIdx.x;
    if (id >= tile_blocks) {
        small_partials_block(a, (id - tile_blocks) % a.small_blocks, (id - tile_blocks) / a.small_blocks);
        return;
    }
    const int pair = a.xcd_map ? (id % 8) * per + id / 8 : id;
    if (pair >= P) return;
    // locate the job of this block
    const int split = pair / a.n_wtiles, tile = pair - split * a.n_wtiles;
    int j = 0;
    for (int k = 1; k < a.n_jobs; ++k)
        if (a.jobs[k].tile0 <= tile) j = k;
    const DtqnWJob& job = a.jobs[j];
    const int local = tile - job.tile0;
    const int bn = local / job.tiles_k, bk = local - bn * job.tiles_k;
    const int nbase = bn * 64, kbase = bk * 64;
    const float* xbase = (job.x_in_act ? a.act : a.grd) + job.x_off;
    const size_t xstride = job.x_in_act ? (size_t)net.act_stride : (size_t)net.grd_stride;
    const float* ybase = a.grd + job.dy_off;
    const size_t ystride = (size_t)net.grd_stride;
    // this lane's 4 consecutive dY columns / X columns
    const int ycol = nbase + 4 * t.i, xcol = kbase + 4 * t.i;
    const bool yok = ycol < job.ldy, xok = xcol < job.ldx;     // ld* are multiples of 4: whole float4 in range

    f32x4 acc[4][4];
#pragma unroll
    for (int cn = 0; cn < 4; ++cn)
#pragma unroll
        for (int ck = 0; ck < 4; ++ck) acc[cn][ck] = zero4();
    float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);

    const int b_lo = (int)((long long)a.batch * split / a.n_split);
    const int b_hi = (int)((long long)a.batch * (split + 1) / a.n_split);
    // work unit = (sequence of this split, 16 of its tokens = four 4-token MFMA steps); units are dealt to the 4 waves, so
    // even a one-sequence split keeps every wave busy.  The eight operand loads of a unit go in flight at once and units are
    // fetched ahead of the one that multiplies -- two ahead for the long contexts of the row-block path (three register buffers, alternating by name): the records come out of
    // MALL / HBM (memory-side latency of microseconds), a unit's 64 MFMAs last about one.  Padded contexts that are not a
    // multiple of 16 rows (none today) take the step-wise loop below.
    const bool fast = LP % 16 == 0;
    const int NSUB = fast ? LP / 16 : 4;
    const int steps_per_sub = fast ? 4 : (LP / 4 + NSUB - 1) / NSUB;
    const int units = (b_hi - b_lo) * NSUB * job.n_layers;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    auto unit_ptrs = [&](int u, const float*& yp, const float*& xp, int& s_lo, int& s_hi) {
        const int lyr = u / ((b_hi - b_lo) * NSUB), ul = u - lyr * (b_hi - b_lo) * NSUB;
        const int b = b_lo + ul / NSUB, sub = ul - (ul / NSUB) * NSUB;
        s_lo = sub * steps_per_sub;
        s_hi = min(LP / 4, s_lo + steps_per_sub);
        yp = ybase + (size_t)b * ystride + (size_t)lyr * job.dy_lstride + (size_t)t.kq * job.ldy + ycol;
        xp = xbase + (size_t)b * xstride + (size_t)lyr * job.x_lstride + (size_t)t.kq * job.ldx + xcol;
    };
    if (fast) {
        float4 av[3][4], bv[3][4];
        auto unit_load = [&](int u, float4 (&a4)[4], float4 (&b4)[4]) {
            const float *yp, *xp;
            int s_lo, s_hi;
            unit_ptrs(u, yp, xp, s_lo, s_hi);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                a4[k] = yok ? ld4(yp + (size_t)4 * (s_lo + k) * job.ldy) : z4;
                b4[k] = xok ? ld4(xp + (size_t)4 * (s_lo + k) * job.ldx) : z4;
            }
        };
        auto unit_mma = [&](const float4 (&a4)[4], const float4 (&b4)[4]) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                bsum.x += a4[k].x; bsum.y += a4[k].y; bsum.z += a4[k].z; bsum.w += a4[k].w;
                const float aa[4] = {a4[k].x, a4[k].y, a4[k].z, a4[k].w};
                const float bb[4] = {b4[k].x, b4[k].y, b4[k].z, b4[k].w};
#pragma unroll
                for (int cn = 0; cn < 4; ++cn)
#pragma unroll
                    for (int ck = 0; ck < 4; ++ck) acc[cn][ck] = mfma16(aa[cn], bb[ck], acc[cn][ck]);
            }
        };
        constexpr int W = DTQN_WAVES;
        int u = t.wave;
        if (u < units) unit_load(u, av[0], bv[0]);
        if (LP <= 64) {
            // 64-row records (whole-sequence kernels): one unit ahead is enough (measured: two ahead costs cfg 2 / 3 about 1 %)
            for (; u < units; u += 2 * W) {
                if (u + W < units) unit_load(u + W, av[1], bv[1]);
                unit_mma(av[0], bv[0]);
                if (u + W < units) {
                    if (u + 2 * W < units) unit_load(u + 2 * W, av[0], bv[0]);
                    unit_mma(av[1], bv[1]);
                }
            }
        } else {
            if (u + W < units) unit_load(u + W, av[1], bv[1]);
            for (; u < units; u += 3 * W) {                // three units per trip: the buffers rotate without dynamic indexing
                if (u + 2 * W < units) unit_load(u + 2 * W, av[2], bv[2]);
                unit_mma(av[0], bv[0]);
                if (u + W < units) {
                    if (u + 3 * W < units) unit_load(u + 3 * W, av[0], bv[0]);
                    unit_mma(av[1], bv[1]);
                    if (u + 2 * W < units) {
                        if (u + 4 * W < units) unit_load(u + 4 * W, av[1], bv[1]);
                        unit_mma(av[2], bv[2]);
                    }
                }
            }
        }
    }
    for (int u = fast ? units : t.wave; u < units; u += DTQN_WAVES) {
        const float *yp, *xp;
        int s_lo, s_hi;
        unit_ptrs(u, yp, xp, s_lo, s_hi);
        // general shape: explicit 2-deep pipeline, the operands of step s+1 are in flight while step s multiplies
        float4 av = (yok && s_lo < s_hi) ? ld4(yp + (size_t)4 * s_lo * job.ldy) : z4;
        float4 bv = (xok && s_lo < s_hi) ? ld4(xp + (size_t)4 * s_lo * job.ldx) : z4;
#pragma unroll 2
        for (int s = s_lo; s < s_hi; ++s) {
            const float4 an = (yok && s + 1 < s_hi) ? ld4(yp + (size_t)4 * (s + 1) * job.ldy) : z4;
            const float4 bn = (xok && s + 1 < s_hi) ? ld4(xp + (size_t)4 * (s + 1) * job.ldx) : z4;
            bsum.x += av.x; bsum.y += av.y; bsum.z += av.z; bsum.w += av.w;
            const float aa[4] = {av.x, av.y, av.z, av.w};
            const float bb[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
            for (int cn = 0; cn < 4; ++cn)
#pragma unroll
                for (int ck = 0; ck < 4; ++ck) acc[cn][ck] = mfma16(aa[cn], bb[ck], acc[cn][ck]);
            av = an;
            bv = bn;
        }
    }
    // bias: sum over the 4 token phases (kq) of this lane's columns (dY column n = nbase + 4*i + c)
    bsum.x += __shfl_xor(bsum.x, 16); bsum.y += __shfl_xor(bsum.y, 16); bsum.z += __shfl_xor(bsum.z, 16); bsum.w += __shfl_xor(bsum.w, 16);
    bsum.x += __shfl_xor(bsum.x, 32); bsum.y += __shfl_xor(bsum.y, 32); bsum.z += __shfl_xor(bsum.z, 32); bsum.w += __shfl_xor(bsum.w, 32);

    // cross-wave sum through LDS: every wave publishes its 64x64 slab (+ a bias row) in output layout, then all
    // 256 threads add the four slabs in a fixed order and store coalesced rows (deterministic, few registers)
    constexpr int SLD = 68;                                    // slab leading dimension (floats)
    float* slab = reinterpret_cast<float*>(dtqn_smem) + (size_t)t.wave * 65 * SLD;
    // acc[cn][ck][r]: n = 4*(kq*4 + r) + cn,  k = 4*i + ck
#pragma unroll
    for (int cn = 0; cn < 4; ++cn)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            st4(slab + (4 * (t.kq * 4 + r) + cn) * SLD + 4 * t.i,
                make_float4(acc[cn][0][r], acc[cn][1][r], acc[cn][2][r], acc[cn][3][r]));
    if (t.kq == 0) st4(slab + 64 * SLD + 4 * t.i, bsum);
    __syncthreads();
    const float* s0 = reinterpret_cast<const float*>(dtqn_smem);
    float* out = a.gsplit + (size_t)split * net.n_trainable;
    for (int idx = t.tid; idx < 64 * 16; idx += DTQN_THREADS) {
        const int nl = idx >> 4, k4 = (idx & 15) * 4;
        const int n = nbase + nl, k = kbase + k4;
        if (n < job.N && k < job.K) {
            float4 v = ld4(s0 + nl * SLD + k4);
#pragma unroll
            for (int wv = 1; wv < DTQN_WAVES; ++wv) {
                const float4 x = ld4(s0 + (size_t)wv * 65 * SLD + nl * SLD + k4);
                v.x += x.x; v.y += x.y; v.z += x.z; v.w += x.w;
            }
            float* op = out + job.w_off + (size_t)n * job.K + k;
            if (k + 3 < job.K && (job.K & 3) == 0) {
                st4(op, v);
            } else {
                const float vv[4] = {v.x, v.y, v.z, v.w};
                for (int c = 0; c < 4; ++c)
                    if (k + c < job.K) op[c] = vv[c];
            }
        }
    }
    if (job.b_off >= 0 && bk == 0 && t.tid < 64) {
        const int n = nbase + t.tid;
        if (n < job.N) {
            float v = 0.f;
#pragma unroll
            for (int wv = 0; wv < DTQN_WAVES; ++wv) v += s0[(size_t)wv * 65 * SLD + 64 * SLD + t.tid];
            out[job.b_off + n] = v;
        }
    }
}

// ---- large matrices of the row-block networks: 128 x 128 output tiles, operands staged through LDS (round 6) -------------------------
// dtqn_wgrad_kernel above gives every WAVE its own pair of operand loads per 16 MFMAs: 64 x 64 tiles straight out of L2, 8 waves per
// compute unit (160 registers each), 0.55 of the f32 matrix peak at BASELINE configs 3 - 5, where it is 15 - 17 % of the update.  Here a
// workgroup of 8 waves owns a 128 x 128 block of one dW and one split of the batch; a chunk of 32 tokens of its dY columns
// ([32][128]) and X columns ([32][128]) is loaded ONCE by the whole workgroup as whole 512-byte row pieces -- twice the MACs per byte
// fetched -- into registers while the previous chunk multiplies, dropped into LDS between two barriers, and read from there as MFMA
// operands: wave (wn, wk) owns rows [64 wn, 64 wn + 64) x columns [32 wk, 32 wk + 32) of the block as 4 x 2 interleaved 16 x 16 tiles
// (tile cn holds rows 4 i + cn, tile ck columns 2 i + ck), so that one 16-byte and one 8-byte LDS read feed 8 MFMAs.  Leading
// dimensions 128 / 160 words: the lane groups of ds_read_b128 / ds_read_b64 each cover the 64 banks once (MI355X_MICROARCH.md, LDS).
// Same contraction as loss.backward() (dtqn/agents/dtqn.py:256); splits are summed by dtqn_td_reduce in a fixed order.
constexpr int kLdsTK = 32, kLdsLDY = 128, kLdsLDX = 160, kLdsThreads = 512;
struct WgradLdsJob {
    DtqnWJob j;
    int tile0;              // first 128 x 128 tile of this job in the launch
};
struct WgradLdsArgs {
    DtqnNet net;
    WgradLdsJob jobs[kMaxWJobs];
    const float* act;
    const float* grd;
    float* gsplit;
    int batch, n_split, n_jobs, n_tiles;
};
__global__ __launch_bounds__(kLdsThreads, 2) void dtqn_wgrad_lds_kernel(WgradLdsArgs a) {
    const Thr t = make_thr();
    const DtqnNet& net = a.net;
    const int LP = net.lp;
    float* Ys = reinterpret_cast<float*>(dtqn_smem);               // [32][128] dY columns of the chunk
    float* Xs = Ys + kLdsTK * kLdsLDY;                             // [32][160] X columns of the chunk
    const int pair = (int)blockIdx.x, split = pair / a.n_tiles, tile = pair - split * a.n_tiles;
    int jx = 0;
    for (int k = 1; k < a.n_jobs; ++k)
        if (a.jobs[k].tile0 <= tile) jx = k;
    const DtqnWJob& job = a.jobs[jx].j;
    const int local = tile - a.jobs[jx].tile0, tk = job.K / 128;
    const int bn = local / tk, bk = local - bn * tk, nbase = bn * 128, kbase = bk * 128;
    const float* xbase = (job.x_in_act ? a.act : a.grd) + job.x_off + kbase;
    const size_t xstride = job.x_in_act ? (size_t)net.act_stride : (size_t)net.grd_stride;
    const float* ybase = a.grd + job.dy_off + nbase;
    const size_t ystride = (size_t)net.grd_stride;
    const int b_lo = (int)((long long)a.batch * split / a.n_split), b_hi = (int)((long long)a.batch * (split + 1) / a.n_split);
    // chunks of 32 rows; the rows behind the context carry zero gradients (the loss runs over the L real positions): chunks that
    // hold none of the first L rows are skipped (a bag network's embedding job contracts up to LP bag entries: it keeps every chunk)
    const int live_rows = net.bag_size > 0 ? LP : net.ctx_len;
    const int cps = (live_rows + kLdsTK - 1) / kLdsTK, nb = b_hi - b_lo, nchunks = job.n_layers * nb * cps;
    f32x4 acc[4][2];
#pragma unroll
    for (int cn = 0; cn < 4; ++cn) { acc[cn][0] = zero4(); acc[cn][1] = zero4(); }
    float bsum = 0.f;
    const bool want_bias = job.b_off >= 0 && bk == 0;
    // staging: thread -> (row, 16-byte piece) of each tile, two pieces per tile
    const int sr = t.tid >> 5, sc = (t.tid & 31) * 4;
    float4 yr[2], xr[2];
    auto chunk_load = [&](int c) {
        const int lyr = c / (nb * cps), rem = c - lyr * nb * cps, b = rem / cps, r0 = (rem - b * cps) * kLdsTK;
        const float* yp = ybase + (size_t)(b_lo + b) * ystride + (size_t)lyr * job.dy_lstride + (size_t)r0 * job.ldy + sc;
        const float* xp = xbase + (size_t)(b_lo + b) * xstride + (size_t)lyr * job.x_lstride + (size_t)r0 * job.ldx + sc;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            yr[k] = ld4(yp + (size_t)(sr + 16 * k) * job.ldy);
            xr[k] = ld4(xp + (size_t)(sr + 16 * k) * job.ldx);
        }
    };
    const int wn = t.wave >> 2, wk = t.wave & 3;
    const float* ya = Ys + t.kq * kLdsLDY + 64 * wn + 4 * t.i;
    const float* xa = Xs + t.kq * kLdsLDX + 32 * wk + 2 * t.i;
    if (nchunks > 0) chunk_load(0);
    for (int c = 0; c < nchunks; ++c) {
        __syncthreads();                                               // the previous chunk's tiles are consumed
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            st4(Ys + (sr + 16 * k) * kLdsLDY + sc, yr[k]);
            st4(Xs + (sr + 16 * k) * kLdsLDX + sc, xr[k]);
        }
        __syncthreads();
        if (c + 1 < nchunks) chunk_load(c + 1);                        // in flight while this chunk multiplies
        // 4-row steps of this chunk that hold a live row (L = 50: rows 32 .. 49 of the second chunk -> 5 of its 8 steps; the rows behind
        // the context carry zero gradients)
        const int r0c = (c % cps) * kLdsTK, smax = live_rows - r0c >= kLdsTK ? kLdsTK / 4 : (live_rows - r0c + 3) / 4;
#pragma unroll
        for (int s = 0; s < kLdsTK / 4; ++s) {
            if (s >= smax) break;
            const float4 a4 = ld4(ya + 4 * s * kLdsLDY);
            const float2 b2 = *reinterpret_cast<const float2*>(xa + 4 * s * kLdsLDX);
            const float aa[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
            for (int cn = 0; cn < 4; ++cn) {
                acc[cn][0] = mfma16(aa[cn], b2.x, acc[cn][0]);
                acc[cn][1] = mfma16(aa[cn], b2.y, acc[cn][1]);
            }
        }
        if (want_bias && t.tid < 128) {
#pragma unroll 8
            for (int r = 0; r < kLdsTK; ++r) bsum += Ys[r * kLdsLDY + t.tid];
        }
    }
    // acc[cn][ck][r] at lane (i, kq) = dW[nbase + 64 wn + 4 (4 kq + r) + cn][kbase + 32 wk + 2 i + ck]
    float* out = a.gsplit + (size_t)split * net.n_trainable;
#pragma unroll
    for (int cn = 0; cn < 4; ++cn)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = nbase + 64 * wn + 4 * (4 * t.kq + r) + cn, k = kbase + 32 * wk + 2 * t.i;
            *reinterpret_cast<float2*>(out + job.w_off + (size_t)n * job.K + k) = make_float2(acc[cn][0][r], acc[cn][1][r]);
        }
    if (want_bias && t.tid < 128) out[job.b_off + nbase + t.tid] = bsum;
}
// jobs dtqn_wgrad_lds_kernel takes: both dimensions whole 128-blocks (in_proj, out_proj, ffn.0, ffn.2, the first head matrix and
// the GRU gate matrices of a row-block network with d_model 128 / 256)
static inline bool wgrad_lds_job(const DtqnNet& net, const DtqnWJob& j) {
    return net.tiled && j.N % 128 == 0 && j.K % 128 == 0 && j.ldy % 4 == 0 && j.ldx % 4 == 0;
}
static inline bool wgrad_lds_enabled(const DtqnNet& net) {
    const char* e = getenv("DTQN_WGRAD_LDS");
    return net.tiled && net.d_model % 128 == 0 && net.img_c <= 0 && !(e != nullptr && atoi(e) == 0);
}

// ---- small batches: one launch, no split-K -----------------------------------------------------------------------------
// With B * LP <= kDirectMaxTokens (2048) the token axis is short enough for ONE workgroup to contract it all, so the splits and
// the dtqn_td_reduce launch disappear: the output is cut into many SMALL tiles instead (16 dY columns x 32 X columns, a
// few hundred workgroups), each written straight into `grad` together with its sum of squares for the global norm.
// Small tiles re-read the records once per tile; to keep those re-reads inside one L2, consecutive tiles (= the tiles of
// one weight matrix, which share their dY / X columns) are dealt to the SAME XCD: workgroup b runs on XCD b % 8
// (round-robin dispatch) and takes tile (b % 8) * per_xcd + b / 8 (direct_plan).  Inside a matrix the tiles run along its
// SHORTER tile axis first, so that a run of consecutive tiles touches as few distinct column blocks as possible.
// The tile / partial-block bodies live in dtqn_wgrad_direct.hpp: the row-slice backward launch runs the same code in its
// own weight-gradient workgroups (dtqn_td_wgrad_is_fused), and this launch then does not happen at all.
struct WgradDirectArgs {
    DtqnNet net;
    DtqnWJob jobs[kMaxWJobs];
    int dtile0[kMaxWJobs];  // first direct tile of each job
    const float* act;
    const float* grd;
    float* grad;            // [n_trainable]
    const float* small;
    float* norm_partial;    // [n_parts]
    int32_t* step_counter;
    int batch, n_jobs, n_small, row_split;
    int n_tiles, n_parts;
    int slots;              // tile workgroups per XCD: workgroup b (on XCD b % 8) takes tile xcd_tile0[b % 8] + b / 8
    int xcd_tile0[8], xcd_ntiles[8];
    int xcd_map;            // 0: tile = workgroup index (A/B timing)
    int small_per;          // per-sequence-partial elements per workgroup: kDSmall, or 2 kDSmall where that brings the grid down to one round
};

__global__ __launch_bounds__(kDirectThreads) void dtqn_wgrad_direct_kernel(WgradDirectArgs a) {
    const Thr t = make_thr();
    const DtqnNet& net = a.net;
    const int tid = t.tid;
    float* red = reinterpret_cast<float*>(dtqn_smem);                     // [waves] block reduction of the sum of squares
    float* slabs = red + kDirectWaves;
    float ss = 0.f;                                                       // this thread's share of sum(g^2)
    const int b = (int)blockIdx.x, tile_blocks = a.slots * 8;
    if (b == 0 && tid == 0) a.step_counter[0] = a.step_counter[1];        // publish the step count of the previous update
    if (b == 0)                                                           // partials nobody writes this time
        for (int i = (int)gridDim.x + tid; i < a.n_parts; i += kDirectThreads) a.norm_partial[i] = 0.f;
    const int tile = b >= tile_blocks ? a.n_tiles : !a.xcd_map ? b : b / 8 < a.xcd_ntiles[b % 8] ? a.xcd_tile0[b % 8] + b / 8 : a.n_tiles;
    const DirectCtx ctx{a.act, a.grd, a.small, a.grad, a.batch, a.row_split, a.n_small};
    if (b >= tile_blocks) {
        // (same sums in the same order whatever the block size: an element's records are added wave by wave, then the waves in order)
        ss = a.small_per == 2 * kDSmall ? direct_small<false, 2 * kDSmall, kDirectWaves>(net, ctx, b - tile_blocks, slabs, t)
                                        : direct_small<false, kDSmall, kDirectWaves>(net, ctx, b - tile_blocks, slabs, t);
    } else if (tile < a.n_tiles) {
        int j = 0;
        for (int k = 1; k < a.n_jobs; ++k)
            if (a.dtile0[k] <= tile) j = k;
        ss = direct_tile<false, kDirectWaves>(net, ctx, a.jobs[j], tile - a.dtile0[j], slabs, t);
    }
    direct_norm_partial<kDirectWaves>(ss, red, a.norm_partial, b, t);
}

// direct tiles of a net: per job ceil(N / 16) * ceil(K / 32)
static int direct_tiles(const DtqnNet* net, const DtqnWJob* jobs, int* dtile0) {
    int tiles = 0;
    for (int j = 0; j < net->n_wjobs; ++j) {
        if (dtile0) dtile0[j] = tiles;
        tiles += direct_tiles_of(jobs[j]);
    }
    return tiles;
}
struct DirectPlan {
    int dtile0[kMaxWJobs];
    int n_tiles, slots, grid;
    int small_per;          // elements of the per-sequence partials per workgroup
    int grid_max;           // the grid with the smaller blocks (what norm_partial is sized for)
    int xcd_tile0[8], xcd_ntiles[8];
};
// Which tiles each XCD takes: eight equal runs of consecutive tiles.  (Measured at cfg 1, batch 32: tile = workgroup
// index, i.e. every XCD sees every matrix, 116.7 us / update; equal runs 113.3; whole matrices per XCD, which leaves
// XCDs with 8 - 32 tiles, 116.4.)  DTQN_WGRAD_XCD=0 gives the first for A/B timing.
static DirectPlan direct_plan(const DtqnNet* net, const DtqnWJob* jobs) {
    DirectPlan p;
    p.n_tiles = direct_tiles(net, jobs, p.dtile0);
    const int per = (p.n_tiles + 7) / 8;
    for (int x = 0; x < 8; ++x) {
        p.xcd_tile0[x] = x * per < p.n_tiles ? x * per : p.n_tiles;
        p.xcd_ntiles[x] = p.n_tiles - p.xcd_tile0[x] < per ? p.n_tiles - p.xcd_tile0[x] : per;
    }
    p.slots = per;
    // The blocks that sum the per-sequence partials are short; with kDSmall elements each, BASELINE config 1 is 240 + 29 = 269
    // workgroups on 256 compute units -- 13 of them wait for a second round.  Twice the elements per block (255 workgroups) is one
    // round; taken only when it makes that difference.  DTQN_DSMALL=<kDSmall | 2 kDSmall> overrides (A/B timing).
    static_assert(2 * kDSmall / 64 <= kDirectWaves && (size_t)(2 * kDSmall / 64) * kDirectWaves * 64 <= kDirectLdsFloats - kDirectWaves, "small-partial block fits");
    const int n_small = small_elems(net), g1 = p.slots * 8 + (n_small + kDSmall - 1) / kDSmall, g2 = p.slots * 8 + (n_small + 2 * kDSmall - 1) / (2 * kDSmall);
    const int cus = fuse_cu_count();
    p.small_per = (cus > 0 && g1 > cus && g2 <= cus) ? 2 * kDSmall : kDSmall;
    const char* e = getenv("DTQN_DSMALL");
    if (e != nullptr && (atoi(e) == kDSmall || atoi(e) == 2 * kDSmall)) p.small_per = atoi(e);
    p.grid = p.small_per == kDSmall ? g1 : g2;
    p.grid_max = g1;
    return p;
}

}  // namespace dtqn

using namespace dtqn;

// Batch splits of the large-batch weight-gradient launches that fill the chip once: with dtqn_wgrad_lds_kernel in play (row-block
// networks of d_model 128 / 256) as many as put one round of its 128 x 128 tiles on the 2 x 256 slots; else min(batch, 16).
extern "C" int dtqn_td_wgrad_splits(const DtqnNet* net, int batch) {
    if (!net || batch < 1) return 1;
    int n = batch < 16 ? batch : 16;
    if (net->n_wjobs <= kMaxWJobs && wgrad_lds_enabled(*net)) {
        DtqnWJob jobs[kMaxWJobs];
        if (dtqn_net_wjobs(net, jobs) == DTQN_OK) {
            int tiles = 0;
            for (int j = 0; j < net->n_wjobs; ++j)
                if (wgrad_lds_job(*net, jobs[j])) tiles += (jobs[j].N / 128) * (jobs[j].K / 128);
            if (tiles > 0) {
                const int cus = device_cu_count() > 0 ? device_cu_count() : 256;
                const char* e = getenv("DTQN_WGRAD_SLOTS");            // resident workgroups per compute unit the split count aims at (A/B timing)
                const int per_cu = e != nullptr && atoi(e) > 0 ? atoi(e) : 2;
                n = per_cu * cus / tiles;
                n = n < 1 ? 1 : (n > batch ? batch : n);
                n = n > 64 ? 64 : n;
            }
        }
    }
    return n;
}

// One launch, no splits, no dtqn_td_reduce: when the whole batch is at most kDirectMaxTokens tokens (DTQN_WGRAD_DIRECT=0/1
// overrides, for A/B timing).
extern "C" int dtqn_td_wgrad_is_direct(const DtqnNet* net, int batch) {
    if (!net || batch < 1 || net->n_wjobs > kMaxWJobs) return 0;
    if (net->img_c > 0) return 0;      // image nets: the encoder's gradients join the others in dtqn_td_reduce (split 0 of gsplit)
    const char* e = getenv("DTQN_WGRAD_DIRECT");
    if (e != nullptr) return atoi(e) != 0 ? 1 : 0;
    // shared GRU gate matrices contract over the tokens of every layer (measured at cfg-1 shapes with GRU gates, 2 layers:
    // 257 us per update direct, 247 us split)
    // (the embedding job of a bag network contracts the context tokens and the bag entries: two token sets)
    const int gru_layers = net->gate == DTQN_GATE_GRU ? net->num_layers : 1, bag_sets = net->bag_size > 0 ? 2 : 1;
    const int layers = gru_layers > bag_sets ? gru_layers : bag_sets;
    return (long long)batch * net->lp * layers <= kDirectMaxTokens ? 1 : 0;
}

// Length of DtqnTd.norm_partial: the sum-of-squares partials of whichever kernel assembles `grad` (dtqn_td_reduce /
// dtqn_td_gradnorm: one per Adam block; the direct weight-gradient kernel: one per workgroup).
extern "C" int dtqn_td_norm_partials(const DtqnNet* net) {
    if (!net || net->n_wjobs > kMaxWJobs) return 0;
    DtqnWJob jobs[kMaxWJobs];
    if (dtqn_net_wjobs(net, jobs) != DTQN_OK) return 0;
    const int opt = (net->n_trainable + 1023) / 1024, dir = direct_plan(net, jobs).grid_max;
    return opt > dir ? opt : dir;
}

static int wgrad_direct(const DtqnNet* net, const DtqnTd* td, hipStream_t stream) {
    WgradDirectArgs a;
    a.net = *net;
    if (dtqn_net_wjobs(net, a.jobs) != DTQN_OK) return DTQN_ERR_CONFIG;
    for (int j = 0; j < net->n_wjobs; ++j)      // a wave's units must fit the register buffers direct_tile was compiled with
        if ((long long)td->batch * net->lp * a.jobs[j].n_layers > kDirectMaxTokens) return DTQN_ERR_CONFIG;
    const DirectPlan plan = direct_plan(net, a.jobs);
    a.n_tiles = plan.n_tiles; a.slots = plan.slots; a.small_per = plan.small_per;
    for (int j = 0; j < net->n_wjobs; ++j) a.dtile0[j] = plan.dtile0[j];
    for (int x = 0; x < 8; ++x) { a.xcd_tile0[x] = plan.xcd_tile0[x]; a.xcd_ntiles[x] = plan.xcd_ntiles[x]; }
    a.act = td->act; a.grd = td->grd; a.grad = td->grad; a.small = td->small;
    a.norm_partial = td->norm_partial; a.step_counter = td->step_counter;
    a.batch = td->batch; a.n_jobs = net->n_wjobs; a.n_small = small_elems(net);
    a.row_split = td->row_split > 1 ? td->row_split : 1;
    a.n_parts = dtqn_td_norm_partials(net);
    const char* xm = getenv("DTQN_WGRAD_XCD");
    a.xcd_map = xm != nullptr ? atoi(xm) : 1;
    const int grid = plan.grid;
    const size_t lds = kDirectLdsFloats * sizeof(float);
    (void)hipGetLastError();   // drop stale errors left by other users of the runtime
    hipLaunchKernelGGL(dtqn_wgrad_direct_kernel, dim3(grid), dim3(kDirectThreads), lds, stream, a);
    return hipGetLastError() == hipSuccess ? DTQN_OK : DTQN_ERR_LAUNCH;
}

// Fork-join lane of the split weight gradients (round 6).  The small jobs that stay on dtqn_wgrad_kernel next to dtqn_wgrad_lds_kernel
// (embedding matrix, last head matrix, the small-partial sums) are a hundred-odd workgroups whose waves walk a split's tokens one 16-token
// unit after the other: 37 / 66 / 52 us at BASELINE configs 4 / 3 / 5, latency-bound, on a chip the launch does not fill.  They depend
// on nothing the large-matrix kernel writes (disjoint ranges of gsplit), so they run on a second stream BESIDE it: fork event on the
// caller's stream, the small kernel on the lane's stream, join event back.  One lane per device, created on first use (a stream and two
// events: no device memory); DTQN_WGRAD_SIDE=0 keeps both launches on the caller's stream.
struct WgradLane {
    hipStream_t s;
    hipEvent_t fork, join;
    int state;                         // 0: not tried, 1: ready, -1: unavailable
};
static WgradLane* wgrad_lane() {
    static WgradLane lanes[kMaxDevices] = {};
    const char* e = getenv("DTQN_WGRAD_SIDE");
    if (e != nullptr && atoi(e) == 0) return nullptr;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return nullptr;
    WgradLane& l = lanes[dev];
    if (l.state == 0) {
        l.state = (hipStreamCreateWithFlags(&l.s, hipStreamNonBlocking) == hipSuccess &&
                   hipEventCreateWithFlags(&l.fork, hipEventDisableTiming) == hipSuccess &&
                   hipEventCreateWithFlags(&l.join, hipEventDisableTiming) == hipSuccess) ? 1 : -1;
        (void)hipGetLastError();
    }
    return l.state == 1 ? &l : nullptr;
}

extern "C" int dtqn_td_wgrad(const DtqnNet* net, const DtqnTd* td, void* stream) {
    if (!net || !td || td->batch < 1 || td->n_split < 1) return DTQN_ERR_ARG;
    if (net->n_wjobs > kMaxWJobs) return DTQN_ERR_CONFIG;
    if (dtqn_td_wgrad_is_fused(net, td)) return DTQN_OK;       // the backward launch wrote grad / norm_partial itself
    if (dtqn_td_wgrad_is_direct(net, td->batch)) return wgrad_direct(net, td, (hipStream_t)stream);
    WgradArgs a;
    a.net = *net;
    if (dtqn_net_wjobs(net, a.jobs) != DTQN_OK) return DTQN_ERR_CONFIG;
    a.n_jobs = net->n_wjobs;
    a.n_wtiles = net->n_wtiles;
    WgradLdsArgs la;
    bool have_lds = false;
    if (wgrad_lds_enabled(*net)) {
        // the large matrices go to dtqn_wgrad_lds_kernel, the rest (embedding, last head matrix) stay here with their tiles renumbered
        la.net = *net;
        la.n_jobs = 0; la.n_tiles = 0;
        int keep = 0, tiles = 0;
        for (int j = 0; j < net->n_wjobs; ++j) {
            if (wgrad_lds_job(*net, a.jobs[j])) {
                la.jobs[la.n_jobs].j = a.jobs[j];
                la.jobs[la.n_jobs].tile0 = la.n_tiles;
                la.n_tiles += (a.jobs[j].N / 128) * (a.jobs[j].K / 128);
                ++la.n_jobs;
            } else {
                a.jobs[keep] = a.jobs[j];
                a.jobs[keep].tile0 = tiles;
                tiles += a.jobs[keep].tiles_n * a.jobs[keep].tiles_k;
                ++keep;
            }
        }
        if (la.n_jobs > 0) {
            a.n_jobs = keep; a.n_wtiles = tiles;
            la.act = td->act; la.grd = td->grd; la.gsplit = td->gsplit; la.batch = td->batch; la.n_split = td->n_split;
            have_lds = true;
        }
    }
    // the small jobs first, beside the large-matrix kernel when the lane is there
    WgradLane* lane = have_lds ? wgrad_lane() : nullptr;
    static std::mutex lane_mu;         // fork / launch / join of one update are not interleaved with another host thread's
    std::unique_lock<std::mutex> lk(lane_mu, std::defer_lock);
    hipStream_t small_stream = (hipStream_t)stream;
    if (lane != nullptr) {
        lk.lock();
        if (hipEventRecord(lane->fork, (hipStream_t)stream) == hipSuccess && hipStreamWaitEvent(lane->s, lane->fork, 0) == hipSuccess) small_stream = lane->s;
        else { lane = nullptr; (void)hipGetLastError(); }
    }
    a.act = td->act; a.grd = td->grd; a.gsplit = td->gsplit; a.small = td->small;
    a.batch = td->batch; a.n_split = td->n_split;
    a.row_split = td->row_split > 1 ? td->row_split : 1;
    a.n_small = net->num_layers * 4 * net->d_model + (net->discrete ? net->vocab * net->embed_per_obs : 0) +
                (net->action_dim > 0 ? net->num_actions * net->action_dim : 0) +
                (net->pos == DTQN_POS_LEARNED ? net->ctx_len * net->d_model : 0);
    const int small_blocks = (a.n_small + 1023) / 1024;
    a.small_blocks = small_blocks;
    const char* xm = getenv("DTQN_WGRAD_XCD");
    a.xcd_map = xm != nullptr ? atoi(xm) : 1;
    const int tile_blocks = (a.n_wtiles * td->n_split + 7) / 8 * 8;
    const size_t lds = (size_t)DTQN_WAVES * 65 * 68 * sizeof(float);
    static size_t attr_lds[kMaxDevices] = {};    // per device
    raise_lds_limit(reinterpret_cast<const void*>(&dtqn_wgrad_kernel), lds, attr_lds);
    (void)hipGetLastError();   // drop stale errors left by other users of the runtime
    hipLaunchKernelGGL(dtqn_wgrad_kernel, dim3(tile_blocks + small_blocks * td->n_split), dim3(DTQN_THREADS), lds, small_stream, a);
    if (hipGetLastError() != hipSuccess) return DTQN_ERR_LAUNCH;
    if (lane != nullptr && hipEventRecord(lane->join, lane->s) != hipSuccess) return DTQN_ERR_LAUNCH;
    if (have_lds) {
        const size_t llds = (size_t)kLdsTK * (kLdsLDY + kLdsLDX) * sizeof(float);
        (void)hipGetLastError();
        hipLaunchKernelGGL(dtqn_wgrad_lds_kernel, dim3(la.n_tiles * td->n_split), dim3(kLdsThreads), llds, (hipStream_t)stream, la);
        if (hipGetLastError() != hipSuccess) return DTQN_ERR_LAUNCH;
    }
    if (lane != nullptr && hipStreamWaitEvent((hipStream_t)stream, lane->join, 0) != hipSuccess) return DTQN_ERR_LAUNCH;
    return DTQN_OK;
}
