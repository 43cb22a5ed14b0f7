// Small-batch weight gradients (B * LP <= 2048 tokens): the work of ONE workgroup on one output tile / one block of the
// per-sequence partials, shared by dtqn_wgrad_direct_kernel (dtqn_wgrad.hip: its own launch behind the backward) and by the
// weight-gradient workgroups that ride in the row-slice BACKWARD launch (dtqn_backward.hip, "fused" mode: they start on a
// layer's tiles as soon as every sequence has published that layer's gradient records, while the data-gradient chain is
// still working on the layers below).
//
// Replaces the parameter-gradient half of loss.backward() (dtqn/agents/dtqn.py:256); see dtqn_wgrad.hip for the layout.
#pragma once
#include "dtqn_device.hpp"

namespace dtqn {

// waves per workgroup: a wave can have 64 vector-memory instructions outstanding (vmcnt), a 2048-token tile is 128 16-token units of
// 8 loads each -- with 16 waves every load of the tile is in flight at once (8 units per wave), with 8 waves half of them
// (measured round 3, cfg 1: 108.0 -> 106.6 us per update); the workgroups that ride in the backward launch have that kernel's 8 waves
#ifndef DTQN_DIRECT_WAVES
#define DTQN_DIRECT_WAVES 16
#endif
constexpr int kDirectWaves = DTQN_DIRECT_WAVES;
constexpr int kFuseWaves = 8;
constexpr int kDirectThreads = kDirectWaves * 64;
constexpr int kDirectMaxTokens = 2048;
constexpr int kDTN = 16, kDTK = 32;         // tile: dY columns x X columns
constexpr int kDGroup = 8;                  // 16-token units in flight per wave and buffer
#ifndef DTQN_DSMALL
#define DTQN_DSMALL 128
#endif
constexpr int kDSmall = DTQN_DSMALL;        // per-sequence-partial elements per workgroup (own launch)
constexpr int kDSmallFused = 256;           //   ... in the backward launch (fewer, larger blocks: at most one late item per workgroup)
constexpr size_t direct_lds_floats(int waves) { return waves + (size_t)waves * (kDTN + 1) * (kDTK + 4); }
constexpr size_t kDirectLdsFloats = direct_lds_floats(kDirectWaves);

struct DirectCtx {
    const float* act;
    const float* grd;
    const float* small;
    float* grad;
    int batch, row_split, n_small;
};

// SC1: the grd / small records were written by OTHER workgroups of the same launch with write-through (sc1) stores; they are read
// with sc1 loads (L1 bypassed; this XCD's L2 holds no older copy: nothing of this launch read them before their flag went up)
template <bool SC1>
__device__ __forceinline__ float direct_ld(const float* base, const DtqnRsrc& rs, size_t idx) {
    if constexpr (SC1) return dtqn_xch_load1(rs, (int)(idx * 4));
    else return base[idx];
}

// One 16 x 32 tile of one dW (+ its slice of db): contracts every token of the batch, writes `grad`, returns this thread's share
// of the tile's sum of squares.  `slabs`: kDirectWaves * (kDTN + 1) * (kDTK + 4) floats of LDS.  Contains one __syncthreads().
template <bool SC1, int WV>
__device__ __forceinline__ float direct_tile(const DtqnNet& net, const DirectCtx& a, const DtqnWJob& job, int local, float* slabs, const Thr& t) {
    const int LP = net.lp, tid = t.tid;
    float ss = 0.f;
    const int tiles_k = (job.K + kDTK - 1) / kDTK;
    const int tiles_n = (job.N + kDTN - 1) / kDTN;
    const bool k_major = tiles_k > tiles_n;                            // e.g. ffn.2: 4 x 8 tiles, walk the 4 first
    const int bn = k_major ? local % tiles_n : local / tiles_k, bk = k_major ? local / tiles_n : local - bn * tiles_k;
    const int nbase = bn * kDTN, kbase = bk * kDTK;
    const float* xbase = (job.x_in_act ? a.act : a.grd) + job.x_off;
    const size_t xstride = job.x_in_act ? (size_t)net.act_stride : (size_t)net.grd_stride;
    const size_t ystride = (size_t)net.grd_stride;
    const DtqnRsrc grs = DTQN_XCH_RSRC(a.grd, (size_t)a.batch * net.grd_stride * 4);
    // MFMA 16x16x4: A[n = i][token = kq] = dY[4s + kq][nbase + i];  B[token = kq][k-tile c, column i] = X[4s + kq][kbase + 2i + c]
    const int ycol = nbase + t.i, xcol = kbase + 2 * t.i;
    const bool yok = ycol < job.ldy, xok = xcol < job.ldx;            // ldx is a multiple of 4: the whole float2 is in range
    f32x4 acc[2] = {zero4(), zero4()};
    float bsum = 0.f;
    const int nsub = LP / 16;                                         // 16-token units per sequence
    const int per_layer = a.batch * nsub, units = per_layer * job.n_layers;
    const int mine = (units - t.wave + WV - 1) / WV;   // units w, w + 8, ... of this wave
    // groups of kDGroup units a wave can get (B * LP * layers <= kDirectMaxTokens): with one, the second buffer and the loop fold away
    constexpr int kMaxGroups = (kDirectMaxTokens / 16 + WV * kDGroup - 1) / (WV * kDGroup);
    constexpr int NBUF = kMaxGroups > 1 ? 2 : 1;
    float av[NBUF][kDGroup][4];
    float2 bv[NBUF][kDGroup][4];
    // every load is issued unconditionally (out-of-range columns / units read a valid stand-in address and are zeroed
    // afterwards): a load under a branch would hide the number of outstanding loads from the compiler and turn
    // every wait into "wait for all of them"
    const int ycol_c = yok ? ycol : 0, xcol_c = xok ? xcol : 0;
    // units w, w + 8, ... of this wave, walked incrementally as (layer, sequence, 16-token block): no divisions in
    // front of the loads; all of it is wave-uniform (scalar registers)
    const int wv = __builtin_amdgcn_readfirstlane(t.wave);
    int u_lyr = 0, u_sq = wv / nsub, u_sub = wv - u_sq * nsub, u_m = 0;
    while (u_sq >= a.batch) { u_sq -= a.batch; ++u_lyr; }
    const int inc_sq = WV / nsub, inc_sub = WV - inc_sq * nsub;
    const size_t ylane = (size_t)job.dy_off + (size_t)t.kq * job.ldy + ycol_c;       // float index into grd
    const float* xlane = xbase + (size_t)t.kq * job.ldx + xcol_c;
    auto group_load = [&](float (&a4)[kDGroup][4], float2 (&b4)[kDGroup][4]) {
#pragma unroll
        for (int q = 0; q < kDGroup; ++q) {
            const bool live = u_m < mine;                             // past the end: re-read unit (0, 0, 0), zeroed in group_mma
            const int lyr = live ? u_lyr : 0, sq = live ? u_sq : 0, sub = live ? u_sub : 0;
            const size_t yp = ylane + (size_t)sq * ystride + (size_t)lyr * job.dy_lstride + (size_t)(sub * 16) * job.ldy;
            const float* xp = xlane + (size_t)sq * xstride + (size_t)lyr * job.x_lstride + (size_t)(sub * 16) * job.ldx;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                a4[q][k] = direct_ld<SC1>(a.grd, grs, yp + (size_t)4 * k * job.ldy);
                b4[q][k] = *reinterpret_cast<const float2*>(xp + (size_t)4 * k * job.ldx);
            }
            ++u_m;
            u_sub += inc_sub; u_sq += inc_sq;
            if (u_sub >= nsub) { u_sub -= nsub; ++u_sq; }
            while (u_sq >= a.batch) { u_sq -= a.batch; ++u_lyr; }
        }
    };
    auto group_mma = [&](int g, const float (&a4)[kDGroup][4], const float2 (&b4)[kDGroup][4]) {
#pragma unroll
        for (int q = 0; q < kDGroup; ++q) {
            const float keep = (yok && g * kDGroup + q < mine) ? 1.f : 0.f;     // zero A: the product and the bias sum vanish
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float av_ = keep != 0.f ? a4[q][k] : 0.f;
                bsum += av_;
                acc[0] = mfma16(av_, b4[q][k].x, acc[0]);
                acc[1] = mfma16(av_, b4[q][k].y, acc[1]);
            }
        }
    };
    const int ngroups = (mine + kDGroup - 1) / kDGroup;
    if constexpr (kMaxGroups == 1) {
        group_load(av[0], bv[0]);
        DTQN_SCHED_FENCE();
        group_mma(0, av[0], bv[0]);
    } else {
        constexpr int B1 = NBUF - 1;
        if (ngroups > 0) group_load(av[0], bv[0]);
        DTQN_SCHED_FENCE();
        int g = 0;
        for (; g + 2 < ngroups; g += 2) {                                 // buffers alternate without dynamic indexing
            group_load(av[B1], bv[B1]);
            DTQN_SCHED_FENCE();
            group_mma(g, av[0], bv[0]);
            DTQN_SCHED_FENCE();
            group_load(av[0], bv[0]);
            DTQN_SCHED_FENCE();
            group_mma(g + 1, av[B1], bv[B1]);
            DTQN_SCHED_FENCE();
        }
        if (g + 1 < ngroups) {                                            // last pair (cfg 1: the only one -- all 16 units in flight at once)
            group_load(av[B1], bv[B1]);
            DTQN_SCHED_FENCE();
            group_mma(g, av[0], bv[0]);
            DTQN_SCHED_FENCE();
            group_mma(g + 1, av[B1], bv[B1]);
        } else if (g < ngroups) {
            group_mma(g, av[0], bv[0]);
        }
    }
    // bias: sum over the 4 token phases of this lane's dY column
    bsum += __shfl_xor(bsum, 16);
    bsum += __shfl_xor(bsum, 32);
    // cross-wave sum through LDS, fixed order: slab[n][k] (+ a bias row) per wave
    constexpr int SLD = kDTK + 4;
    float* slab = slabs + (size_t)t.wave * (kDTN + 1) * SLD;
#pragma unroll
    for (int r = 0; r < 4; ++r)
        *reinterpret_cast<float2*>(slab + (t.kq * 4 + r) * SLD + 2 * t.i) = make_float2(acc[0][r], acc[1][r]);
    if (t.kq == 0) slab[kDTN * SLD + t.i] = bsum;
    __syncthreads();
    if (tid < kDTN * (kDTK / 4)) {
        const int nl = tid / (kDTK / 4), k4 = (tid % (kDTK / 4)) * 4;
        const int n = nbase + nl, k = kbase + k4;
        if (n < job.N && k < job.K) {
            float4 v = ld4(slabs + nl * SLD + k4);
#pragma unroll
            for (int w = 1; w < WV; ++w) {
                const float4 x = ld4(slabs + (size_t)w * (kDTN + 1) * SLD + nl * SLD + k4);
                v.x += x.x; v.y += x.y; v.z += x.z; v.w += x.w;
            }
            float* op = a.grad + job.w_off + (size_t)n * job.K + k;
            if (k + 3 < job.K && (job.K & 3) == 0) {
                st4(op, v);
                ss = (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
            } else {
                const float vv[4] = {v.x, v.y, v.z, v.w};
                for (int c = 0; c < 4; ++c)
                    if (k + c < job.K) { op[c] = vv[c]; ss += vv[c] * vv[c]; }
            }
        }
    } else if (tid >= 256 && tid < 256 + kDTN && job.b_off >= 0 && bk == 0) {
        const int nl = tid - 256, n = nbase + nl;
        if (n < job.N) {
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < WV; ++w) v += slabs[(size_t)w * (kDTN + 1) * SLD + kDTN * SLD + nl];
            a.grad[job.b_off + n] = v;
            ss = v * v;
        }
    }
    return ss;
}

// Block `sb` of the per-sequence partials of the backward kernel (LayerNorm affine, embedding tables, learned positions = dL/dx0):
// PER elements per workgroup (PER / 64 per lane); wave p sums records p, p + 8, ... (independent loads in flight), then the
// eight partial sums are added in wave order.  `slabs`: (PER / 64) * WV * 64 floats.  Contains one __syncthreads().
template <bool SC1, int PER, int WV>
__device__ __forceinline__ float direct_small(const DtqnNet& net, const DirectCtx& a, int sb, float* slabs, const Thr& t) {
    constexpr int HN = PER / 64;
    static_assert(HN <= WV, "one finishing wave per 64 elements");
    const int D = net.d_model;
    const int n_ln = net.num_layers * 4 * D;
    const int n_tab = net.discrete ? net.vocab * net.embed_per_obs : 0;
    const int n_act = net.action_dim > 0 ? net.num_actions * net.action_dim : 0;
    const DtqnRsrc grs = DTQN_XCH_RSRC(a.grd, (size_t)a.batch * net.grd_stride * 4);
    const DtqnRsrc srs = DTQN_XCH_RSRC(a.small, (size_t)a.batch * a.row_split * net.sp_parts * net.sp_stride * 4);
    float ss = 0.f;
    int dst[HN];
#pragma unroll
    for (int h = 0; h < HN; ++h) {
        dst[h] = -1;
        const int e = sb * PER + h * 64 + t.lane;
        float v = 0.f;
        if (e < a.n_small) {
            int src;
            bool in_small = true;
            size_t stride = (size_t)net.sp_stride;
            if (e < n_ln) {
                const int l = e / (4 * D);
                dst[h] = net.off_layer0 + l * net.layer_stride + (e - l * 4 * D);
                src = net.so_ln + e;
            } else if (e < n_ln + n_tab) {
                dst[h] = net.off_obs_tab + (e - n_ln);
                src = net.so_tab + (e - n_ln);
            } else if (e < n_ln + n_tab + n_act) {
                dst[h] = net.off_act_emb + (e - n_ln - n_tab);
                src = net.so_act + (e - n_ln - n_tab);
            } else {
                dst[h] = net.off_pos + (e - n_ln - n_tab - n_act);
                src = net.go_dx0 + (e - n_ln - n_tab - n_act);
                in_small = false;
                stride = (size_t)net.grd_stride;
            }
            const int cnt = a.batch * (in_small ? a.row_split * net.sp_parts : 1);
#pragma unroll 8
            for (int q = t.wave; q < cnt; q += WV)
                v += in_small ? direct_ld<SC1>(a.small, srs, (size_t)q * stride + src) : direct_ld<SC1>(a.grd, grs, (size_t)q * stride + src);
        }
        slabs[(h * WV + t.wave) * 64 + t.lane] = v;
    }
    __syncthreads();
    if (t.wave < HN) {
        int d = -1;
#pragma unroll
        for (int h = 0; h < HN; ++h) d = t.wave == h ? dst[h] : d;
        if (d >= 0) {
            float tot = 0.f;
#pragma unroll
            for (int w = 0; w < WV; ++w) tot += slabs[(t.wave * WV + w) * 64 + t.lane];
            a.grad[d] = tot;
            ss = tot * tot;
        }
    }
    return ss;
}

// sum of squares of everything this workgroup wrote (fixed order: lanes, then waves) -> norm_partial[slot].  `red`: WV floats of LDS.
template <int WV>
__device__ __forceinline__ void direct_norm_partial(float ss, float* red, float* norm_partial, int slot, const Thr& t) {
    ss = wave_sum(ss);
    __syncthreads();
    if (t.lane == 0) red[t.wave] = ss;
    __syncthreads();
    if (t.tid == 0) {
        float tot = 0.f;
        for (int w = 0; w < WV; ++w) tot += red[w];
        norm_partial[slot] = tot;
    }
}

// per job ceil(N / 16) * ceil(K / 32) direct tiles
inline int direct_tiles_of(const DtqnWJob& j) { return ((j.N + kDTN - 1) / kDTN) * ((j.K + kDTK - 1) / kDTK); }
inline int small_elems(const DtqnNet* net) {
    return net->num_layers * 4 * net->d_model + (net->discrete ? net->vocab * net->embed_per_obs : 0) +
           (net->action_dim > 0 ? net->num_actions * net->action_dim : 0) +
           (net->pos == DTQN_POS_LEARNED ? net->ctx_len * net->d_model : 0);
}

// ---- the weight-gradient workgroups of the fused backward launch ---------------------------------------------------------
// Events: e < num_layers = "every workgroup of the data-gradient chain has published the gradient records of layer NL - 1 - e
// (and, e = 0, of the Q head)"; e = num_layers = everything (layer 0, dL/dx0, the per-sequence partials).  counters[e] counts the
// workgroups that have reached the event; counters[kFuseDeparted] the weight-gradient workgroups that are done waiting: the
// last one zeroes the lot, so the next launch starts clean without a memset in front of it.
constexpr int kFuseMaxJobs = 16;
constexpr int kFuseMaxEvents = 12;
constexpr int kFuseDeparted = kFuseMaxEvents;
constexpr int kFuseWords = 16;               // counters appended to DtqnTd.xflags (dtqn_td_xch_flags)
struct FuseArgs {
    int n_role;                              // weight-gradient workgroups behind the batch * RS chain workgroups (0: not fused)
    int n_chain;                             // chain workgroups = arrivals per event
    int n_jobs, n_tiles, n_items, n_small, n_parts;
    DtqnWJob jobs[kFuseMaxJobs];             // in order of readiness: head, last layer ... first layer, embedding
    int dtile0[kFuseMaxJobs];
    int job_event[kFuseMaxJobs];
    float* grad;
    float* norm_partial;
    int32_t* step_counter;
    int32_t* counters;                       // kFuseWords ints, zero between launches
};

__device__ __forceinline__ void fuse_wait(int32_t* counter, int target, const Thr& t) {
    if (t.tid == 0)
        while (DTQN_AGENT_LOAD(counter) < target) DTQN_SPIN_PAUSE_LONG();
    __syncthreads();
}

template <int NW>
__device__ __forceinline__ void fuse_role(const DtqnNet& net, const FuseArgs& f, const DirectCtx& ctx, int w, float* smem, const Thr& t) {
    static_assert(NW == kFuseWaves, "the weight-gradient workgroups of the backward launch have its eight waves");
    float* red = smem;
    float* slabs = smem + NW;
    float ss = 0.f;
    if (w == 0 && t.tid == 0) f.step_counter[0] = f.step_counter[1];        // publish the step count of the previous update
    if (w == 0)                                                            // partials nobody writes this time
        for (int i = f.n_role + t.tid; i < f.n_parts; i += NW * 64) f.norm_partial[i] = 0.f;
    // item p of the readiness-ordered list goes to workgroup perm^-1(p % n_role) in round p / n_role; perm keeps the 16 consecutive
    // items of a round on one XCD (workgroup b runs on XCD b % 8; n_role is a multiple of 8): they share their X / dY columns
    const int per_xcd = f.n_role / 8, slot = (w % 8) * per_xcd + w / 8;
    int waited = -1;
#ifndef DTQN_FUSE_DEBUG_PLAINLD
#define DTQN_FUSE_DEBUG_PLAINLD 0      /* timing experiments only: 1 = the role reads with plain loads (may read stale records) */
#endif
#ifdef DTQN_FUSE_DEBUG_NOROLE          /* timing experiments only: the role does no work (grad stays unwritten) */
    for (int p = f.n_items; p < f.n_items; p += f.n_role) {
#else
    for (int p = slot; p < f.n_items; p += f.n_role) {
#endif
        __syncthreads();                                                   // LDS of the previous item consumed
        if (p < f.n_tiles) {
            int j = 0;
            for (int k = 1; k < f.n_jobs; ++k)
                if (f.dtile0[k] <= p) j = k;
            const int ev = f.job_event[j];
            if (ev > waited) { fuse_wait(f.counters + ev, f.n_chain, t); waited = ev; }
            ss += direct_tile<!DTQN_FUSE_DEBUG_PLAINLD, NW>(net, ctx, f.jobs[j], p - f.dtile0[j], slabs, t);
        } else {
            if (net.num_layers > waited) { fuse_wait(f.counters + net.num_layers, f.n_chain, t); waited = net.num_layers; }
            ss += direct_small<!DTQN_FUSE_DEBUG_PLAINLD, kDSmallFused, NW>(net, ctx, p - f.n_tiles, slabs, t);
        }
    }
    direct_norm_partial<NW>(ss, red, f.norm_partial, w, t);
    // depart: every weight-gradient workgroup sees the last event before it leaves, so nobody reads a counter after the reset
    if (net.num_layers > waited) fuse_wait(f.counters + net.num_layers, f.n_chain, t);
    if (t.tid == 0) {
        const int before = DTQN_AGENT_ADD(f.counters + kFuseDeparted, (int32_t)1);
        if (before == f.n_role - 1)
            for (int e = 0; e <= kFuseDeparted; ++e) DTQN_AGENT_STORE(f.counters + e, (int32_t)0);
    }
}

// chain side: one arrival at event `e` per workgroup, behind a point where all of its write-through stores are acknowledged
__device__ __forceinline__ void fuse_arrive(int32_t* counters, int e, const Thr& t) {
    if (t.tid == 0) DTQN_AGENT_ADD(counters + e, (int32_t)1);
}

// ---- host: which launches fuse, and the readiness-ordered item list --------------------------------------------------------
inline int fuse_cu_count() { return device_cu_count(); }
// Fused when: four row slices per sequence (latency mode of the backward), residual gates, the plain context network, the whole batch
// short enough for the one-launch weight gradients, and at least eight compute units left over behind the chain's workgroups (every
// workgroup of the launch has a CU of its own: the weight-gradient workgroups spin on the chain's events).  OPT-IN with DTQN_WGRAD_FUSED=1
// (round 3: correct on the emulation and on the GPU, but 127 us per update against 108 with the separate launch -- the write-through
// stores and the event counters cost the chain 9 us and a tile still takes 10 us behind the last event);
// DTQN_FUSE_ROLE_WGS=<n> overrides the number of weight-gradient workgroups.
inline bool fuse_plan(const DtqnNet* net, const DtqnTd* td, FuseArgs* f) {
    f->n_role = 0;
    if (!net || !td || td->row_split != 4 || net->tiled || net->gate == DTQN_GATE_GRU || net->identity || net->bag_size > 0 || net->img_c > 0 ||
        dtqn_ws_lite(net->tiled, net->d_model, net->head_dim, net->d_real))
        return false;
    if (!td->grad || !td->norm_partial || !td->step_counter || !td->small || !td->xflags || !td->act || !td->grd) return false;
    if (!dtqn_td_wgrad_is_direct(net, td->batch) || waves_for(*net) != kFuseWaves) return false;
    const int NL = net->num_layers;
    if (net->n_wjobs != 1 + 4 * NL + 2 || net->n_wjobs > kFuseMaxJobs || NL + 1 > kFuseMaxEvents) return false;
    const char* e = getenv("DTQN_WGRAD_FUSED");
    if (e == nullptr || atoi(e) == 0) return false;          // opt-in: measured slower than the separate launch
    DtqnWJob jobs[kFuseMaxJobs];
    if (dtqn_net_wjobs(net, jobs) != DTQN_OK) return false;
    // no 128-byte line may hold records of two different events
    // (records of one event are contiguous: head | dx0 | layer 0 | layer 1 ...; every tensor is a multiple of lp >= 32 floats long)
    if (net->grd_stride % 32 != 0 || net->go_dq % 32 != 0 || net->go_dhh % 32 != 0 || net->go_dx0 % 32 != 0 ||
        net->go_layer0 % 32 != 0 || net->grd_layer_stride % 32 != 0 || net->lp % 32 != 0)
        return false;
    for (int j = 0; j < net->n_wjobs; ++j)
        if (!jobs[j].x_in_act) return false;
    const int n_chain = td->batch * 4;
    int n_role = ((fuse_cu_count() - n_chain) / 8) * 8;
    const char* r = getenv("DTQN_FUSE_ROLE_WGS");
    if (r != nullptr) n_role = (atoi(r) / 8) * 8;
    const int n_parts = dtqn_td_norm_partials(net);
    if (n_role > n_parts) n_role = (n_parts / 8) * 8;
    if (n_role < 8) return false;
    // wjobs come as [embedding, layer 0 (in, out, ffn.1, ffn.2), ..., layer NL - 1, head.1, head.2]: the reverse is the order in which
    // the chain completes their dY records
    int tiles = 0;
    for (int k = 0; k < net->n_wjobs; ++k) {
        const int j = net->n_wjobs - 1 - k;
        f->jobs[k] = jobs[j];
        f->dtile0[k] = tiles;
        tiles += direct_tiles_of(jobs[j]);
        const int layer = j == 0 ? 0 : j > 4 * NL ? NL - 1 : (j - 1) / 4;      // head: with the last layer
        f->job_event[k] = (j == 0 || layer == 0) ? NL : NL - 1 - layer;
    }
    f->n_jobs = net->n_wjobs;
    f->n_tiles = tiles;
    f->n_small = small_elems(net);
    f->n_items = tiles + (f->n_small + kDSmallFused - 1) / kDSmallFused;
    f->n_parts = n_parts;
    f->n_chain = n_chain;
    f->grad = td->grad;
    f->norm_partial = td->norm_partial;
    f->step_counter = td->step_counter;
    f->counters = td->xflags + dtqn_td_xch_flags(net, td->batch) - kFuseWords;
    f->n_role = n_role;
    return true;
}

}  // namespace dtqn
