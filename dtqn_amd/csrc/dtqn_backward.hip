// Double-DQN loss + hand-written data-gradient chain of the DTQN policy network (gfx950).
// One workgroup per sampled sequence; dL/dx of the residual stream lives in LDS for the whole pass.
//
// Replaces (dtqn/agents/dtqn.py): q.gather / argmax / target gather / Bellman / mse_loss (:219-243),
// the seven logged statistics (:245-253) and the autograd backward of forward #1 (:256) -- restricted
// to the DATA gradients.  Every pre-activation gradient that feeds a weight matrix is written to the
// per-sequence `grd` record; dtqn_wgrad.hip contracts those against the saved activations over all
// B*L tokens, so no per-sequence weight-gradient partials exist.
#include "dtqn_device.hpp"
#include "dtqn_bwd_device.hpp"
#include "dtqn_gru.hpp"
#include "dtqn_wgrad_direct.hpp"
#include "dtqn_forward_body.hpp"      // AHEAD: the next update's target pass rides in this launch (forward_body_wl)

#ifndef DTQN_SPLIT_ATTN_MFMA
#define DTQN_SPLIT_ATTN_MFMA 1
#endif
// Item guard of the register-accumulated stages.  -DDTQN_BWD_FOLD_GUARD=1 folds it for full rounds of items (as the
// forward does): see DESIGN.md section 6 for what that exposed in the <D = 128, 16-row slice> instantiation.
// Row slices, attention backward: 1 = every slice broadcasts its dO tile (+ delta) to the slices below it BEFORE the attention passes
// and owns dK | dV of its own rows (attention_backward_group_mfma, "own keys" form); 0 = the round-1 / round-2 scheme, every slice
// hands its queries' share of dK | dV down AFTER the passes.
#ifndef DTQN_BWD_DOX
#define DTQN_BWD_DOX 1
#endif
// 1: the sender waits for its write-through stores and raises its flags BEFORE its own pass 1 (receivers never wait for a sender's
// attention pass); 0: behind pass 1 (the acknowledgements are hidden, the receivers wait longer)
#ifndef DTQN_BWD_DOX_EARLY_FLAG
#define DTQN_BWD_DOX_EARLY_FLAG 1
#endif
#ifndef DTQN_BWD_FOLD_GUARD
#define DTQN_BWD_FOLD_GUARD 0
#endif
#if DTQN_BWD_FOLD_GUARD
#define DTQN_BWD_GUARD(w, q) Own::valid_fast(w, q)
#else
#define DTQN_BWD_GUARD(w, q) Own::valid(w, q)
#endif

namespace dtqn {

struct BwdArgs {
    DtqnNet net;
    const float* theta;          // policy parameters
    const float* act;            // [B][act_stride] saved by the training forward
    float* grd;                  // [B][grd_stride]
    float* small;                // [B][sp_stride]
    const float* q3;             // [3][B][LP][AP]
    float* stats_partial;        // [B][8]
    const float* obs;            // replay arrays (actions / rewards / dones of the sampled window)
    const uint8_t* actions;
    const float* rewards;
    const uint8_t* dones;
    long long obs_ep_stride, act_ep_stride, rew_ep_stride;
    const int32_t* ep_idx;
    const int32_t* start;
    int batch, history;
    float gamma;
    long long* prof;             // debug stage clock
    float* xch;                  // row-split hand-over buffer / flags (RS > 1 only)
    int32_t* xflags;
    // dropout of the training forward, recomputed here: step = step_counter[1]
    uint32_t drop_thresh, drop_seed;
    float drop_scale;
    const int32_t* step_counter;
    FuseArgs fuse;               // weight-gradient workgroups in this launch (n_role == 0: none)
};

// RS = row slices per sequence (see dtqn_forward.hip).  RS == 2: the workgroup owns rows [R0, R0 + LP); attention is
// the only stage that looks below R0 (keys / values of the lower rows, read from the forward's record) and the only one
// that hands something over: the upper slice's contribution to dK | dV of the lower rows (slice 1 -> slice 0).
// DROP: the training forward ran with dropout (compile-time: the default build of a network carries no keep-mask code)
// FUSE: the launch carries a.fuse.n_role extra workgroups behind the batch * RS of the data-gradient chain; they compute the weight
// gradients (dtqn_wgrad_direct.hpp), layer by layer as the chain publishes its gradient records: every record store of the chain is
// then a write-through (sc1) store, and the chain counts itself in at an event once a layer's stores are acknowledged.
// AHEAD: the launch carries 4 * batch more workgroups behind the batch * RS of the data-gradient chain: the TARGET pass of the NEXT
// update (forward_body_wl, four 16-row slices per sequence; dtqn_td_backward_ahead).  It depends on nothing this update computes,
// the chain leaves half the compute units idle at batch 32, and riding in the same launch it needs no second stream, no event and no
// launch of its own: its Q rows are simply there when the next update's loss stage wants them.
template <bool AHEAD> struct AheadArgs {};
template <> struct AheadArgs<true> { FwdArgs f; };

// PAD: width-padded network (DtqnNet.d_real > 0; four-slice residual-gate chain only): LayerNorm backward over the d_real real columns,
// softmax scale of the real head width.
template <int D, int MT, int HD, int NW, bool GRU, int RS, bool DROP, bool FUSE = false, bool AHEAD = false, bool PAD = false>
__global__ __launch_bounds__(NW * 64) void dtqn_backward_kernel(BwdArgs a, AheadArgs<AHEAD> ahead) {
    static_assert(RS == 1 || RS == 2 || RS == 4, "one, two or four row slices");
    static_assert(!PAD || (RS == 4 && MT == 1 && !GRU && !DROP && !FUSE && D == 64), "width padding rides with the four-slice residual-gate chain");
    if constexpr (AHEAD) {
        static_assert(RS == 4 && MT == 1 && !GRU && !DROP && !FUSE && D <= 64, "the target pass ahead rides with the four-slice residual-gate chain");
        if ((int)blockIdx.x >= a.batch * RS) {
            forward_body_wl<D, 1, HD, NW, 4, false, PAD>(ahead.f);
            return;
        }
    }
    constexpr int NT = NW * 64;
    constexpr int LP = MT * 16;                   // rows this workgroup owns
    constexpr int LPF = LP * RS;                  // padded rows of the whole sequence (= net.lp)
    constexpr int LDX = D + 4;
    constexpr int GW = D >= 64 ? 64 : D;          // attention head-group width (columns)
    constexpr int NG = D / GW;
    constexpr int NC = D >= 128 ? D / 2 : 2 * D;  // FFN hidden columns per pass (D = 128: 2D-wide fragments would be 128 VGPRs a pair)
    constexpr bool OST = D <= 64;                 // stage the attention output o in LDS too (sixth W5 tile) when it fits
    constexpr int W5C = ((OST ? 6 : 5) * GW > NC ? (OST ? 6 : 5) * GW : NC);
    constexpr int LD5 = W5C + 4;
    constexpr int MGX = pick_mg(D / 16, MT, NW);
    using Own = Owned<D, MT, MGX, NW>;          // fixed ownership of a [LP][D] register-accumulated output
    const DtqnNet& net = a.net;
    const Thr t = make_thr();
    const Thr& t_outer = t;
    if constexpr (FUSE) {
        if ((int)blockIdx.x >= a.batch * RS) {
            const DirectCtx ctx{a.act, a.grd, a.small, a.fuse.grad, a.batch, RS, a.fuse.n_small};
            fuse_role<NW>(net, a.fuse, ctx, (int)blockIdx.x - a.batch * RS, reinterpret_cast<float*>(dtqn_smem), t);
            return;
        }
    }
    int b, pos;                                    // the upper slice (the producer of this kernel) first; a sequence's slices on one XCD
    slice_block_map((int)blockIdx.x, a.batch, RS, b, pos);
    const int slice = RS - 1 - pos;
    const int R0 = slice * LP;
    const int Lfull = net.ctx_len, A = net.num_actions, AP = net.ap, H = net.num_heads, adim = net.action_dim;
    const int L = Lfull - R0 < LP ? Lfull - R0 : LP;   // live rows of this slice (may be <= 0)
    const bool ident = RS > 1 ? false : net.identity != 0;     // row slices are dispatched for post-LN nets only: folds away
    const int dreal = PAD ? net.d_real : D;                    // LayerNorm width
    constexpr bool gru = GRU;                      // gate type is a template parameter: the ResGate build carries no GRU code
    const float* __restrict__ theta = a.theta;
    const float* rec = a.act + (size_t)b * net.act_stride;
    float* grec = a.grd + (size_t)b * net.grd_stride;
    float* srec = a.small + ((size_t)b * RS + slice) * net.sp_stride;
    // a [LPF][w] record tensor at the first row of this slice; ReLU ballot records at its first row tile
    auto rf = [&](const float* base, int off, int w) -> const float* { return base + off + (size_t)R0 * w; };
    auto gf = [&](float* base, int off, int w) -> float* { return base + off + (size_t)R0 * w; };
    auto mf = [&](const float* base, int off, int ctiles) -> const float* { return base + off + (size_t)(R0 / 16) * ctiles * 8; };
    // record stores: plain, or write-through through a descriptor of this sequence's record (FUSE)
    const DtqnRsrc grs = DTQN_XCH_RSRC(grec, (size_t)net.grd_stride * 4);
    // gradient records as write-through (sc1) stores.  Not for the GRU-gated chain without dropout on 32- / 64-row tiles of eight waves:
    // with the descriptor stores the register allocator ends at 2.2 - 4.6 KB of scratch per lane there (round 6 bisect over DTQN_OPT:
    // bit 32 alone; 148 - 600 B with plain stores, profiles/r06_kernel_resources.md), and GRU nets do not ride the pipelined update
    constexpr bool WT = FUSE || (kOptBwdWT && !(GRU && !DROP && MT >= 2 && NW == 8));
    auto g_store4 = [&](float* p, float4 v) {
        if constexpr (WT) dtqn_xch_store4(grs, (int)(p - grec) * 4, v);
        else st4(p, v);
    };
    auto g_tile_store = [&](const float* s_, int ld_, float* g_, int rows, int cols, int gld = 0) {
        if constexpr (WT) {
            const int c4 = cols >> 2;
            if (gld == 0) gld = cols;
            const int base = (int)(g_ - grec);
            for (int idx = t_outer.tid; idx < rows * c4; idx += NT) {
                const int r = idx / c4, c = (idx - r * c4) * 4;
                dtqn_xch_store4(grs, (base + r * gld + c) * 4, ld4(s_ + r * ld_ + c));
            }
        } else {
            tile_store<NW>(s_, ld_, g_, rows, cols, t_outer, gld);
        }
    };
    auto s_store = [&](float* p, float v) {      // per-sequence partials (`small` record)
        if constexpr (FUSE) DTQN_AGENT_STORE(p, v);
        else *p = v;
    };

    float* DX = reinterpret_cast<float*>(dtqn_smem);   // dL/d(residual stream)        [LP][LDX]
    float* T2 = DX + LP * LDX;                         // narrow temp                   [LP][LDX]
    float* W5 = T2 + LP * LDX;                         // wide temp, attention tiles on GLOBAL rows [LPF][LD5]
    float* dq_s = W5 + LPF * LD5;                      // dL/dQ                         [LP][AP]
    float* delta_s = dq_s + LP * AP;                   // attention row terms           [GW/HD][LPF] (global rows)
    float* lse_s = delta_s + (GW / HD) * LPF;
    float* red = lse_s + (GW / HD) * LPF;              // LN column-sum scratch         [PARTS][2][D]
    constexpr int PARTS = NT / D >= 1 ? NT / D : 1;
    float* st_s = red + PARTS * 2 * D;                 // LayerNorm (mean, rstd) of this layer [2][LP][2]
    float* DU = st_s + 4 * LP;                         // identity only: branch grad    [LP][LDX]

    const int ep = a.ep_idx[b], st0 = a.start[b] + R0;
    Drop dr = drop_off();
    if constexpr (DROP) dr = Drop{a.drop_thresh, a.drop_scale, a.drop_seed, (uint32_t)a.step_counter[1], (uint32_t)b};
    int ps = 0;
    DTQN_PROF(a.prof, ps++);
    // everything the head stage needs goes in flight before the (latency-bound, one-wave) loss stage
    // delta = dO . o per (row, head) comes out of the dO product's epilogue: a head inside one 16-column tile is summed across its lanes;
    // head width 32 spans two tiles (two different waves): each adds its half into the zero-filled entry -- two addends, so the sum
    // does not depend on who comes first
    static_assert(HD <= 16 || (HD == 32 && RS == 4 && MT == 1), "delta-in-epilogue: a head inside one 16-column tile, or two tiles wide on the four-slice chain");
    // (16-row slices: D / 16 column tiles on twice as many waves -> two waves per tile, half the contraction each)
    constexpr bool SPLITS = kOptSplitS && MT == 1 && 2 * (D / 16) == NW && 2 * (GW / 16) == NW;
    std::conditional_t<SPLITS, StageDyWSplit<D, NW, D / 16>, StageDyW<D, MT, pick_mg(D / 16, MT, NW), NW, D / 16>> g_h1;
    g_h1.prefetch(theta + net.off_head1_w, D, t);
    TileRegs<NW, LP, D> tr;                                // saved-activation tile in flight
    TileRegs<NW, LP, D> trh;
    trh.load(rf(rec, net.ao_hh, D), D, t);
    if (!ident) tr.load(rf(rec, net.ao_layer0 + (net.num_layers - 1) * net.act_layer_stride + net.al_s2, D), D, t);

    // (mean, rstd) of a layer's two LayerNorms, one float per thread, fetched a whole layer ahead of their use -- the last layer's
    // here, in front of the loss stage (round 3: it was fetched right in front of the layer loop; 107.2 -> 106.4 us per update)
    static_assert(4 * LP <= NT, "one statistics value per thread");
    auto st_fetch = [&](int l) -> float {
        const float* lr = rec + net.ao_layer0 + (size_t)l * net.act_layer_stride;
        if (t.tid >= 4 * LP) return 0.f;
        return t.tid < 2 * LP ? rf(lr, net.al_st1, 2)[t.tid] : rf(lr, net.al_st2, 2)[t.tid - 2 * LP];
    };
    float st_next = st_fetch(net.num_layers - 1);

    // ---------------- B0: double-DQN target, loss, dL/dQ, statistics (dtqn.py:219-253) ----------------
    {
        const float* q0 = a.q3 + (((size_t)0 * a.batch + b) * LPF + R0) * AP;
        const float* q1 = a.q3 + (((size_t)1 * a.batch + b) * LPF + R0) * AP;
        const float* q2 = a.q3 + (((size_t)2 * a.batch + b) * LPF + R0) * AP;
        const float inv_count = 1.0f / ((float)a.batch * (float)a.history);
        DTQN_PROF(a.prof, 24);   // (finer marks of the stages that carry a first-trip penalty: slots 24..30, printed raw by stage_profile.py)
        for (int idx = t.tid; idx < LP * AP; idx += NT) dq_s[idx] = 0.f;
        __syncthreads();
        DTQN_PROF(a.prof, 25);
        if (t.wave == 0)
            td_loss_wave(q0, q1, q2, AP, A, Lfull - R0, LP, a.history, a.gamma, inv_count,     // window test in slice-local rows
                         a.actions + (size_t)ep * a.act_ep_stride + st0, a.rewards + (size_t)ep * a.rew_ep_stride + st0,
                         a.dones + (size_t)ep * a.rew_ep_stride + st0, dq_s, a.stats_partial + ((size_t)b * RS + slice) * 8, t.lane);
        DTQN_PROF(a.prof, 26);
        __syncthreads();
        DTQN_PROF(a.prof, 27);
        if constexpr (WT) {
            for (int idx = t.tid; idx < LP * AP / 4; idx += NT) g_store4(gf(grec, net.go_dq, AP) + 4 * idx, ld4(dq_s + 4 * idx));   // AP = up4(A)
        } else {
            for (int idx = t.tid; idx < LP * AP; idx += NT) gf(grec, net.go_dq, AP)[idx] = dq_s[idx];
        }
    }

    DTQN_PROF(a.prof, ps++);   // loss done
    // ---------------- B1: Q head backward ----------------
    // Memory discipline of this kernel: every global LOAD (weight fragments, saved activations) is
    // issued one stage before its use; every global STORE of a gradient tensor is a coalesced copy of
    // the LDS tile, issued at the start of the NEXT stage after the prefetched fragment has been
    // retired, so that no s_waitcnt ever sits behind a freshly issued store.
    {
        const float* __restrict__ W2 = theta + net.off_head2_w;
        constexpr int C4 = D / 4;
#pragma unroll
        for (int k4 = 0; k4 < TileRegs<NW, LP, D>::N; ++k4) {
            const int idx4 = t.tid + k4 * NT;
            if (idx4 < LP * C4) {
                const int r = idx4 / C4, c0 = (idx4 - r * C4) * 4;
                const float hv[4] = {trh.v[k4].x, trh.v[k4].y, trh.v[k4].z, trh.v[k4].w};
                float g4[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float g = 0.f;
                    if (hv[e] > 0.f)
                        for (int c = 0; c < A; ++c) g = fmaf(dq_s[r * AP + c], W2[c * D + c0 + e], g);
                    g4[e] = g;
                }
                st4(T2 + r * LDX + c0, make_float4(g4[0], g4[1], g4[2], g4[3]));
            }
        }
    }
    __syncthreads();
    g_h1.retire();
    g_tile_store(T2, LDX, gf(grec, net.go_dhh, D), LP, D);
    if constexpr (SPLITS) g_h1.run(T2, LDX, t, red, [&](int r, int c, float v) { DX[r * LDX + c] = v; });      // (red: free until the first LayerNorm)
    else g_h1.run(T2, LDX, t, [&](int r, int c, float v) { DX[r * LDX + c] = v; });
    __syncthreads();
    DTQN_PROF(a.prof, ps++);   // head done

    // ---------------- layers, last to first ----------------
    for (int l = net.num_layers - 1; l >= 0; --l) {
        // D >= 128: the thread coordinates of this iteration are made opaque to the optimiser, at the top of the layer and of its two
        // big stages.  Otherwise it hoists dozens of per-thread address computations out of the layer loop and keeps them live across
        // the whole body; at D = 128 that ended in 316 bytes of scratch per lane (now 0; backward 748 -> 692 us at BASELINE config 3
        // shapes).  At D <= 64 the hoisted addresses are the faster code even where they spill a little (measured round 3: the 16-row
        // slices of latency mode lose 2 us with the same trick, the full 64-row tile is unchanged), so those keep them.
        Thr t = t_outer;
        // (round 6: the one-workgroup-per-sequence backward of d_model 64 -- BASELINE config 2 -- and the GRU-gated two-slice one re-derive the
        //  coordinates from ONE opaque register instead: 148 - 624 B of scratch per lane -> 0.  Measured, us per update old | new
        //  (tests/perf/time_ws_shapes.py): residual gate batch 192 297 | 287, 256 323 | 312 (config 2: 3 110 -> 3 228 TD-updates/s); GRU batch 96
        //  402 | 334, 128 424 | 353, 192 642 | 580, 256 728 | 665.  The residual-gate two-slice kernels lose 0.6 % with it (188 | 189, 196 | 197)
        //  and keep the hoisted addresses; the same in the weights-through-LDS forward also reaches 0 B and is slower (3 228 -> 3 131): not kept)
#define DTQN_RELAUNDER() do { if constexpr (D >= 128 && MT >= 4) { DTQN_ASM_KEEP(t.tid); DTQN_ASM_KEEP(t.lane); DTQN_ASM_KEEP(t.wave); DTQN_ASM_KEEP(t.i); DTQN_ASM_KEEP(t.kq); } \
        else if constexpr (D == 64 && MT >= 2 && NW <= 8 && (RS == 1 || (GRU && RS == 2))) { int tid_ = (int)threadIdx.x; DTQN_ASM_KEEP(tid_); t.tid = tid_; t.lane = tid_ & 63; t.wave = tid_ >> 6; t.i = t.lane & 15; t.kq = t.lane >> 4; } } while (0)
        DTQN_RELAUNDER();
        const float* __restrict__ th = layer_theta(net, theta, l);
        const float* lrec = rec + net.ao_layer0 + (size_t)l * net.act_layer_stride;
        float* lgrd = grec + net.go_layer0 + (size_t)l * net.grd_layer_stride;
        float* lsm = srec + net.so_ln + l * 4 * D;
        const float* __restrict__ W1 = th + net.lo_f1_w;
        const float* __restrict__ W2 = th + net.lo_f2_w;
        StageDyW<D, MT, pick_mg(NC / 16, MT, NW), NW, NC / 16> g_dh;      // dh = df W2[:, chunk]
        g_dh.prefetch(W2, 4 * D, t);
        // (mean, rstd) of both LayerNorms of this layer -> LDS (published by the next barrier); the previous layer's
        // reads of st_s ended behind a barrier, and the next layer's values go in flight now
        if (t.tid < 4 * LP) st_s[t.tid] = st_next;
        if (l > 0) st_next = st_fetch(l - 1);

        if (!ident) {   // x_out = LN2(s2): dL/ds2   (s2 was put in flight one stage ago)
            if (l == net.num_layers - 1) DTQN_PROF(a.prof, 28);
            tr.to_lds(T2, LDX, t);
            if (l == net.num_layers - 1) DTQN_PROF(a.prof, 29);
            __syncthreads();
            if (l == net.num_layers - 1) DTQN_PROF(a.prof, 30);
            layernorm_backward<D, NW, FUSE, PAD>(DX, T2, DX, false, LDX, LP, st_s + 2 * LP, th + net.lo_ln2_w, lsm + 2 * D, red, t, false, dreal);
            if (l == net.num_layers - 1) DTQN_PROF(a.prof, 31);
            __syncthreads();
        }
        DTQN_PROF(a.prof, ps++);   // LN2 bwd done
        tr.load(rf(lrec, net.al_s1, D), D, t);         // needed after the FFN: in flight during it
        // mlp gate.  res: s2 = x1 + relu(f)  ->  df = ds2 * [y2 > 0], the skip path keeps DX.
        // gru: DX <- dL/dx (skip path), T2 <- dL/dy, then the same ReLU mask.
        if (gru) {
            gru_gate_backward<D, MT, NW>(DX, T2, LDX, W5, LD5, theta + net.off_gate_mlp, net, lrec + net.al_gate2, lgrd + net.gl_gate2, t, LPF, R0);
            __syncthreads();
        }
        {
            const float* m2 = mf(lrec, net.al_m2, D / 16);
            for (int idx = t.tid; idx < LP * D; idx += NT) {
                const int r = idx / D, c = idx - r * D;
                const float dy = gru ? T2[r * LDX + c] : DX[r * LDX + c];
                // y2 = relu(dropout(f)): the ReLU pattern, then the keep mask of the FFN output
                T2[r * LDX + c] = mask_bit(m2, D / 16, r, c) ? drop_apply(dr, DROP_FFN, l, (uint32_t)((R0 + r) * D + c), dy) : 0.f;
            }
        }
        __syncthreads();
        // FFN backward: dh = df W2 (masked by h > 0), du2 = dh W1, in hidden-column passes
        DTQN_RELAUNDER();
        {
            f32x4 xacc[Own::PER_WAVE][MGX];
#pragma unroll
            for (int q = 0; q < Own::PER_WAVE; ++q)
#pragma unroll
                for (int m = 0; m < MGX; ++m) xacc[q][m] = zero4();
            float w1f[2][NC / 4];
            // 16-row slices (four column tiles of du2 on eight waves): wave w takes tile w % 4 and half w / 4 of the chunk's hidden
            // columns; the halves are summed through W5 behind the loop
            constexpr bool SPLITK = kOptSplitK && Own::ITEMS * 2 == NW && Own::PER_WAVE == 1 && MGX == 1 && NC % 32 == 0;
            constexpr bool HALFW = kOptHalfW && !SPLITK && NC % 32 == 0;      // half fragments in time: [2][NC / 8] registers instead of [2][NC / 4]
            const int sk_it = t.wave % Own::ITEMS, sk_kh = t.wave / Own::ITEMS;
            const unsigned long long* mh = reinterpret_cast<const unsigned long long*>(mf(lrec, net.al_mh, 4 * D / 16));
            constexpr int MGH = pick_mg(NC / 16, MT, NW);
#pragma clang loop unroll_count(D >= 128 ? 1 : 4)
            for (int c0 = 0; c0 < 4 * D; c0 += NC) {   // unrolled (D <= 64): exact s_waitcnt counts across the chunk boundary; at D = 128
                                                       // the four unrolled chunks let the scheduler hoist loads until 630 bytes per lane spilled
                g_dh.retire();
                if (c0 == 0) g_tile_store(T2, LDX, gf(lgrd, net.gl_df, D), LP, D);
                unsigned long long mw[MGH][4];         // ReLU ballots of the item's accumulator registers
                g_dh.run(T2, LDX, t,
                         [&](int kt, int mg) {
#pragma unroll
                             for (int m = 0; m < MGH; ++m)
#pragma unroll
                                 for (int r = 0; r < 4; ++r)
                                     mw[m][r] = mh[((mg * MGH + m) * (4 * D / 16) + (c0 >> 4) + kt) * 4 + r];
                         },
                         [&](int r, int c, float v) {
                             const unsigned long long w = mw[(r >> 4) % MGH][r & 3];
                             W5[r * LD5 + c] = ((w >> t.lane) & 1ull) ? v : 0.f;
                         });
                if constexpr (SPLITK) {
                    float (&wh)[NC / 8] = *reinterpret_cast<float (*)[NC / 8]>(&w1f[0][0]);
                    frag_dyw_fetch<NC / 2>(wh, W1 + (size_t)(c0 + sk_kh * (NC / 2)) * D + sk_it * 16 + t.i, D, t);
                } else if constexpr (HALFW) {
                    float (&wh)[NC / 8] = *reinterpret_cast<float (*)[NC / 8]>(&w1f[0][0]);
                    if (DTQN_BWD_GUARD(t.wave, 0)) frag_dyw_fetch<NC / 2>(wh, W1 + (size_t)c0 * D + Own::nt(t.wave, 0) * 16 + t.i, D, t);
                } else if (DTQN_BWD_GUARD(t.wave, 0))
                    frag_dyw_fetch<NC>(w1f[0], W1 + (size_t)c0 * D + Own::nt(t.wave, 0) * 16 + t.i, D, t);
                if constexpr (FUSE) {
                    // the layer above is complete in memory once every wave has drained its vmcnt here: its last record stores (dq | dk | dv)
                    // were issued before this layer's first fragment fetch.  Waiting for w1f[0] in front of the barrier instead of behind
                    // it costs nothing: every wave needs it right after.
                    if (c0 == 0 && l + 1 < net.num_layers) DTQN_WAIT_VMEM();
                }
                __syncthreads();
                if constexpr (FUSE) {
                    if (c0 == 0 && l + 1 < net.num_layers) fuse_arrive(a.fuse.counters, net.num_layers - 2 - l, t);
                }
                if (SPLITK || DTQN_BWD_GUARD(t.wave, 0)) {
#pragma unroll
                    for (int q = 0; q < (SPLITK || HALFW ? NC / 8 : NC / 4); ++q) DTQN_ASM_KEEP(w1f[0][q]);
                }
                g_tile_store(W5, LD5, gf(lgrd, net.gl_dhp, 4 * D) + c0, LP, NC, 4 * D);
                if constexpr (SPLITK) {
                    const float (&wh)[NC / 8] = *reinterpret_cast<const float (*)[NC / 8]>(&w1f[0][0]);
                    frag_dyw_mma<NC / 2, 1>(W5 + sk_kh * (NC / 2), LD5, wh, t, xacc[0]);
                } else if constexpr (HALFW) {
                    float (&wa)[NC / 8] = *reinterpret_cast<float (*)[NC / 8]>(&w1f[0][0]);
                    float (&wb)[NC / 8] = *reinterpret_cast<float (*)[NC / 8]>(&w1f[1][0]);
#pragma unroll
                    for (int q = 0; q < Own::PER_WAVE; ++q) {
                        const float* wcol = W1 + (size_t)c0 * D + Own::nt(t.wave, q) * 16 + t.i;
                        const float* rows = W5 + Own::mg(t.wave, q) * MGX * 16 * LD5;
                        if (DTQN_BWD_GUARD(t.wave, q)) {
                            frag_dyw_fetch<NC / 2>(wb, wcol + (size_t)(NC / 2) * D, D, t);
                            frag_dyw_mma<NC / 2, MGX>(rows, LD5, wa, t, xacc[q]);
                        }
                        if (q + 1 < Own::PER_WAVE && DTQN_BWD_GUARD(t.wave, q + 1))
                            frag_dyw_fetch<NC / 2>(wa, W1 + (size_t)c0 * D + Own::nt(t.wave, q + 1) * 16 + t.i, D, t);
                        if (DTQN_BWD_GUARD(t.wave, q)) frag_dyw_mma<NC / 2, MGX>(rows + NC / 2, LD5, wb, t, xacc[q]);
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < Own::PER_WAVE; ++q) {
                        if (q + 1 < Own::PER_WAVE && DTQN_BWD_GUARD(t.wave, q + 1))
                            frag_dyw_fetch<NC>(w1f[(q + 1) & 1], W1 + (size_t)c0 * D + Own::nt(t.wave, q + 1) * 16 + t.i, D, t);
                        if (DTQN_BWD_GUARD(t.wave, q))
                            frag_dyw_mma<NC, MGX>(W5 + Own::mg(t.wave, q) * MGX * 16 * LD5, LD5, w1f[q & 1], t, xacc[q]);
                    }
                }
                if (c0 + NC < 4 * D) g_dh.prefetch(W2 + c0 + NC, 4 * D, t);
                __syncthreads();
            }
            if constexpr (SPLITK) {      // (W5 is free: the loop ended on a barrier)
                if (t.wave >= Own::ITEMS) {
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) W5[((t.wave - Own::ITEMS) * 4 + r4) * 64 + t.lane] = xacc[0][0][r4];
                }
                __syncthreads();
                if (t.wave < Own::ITEMS) {
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) xacc[0][0][r4] += W5[(t.wave * 4 + r4) * 64 + t.lane];
                }
            }
            float* dst = ident ? DU : DX;
#pragma unroll
            for (int q = 0; q < Own::PER_WAVE; ++q) {
                if (DTQN_BWD_GUARD(t.wave, q)) {
                    const int c = Own::nt(t.wave, q) * 16 + t.i;
#pragma unroll
                    for (int m = 0; m < MGX; ++m)
#pragma unroll
                        for (int r4 = 0; r4 < 4; ++r4) {
                            const int r = (Own::mg(t.wave, q) * MGX + m) * 16 + t.kq * 4 + r4;
                            if (ident) dst[r * LDX + c] = xacc[q][m][r4];
                            else dst[r * LDX + c] += xacc[q][m][r4];
                        }
                }
            }
        }
        // LayerNorm in front of / behind the FFN  (T2 is free again: nobody reads df any more)
        tr.to_lds(T2, LDX, t);                         // s1
        // q | k | v (| o) of head group 0 go in flight now; they land in W5 after the LayerNorm backward
        // q, o: this slice's rows; k, v: every row up to the end of the slice, in LP-row chunks (chunk j = rows j*LP..)
        constexpr bool DOX = RS > 1 && DTQN_BWD_DOX && DTQN_SPLIT_ATTN_MFMA;
        TileRegs<NW, LP, GW> tq, to, tk[RS], tv[RS], tqu[DOX ? RS : 1];
        constexpr int NLSE = (GW / HD) * LPF, LSE_N = (NLSE + NT - 1) / NT;      // log-sum-exp rows of a head group, in flight with its tiles
        float lse_r[LSE_N];
        auto load_group = [&](int g) {
            if constexpr (kOptLse) {
#pragma unroll
                for (int k = 0; k < LSE_N; ++k) {
                    const int idx = t.tid + k * NT;
                    lse_r[k] = lrec[net.al_lse + g * NLSE + (idx < NLSE ? idx : 0)];
                }
            }
            const float* qkv0 = lrec + net.al_qkv + g * GW;
            tq.load(qkv0 + (size_t)R0 * 3 * D, 3 * D, t);
            if constexpr (DOX) {       // queries of the slices ABOVE this one: pass 2 runs this slice's keys against them
#pragma unroll
                for (int j = 1; j < RS; ++j)
                    if (j > slice) tqu[j].load(qkv0 + (size_t)j * LP * 3 * D, 3 * D, t);
            }
#pragma unroll
            for (int j = 0; j < RS; ++j)
                if (j <= slice) {
                    tk[j].load(qkv0 + (size_t)j * LP * 3 * D + D, 3 * D, t);
                    tv[j].load(qkv0 + (size_t)j * LP * 3 * D + 2 * D, 3 * D, t);
                }
            if (OST) to.load(rf(lrec, net.al_o, D) + g * GW, D, t);
        };
        load_group(0);
        const float* __restrict__ Wo = th + net.lo_out_w;
        const float* __restrict__ Win = th + net.lo_in_w;
        std::conditional_t<SPLITS, StageDyWSplit<D, NW, GW / 16>, StageDyW<D, MT, pick_mg(GW / 16, MT, NW), NW, GW / 16>> g_do;   // dO = da W_o[:, group]
        g_do.prefetch(Wo, D, t);
        __syncthreads();
        DTQN_PROF(a.prof, ps++);   // FFN bwd done
        if (!ident)   // u2 = LN1(s1): DX currently holds dL/du2 (skip + FFN branch)
            layernorm_backward<D, NW, FUSE, PAD>(DX, T2, DX, false, LDX, LP, st_s, th + net.lo_ln1_w, lsm, red, t, false, dreal);
        else          // u2 = LN2(s1) feeds only the FFN branch: stream grad += LN2'(DU)
            layernorm_backward<D, NW, FUSE, PAD>(DU, T2, DX, true, LDX, LP, st_s + 2 * LP, th + net.lo_ln2_w, lsm + 2 * D, red, t, false, dreal);
        __syncthreads();
        DTQN_PROF(a.prof, ps++);   // LN1 bwd done
        // attention gate.  res: s1 = x_in + relu(attn)  ->  da = ds1 * [y1 > 0]; gru as above.
        if (gru) {
            gru_gate_backward<D, MT, NW>(DX, T2, LDX, W5, LD5, theta + net.off_gate_attn, net, lrec + net.al_gate1, lgrd + net.gl_gate1, t, LPF, R0);
            __syncthreads();
        }
        {
            const float* m1 = mf(lrec, net.al_m1, D / 16);
            for (int idx = t.tid; idx < LP * D; idx += NT) {
                const int r = idx / D, c = idx - r * D;
                const float dy = gru ? T2[r * LDX + c] : DX[r * LDX + c];
                T2[r * LDX + c] = mask_bit(m1, D / 16, r, c) ? dy : 0.f;
            }
        }
        // attention backward, one head group (GW columns) at a time; du1 = dqkv W_in accumulates in registers
        DTQN_RELAUNDER();
        {
            f32x4 xacc[Own::PER_WAVE][MGX];
#pragma unroll
            for (int q = 0; q < Own::PER_WAVE; ++q)
#pragma unroll
                for (int m = 0; m < MGX; ++m) xacc[q][m] = zero4();
            float winf[2][GW / 4];
            // 16-row slices: the 3 GW contraction rows of du1 = [dq | dk | dv] W_in (q, k, v rows of W_in) as six pieces of GW / 2 rows,
            // three per half of the waves (wave w: column tile w % 4, pieces 3 (w / 4) ...): half the rows, half the fragment per wave
            constexpr bool SPLITW = kOptSplitW && Own::ITEMS * 2 == NW && Own::PER_WAVE == 1 && MGX == 1 && GW % 32 == 0;
            constexpr int WPC = GW / 2;                                      // rows of a piece
            const int sw_it = t.wave % Own::ITEMS, sw_kh = t.wave / Own::ITEMS;
            float wpc[2][WPC / 4];
            auto piece_fetch = [&](float (&wf)[WPC / 4], int g_, int j) {      // piece j = 0..2 of this wave's half
                const int u = sw_kh * 3 + j, part = u >> 1, off = (u & 1) * WPC;
                frag_dyw_fetch<WPC>(wf, Win + (size_t)(part * D + g_ * GW + off) * D + sw_it * 16 + t.i, D, t);
            };
            const float* o_g = rf(lrec, net.al_o, D);
            float* W5r = W5 + R0 * LD5;                // this slice's rows of the attention tiles
            constexpr int MGO = pick_mg(GW / 16, MT, NW);
#pragma unroll
            for (int g = 0; g < NG; ++g) {          // unrolled: the g == 0 / g + 1 < NG cases fold
                // q, k, v (o) of this head group -> W5   (W5 is free: the FFN / previous group are behind a barrier)
                tq.to_lds(W5r, LD5, t);
                if constexpr (DOX) {
#pragma unroll
                    for (int j = 1; j < RS; ++j)
                        if (j > slice) tqu[j].to_lds(W5 + j * LP * LD5, LD5, t);
                }
#pragma unroll
                for (int j = 0; j < RS; ++j)
                    if (j <= slice) {
                        tk[j].to_lds(W5 + j * LP * LD5 + GW, LD5, t);
                        tv[j].to_lds(W5 + j * LP * LD5 + 2 * GW, LD5, t);
                    }
                if (OST) to.to_lds(W5r + 5 * GW, LD5, t);
                // log-sum-exp of this group's heads -> LDS
                if constexpr (kOptLse) {
#pragma unroll
                    for (int k = 0; k < LSE_N; ++k) {
                        const int idx = t.tid + k * NT;
                        if (idx < NLSE) lse_s[idx] = lse_r[k];
                    }
                } else {
                    for (int idx = t.tid; idx < NLSE; idx += NT) lse_s[idx] = lrec[net.al_lse + g * NLSE + idx];
                }
                if constexpr (HD > 16) {           // delta of this slice's rows is summed up from two column tiles (the dO epilogue below)
                    for (int idx = t.tid; idx < (GW / HD) * LP; idx += NT) delta_s[(idx / LP) * LPF + R0 + idx % LP] = 0.f;
                }
                __syncthreads();                   // da (T2) visible
                g_do.retire();
                if (g == 0 && !DOX) g_tile_store(T2, LDX, gf(lgrd, net.gl_da, D), LP, D);      // (dO broadcast: behind the send, below)
                // do = da W_o restricted to this group's columns -> W5[:, 3GW:4GW];  delta = do . o per (row, head)
                float ov[MGO][4];
                auto do_pre = [&](int kt, int mg) {
                    if (!OST) {
#pragma unroll
                        for (int m = 0; m < MGO; ++m)
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                ov[m][r] = o_g[(size_t)((mg * MGO + m) * 16 + t.kq * 4 + r) * D + g * GW + kt * 16 + t.i];
                    }
                };
                auto do_epi = [&](int r, int c, float v) {
                    W5r[r * LD5 + 3 * GW + c] = v;
                    float p = v * (OST ? W5r[r * LD5 + 5 * GW + c] : ov[(r >> 4) % MGO][r & 3]);
#pragma unroll
                    for (int m = 1; m < (HD < 16 ? HD : 16); m <<= 1) p += __shfl_xor(p, m);
                    if constexpr (HD <= 16) {
                        if ((t.i & (HD - 1)) == 0) delta_s[(c / HD) * LPF + R0 + r] = p;
                    } else {
                        if (t.i == 0) atomicAdd(&delta_s[(c / HD) * LPF + R0 + r], p);
                    }
                };
                if constexpr (SPLITS) g_do.run(T2, LDX, t, red, do_pre, do_epi);      // (red: the LayerNorm scratch, idle during the attention stage)
                else g_do.run(T2, LDX, t, do_pre, do_epi);
                // first W_in fragment of this wave in flight during the attention passes
                if constexpr (SPLITW) piece_fetch(wpc[0], g, 0);
                else if (DTQN_BWD_GUARD(t.wave, 0))
                    frag_dyw_fetch<GW>(winf[0], Win + (size_t)(0 * D + g * GW) * D + Own::nt(t.wave, 0) * 16 + t.i, D, t);
                __syncthreads();
                DTQN_PROF(a.prof, ps++);   // qkv load + dO gemm done
                if constexpr (DOX) {
                    // dO (own rows, this head group) and delta go to the slices below: sc1 stores now, acknowledged behind pass 1;
                    // what the slices above sent is picked up between the passes
                    constexpr int PAIRS = RS * (RS - 1) / 2, SEG = LP * GW + (GW / HD) * LP;
                    const size_t grp = ((size_t)b * net.num_layers + l) * NG + g;
                    float* xg = a.xch + grp * PAIRS * LP * 2 * GW;          // (RS - 1) segments of SEG floats fit the pair regions
                    int32_t* fg = a.xflags + grp * PAIRS;
                    auto pair_id = [](int s_, int r_) { return s_ * (s_ - 1) / 2 + r_; };
                    static_assert((RS - 1) * SEG <= PAIRS * LP * 2 * GW, "dO segments fit the exchange region");
                    if (slice > 0) {
                        const DtqnRsrc rs = DTQN_XCH_RSRC(xg + (size_t)(slice - 1) * SEG, SEG * 4);
                        constexpr int C4 = GW / 4;
                        for (int idx = t.tid; idx < LP * C4; idx += NT) {
                            const int r = idx / C4, c = (idx - r * C4) * 4;
                            dtqn_xch_store4(rs, idx * 16, ld4(W5r + r * LD5 + 3 * GW + c));
                        }
                        for (int idx = t.tid; idx < (GW / HD) * LP / 4; idx += NT) {
                            const int h = idx / (LP / 4), r4 = (idx - h * (LP / 4)) * 4;
                            dtqn_xch_store4(rs, (LP * GW + h * LP + r4) * 4, ld4(delta_s + h * LPF + R0 + r4));
                        }
#if DTQN_BWD_DOX_EARLY_FLAG
                        DTQN_WAIT_VMEM();
                        __syncthreads();
                        if (t.tid < slice) DTQN_AGENT_STORE(fg + pair_id(slice, t.tid), (int32_t)1);
#endif
                    }
                    auto between = [&]() {
#if !DTQN_BWD_DOX_EARLY_FLAG
                        if (slice > 0) {
                            DTQN_WAIT_VMEM();
                            __syncthreads();
                            if (t.tid < slice) DTQN_AGENT_STORE(fg + pair_id(slice, t.tid), (int32_t)1);
                        }
#endif
                        // every sender above this slice at once: one thread per flag spins, then one pass over all their segments
                        const int nsnd = RS - 1 - slice;
                        if (nsnd > 0) {
                            if (t.tid < nsnd)
                                while (DTQN_AGENT_LOAD(fg + pair_id(slice + 1 + t.tid, slice)) == 0) DTQN_SPIN_PAUSE();
                            __syncthreads();
                            constexpr int C4 = GW / 4, TILE4 = LP * C4, DEL4 = (GW / HD) * LP / 4, SEG4 = TILE4 + DEL4;
                            const DtqnRsrc rs = DTQN_XCH_RSRC(xg + (size_t)slice * SEG, nsnd * SEG * 4);      // segments of senders slice + 1 ...
                            for (int idx = t.tid; idx < nsnd * SEG4; idx += NT) {
                                const int k = idx / SEG4, e = idx - k * SEG4, sndr = slice + 1 + k;
                                const float4 v = dtqn_xch_load4(rs, (k * SEG + e * 4) * 4);
                                if (e < TILE4) {
                                    const int r = e / C4, c = (e - r * C4) * 4;
                                    st4(W5 + (sndr * LP + r) * LD5 + 3 * GW + c, v);
                                } else {
                                    const int d = e - TILE4, h = d / (LP / 4), r4 = (d - h * (LP / 4)) * 4;
                                    st4(delta_s + h * LPF + sndr * LP + r4, v);
                                }
                            }
                            __syncthreads();
                            if (t.tid < nsnd) DTQN_AGENT_STORE(fg + pair_id(slice + 1 + t.tid, slice), (int32_t)0);
                        }
                    };
                    if constexpr (PAD)      // heads zero-padded to HD columns: the scale of the real head width
                        attention_backward_group_mfma<HD, NW>(W5, LD5, GW, LP, Lfull, delta_s, lse_s, t, nullptr, 0, R0, LPF, dr, l, g * (GW / HD),
                                                              true, LPF / 16, between, (float)net.hd_real);
                    else
                    attention_backward_group_mfma<HD, NW>(W5, LD5, GW, LP, Lfull, delta_s, lse_s, t, nullptr, 0, R0, LPF, dr, l, g * (GW / HD),
                                                          true, LPF / 16, between);
                    __syncthreads();
                } else {
                attention_backward_group<HD, NW, (HD >= kAttnMfmaMinHeadDim) || (RS > 1 && DTQN_SPLIT_ATTN_MFMA)>(W5, LD5, GW, LP, Lfull, delta_s, lse_s, t, nullptr, 0, R0, LPF, dr, l, g * (GW / HD));
                __syncthreads();
                if (RS > 1) {
                    // every slice s holds, in the k / v tiles of the rows BELOW it, its queries' share of their dK | dV: one
                    // hand-over per pair (s -> r), r < s, LP rows each; a slice first sends (it waits for nobody to do
                    // so), then adds what the slices above it sent, in slice order (deterministic)
                    constexpr int PAIRS = RS * (RS - 1) / 2;
                    const size_t grp = ((size_t)b * net.num_layers + l) * NG + g;
                    float* xg = a.xch + grp * PAIRS * LP * 2 * GW;
                    int32_t* fg = a.xflags + grp * PAIRS;
                    auto pair_id = [](int s_, int r_) { return s_ * (s_ - 1) / 2 + r_; };
                    if (slice > 0) {
                        // rows [0, R0) in receiver order are exactly pairs pair_id(slice, 0 .. slice-1): one contiguous region
                        const DtqnRsrc rs = DTQN_XCH_RSRC(xg + (size_t)pair_id(slice, 0) * LP * 2 * GW, R0 * 2 * GW * 4);
                        constexpr int C4 = 2 * GW / 4;
                        for (int idx = t.tid; idx < R0 * C4; idx += NT) {
                            const int r = idx / C4, c = (idx - r * C4) * 4;            // r: global row < R0
                            dtqn_xch_store4(rs, idx * 16, ld4(W5 + r * LD5 + GW + c));
                        }
                        DTQN_WAIT_VMEM();
                        __syncthreads();
                        if (t.tid < slice) DTQN_AGENT_STORE(fg + pair_id(slice, t.tid), (int32_t)1);
                    }
                    for (int sndr = slice + 1; sndr < RS; ++sndr)
                        xch_recv<NW, true>(W5r + GW, LD5, xg + (size_t)pair_id(sndr, slice) * LP * 2 * GW, LP, 2 * GW,
                                           fg + pair_id(sndr, slice), t);
                }
                }
                DTQN_PROF(a.prof, ps++);   // attention bwd done
                if constexpr (SPLITW) {
#pragma unroll
                    for (int q = 0; q < WPC / 4; ++q) DTQN_ASM_KEEP(wpc[0][q]);
                } else if (DTQN_BWD_GUARD(t.wave, 0)) {
#pragma unroll
                    for (int q = 0; q < GW / 4; ++q) DTQN_ASM_KEEP(winf[0][q]);
                }
                // (dO broadcast: the da record leaves HERE, behind the hand-overs -- in front of them the sender's drain before its flags
                //  and the receivers' loads would queue behind the acknowledgements of these write-through stores)
                if (g == 0 && DOX) g_tile_store(T2, LDX, gf(lgrd, net.gl_da, D), LP, D);
                // dq | dk | dv of the group -> grd record (columns of the packed [LP][3D] layout)
                for (int idx = t.tid; idx < LP * 3 * (GW / 4); idx += NT) {
                    const int r = idx / (3 * (GW / 4)), rem = idx - r * (3 * (GW / 4));
                    const int which = rem / (GW / 4), c = (rem - which * (GW / 4)) * 4;
                    const float* sp = W5r + r * LD5 + (which == 0 ? 4 * GW : which * GW) + c;
                    g_store4(gf(lgrd, net.gl_dqkv, 3 * D) + (size_t)r * 3 * D + which * D + g * GW + c, ld4(sp));
                }
                // du1 += dq W_in[q rows] + dk W_in[k rows] + dv W_in[v rows]: 3 fragments per owned item, double-buffered
                if constexpr (SPLITW) {
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        if (j + 1 < 3) piece_fetch(wpc[(j + 1) & 1], g, j + 1);
                        const int u = sw_kh * 3 + j, part = u >> 1, off = (u & 1) * WPC;
                        frag_dyw_mma<WPC, 1>(W5r + (part == 0 ? 4 * GW : part * GW) + off, LD5, wpc[j & 1], t, xacc[0]);
                    }
                } else
#pragma unroll
                for (int q = 0; q < Own::PER_WAVE; ++q) {
#pragma unroll
                    for (int part = 0; part < 3; ++part) {
                        const int step = q * 3 + part;
                        const int nq = part < 2 ? q : q + 1, np = part < 2 ? part + 1 : 0;
                        if (nq < Own::PER_WAVE && DTQN_BWD_GUARD(t.wave, nq))
                            frag_dyw_fetch<GW>(winf[(step + 1) & 1], Win + (size_t)(np * D + g * GW) * D + Own::nt(t.wave, nq) * 16 + t.i, D, t);
                        if (DTQN_BWD_GUARD(t.wave, q)) {
                            const float* rows = W5r + Own::mg(t.wave, q) * MGX * 16 * LD5 + (part == 0 ? 4 * GW : part * GW);
                            frag_dyw_mma<GW, MGX>(rows, LD5, winf[step & 1], t, xacc[q]);
                        }
                    }
                }
                if (g + 1 < NG) {
                    g_do.prefetch(Wo + (g + 1) * GW, D, t);
                    load_group(g + 1);
                }
                __syncthreads();
            }
            // next stage's saved activation goes in flight now: s2 of the layer below (post-LN), or the
            // LN1 input of this layer (identity)
            if (!ident) {
                if (l > 0) tr.load(rf(rec, net.ao_layer0 + (l - 1) * net.act_layer_stride + net.al_s2, D), D, t);
            } else {
                tr.load(l == 0 ? rf(rec, net.ao_x0, D) : rf(rec, net.ao_layer0 + (l - 1) * net.act_layer_stride + net.al_s2, D), D, t);
            }
            if constexpr (SPLITW) {      // (W5 is free: the group loop ended on a barrier)
                if (t.wave >= Own::ITEMS) {
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) W5[((t.wave - Own::ITEMS) * 4 + r4) * 64 + t.lane] = xacc[0][0][r4];
                }
                __syncthreads();
                if (t.wave < Own::ITEMS) {
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) xacc[0][0][r4] += W5[(t.wave * 4 + r4) * 64 + t.lane];
                }
            }
            float* dst = ident ? DU : DX;
#pragma unroll
            for (int q = 0; q < Own::PER_WAVE; ++q) {
                if (DTQN_BWD_GUARD(t.wave, q)) {
                    const int c = Own::nt(t.wave, q) * 16 + t.i;
#pragma unroll
                    for (int m = 0; m < MGX; ++m)
#pragma unroll
                        for (int r4 = 0; r4 < 4; ++r4) {
                            const int r = (Own::mg(t.wave, q) * MGX + m) * 16 + t.kq * 4 + r4;
                            if (ident) dst[r * LDX + c] = xacc[q][m][r4];
                            else dst[r * LDX + c] += xacc[q][m][r4];
                        }
                }
            }
        }
        __syncthreads();
        DTQN_PROF(a.prof, ps++);   // dqkv W_in done
        if (ident) {   // u1 = LN1(x_in): stream grad += LN1'(DU), x_in = layer input stream
            tr.to_lds(T2, LDX, t);
            __syncthreads();
            layernorm_backward<D, NW, FUSE, PAD>(DU, T2, DX, true, LDX, LP, st_s, th + net.lo_ln1_w, lsm, red, t, false, dreal);
            __syncthreads();
        }
    }

    // ---------------- embedding: dL/dx0 -> record; table / action-embedding partials ----------------
    DTQN_PROF(a.prof, ps++);
    if (dr.thresh != 0u) {       // x0 = dropout(embedding + position): dL/d(embedding) takes the keep mask
        for (int idx = t.tid; idx < LP * D; idx += NT) {
            const int r = idx / D, c = idx - r * D;
            DX[r * LDX + c] = drop_apply(dr, DROP_EMB, 0, (uint32_t)((R0 + r) * D + c), DX[r * LDX + c]);
        }
        __syncthreads();
    }
    g_tile_store(DX, LDX, gf(grec, net.go_dx0, D), LP, D);
    const float* obs_rows = a.obs + (size_t)ep * a.obs_ep_stride + (size_t)st0 * net.obs_dim;
    const uint8_t* act_rows = a.actions + (size_t)ep * a.act_ep_stride + st0;
    if (net.discrete) {
        const int KE = net.ke, KEP = net.kep, e = net.embed_per_obs, V = net.vocab, O = net.obs_dim;
        const float* __restrict__ We = theta + net.off_obs_w;
        float* dein = W5;    // [LP][KEP]: dL/d(gathered table rows) = dx0[:, a:] W_e
        for (int idx = t.tid; idx < LP * KEP; idx += NT) {
            const int r = idx / KEP, k = idx - r * KEP;
            float g = 0.f;
            if (r < L && k < KE)
                for (int d = 0; d < D - adim; ++d) g = fmaf(DX[r * LDX + adim + d], We[(size_t)d * KE + k], g);
            dein[idx] = g;
        }
        __syncthreads();
        for (int idx = t.tid; idx < V * e; idx += NT) {
            const int v = idx / e, c = idx - v * e;
            float g = 0.f;
            for (int r = 0; r < L; ++r)
                for (int j = 0; j < O; ++j) {
                    int tok = (int)obs_rows[(size_t)r * O + j];
                    tok = tok < 0 ? 0 : (tok >= V ? V - 1 : tok);
                    if (tok == v) g += dein[r * KEP + j * e + c];
                }
            s_store(srec + net.so_tab + idx, g);
        }
    }
    if (adim > 0) {
        for (int idx = t.tid; idx < A * adim; idx += NT) {
            const int v = idx / adim, c = idx - v * adim;
            float g = 0.f;
            if (Lfull == 1) {
                if ((int)act_rows[0] == v) g = DX[c];
            } else {
                for (int r = R0 > 0 ? 0 : 1; r < L; ++r)      // global row >= 1: the action that led to this observation
                    if ((int)act_rows[r - 1] == v) g += DX[r * LDX + c];
            }
            s_store(srec + net.so_act + idx, g);
        }
    }
    if constexpr (FUSE) {     // everything this workgroup publishes is in memory: the last event
        DTQN_WAIT_VMEM();
        __syncthreads();
        fuse_arrive(a.fuse.counters, net.num_layers, t);
    }
}

static size_t bwd_lds_bytes(const DtqnNet* net) {
    const int LP = net->lp, D = net->d_model, HD = net->head_dim;
    const int GW = D >= 64 ? 64 : D, NC = D >= 128 ? D : 2 * D;
    const int ntile = D <= 64 ? 6 : 5;
    const int W5C = ntile * GW > NC ? ntile * GW : NC;
    const int NT = waves_for(*net) * 64;
    const int PARTS = NT / D >= 1 ? NT / D : 1;
    size_t fl = 2 * (size_t)LP * (D + 4) + (size_t)LP * (W5C + 4) + (size_t)LP * net->ap + 2 * (size_t)(GW / HD) * LP +
                (size_t)PARTS * 2 * D + 4 * (size_t)LP;
    if (net->identity) fl += (size_t)LP * (D + 4);
    return fl * sizeof(float);
}

template <int D, int MT, int HD, int NW, bool GRU, int RS, bool DROP, bool FUSE = false>
static int launch_bwd3(const BwdArgs& a, hipStream_t stream) {
    const size_t lds = bwd_lds_bytes(&a.net);
    static size_t attr_lds[kMaxDevices] = {};    // per instantiation and device
    raise_lds_limit(reinterpret_cast<const void*>(&dtqn_backward_kernel<D, MT, HD, NW, GRU, RS, DROP, FUSE>), lds, attr_lds);
    (void)hipGetLastError();   // drop stale errors left by other users of the runtime
    const int grid = a.batch * RS + (FUSE ? a.fuse.n_role : 0);
    hipLaunchKernelGGL((dtqn_backward_kernel<D, MT, HD, NW, GRU, RS, DROP, FUSE>), dim3(grid), dim3(NW * 64), lds, stream, a, AheadArgs<false>{});
    return hipGetLastError() == hipSuccess ? DTQN_OK : DTQN_ERR_LAUNCH;
}
// the chain of this update + the target pass of the next one (four slices per sequence each: 8 * batch workgroups)
template <int HD, bool PAD = false>
static int launch_bwd_ahead(const BwdArgs& a, const FwdArgs& f, hipStream_t stream) {
    const size_t lb = bwd_lds_bytes(&a.net), lf = fwd_lds_bytes(&a.net, true);
    const size_t lds = lb > lf ? lb : lf;
    if (lds > 160 * 1024) return DTQN_ERR_CONFIG;
    static size_t attr_lds[kMaxDevices] = {};
    raise_lds_limit(reinterpret_cast<const void*>(&dtqn_backward_kernel<64, 1, HD, 8, false, 4, false, false, true, PAD>), lds, attr_lds);
    (void)hipGetLastError();
    AheadArgs<true> ah;
    ah.f = f;
    ah.f.block0 = a.batch * 4;
    hipLaunchKernelGGL((dtqn_backward_kernel<64, 1, HD, 8, false, 4, false, false, true, PAD>), dim3(a.batch * 4 + f.batch * 4), dim3(8 * 64), lds, stream, a, ah);
    return hipGetLastError() == hipSuccess ? DTQN_OK : DTQN_ERR_LAUNCH;
}
// the chain alone of the shapes that exist as four-slice kernels only (dtqn_limits.h, dtqn_ws_lite)
template <int HD, bool PAD>
static int launch_bwd_lite(const BwdArgs& a, hipStream_t stream) {
    const size_t lds = bwd_lds_bytes(&a.net);
    static size_t attr_lds[kMaxDevices] = {};
    raise_lds_limit(reinterpret_cast<const void*>(&dtqn_backward_kernel<64, 1, HD, 8, false, 4, false, false, false, PAD>), lds, attr_lds);
    (void)hipGetLastError();
    hipLaunchKernelGGL((dtqn_backward_kernel<64, 1, HD, 8, false, 4, false, false, false, PAD>), dim3(a.batch * 4), dim3(8 * 64), lds, stream, a, AheadArgs<false>{});
    return hipGetLastError() == hipSuccess ? DTQN_OK : DTQN_ERR_LAUNCH;
}
template <int D, int MT, int HD, int NW, bool GRU, int RS>
static int launch_bwd2(const BwdArgs& a, hipStream_t stream) {
    if constexpr (RS == 4 && !GRU && NW == kFuseWaves) {
        if (a.fuse.n_role > 0) {
            if (bwd_lds_bytes(&a.net) < (kFuseWaves + (size_t)(kDSmallFused / 64) * kFuseWaves * 64) * sizeof(float) ||
                bwd_lds_bytes(&a.net) < direct_lds_floats(kFuseWaves) * sizeof(float)) return DTQN_ERR_CONFIG;
            if (a.drop_thresh != 0u) return launch_bwd3<D, MT, HD, NW, GRU, RS, true, true>(a, stream);
            return launch_bwd3<D, MT, HD, NW, GRU, RS, false, true>(a, stream);
        }
    }
    if (a.fuse.n_role > 0) return DTQN_ERR_CONFIG;
    if (a.drop_thresh != 0u) return launch_bwd3<D, MT, HD, NW, GRU, RS, true>(a, stream);
    return launch_bwd3<D, MT, HD, NW, GRU, RS, false>(a, stream);
}
template <int D, int MT, int HD, int NW>
static int launch_bwd(const BwdArgs& a, hipStream_t stream) {
    if (a.net.gate == DTQN_GATE_GRU) {
        if constexpr (D <= 64) return launch_bwd2<D, MT, HD, NW, true, 1>(a, stream);
        else return DTQN_ERR_CONFIG;
    }
    return launch_bwd2<D, MT, HD, NW, false, 1>(a, stream);
}

}  // namespace dtqn

using namespace dtqn;

extern "C" int dtqn_lds_bytes_backward(const DtqnNet* net) {
    if (!net) return 0;
    const size_t b = bwd_lds_bytes(net);
    return b <= 160 * 1024 ? (int)b : 0;
}

// 1: dtqn_td_backward also computes the weight gradients (grad, norm_partial, step_counter[0]) in the same launch and
// dtqn_td_wgrad has nothing left to do
extern "C" int dtqn_td_wgrad_is_fused(const DtqnNet* net, const DtqnTd* td) {
    FuseArgs f;
    return fuse_plan(net, td, &f) ? 1 : 0;
}

static int td_backward(const DtqnNet* net, const DtqnReplay* rp, const DtqnTd* td, const DtqnTd* td_next, int draw_step_next, void* stream);

extern "C" int dtqn_td_backward(const DtqnNet* net, const DtqnReplay* rp, const DtqnTd* td, void* stream) {
    return td_backward(net, rp, td, nullptr, -1, stream);
}

extern "C" int dtqn_td_backward_ahead(const DtqnNet* net, const DtqnReplay* rp, const DtqnTd* td, const DtqnTd* td_next, int draw_step_next,
                                      void* stream) {
    if (!td_next || draw_step_next < 0) return DTQN_ERR_ARG;
    return td_backward(net, rp, td, td_next, draw_step_next, stream);
}

static int td_backward(const DtqnNet* net, const DtqnReplay* rp, const DtqnTd* td, const DtqnTd* td_next, int draw_step_next, void* stream) {
    if (!net || !rp || !td || td->batch < 1) return DTQN_ERR_ARG;
    if (td->history < 1 || td->history > net->ctx_len) return DTQN_ERR_ARG;
    if (net->tiled) return td_next ? DTQN_ERR_CONFIG : tiled_td_backward(net, rp, td, (hipStream_t)stream);
    if (bwd_lds_bytes(net) > 160 * 1024) return DTQN_ERR_CONFIG;
    BwdArgs a;
    a.net = *net;
    a.theta = td->theta_pol;
    a.act = td->act; a.grd = td->grd; a.small = td->small; a.q3 = td->q3; a.stats_partial = td->stats_partial;
    a.obs = rp->obs; a.actions = rp->actions; a.rewards = rp->rewards; a.dones = rp->dones;
    a.obs_ep_stride = (long long)(rp->max_steps + 1) * rp->obs_dim;
    a.act_ep_stride = rp->max_steps + 1;
    a.rew_ep_stride = rp->max_steps;
    a.ep_idx = td->ep_idx; a.start = td->start;
    a.batch = td->batch; a.history = td->history; a.gamma = td->gamma;
    a.prof = dtqn_debug_profile_buffer() ? static_cast<long long*>(dtqn_debug_profile_buffer()) + 64 : nullptr;
    a.xch = td->xch; a.xflags = td->xflags;
    a.drop_thresh = net->dropout > 0.f ? (uint32_t)((double)net->dropout * 4294967296.0) : 0u;
    a.drop_scale = net->dropout > 0.f ? 1.0f / (1.0f - net->dropout) : 1.0f;
    a.drop_seed = td->dropout_seed; a.step_counter = td->step_counter;
    fuse_plan(net, td, &a.fuse);
    const int D = net->d_model, MT = net->lp / 16, HD = net->head_dim, NW = waves_for(*net);
    hipStream_t s = (hipStream_t)stream;
    if (dtqn_ws_lite(net->tiled, D, HD, net->d_real)) {      // head width 32 / width-padded: the four-slice chain, alone or with the pass ahead
        if (td->row_split != 4 || net->lp != 64 || net->identity || net->gate != DTQN_GATE_RES || !a.xch || !a.xflags || a.drop_thresh != 0u ||
            a.fuse.n_role > 0 || NW != 8) return DTQN_ERR_CONFIG;
        const bool pad = net->d_real > 0;
        if (td_next != nullptr) {
            if (!dtqn_td_fwd_slices4_ok(net) || !td_next->sample_in_kernel || td_next->batch != td->batch || !td_next->xch || !td_next->xflags ||
                !td_next->q3) return DTQN_ERR_CONFIG;
            FwdArgs f;
            td_forward_args(net, rp, td_next, 2, draw_step_next, &f);
            f.nseq = td_next->batch;
            f.prof = nullptr;
            if (HD == 8 && pad) return launch_bwd_ahead<8, true>(a, f, s);
            if (HD == 16 && pad) return launch_bwd_ahead<16, true>(a, f, s);
            if (HD == 32) return pad ? launch_bwd_ahead<32, true>(a, f, s) : launch_bwd_ahead<32, false>(a, f, s);
            return DTQN_ERR_CONFIG;
        }
        if (HD == 8 && pad) return launch_bwd_lite<8, true>(a, s);
        if (HD == 16 && pad) return launch_bwd_lite<16, true>(a, s);
        if (HD == 32) return pad ? launch_bwd_lite<32, true>(a, s) : launch_bwd_lite<32, false>(a, s);
        return DTQN_ERR_CONFIG;
    }
    if (td->row_split == 2 || td->row_split == 4) {   // several workgroups per sequence (dtqn_td_row_split)
        if (net->lp != 64 || net->identity || !a.xch || !a.xflags) return DTQN_ERR_CONFIG;
        if (net->gate == DTQN_GATE_GRU) {
            if (td->row_split == 4) {
                if (D == 64 && HD == 8) return launch_bwd2<64, 1, 8, 8, true, 4>(a, s);
                if (D == 64 && HD == 16) return launch_bwd2<64, 1, 16, 8, true, 4>(a, s);
                return DTQN_ERR_CONFIG;
            }
            if (D == 64 && HD == 8) return launch_bwd2<64, 2, 8, 8, true, 2>(a, s);
            if (D == 64 && HD == 16) return launch_bwd2<64, 2, 16, 8, true, 2>(a, s);
            return DTQN_ERR_CONFIG;
        }
        if (td_next != nullptr) {      // + the next update's target pass in the same launch
            if (td->row_split != 4 || net->gate != DTQN_GATE_RES || a.drop_thresh != 0u || a.fuse.n_role > 0 || !dtqn_td_fwd_slices4_ok(net) ||
                !td_next->sample_in_kernel || td_next->batch != td->batch || !td_next->xch || !td_next->xflags || !td_next->q3 || NW != 8)
                return DTQN_ERR_CONFIG;
            FwdArgs f;
            td_forward_args(net, rp, td_next, 2, draw_step_next, &f);
            f.nseq = td_next->batch;
            f.prof = nullptr;
            if (D == 64 && HD == 8) return launch_bwd_ahead<8>(a, f, s);
            if (D == 64 && HD == 16) return launch_bwd_ahead<16>(a, f, s);
            return DTQN_ERR_CONFIG;
        }
        if (td->row_split == 4) {
            if (D == 64 && HD == 8) return launch_bwd2<64, 1, 8, 8, false, 4>(a, s);
            if (D == 64 && HD == 16) return launch_bwd2<64, 1, 16, 8, false, 4>(a, s);
            if (D == 128 && HD == 16) return launch_bwd2<128, 1, 16, 8, false, 4>(a, s);
            return DTQN_ERR_CONFIG;
        }
        if (D == 64 && HD == 8) return launch_bwd2<64, 2, 8, 8, false, 2>(a, s);
        if (D == 64 && HD == 16) return launch_bwd2<64, 2, 16, 8, false, 2>(a, s);
        if (D == 128 && HD == 16) return launch_bwd2<128, 2, 16, 8, false, 2>(a, s);
        return DTQN_ERR_CONFIG;
    }
#define DTQN_BWD_CASE(d, mt, hd, nw) \
    if (D == d && MT == mt && HD == hd && NW == nw) return launch_bwd<d, mt, hd, nw>(a, s);
    DTQN_WS_TRAIN_INSTANCES(DTQN_BWD_CASE)
#undef DTQN_BWD_CASE
    return DTQN_ERR_CONFIG;
}
