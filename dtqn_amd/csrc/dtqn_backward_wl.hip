// Weights-through-LDS variant of the whole-sequence backward kernel (dtqn_backward.hip): residual gate, post-LN layers,
// D <= 64.  Same stages, records, hand-overs and results; what changes is where the B operand of every `dY W` product
// comes from.
//
// The register-direct stages (StageDyW) read W column-wise: one 4-byte load per contraction step and lane, one item
// ahead of its use -- at 16-row slices every fragment feeds a single MFMA chain, so a stage is a chain of exposed L2 round
// trips (stage clocks, profiles/r02_stage_profile_cfg1_*.txt: dO 1.9 us against 0.26 us of MFMA issue).  Here the tile
// of stage s+1 is pulled global -> registers by all threads during stage s (coalesced 64-byte row segments), dropped
// into LDS TRANSPOSED (T[c][n] = W[n][c]: the contraction index becomes contiguous) once its region is free, and the stage
// is then exactly the forward's `X W'^T` stage on W' = W^T (StageXwL, both operands ds_read_b128).
//
// Arena (4 D (D + 4) floats, as the forward's):   region A = [0, 2D(D+4)),  region B = [2D(D+4), 4D(D+4))
//   A: head W_1^T  ->  FFN-2^T chunk 0 [NC][D+4]  ->  chunk 1  ->  (-> next layer's FFN-2^T chunk 0)
//   B: FFN-1^T chunk 0 [D][NC+4]  ->  chunk 1  ->  (-> next layer ...)
//   attention phase (FFN done, whole arena free): W_out^T [D][D+4] at 0, behind it W_in^T of the current head group, 3 x [D][GW+4]
// Attention head groups are GW = 32 columns wide here (64 in dtqn_backward.hip): the q | k | v | dO | dq | o tile set of a
// group then takes 50 KB instead of 99 KB, which is what makes room for the arena.
#include "dtqn_backward_args.hpp"
#include "dtqn_bwd_device.hpp"
#include "dtqn_wl.hpp"

#ifndef DTQN_SPLIT_ATTN_MFMA
#define DTQN_SPLIT_ATTN_MFMA 1
#endif

namespace dtqn {

// A [ROWS (n)][COLS (c)] global tile (row stride gld) parked in registers and dropped TRANSPOSED into LDS: dst[c][n],
// leading dim ldt.  One wave-instruction moves a 16 x 16-element block: lane = (n & 15) | (c4 << 4), so the global loads
// are 16 rows x 64 contiguous bytes and the four ds_write_b32 of a lane group hit 32 distinct banks (2-way at worst).
template <int NW, int ROWS, int COLS>
struct TileRegsT {
    static_assert(ROWS % 16 == 0 && COLS % 16 == 0, "16 x 16 blocks");
    static constexpr int CB = COLS / 16;
    static constexpr int BLK = (ROWS / 16) * CB;
    static constexpr int N = (BLK + NW - 1) / NW;
    float4 v[N];
    __device__ __forceinline__ void load(const float* __restrict__ g, int gld, const Thr& t) {
#pragma unroll
        for (int k = 0; k < N; ++k) {
            const int blk = t.wave + k * NW;
            if (BLK >= (k + 1) * NW || blk < BLK) {
                const int rb = blk / CB, cb = blk - rb * CB;
                v[k] = ld4(g + (size_t)(rb * 16 + (t.lane & 15)) * gld + cb * 16 + (t.lane >> 4) * 4);
            }
        }
    }
    __device__ __forceinline__ void to_lds_t(float* s, int ldt, const Thr& t) {
#pragma unroll
        for (int k = 0; k < N; ++k) {
            const int blk = t.wave + k * NW;
            if (BLK >= (k + 1) * NW || blk < BLK) {
                const int rb = blk / CB, cb = blk - rb * CB;
                float* d = s + (cb * 16 + (t.lane >> 4) * 4) * ldt + rb * 16 + (t.lane & 15);
                d[0] = v[k].x; d[ldt] = v[k].y; d[2 * ldt] = v[k].z; d[3 * ldt] = v[k].w;
            }
        }
    }
};

constexpr int bwl_gw(int D) { return D >= 32 ? 32 : D; }

template <int D, int MT, int HD, int NW, int RS>
__global__ __launch_bounds__(NW * 64) void dtqn_backward_wl_kernel(BwdArgs a) {
    static_assert(RS == 1 || RS == 2 || RS == 4, "one, two or four row slices");
    static_assert(D <= 64, "the weight arena is sized for D <= 64");
    constexpr int NT = NW * 64;
    constexpr int LP = MT * 16;
    constexpr int LPF = LP * RS;
    constexpr int LDX = D + 4;
    constexpr int GW = bwl_gw(D);                 // attention head-group width (columns)
    constexpr int NG = D / GW;
    constexpr int NC = 2 * D;                     // FFN hidden columns per pass
    constexpr int W5C = 6 * GW > NC ? 6 * GW : NC;     // q | k | v | dO | dq | o tiles of a head group, or a hidden chunk
    constexpr int LD5 = W5C + 4;
    constexpr int MGX = pick_mg(D / 16, MT, NW);
    constexpr int LWD = D + 4, LWC = NC + 4, LWG = GW + 4;
    constexpr int OFF_B = 2 * D * LWD;
    constexpr int PSB = 4 * D;                    // ln1 w, b | ln2 w, b of a layer
    static_assert(HD <= 16 && GW % HD == 0, "whole heads inside a group; a head inside one 16-column tile");
    constexpr int OFF_WIN = D * LWD;              // W_in^T of a head group sits right behind W_out^T (both live in the attention phase)
    static_assert(D * LWC <= 2 * D * LWD && OFF_WIN + 3 * D * LWG <= 4 * D * LWD, "FFN-1^T chunk fits region B, W_out^T | W_in^T the arena");
    using Own = Owned<D, MT, MGX, NW>;
    const DtqnNet& net = a.net;
    const Thr t = make_thr();
    const int b = (int)blockIdx.x / RS;
    const int slice = RS - 1 - ((int)blockIdx.x - b * RS);     // the upper slice (the producer of this kernel) first
    const int R0 = slice * LP;
    const int Lfull = net.ctx_len, A = net.num_actions, AP = net.ap, adim = net.action_dim;
    const int L = Lfull - R0 < LP ? Lfull - R0 : LP;
    const float* __restrict__ theta = a.theta;
    const float* rec = a.act + (size_t)b * net.act_stride;
    float* grec = a.grd + (size_t)b * net.grd_stride;
    float* srec = a.small + ((size_t)b * RS + slice) * net.sp_stride;
    auto rf = [&](const float* base, int off, int w) -> const float* { return base + off + (size_t)R0 * w; };
    auto gf = [&](float* base, int off, int w) -> float* { return base + off + (size_t)R0 * w; };
    auto mf = [&](const float* base, int off, int ctiles) -> const float* { return base + off + (size_t)(R0 / 16) * ctiles * 8; };

    float* DX = reinterpret_cast<float*>(dtqn_smem);   // dL/d(residual stream)        [LP][LDX]
    float* T2 = DX + LP * LDX;                         // narrow temp                   [LP][LDX]
    float* W5 = T2 + LP * LDX;                         // wide temp, attention tiles on GLOBAL rows [LPF][LD5]
    float* dq_s = W5 + LPF * LD5;                      // dL/dQ                         [LP][AP]
    float* delta_s = dq_s + LP * AP;                   // attention row terms           [GW/HD][LPF]
    float* lse_s = delta_s + (GW / HD) * LPF;
    float* red = W5;                                   // LN column-sum scratch [PARTS][2][D]: W5 is idle during both LayerNorm backwards
    constexpr int PARTS = NT / D >= 1 ? NT / D : 1;
    static_assert(PARTS * 2 * D <= LPF * LD5, "LayerNorm scratch fits the wide temp");
    float* st_s = lse_s + (GW / HD) * LPF;             // LayerNorm (mean, rstd) of this layer [2][LP][2]
    float* Ps = st_s + 4 * LP;                         // LayerNorm affines              [2][4 D]
    float* Ar = Ps + 2 * PSB;                          // weight arena                   [4 D (D + 4)]
    float* ArB = Ar + OFF_B;
    float* ArW = Ar + OFF_WIN;

    const int ep = a.ep_idx[b], st0 = a.start[b] + R0;
    const int top = net.num_layers - 1;
    // the head's W_1^T and the top layer's LayerNorm affines go in flight before the (one-wave, latency-bound) loss stage
    TileRegsT<NW, D, D> tw_dd;                         // a [D][D] matrix: head W_1, later W_out
    tw_dd.load(theta + net.off_head1_w, D, t);
    float4 ps_reg = make_float4(0.f, 0.f, 0.f, 0.f);
    if (t.tid < PSB / 4) ps_reg = ld4(layer_theta(net, theta, top) + net.lo_ln1_w + 4 * t.tid);
    TileRegs<NW, LP, D> tr, trh;
    trh.load(rf(rec, net.ao_hh, D), D, t);
    tr.load(rf(rec, net.ao_layer0 + top * net.act_layer_stride + net.al_s2, D), D, t);

    // ---------------- B0: double-DQN target, loss, dL/dQ, statistics (dtqn.py:219-253) ----------------
    {
        const float* q0 = a.q3 + (((size_t)0 * a.batch + b) * LPF + R0) * AP;
        const float* q1 = a.q3 + (((size_t)1 * a.batch + b) * LPF + R0) * AP;
        const float* q2 = a.q3 + (((size_t)2 * a.batch + b) * LPF + R0) * AP;
        const float inv_count = 1.0f / ((float)a.batch * (float)a.history);
        for (int idx = t.tid; idx < LP * AP; idx += NT) dq_s[idx] = 0.f;
        __syncthreads();
        if (t.wave == 0)
            td_loss_wave(q0, q1, q2, AP, A, Lfull - R0, LP, a.history, a.gamma, inv_count,
                         a.actions + (size_t)ep * a.act_ep_stride + st0, a.rewards + (size_t)ep * a.rew_ep_stride + st0,
                         a.dones + (size_t)ep * a.rew_ep_stride + st0, dq_s, a.stats_partial + ((size_t)b * RS + slice) * 8, t.lane);
        tw_dd.to_lds_t(Ar, LWD, t);
        if (t.tid < PSB / 4) st4(Ps + (top & 1) * PSB + 4 * t.tid, ps_reg);
        __syncthreads();
        for (int idx = t.tid; idx < LP * AP; idx += NT) gf(grec, net.go_dq, AP)[idx] = dq_s[idx];
    }

    // ---------------- B1: Q head backward ----------------
    TileRegsT<NW, D, NC> tw_f2;                        // W_2[:, chunk]   ([D rows n][NC cols c])  ->  A as [NC][D + 4]
    TileRegsT<NW, NC, D> tw_f1;                        // W_1[chunk, :]   ([NC rows n][D cols c])  ->  B as [D][NC + 4]
    TileRegsT<NW, GW, D> tw_in[3];                     // W_in[part*D + g*GW + n][c]               ->  B as 3 x [D][GW + 4]
    {
        const float* __restrict__ W2h = theta + net.off_head2_w;
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
                        for (int c = 0; c < A; ++c) g = fmaf(dq_s[r * AP + c], W2h[c * D + c0 + e], g);
                    g4[e] = g;
                }
                st4(T2 + r * LDX + c0, make_float4(g4[0], g4[1], g4[2], g4[3]));
            }
        }
    }
    tw_f2.load(layer_theta(net, theta, top) + net.lo_f2_w, 4 * D, t);      // top layer's FFN-2^T chunk 0, in flight during the head
    __syncthreads();
    tile_store<NW>(T2, LDX, gf(grec, net.go_dhh, D), LP, D, t);
    StageXwL<D, MT, pick_mg(D / 16, MT, NW), NW, D / 16>::run(T2, LDX, Ar, nullptr, t, [&](int r, int c, float v) { DX[r * LDX + c] = v; });
    __syncthreads();

    // ---------------- layers, last to first ----------------
    static_assert(4 * LP <= NT, "one statistics value per thread");
    auto st_fetch = [&](int l) -> float {
        const float* lr = rec + net.ao_layer0 + (size_t)l * net.act_layer_stride;
        if (t.tid >= 4 * LP) return 0.f;
        return t.tid < 2 * LP ? rf(lr, net.al_st1, 2)[t.tid] : rf(lr, net.al_st2, 2)[t.tid - 2 * LP];
    };
    float st_next = st_fetch(top);
    using GF2 = StageXwL<D, MT, pick_mg(NC / 16, MT, NW), NW, NC / 16>;      // dh = df W_2[:, chunk]
    using GDo = StageXwL<D, MT, pick_mg(GW / 16, MT, NW), NW, GW / 16>;      // dO = da W_out[:, group]
    constexpr int MGH = pick_mg(NC / 16, MT, NW);
    for (int l = top; l >= 0; --l) {
        const float* __restrict__ th = layer_theta(net, theta, l);
        const float* lrec = rec + net.ao_layer0 + (size_t)l * net.act_layer_stride;
        float* lgrd = grec + net.go_layer0 + (size_t)l * net.grd_layer_stride;
        float* lsm = srec + net.so_ln + l * 4 * D;
        const float* sm = Ps + (l & 1) * PSB;              // ln1 w | ln1 b | ln2 w | ln2 b
        const float* __restrict__ W1 = th + net.lo_f1_w;
        const float* __restrict__ W2 = th + net.lo_f2_w;
        const float* __restrict__ Wo = th + net.lo_out_w;
        const float* __restrict__ Win = th + net.lo_in_w;
        // ---- s0: x_out = LN2(s2): dL/ds2 ----
        tw_f2.to_lds_t(Ar, LWD, t);                        // FFN-2^T chunk 0 -> A (free: head / previous layer's W_out^T are behind a barrier)
        tw_f1.load(W1, D, t);                              // FFN-1^T chunk 0, in flight during the LayerNorm backward
        if (t.tid < 4 * LP) st_s[t.tid] = st_next;
        if (l > 0) st_next = st_fetch(l - 1);
        tr.to_lds(T2, LDX, t);
        __syncthreads();
        layernorm_backward<D, NW>(DX, T2, DX, false, LDX, LP, st_s + 2 * LP, sm + 2 * D, lsm + 2 * D, red, t);
        __syncthreads();
        // ---- s1: mlp gate  s2 = x1 + relu(f)  ->  df = ds2 * [y2 > 0] ----
        tw_f1.to_lds_t(ArB, LWC, t);                       // FFN-1^T chunk 0 -> B (free since the previous layer's last W_in^T product)
        tr.load(rf(lrec, net.al_s1, D), D, t);
        {
            const float* m2 = mf(lrec, net.al_m2, D / 16);
            for (int idx = t.tid; idx < LP * D; idx += NT) {
                const int r = idx / D, c = idx - r * D;
                T2[r * LDX + c] = mask_bit(m2, D / 16, r, c) ? DX[r * LDX + c] : 0.f;
            }
        }
        __syncthreads();
        // ---- s2..s5: FFN backward, dh = df W_2 (masked by h > 0), du2 = dh W_1, in two hidden-column passes ----
        f32x4 xacc[Own::PER_WAVE][MGX];
#pragma unroll
        for (int q = 0; q < Own::PER_WAVE; ++q)
#pragma unroll
            for (int m = 0; m < MGX; ++m) xacc[q][m] = zero4();
        const unsigned long long* mh = reinterpret_cast<const unsigned long long*>(mf(lrec, net.al_mh, 4 * D / 16));
#pragma unroll
        for (int c0 = 0; c0 < 4 * D; c0 += NC) {
            if (c0 == 0) {
                tw_f2.load(W2 + NC, 4 * D, t);             // chunk 1 of FFN-2^T, in flight during chunk 0
                tile_store<NW>(T2, LDX, gf(lgrd, net.gl_df, D), LP, D, t);
            } else {
                tw_dd.load(Wo, D, t);                      // W_out^T, in flight during FFN chunk 1
            }
            unsigned long long mw[MGH][4];                 // ReLU ballots of the item's accumulator registers
            GF2::run(T2, LDX, Ar, nullptr, t,
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
            __syncthreads();                               // dh chunk visible; region A free
            if (c0 == 0) {
                tw_f2.to_lds_t(Ar, LWD, t);                // FFN-2^T chunk 1 -> A
                tw_f1.load(W1 + (size_t)NC * D, D, t);     // FFN-1^T chunk 1, in flight during this chunk's second product
            } else {
                // head group 0's W_in^T parts, in flight during the second product of chunk 1
#pragma unroll
                for (int part = 0; part < 3; ++part) tw_in[part].load(Win + (size_t)(part * D) * D, D, t);
                if (l > 0 && t.tid < PSB / 4) ps_reg = ld4(layer_theta(net, theta, l - 1) + net.lo_ln1_w + 4 * t.tid);
            }
            tile_store<NW>(W5, LD5, gf(lgrd, net.gl_dhp, 4 * D) + c0, LP, NC, t, 4 * D);
#pragma unroll
            for (int q = 0; q < Own::PER_WAVE; ++q)
                if (Own::valid_fast(t.wave, q))
                    frag_xwl_mma<NC, MGX>(W5 + Own::mg(t.wave, q) * MGX * 16 * LD5, LD5, ArB + (Own::nt(t.wave, q) * 16 + t.i) * LWC, t, xacc[q]);
            if (c0 == 0) {
                __syncthreads();                           // chunk 0 of dh and of FFN-1^T consumed; FFN-2^T chunk 1 visible
                tw_f1.to_lds_t(ArB, LWC, t);               // FFN-1^T chunk 1 -> B
            }
        }
#pragma unroll
        for (int q = 0; q < Own::PER_WAVE; ++q) {
            if (Own::valid_fast(t.wave, q)) {
                const int c = Own::nt(t.wave, q) * 16 + t.i;
#pragma unroll
                for (int m = 0; m < MGX; ++m)
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) DX[((Own::mg(t.wave, q) * MGX + m) * 16 + t.kq * 4 + r4) * LDX + c] += xacc[q][m][r4];
            }
        }
        // ---- s6: u2 = LN1(s1): DX holds dL/du2 (skip + FFN branch) ----
        tr.to_lds(T2, LDX, t);                             // s1
        // q | k | v | o of head group 0 go in flight now; they land in W5 after the LayerNorm backward
        TileRegs<NW, LP, GW> tq, to, tk[RS], tv[RS];
        auto load_group = [&](int g) {
            const float* qkv0 = lrec + net.al_qkv + g * GW;
            tq.load(qkv0 + (size_t)R0 * 3 * D, 3 * D, t);
#pragma unroll
            for (int j = 0; j < RS; ++j)
                if (j <= slice) {
                    tk[j].load(qkv0 + (size_t)j * LP * 3 * D + D, 3 * D, t);
                    tv[j].load(qkv0 + (size_t)j * LP * 3 * D + 2 * D, 3 * D, t);
                }
            to.load(rf(lrec, net.al_o, D) + g * GW, D, t);
        };
        load_group(0);
        __syncthreads();                                   // FFN products done: the whole arena is free
        tw_dd.to_lds_t(Ar, LWD, t);                        // W_out^T -> A, all head groups: group g reads rows [g GW, (g+1) GW)
#pragma unroll
        for (int part = 0; part < 3; ++part) tw_in[part].to_lds_t(ArW + part * D * LWG, LWG, t);     // W_in^T of group 0, behind W_out^T
        layernorm_backward<D, NW>(DX, T2, DX, false, LDX, LP, st_s, sm, lsm, red, t);
        __syncthreads();
        // ---- s7: attention gate  s1 = x_in + relu(attn)  ->  da = ds1 * [y1 > 0] ----
        {
            const float* m1 = mf(lrec, net.al_m1, D / 16);
            for (int idx = t.tid; idx < LP * D; idx += NT) {
                const int r = idx / D, c = idx - r * D;
                T2[r * LDX + c] = mask_bit(m1, D / 16, r, c) ? DX[r * LDX + c] : 0.f;
            }
        }
        // ---- attention backward, one head group (GW columns) at a time; du1 = dqkv W_in accumulates in registers ----
#pragma unroll
        for (int q = 0; q < Own::PER_WAVE; ++q)
#pragma unroll
            for (int m = 0; m < MGX; ++m) xacc[q][m] = zero4();
        float* W5r = W5 + R0 * LD5;                        // this slice's rows of the attention tiles
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            tq.to_lds(W5r, LD5, t);
#pragma unroll
            for (int j = 0; j < RS; ++j)
                if (j <= slice) {
                    tk[j].to_lds(W5 + j * LP * LD5 + GW, LD5, t);
                    tv[j].to_lds(W5 + j * LP * LD5 + 2 * GW, LD5, t);
                }
            to.to_lds(W5r + 5 * GW, LD5, t);
            for (int idx = t.tid; idx < (GW / HD) * LPF; idx += NT) lse_s[idx] = lrec[net.al_lse + g * (GW / HD) * LPF + idx];
            if (g > 0) {
#pragma unroll
                for (int part = 0; part < 3; ++part) tw_in[part].to_lds_t(ArW + part * D * LWG, LWG, t);    // this group's W_in^T
            }
            __syncthreads();                               // da (T2), the group's tiles and weight tiles visible
            if (g == 0) tile_store<NW>(T2, LDX, gf(lgrd, net.gl_da, D), LP, D, t);
            // do = da W_out restricted to this group's columns -> W5[:, 3GW:4GW];  delta = do . o per (row, head)
            GDo::run(T2, LDX, Ar + g * GW * LWD, nullptr, t, [&](int r, int c, float v) {
                W5r[r * LD5 + 3 * GW + c] = v;
                float p = v * W5r[r * LD5 + 5 * GW + c];
#pragma unroll
                for (int m = 1; m < HD; m <<= 1) p += __shfl_xor(p, m);
                if ((t.i & (HD - 1)) == 0) delta_s[(c / HD) * LPF + R0 + r] = p;
            });
            __syncthreads();
            attention_backward_group<HD, NW, (HD >= kAttnMfmaMinHeadDim) || (RS > 1 && DTQN_SPLIT_ATTN_MFMA)>(W5, LD5, GW, LP, Lfull, delta_s, lse_s, t, nullptr, 0, R0, LPF);
            __syncthreads();
            if (RS > 1) {
                // every slice s holds, in the k / v tiles of the rows BELOW it, its queries' share of their dK | dV: one
                // hand-over per pair (s -> r), r < s, LP rows each; a slice first sends, then adds what the slices above
                // it sent, in slice order (deterministic)
                constexpr int PAIRS = RS * (RS - 1) / 2;
                const size_t grp = ((size_t)b * net.num_layers + l) * NG + g;
                float* xg = a.xch + grp * PAIRS * LP * 2 * GW;
                int32_t* fg = a.xflags + grp * PAIRS;
                auto pair_id = [](int s_, int r_) { return s_ * (s_ - 1) / 2 + r_; };
                if (slice > 0) {
                    const DtqnRsrc rs = DTQN_XCH_RSRC(xg + (size_t)pair_id(slice, 0) * LP * 2 * GW, R0 * 2 * GW * 4);
                    constexpr int C4 = 2 * GW / 4;
                    for (int idx = t.tid; idx < R0 * C4; idx += NT) {
                        const int r = idx / C4, c = (idx - r * C4) * 4;
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
            // what the next stage needs goes in flight before this group's last product: the next group's tiles and
            // W_in^T parts, or (after the last group) the FFN-2^T chunk 0 of the layer below
            if (g + 1 < NG) {
#pragma unroll
                for (int part = 0; part < 3; ++part) tw_in[part].load(Win + (size_t)(part * D + (g + 1) * GW) * D, D, t);
            } else if (l > 0) {
                tw_f2.load(layer_theta(net, theta, l - 1) + net.lo_f2_w, 4 * D, t);
            }
            // dq | dk | dv of the group -> grd record (columns of the packed [LP][3D] layout)
            for (int idx = t.tid; idx < LP * 3 * (GW / 4); idx += NT) {
                const int r = idx / (3 * (GW / 4)), rem = idx - r * (3 * (GW / 4));
                const int which = rem / (GW / 4), c = (rem - which * (GW / 4)) * 4;
                const float* sp = W5r + r * LD5 + (which == 0 ? 4 * GW : which * GW) + c;
                st4(gf(lgrd, net.gl_dqkv, 3 * D) + (size_t)r * 3 * D + which * D + g * GW + c, ld4(sp));
            }
            // du1 += dq W_in[q rows] + dk W_in[k rows] + dv W_in[v rows]
#pragma unroll
            for (int q = 0; q < Own::PER_WAVE; ++q) {
                if (Own::valid_fast(t.wave, q)) {
#pragma unroll
                    for (int part = 0; part < 3; ++part) {
                        const float* rows = W5r + Own::mg(t.wave, q) * MGX * 16 * LD5 + (part == 0 ? 4 * GW : part * GW);
                        frag_xwl_mma<GW, MGX>(rows, LD5, ArW + part * D * LWG + (Own::nt(t.wave, q) * 16 + t.i) * LWG, t, xacc[q]);
                    }
                }
            }
            if (g + 1 < NG) load_group(g + 1);
            __syncthreads();                               // the group's tiles and W_in^T consumed
        }
        // next stage's saved activation goes in flight now: s2 of the layer below
        if (l > 0) tr.load(rf(rec, net.ao_layer0 + (l - 1) * net.act_layer_stride + net.al_s2, D), D, t);
        if (l > 0 && t.tid < PSB / 4) st4(Ps + ((l - 1) & 1) * PSB + 4 * t.tid, ps_reg);
#pragma unroll
        for (int q = 0; q < Own::PER_WAVE; ++q) {
            if (Own::valid_fast(t.wave, q)) {
                const int c = Own::nt(t.wave, q) * 16 + t.i;
#pragma unroll
                for (int m = 0; m < MGX; ++m)
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) DX[((Own::mg(t.wave, q) * MGX + m) * 16 + t.kq * 4 + r4) * LDX + c] += xacc[q][m][r4];
            }
        }
        __syncthreads();
    }

    // ---------------- embedding: dL/dx0 -> record; table / action-embedding partials ----------------
    tile_store<NW>(DX, LDX, gf(grec, net.go_dx0, D), LP, D, t);
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
            srec[net.so_tab + idx] = g;
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
            srec[net.so_act + idx] = g;
        }
    }
}

static size_t bwd_wl_lds_bytes(const DtqnNet* net, int NW) {
    const int LP = net->lp, D = net->d_model, HD = net->head_dim;       // LPF = LP whatever the slicing: W5 spans the sequence
    const int GW = bwl_gw(D), NC = 2 * D;
    const int W5C = 6 * GW > NC ? 6 * GW : NC;
    (void)NW;
    const size_t fl = 2 * (size_t)LP * (D + 4) + (size_t)LP * (W5C + 4) + (size_t)LP * net->ap + 2 * (size_t)(GW / HD) * LP +
                      4 * (size_t)LP + 2 * 4 * (size_t)D + (size_t)wl_arena_floats(D);
    return fl * sizeof(float);
}

bool bwd_wl_ok(const DtqnNet* net, int rs) {
    (void)rs;
    const int D = net->d_model;
    if (net->tiled || net->gate != DTQN_GATE_RES || net->identity || D > 64) return false;
    if (net->head_dim > 16 || bwl_gw(D) % net->head_dim != 0) return false;
    const char* e = getenv("DTQN_WL");
    if (e != nullptr && (e[0] == '0' || e[0] == 'f')) return false;      // 0: no weights-through-LDS kernels; f: forward only
    const int b = net->lo_ln1_w;
    if (!(net->lo_ln1_b - b == D && net->lo_ln2_w - b == 2 * D && net->lo_ln2_b - b == 3 * D)) return false;
    return bwd_wl_lds_bytes(net, waves_for(*net)) <= 160 * 1024;
}

template <int D, int MT, int HD, int NW, int RS>
static int launch_one(const BwdArgs& a, hipStream_t stream) {
    const size_t lds = bwd_wl_lds_bytes(&a.net, NW);
    static size_t attr_lds[kMaxDevices] = {};
    raise_lds_limit(reinterpret_cast<const void*>(&dtqn_backward_wl_kernel<D, MT, HD, NW, RS>), lds, attr_lds);
    (void)hipGetLastError();   // drop stale errors left by other users of the runtime
    hipLaunchKernelGGL((dtqn_backward_wl_kernel<D, MT, HD, NW, RS>), dim3(a.batch * RS), dim3(NW * 64), lds, stream, a);
    return hipGetLastError() == hipSuccess ? DTQN_OK : DTQN_ERR_LAUNCH;
}

int launch_bwd_wl(const BwdArgs& a, int D, int MT, int HD, int NW, int RS, hipStream_t stream) {
#define DTQN_BWL_CASE(d, mt, hd, nw, rs) \
    if (D == d && MT == mt && HD == hd && NW == nw && RS == rs) return launch_one<d, mt, hd, nw, rs>(a, stream);
    DTQN_BWL_CASE(64, 1, 8, 8, 4)
    DTQN_BWL_CASE(64, 1, 16, 8, 4)
    DTQN_BWL_CASE(64, 2, 8, 8, 2)
    DTQN_BWL_CASE(64, 2, 16, 8, 2)
    DTQN_BWL_CASE(64, 4, 8, 8, 1)
    DTQN_BWL_CASE(64, 4, 16, 8, 1)
    DTQN_BWL_CASE(64, 2, 8, 8, 1)
    DTQN_BWL_CASE(64, 1, 8, 8, 1)
    DTQN_BWL_CASE(16, 1, 8, 4, 1)
    DTQN_BWL_CASE(32, 2, 8, 4, 1)
    DTQN_BWL_CASE(32, 1, 16, 4, 1)
#undef DTQN_BWL_CASE
    return DTQN_ERR_CONFIG;
}

}  // namespace dtqn
