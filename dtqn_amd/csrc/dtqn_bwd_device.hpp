// Backward building blocks shared by the whole-sequence kernel (dtqn_backward.hip) and the row-block tiled
// path (dtqn_tiled.hip): LayerNorm backward and the flash-style attention backward passes.
#pragma once
#include "dtqn_device.hpp"

namespace dtqn {

// LayerNorm backward over the rows of a [LP][ld] tile.
//   dy   : gradient w.r.t. the LN output            (LDS, [LP][ld])
//   xin  : the LN input                             (LDS, [LP][ld])
//   st   : (mean, rstd) per row                     (global, [LP][2])
//   dst  : receives rstd*(g - mean(g) - xhat*mean(g*xhat)), g = gamma*dy; assigned or accumulated
//   dgb  : per-sequence partial of d gamma ([D]) followed (at +D) by d beta ([D])   (global)
// Contains two __syncthreads(); caller must sync before (inputs ready) and after (dst ready).
// WT: dgb is read by other workgroups of the same launch (fused weight gradients): agent-scope (write-through) stores
// PAD (width-padded networks, DtqnNet.d_real): the two row means run over the first d_real columns and the columns behind them get 0
// (their dy and gamma are zero, so the column sums of pass A are zero there by themselves)
template <int D, int NW, bool WT = false, bool PAD = false>
__device__ __forceinline__ void layernorm_backward(const float* dy, const float* xin, float* dst, bool accumulate,
                                                   int ld, int LP, const float* __restrict__ st,
                                                   const float* __restrict__ gamma, float* __restrict__ dgb,
                                                   float* red, const Thr& t, bool dgb_accumulate = false, int d_real = D) {
    constexpr int NT = NW * 64;
    constexpr int PARTS = NT / D >= 1 ? NT / D : 1;
    // pass A: column sums  d gamma[d] = sum_r dy*xhat,  d beta[d] = sum_r dy
    {
        const int d = t.tid % D, part = t.tid / D;
        if (part < PARTS) {
            float sg = 0.f, sb = 0.f;
            for (int r = part; r < LP; r += PARTS) {
                const float mean = st[r * 2], rstd = st[r * 2 + 1];
                const float g = dy[r * ld + d];
                sg = fmaf(g, (xin[r * ld + d] - mean) * rstd, sg);
                sb += g;
            }
            red[(part * 2 + 0) * D + d] = sg;
            red[(part * 2 + 1) * D + d] = sb;
        }
    }
    __syncthreads();
    for (int idx = t.tid; idx < 2 * D; idx += NT) {
        const int which = idx / D, d = idx - which * D;
        float s = 0.f;
        for (int p = 0; p < PARTS; ++p) s += red[(p * 2 + which) * D + d];
        if (dgb_accumulate) dgb[which * D + d] += s;     // later row blocks of the same sequence (tiled path)
        else if constexpr (WT) DTQN_AGENT_STORE(dgb + which * D + d, s);
        else dgb[which * D + d] = s;
    }
    // pass B: rows (LPR lanes per row)
    constexpr int LPR = (NT / DTQN_MAX_LP) < (D / 4) ? (NT / DTQN_MAX_LP) : (D / 4);
    constexpr int NV = D / (4 * LPR);
    float4 o[NV];
    int row = t.tid / LPR;
    const int part = t.tid % LPR;
    const bool valid = row < LP;
    row = valid ? row : 0;
    {
        const float mean = st[row * 2], rstd = st[row * 2 + 1];
        const float* yp = dy + row * ld + part * 4;
        const float* xp = xin + row * ld + part * 4;
        float4 gq[NV], xh[NV];
        float c1 = 0.f, c2 = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const float4 y = ld4(yp + 4 * LPR * j), x = ld4(xp + 4 * LPR * j), gm = ld4(gamma + part * 4 + 4 * LPR * j);
            gq[j] = make_float4(y.x * gm.x, y.y * gm.y, y.z * gm.z, y.w * gm.w);
            xh[j] = make_float4((x.x - mean) * rstd, (x.y - mean) * rstd, (x.z - mean) * rstd, (x.w - mean) * rstd);
            c1 += (gq[j].x + gq[j].y) + (gq[j].z + gq[j].w);
            c2 += (gq[j].x * xh[j].x + gq[j].y * xh[j].y) + (gq[j].z * xh[j].z + gq[j].w * xh[j].w);
        }
#pragma unroll
        for (int m = 1; m < LPR; m <<= 1) { c1 += __shfl_xor(c1, m); c2 += __shfl_xor(c2, m); }
        const float inv_d = PAD ? 1.0f / (float)d_real : (1.0f / D);
        c1 *= inv_d;
        c2 *= inv_d;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            o[j].x = rstd * (gq[j].x - c1 - xh[j].x * c2);
            o[j].y = rstd * (gq[j].y - c1 - xh[j].y * c2);
            o[j].z = rstd * (gq[j].z - c1 - xh[j].z * c2);
            o[j].w = rstd * (gq[j].w - c1 - xh[j].w * c2);
            if constexpr (PAD) {
                const int c0 = part * 4 + 4 * LPR * j;
                o[j].x = c0 < d_real ? o[j].x : 0.f; o[j].y = c0 + 1 < d_real ? o[j].y : 0.f;
                o[j].z = c0 + 2 < d_real ? o[j].z : 0.f; o[j].w = c0 + 3 < d_real ? o[j].w : 0.f;
            }
        }
    }
    __syncthreads();   // every lane has read dy / dst before anyone overwrites dst (dst may alias dy)
    if (valid) {
        float* dp = dst + row * ld + part * 4;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            if (accumulate) {
                const float4 p = ld4(dp + 4 * LPR * j);
                st4(dp + 4 * LPR * j, make_float4(p.x + o[j].x, p.y + o[j].y, p.z + o[j].z, p.w + o[j].w));
            } else {
                st4(dp + 4 * LPR * j, o[j]);
            }
        }
    }
}

// Attention backward for one head group resident in W5 = [q | k | v | do | dq] (GW columns each), on the f32
// matrix core.  Flash-style recomputation from the saved log-sum-exp: P = exp(S - lse), no reductions.
//   delta_s[h][t] = dO . o was produced by the dO GEMM's epilogue; lse_s[h][t] is staged by the caller.
//   pass 1, item = (head, 16-row QUERY tile ti), key tiles tj <= ti:
//       S^T = K Q^T and dP^T = V dO^T come out of the MFMA with lane (i, kq) holding keys s = kq*4 + r of query
//       t = i, so lse / delta are per-lane scalars and dS^T = P^T (dP^T - delta) is directly the B operand of
//       dQ^T[c][t] += K^T[c][s] dS^T[s][t]
//   pass 2, item = (head, 16-row KEY tile tj), query tiles ti >= tj:
//       S = Q K^T and dP = dO V^T (rows t = kq*4 + r, column s = i); P and dS are the B operands of
//       dV^T[c][s] += dO^T[c][t] P[t][s] and dK^T[c][s] += Q^T[c][t] dS[t][s]        (in place over k, v at the end)
// Lane (i, kq) ends up with four consecutive columns c = kq*4.. of output row i: one 16-byte store.
// Row slices, "own keys" form (round 3; p2_own = true): pass 2 covers the KEY tiles of this slice only, against every query tile at or
// above them (q | dO | lse | delta of the rows above this slice must be in the tiles: the caller's `between` functor, run between the
// passes, receives what the upper slices broadcast).  dk | dv of the slice's rows are then complete here and nothing is handed down
// afterwards; the work is (keys below) + (queries above) = RS + 1 tile pairs for EVERY slice instead of 2 ... 2 RS.
struct AttnNoBetween {
    __device__ __forceinline__ void operator()() const {}
};
template <int HD, int NW, typename BT = AttnNoBetween>
__device__ __forceinline__ void attention_backward_group_mfma(float* W5, int ld, int GW, int LP, int n,
                                                              const float* delta_s, const float* lse_s, const Thr& t,
                                                              float* dq_base = nullptr, int dq_ld = 0, int row0 = 0, int sl_ld = 0,
                                                              const Drop& dr = Drop{0u, 1.0f, 0u, 0u, 0u}, int layer = 0, int head0 = 0,
                                                              bool p2_own = false, int q_tiles = 0, BT between = BT(),
                                                              float hd_eff = (float)HD) {          // hd_eff: see attention_forward
    // dr / layer / head0 (global index of the group's first head): attention-probability dropout of the forward, recomputed:
    // o = (P * M) v with M = keep / (1 - p), so dV takes P * M, and dP = M * (dO v^T) before the softmax backward
    // dq goes to W5's fifth tile by default, or to dq_base (row stride dq_ld; may be global memory) when the
    // caller cannot afford a fifth LDS tile.  Row slices as in the VALU version: queries [row0, row0 + LP)
    // (row0 a multiple of 16), keys [0, row0 + LP), global rows everywhere.
    if (dq_base == nullptr) { dq_base = W5 + 4 * GW; dq_ld = ld; }
    if (sl_ld == 0) sl_ld = LP;
    constexpr int KS = HD / 4, CT = (HD + 15) / 16;
    constexpr float LOG2E = 1.4426950408889634f;
    const int HG = GW / HD, MT = LP / 16, last_tile = (n - 1) / 16;
    const int T0 = row0 / 16, MTK = T0 + MT;                 // first query tile; number of key tiles
    const float scale = 1.0f / sqrtf(hd_eff), scale2 = scale * LOG2E;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int item = t.wave; item < HG * MT; item += NW) {
        const int h = item % HG, ti = T0 + item / HG;
        const int t0 = ti * 16, trow = t0 + t.i;
        float* dqp = dq_base + (size_t)trow * dq_ld + h * HD;
        if (ti > last_tile) {
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
                if (ct * 16 + t.kq * 4 < HD) st4(dqp + ct * 16 + t.kq * 4, z4);
            continue;
        }
        float qf[KS], dof[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            qf[s] = W5[trow * ld + h * HD + t.kq * KS + s] * scale2;
            dof[s] = W5[trow * ld + 3 * GW + h * HD + t.kq * KS + s];
        }
        const float lse2 = lse_s[h * sl_ld + trow] * LOG2E, delta = delta_s[h * sl_ld + trow];
        f32x4 acc[CT][2];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) { acc[ct][0] = zero4(); acc[ct][1] = zero4(); }
        for (int tj = 0; tj <= ti; ++tj) {
            const int s0 = tj * 16;
            const float* kp = W5 + (s0 + t.i) * ld + GW + h * HD + t.kq * KS;
            const float* vp = kp + GW;
            f32x4 st = zero4(), dp = zero4();
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                st = mfma16(kp[s], qf[s], st);
                dp = mfma16(vp[s], dof[s], dp);
            }
            float ds[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float p = DTQN_EXP2(st[r] - lse2);
                // masked keys, and PAD query rows (t >= n: their saved lse is 0, so exp2 can overflow and inf * 0 would
                // put a NaN into dq of a pad row, which the weight-gradient contraction over all padded rows would pick up)
                if (s0 + t.kq * 4 + r > trow || trow >= n) p = 0.f;
                const float dpm = dr.thresh == 0u ? dp[r] : (drop_keep(dr, DROP_ATTN, layer, drop_attn_idx(head0 + h, trow, s0 + t.kq * 4 + r)) ? dp[r] * dr.scale : 0.f);
                ds[r] = p * (dpm - delta);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float* krow = W5 + (s0 + t.kq * 4 + r) * ld + GW + h * HD;
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    const int c = ct * 16 + t.i;
                    acc[ct][r & 1] = mfma16(krow[c < HD ? c : 0], ds[r], acc[ct][r & 1]);
                }
            }
        }
        const float f = trow < n ? scale : 0.f;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const int c = ct * 16 + t.kq * 4;
            if (c < HD)
                st4(dqp + c, make_float4((acc[ct][0][0] + acc[ct][1][0]) * f, (acc[ct][0][1] + acc[ct][1][1]) * f,
                                         (acc[ct][0][2] + acc[ct][1][2]) * f, (acc[ct][0][3] + acc[ct][1][3]) * f));
        }
    }
    between();
    __syncthreads();
    const int qt_top = p2_own ? q_tiles - 1 : MTK - 1;                // last query tile pass 2 may look at
    const int ti_hi = last_tile < qt_top ? last_tile : qt_top;        // ... with live rows
    const int kt_lo = p2_own ? T0 : 0;                                // first key tile of pass 2
    for (int item = t.wave; item < HG * (MTK - kt_lo); item += NW) {
        const int h = item % HG, tj = kt_lo + item / HG;
        const int s0 = tj * 16, srow = s0 + t.i;
        float* kout = W5 + srow * ld + GW + h * HD;
        float* vout = kout + GW;
        if (tj > ti_hi) {
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
                if (ct * 16 + t.kq * 4 < HD) { st4(kout + ct * 16 + t.kq * 4, z4); st4(vout + ct * 16 + t.kq * 4, z4); }
            continue;
        }
        float kf[KS], vf[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            kf[s] = kout[t.kq * KS + s] * scale2;
            vf[s] = vout[t.kq * KS + s];
        }
        f32x4 acck[CT][2], accv[CT][2];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) { acck[ct][0] = zero4(); acck[ct][1] = zero4(); accv[ct][0] = zero4(); accv[ct][1] = zero4(); }
        for (int ti = tj > T0 ? tj : T0; ti <= ti_hi; ++ti) {
            const int t0 = ti * 16;
            const float* qp = W5 + (t0 + t.i) * ld + h * HD + t.kq * KS;
            const float* dop = qp + 3 * GW;
            f32x4 st = zero4(), dp = zero4();
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                st = mfma16(qp[s], kf[s], st);
                dp = mfma16(dop[s], vf[s], dp);
            }
            // st[r] = S[t0 + kq*4 + r][s0 + i]
            const float4 l4 = ld4(lse_s + h * sl_ld + t0 + t.kq * 4), d4 = ld4(delta_s + h * sl_ld + t0 + t.kq * 4);
            const float lse4[4] = {l4.x, l4.y, l4.z, l4.w}, del4[4] = {d4.x, d4.y, d4.z, d4.w};
            float p[4], ds[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int tr = t0 + t.kq * 4 + r;
                p[r] = DTQN_EXP2(st[r] - lse4[r] * LOG2E);
                if (srow > tr || tr >= n) p[r] = 0.f;
                float dpm = dp[r];
                if (dr.thresh != 0u) {
                    const bool keep = drop_keep(dr, DROP_ATTN, layer, drop_attn_idx(head0 + h, tr, srow));
                    dpm = keep ? dp[r] * dr.scale : 0.f;
                    ds[r] = p[r] * (dpm - del4[r]);
                    p[r] = keep ? p[r] * dr.scale : 0.f;          // dV takes the dropped probabilities
                } else {
                    ds[r] = p[r] * (dpm - del4[r]);
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float* qrow = W5 + (t0 + t.kq * 4 + r) * ld + h * HD;
                const float* dorow = qrow + 3 * GW;
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    const int c = ct * 16 + t.i;
                    const int cc = c < HD ? c : 0;
                    acck[ct][r & 1] = mfma16(qrow[cc], ds[r], acck[ct][r & 1]);
                    accv[ct][r & 1] = mfma16(dorow[cc], p[r], accv[ct][r & 1]);
                }
            }
        }
        // NOTE: other items of this pass read only q / do / lse / delta, never k or v of another tile
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const int c = ct * 16 + t.kq * 4;
            if (c < HD) {
                st4(kout + c, make_float4((acck[ct][0][0] + acck[ct][1][0]) * scale, (acck[ct][0][1] + acck[ct][1][1]) * scale,
                                          (acck[ct][0][2] + acck[ct][1][2]) * scale, (acck[ct][0][3] + acck[ct][1][3]) * scale));
                st4(vout + c, make_float4(accv[ct][0][0] + accv[ct][1][0], accv[ct][0][1] + accv[ct][1][1],
                                          accv[ct][0][2] + accv[ct][1][2], accv[ct][0][3] + accv[ct][1][3]));
            }
        }
    }
}

// VALU version of the same two passes (narrow heads, see attention_forward_valu): one lane per (row, head) item.
// Attention backward for one head group resident in W5 = [q | k | v | do | dq] (GW columns each).
//   delta_s[h][t] = dO . o was produced by the dO GEMM's epilogue; lse_s[h][t] is staged by the caller.
//   pass 1 (item = query row t, head): dS = P*(dP - delta); dq = scale * dS k
//   pass 2 (item = key row s, head):   dk = scale * dS^T q ; dv = P^T do      (in place over k, v)
template <int HD, int NW>
__device__ __forceinline__ void attention_backward_group_valu(float* W5, int ld, int GW, int LP, int n,
                                                         const float* delta_s, const float* lse_s, const Thr& t,
                                                         float* dq_base = nullptr, int dq_ld = 0, int row0 = 0, int sl_ld = 0,
                                                         const Drop& dr = Drop{0u, 1.0f, 0u, 0u, 0u}, int layer = 0, int head0 = 0,
                                                         float hd_eff = (float)HD) {
    // dq goes to W5's fifth tile by default, or to dq_base (row stride dq_ld; may be global memory) when the
    // caller cannot afford a fifth LDS tile.
    // Row slices: queries [row0, row0 + LP) against keys [0, row0 + LP); all tile rows are GLOBAL rows, delta_s /
    // lse_s are [head][sl_ld] indexed by global row.  dk / dv then hold only this slice's queries' contribution.
    if (dq_base == nullptr) { dq_base = W5 + 4 * GW; dq_ld = ld; }
    if (sl_ld == 0) sl_ld = LP;
    const int HG = GW / HD;
    const float scale = 1.0f / sqrtf(hd_eff);
    const int q_hi = n < row0 + LP ? n : row0 + LP;          // queries of this slice: [row0, q_hi)
    int nblocks = (LP * HG + 63) / 64;
    bool one_round = nblocks <= NW && (64 % HG) == 0;
    int nlive = ((q_hi > row0 ? q_hi - row0 : 0) * HG + 63) / 64;
    for (int it0 = t.tid; it0 < (one_round ? NW * 64 : LP * HG); it0 += NW * 64) {
        int item = it0;
        if (one_round) {
            const int blk = balanced_block<NW>(t.wave, nblocks, nlive, false);
            if (blk < 0) continue;
            item = blk * 64 + t.lane;
            if (item >= LP * HG) continue;
        }
        const int rl = item / HG, hl = item - rl * HG;
        const int row = row0 + rl;
        float* dqp = dq_base + (size_t)row * dq_ld + hl * HD;
        float dq[HD];
#pragma unroll
        for (int c = 0; c < HD; ++c) dq[c] = 0.f;
        if (row < n) {
            const float* qp = W5 + row * ld + hl * HD;
            const float* dop = W5 + row * ld + 3 * GW + hl * HD;
            float q[HD], dO[HD];
#pragma unroll
            for (int c = 0; c < HD; c += 4) {
                const float4 x = ld4(qp + c), g = ld4(dop + c);
                q[c] = x.x * scale; q[c + 1] = x.y * scale; q[c + 2] = x.z * scale; q[c + 3] = x.w * scale;
                dO[c] = g.x; dO[c + 1] = g.y; dO[c + 2] = g.z; dO[c + 3] = g.w;
            }
            const float delta = delta_s[hl * sl_ld + row], lse = lse_s[hl * sl_ld + row];
            const float* kbase = W5 + GW + hl * HD;
            for (int s = 0; s <= row; ++s) {
                const float* kp = kbase + s * ld;
                const float* vp = kp + GW;
                float sc = 0.f, dp = 0.f;
                float kk[HD];
#pragma unroll
                for (int c = 0; c < HD; c += 4) {
                    const float4 k = ld4(kp + c), v = ld4(vp + c);
                    kk[c] = k.x; kk[c + 1] = k.y; kk[c + 2] = k.z; kk[c + 3] = k.w;
                    sc = fmaf(q[c], k.x, sc); sc = fmaf(q[c + 1], k.y, sc); sc = fmaf(q[c + 2], k.z, sc); sc = fmaf(q[c + 3], k.w, sc);
                    dp = fmaf(dO[c], v.x, dp); dp = fmaf(dO[c + 1], v.y, dp); dp = fmaf(dO[c + 2], v.z, dp); dp = fmaf(dO[c + 3], v.w, dp);
                }
                if (dr.thresh != 0u) dp = drop_keep(dr, DROP_ATTN, layer, drop_attn_idx(head0 + hl, row, s)) ? dp * dr.scale : 0.f;
                const float ds = __expf(sc - lse) * (dp - delta);
#pragma unroll
                for (int c = 0; c < HD; ++c) dq[c] = fmaf(ds, kk[c], dq[c]);
            }
        }
#pragma unroll
        for (int c = 0; c < HD; c += 4)
            st4(dqp + c, make_float4(dq[c] * scale, dq[c + 1] * scale, dq[c + 2] * scale, dq[c + 3] * scale));
    }
    __syncthreads();
    const int KR = row0 + LP;                                // key rows of pass 2
    nblocks = (KR * HG + 63) / 64;
    one_round = nblocks <= NW && (64 % HG) == 0;
    nlive = ((q_hi < KR ? q_hi : KR) * HG + 63) / 64;
    for (int it0 = t.tid; it0 < (one_round ? NW * 64 : KR * HG); it0 += NW * 64) {
        int item = it0;
        if (one_round) {
            const int blk = balanced_block<NW>(t.wave, nblocks, nlive, true);
            if (blk < 0) continue;
            item = blk * 64 + t.lane;
            if (item >= KR * HG) continue;
        }
        const int srow = item / HG, hl = item - srow * HG;
        float* kp = W5 + srow * ld + GW + hl * HD;
        float* vp = kp + GW;
        float dk[HD], dv[HD];
#pragma unroll
        for (int c = 0; c < HD; ++c) dk[c] = dv[c] = 0.f;
        if (srow < q_hi) {
            float k[HD], v[HD];
#pragma unroll
            for (int c = 0; c < HD; c += 4) {
                const float4 x = ld4(kp + c), y = ld4(vp + c);
                k[c] = x.x; k[c + 1] = x.y; k[c + 2] = x.z; k[c + 3] = x.w;
                v[c] = y.x; v[c + 1] = y.y; v[c + 2] = y.z; v[c + 3] = y.w;
            }
            const int q_lo = srow > row0 ? srow : row0;
            for (int row = q_hi - 1; row >= q_lo; --row) {
                const float* qp = W5 + row * ld + hl * HD;
                const float* dop = qp + 3 * GW;
                float sc = 0.f, dp = 0.f;
                float qq[HD], dd[HD];
#pragma unroll
                for (int c = 0; c < HD; c += 4) {
                    const float4 x = ld4(qp + c), g = ld4(dop + c);
                    qq[c] = x.x * scale; qq[c + 1] = x.y * scale; qq[c + 2] = x.z * scale; qq[c + 3] = x.w * scale;
                    dd[c] = g.x; dd[c + 1] = g.y; dd[c + 2] = g.z; dd[c + 3] = g.w;
                }
#pragma unroll
                for (int c = 0; c < HD; ++c) { sc = fmaf(qq[c], k[c], sc); dp = fmaf(dd[c], v[c], dp); }
                float p = __expf(sc - lse_s[hl * sl_ld + row]);
                float pv = p;                                   // what multiplied v in the forward
                if (dr.thresh != 0u) {
                    const bool keep = drop_keep(dr, DROP_ATTN, layer, drop_attn_idx(head0 + hl, row, srow));
                    dp = keep ? dp * dr.scale : 0.f;
                    pv = keep ? p * dr.scale : 0.f;
                }
                const float ds = p * (dp - delta_s[hl * sl_ld + row]);
#pragma unroll
                for (int c = 0; c < HD; ++c) { dk[c] = fmaf(ds, qq[c], dk[c]); dv[c] = fmaf(pv, dd[c], dv[c]); }
            }
        }
        // NOTE: other items of this pass read only q / do / lse / delta, never k or v of another row
#pragma unroll
        for (int c = 0; c < HD; c += 4) {
            st4(kp + c, make_float4(dk[c], dk[c + 1], dk[c + 2], dk[c + 3]));
            st4(vp + c, make_float4(dv[c], dv[c + 1], dv[c + 2], dv[c + 3]));
        }
    }
}

template <int HD, int NW, bool MFMA = (HD >= kAttnMfmaMinHeadDim)>
__device__ __forceinline__ void attention_backward_group(float* W5, int ld, int GW, int LP, int n,
                                                         const float* delta_s, const float* lse_s, const Thr& t,
                                                         float* dq_base = nullptr, int dq_ld = 0, int row0 = 0, int sl_ld = 0,
                                                         const Drop& dr = Drop{0u, 1.0f, 0u, 0u, 0u}, int layer = 0, int head0 = 0,
                                                         float hd_eff = (float)HD) {
    if constexpr (MFMA)
        attention_backward_group_mfma<HD, NW>(W5, ld, GW, LP, n, delta_s, lse_s, t, dq_base, dq_ld, row0, sl_ld, dr, layer, head0, false, 0,
                                              AttnNoBetween(), hd_eff);
    else attention_backward_group_valu<HD, NW>(W5, ld, GW, LP, n, delta_s, lse_s, t, dq_base, dq_ld, row0, sl_ld, dr, layer, head0, hd_eff);
}

// Double-DQN target, MSE and dL/dQ of ONE sequence, executed by one wave (dtqn/agents/dtqn.py:219-253).
//   q0 / q1 / q2: Q_pol(o), Q_pol(o'), Q_tgt(o') rows of this sequence, row stride AP
//   dq: [rows][AP] output (LDS or global), must be zero-filled by the caller beforehand
//   sp: 8 floats of per-sequence statistics partials
__device__ __forceinline__ void td_loss_wave(const float* q0, const float* q1, const float* q2, int AP, int A, int L, int LPB,
                                             int history, float gamma, float inv_count, const uint8_t* act, const float* rew,
                                             const uint8_t* don, float* dq, float* sp, int lane) {
    float sq = 0.f, mnq = INFINITY, mxq = -INFINITY, sy = 0.f, mny = INFINITY, mxy = -INFINITY, se = 0.f;
    for (int r = lane; r < LPB; r += 64) {
        if (r < L && r >= L - history) {
            // one round trip: none of these addresses depends on a loaded value
            const int at = (int)act[r];
            const float rw = rew[r];
            const float dn = don[r] ? 1.f : 0.f;
            float q = 0.f, best = q1[r * AP], qt = q2[r * AP];
            for (int c = 0; c < A; ++c) {          // torch.argmax: first maximal index
                const float v0 = q0[r * AP + c], v1 = q1[r * AP + c], v2 = q2[r * AP + c];
                if (c == at) q = v0;
                if (c > 0 && v1 > best) { best = v1; qt = v2; }
            }
            const float y = rw + (1.f - dn) * (qt * gamma);
            const float diff = q - y;
            dq[r * AP + at] = 2.f * diff * inv_count;
            sq += q; mnq = fminf(mnq, q); mxq = fmaxf(mxq, q);
            sy += y; mny = fminf(mny, y); mxy = fmaxf(mxy, y);
            se = fmaf(diff, diff, se);
        }
    }
    sq = wave_sum(sq); sy = wave_sum(sy); se = wave_sum(se);
    mnq = wave_min(mnq); mxq = wave_max(mxq); mny = wave_min(mny); mxy = wave_max(mxy);
    if (lane == 0) {
        sp[0] = se; sp[1] = sq; sp[2] = mxq; sp[3] = mnq; sp[4] = sy; sp[5] = mxy; sp[6] = mny; sp[7] = 0.f;
    }
}

}  // namespace dtqn
