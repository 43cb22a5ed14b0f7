// DTQN forward for gfx950: embed -> +pos -> NL x (causal MHA, gate, LN, FFN, gate, LN) -> Q head,
// one workgroup per sequence, the whole [LP x D] context tile resident in LDS.
//
// Replaces DTQN.forward (dtqn/networks/dtqn.py:158-218), TransformerLayer.forward /
// TransformerIdentityLayer.forward (transformer.py:63-78,86-101), ObservationEmbeddingRepresentation
// (representations.py:17-23), and -- for the TD update -- the window gather of
// ReplayBuffer.sample (replay_buffer.py:160-167): the workgroup reads its (episode, start) window
// straight out of the device-resident replay arrays.
#pragma once
#include "dtqn_device.hpp"
#include "dtqn_gru.hpp"
#include "dtqn_wl.hpp"

#ifndef DTQN_SPLIT_ATTN_MFMA
#define DTQN_SPLIT_ATTN_MFMA 1
#endif

namespace dtqn {

struct FwdArgs {
    DtqnNet net;
    const float* theta_a;       // parameters for which = 0, 1
    const float* theta_b;       // parameters for which = 2 (target network)
    const float* obs;           // row (ep, r) at obs + ep*obs_ep_stride + r*O
    const uint8_t* actions;     // (ep, r) at actions + ep*act_ep_stride + r
    long long obs_ep_stride;
    long long act_ep_stride;
    const int32_t* ep_idx;      // nullptr: ep = sequence index, start = 0
    const int32_t* start;
    int n;                      // real sequence length (<= ctx_len)
    int batch;                  // sequences per `which`
    int nseq;                   // sequences of this launch (passes x batch, or the actors of a batched actor forward)
    int block0;                 // first workgroup of this forward inside its launch (> 0: the passes ride behind another kernel's workgroups,
                                // dtqn_backward.hip AHEAD)
    int pass0;                  // first pass of this launch: which = pass0 + sequence / batch (TD update: 0 policy(o), 1 policy(o'), 2 target(o'))
    int draw_step;              // >= 0: the window draw is keyed by this optimizer step instead of step_counter[1] (a pass launched ahead
                                // of its update, dtqn_td_forward_part)
    float* q_out;               // which-major
    long long q_which_stride, q_seq_stride;
    int q_row_stride;
    float* q_last_host;         // optional, PINNED host memory [batch][num_actions]: Q of the last live row of every sequence (actor)
    const int32_t* last_rows;   // optional [batch]: live rows n_i per sequence (ragged prefixes in one launch): that row is n_i - 1; nullptr = n - 1
    float* act;                 // nullptr: inference; else activation records for which == 0
    float* xch;                 // row-split hand-over buffer / flags (RS == 2 only)
    int32_t* xflags;
    // in-kernel window draw (DtqnTd.sample_in_kernel): ep_len != nullptr
    const int32_t* ep_len;
    const int32_t* step_counter;
    int32_t* ep_out;
    int32_t* start_out;
    int s_n_valid, s_exclude;
    uint32_t s_seed;
    long long* prof;            // debug stage clock (see dtqn_debug_set_profile_buffer)
    // dropout (net.dropout > 0): passes with bit `which` set in drop_passes run in train mode; step from step_counter[1]
    // when that pointer is given (TD update), else drop_step (actor)
    uint32_t drop_thresh, drop_seed, drop_step;
    float drop_scale;
    int drop_passes;
};

__device__ __forceinline__ int lds_ldx(int D) { return D + 4; }
__device__ __forceinline__ int lds_ldw(int D) { return 3 * D + 4; }


// K | V hand-over between the RS row slices of a sequence (forward): slice s needs the keys / values of every row below its own.
// Sender s < RS - 1 publishes its [LP][2 D] K | V tile once (16-byte write-through stores) into segment s of the (sequence, layer)
// buffer and raises one flag per receiver above it; receiver r > 0 waits for the flags of all senders below it, pulls their
// segments into rows [s LP, (s + 1) LP) of its own LDS tile and lowers its flags again (the next launch starts from 0).  A middle
// slice sends first, then receives.  Senders always have the lower blockIdx of a pair, on the receiver's XCD (slice_block_map).
// Pair (s, r), s < r: flag r (r - 1) / 2 + s.
template <int NW, int RS>
__device__ __forceinline__ void kv_send(float* Ws, int ldw, int D, int LP, int slice, float* xb, int32_t* flags, const Thr& t) {
    const int cols = 2 * D, c4 = cols >> 2;
    if (slice < RS - 1) {
        const DtqnRsrc rs = DTQN_XCH_RSRC(xb + (size_t)slice * LP * cols, LP * cols * 4);
        const float* s = Ws + (size_t)slice * LP * ldw + D;
        for (int idx = t.tid; idx < LP * c4; idx += NW * 64) {
            const int r = idx / c4, c = (idx - r * c4) * 4;
            dtqn_xch_store4(rs, idx * 16, ld4(s + r * ldw + c));
        }
        DTQN_WAIT_VMEM();                              // every wave's stores are acknowledged at agent scope ...
        __syncthreads();                               // ... and every wave got here ...
        const int r = slice + 1 + t.tid;               // ... before the flags are raised
        if (t.tid < RS - 1 - slice) DTQN_AGENT_STORE(flags + r * (r - 1) / 2 + slice, (int32_t)1);
    }
}
template <int NW, int RS>
__device__ __forceinline__ void kv_recv(float* Ws, int ldw, int D, int LP, int slice, float* xb, int32_t* flags, const Thr& t) {
    const int cols = 2 * D, c4 = cols >> 2;
    if (slice > 0) {
        if (t.tid < slice)
            while (DTQN_AGENT_LOAD(flags + slice * (slice - 1) / 2 + t.tid) == 0) DTQN_SPIN_PAUSE();
        __syncthreads();
        const DtqnRsrc rs = DTQN_XCH_RSRC(xb, slice * LP * cols * 4);
        for (int idx = t.tid; idx < slice * LP * c4; idx += NW * 64) {
            const int r = idx / c4, c = (idx - r * c4) * 4;         // r: global row below this slice
            st4(Ws + (size_t)r * ldw + D + c, dtqn_xch_load4(rs, idx * 16));
        }
        __syncthreads();
        if (t.tid < slice) DTQN_AGENT_STORE(flags + slice * (slice - 1) / 2 + t.tid, (int32_t)0);
    }
}
template <int NW, int RS>
__device__ __forceinline__ void kv_handover(float* Ws, int ldw, int D, int LP, int slice, float* xb, int32_t* flags, const Thr& t) {
    kv_send<NW, RS>(Ws, ldw, D, LP, slice, xb, flags, t);
    kv_recv<NW, RS>(Ws, ldw, D, LP, slice, xb, flags, t);
}

// floats / flag words of one (sequence, layer) hand-over record for RS slices of LP rows
__host__ __device__ constexpr int kv_xch_floats(int RS, int LP, int D) { return (RS - 1) * LP * 2 * D; }
__host__ __device__ constexpr int kv_xch_flags(int RS) { return RS * (RS - 1) / 2; }

// RS = row slices per sequence.  RS == 1: the workgroup owns the whole sequence (LP = its padded length).
// RS == 2 (latency mode, dtqn_td_row_split): the workgroup owns rows [R0, R0 + LP) of the sequence, LP = half the
// padded length; every stage is row-local except attention, whose K | V of the rows below R0 come from the partner
// workgroup (slice 0 -> slice 1 hand-over through a.xch).  Record tensors keep their full-sequence layout: a slice
// addresses them at row R0.
// TRAIN: this workgroup saves the activation record (policy(o) pass of a TD update).  A template parameter, not a
// pointer test: every record store is then unconditional code, and the compiler can count the stores that sit between a
// prefetched weight fragment and its s_waitcnt instead of falling back to vmcnt(0) (measured: -6 % on the inference pass).
// DROP: the pass may run with dropout (net.dropout > 0; a compile-time switch: with DROP = false the keep-mask code folds away
// and the default path carries none of it)
template <int D, int MT, int HD, int NW, bool GRU, int RS, bool TRAIN, bool DROP>
__device__ __forceinline__ void forward_body(const FwdArgs& a) {
    static_assert(RS == 1 || RS == 2, "one or two row slices");
    constexpr int NT = NW * 64;                    // threads per workgroup
    constexpr int LP = MT * 16;                    // rows this workgroup owns
    constexpr int LPF = LP * RS;                   // padded rows of the whole sequence (= net.lp)
    constexpr int LDX = D + 4, LDW = 3 * D + 4;
    constexpr int NC = D >= 128 ? D : 2 * D;       // FFN hidden columns per pass (D = 128: two 2D-wide FFN-2 fragments alone would be 128 VGPRs)
    const DtqnNet& net = a.net;
    const Thr t = make_thr();
    int seq, slice;                                // slice 0 (the producer) first, a sequence's slices on one XCD (slice_block_map)
    slice_block_map((int)blockIdx.x - a.block0, a.nseq, RS, seq, slice);
    const int R0 = slice * LP;
    const int which = a.pass0 + seq / a.batch;
    const int b = seq - (which - a.pass0) * a.batch;
    Drop dr = drop_off();
    if constexpr (DROP) {
        if (a.drop_thresh != 0u && ((a.drop_passes >> which) & 1))
            dr = Drop{a.drop_thresh, a.drop_scale, a.drop_seed, a.step_counter != nullptr ? (uint32_t)a.step_counter[1] : a.drop_step,
                      ((uint32_t)which << 20) | (uint32_t)b};
    }
    const float* __restrict__ theta = which == 2 ? a.theta_b : a.theta_a;
    const int nfull = a.n, H = net.num_heads, O = net.obs_dim, adim = net.action_dim, A = net.num_actions;
    const int n = nfull - R0;                      // live rows of this slice (may be <= 0: all padding)
    // history_len == 1: no roll, row 0 keeps its own action embedding (dtqn.py:187-191) -- per SEQUENCE when ragged prefixes share a launch
    const bool single = (a.last_rows != nullptr ? a.last_rows[seq] : nfull) == 1;
    const bool ident = RS > 1 ? false : net.identity != 0;     // row slices are dispatched for post-LN nets only: folds away
    constexpr bool gru = GRU;                      // gate type is a template parameter: the ResGate build carries no GRU code
    float* rec = TRAIN ? a.act + (size_t)b * net.act_stride : nullptr;
    // a [LPF][w] record tensor at the first row of this slice
    auto rf = [&](float* base, int off, int w) -> float* { return TRAIN ? base + off + (size_t)R0 * w : nullptr; };
    // ReLU ballot record (64-bit word per accumulator register, dtqn_device.hpp ballot_store) at the slice's first row tile
    auto mf = [&](float* base, int off, int ctiles) -> float* { return TRAIN ? base + off + (size_t)(R0 / 16) * ctiles * 8 : nullptr; };

    float* Xs = reinterpret_cast<float*>(dtqn_smem);   // residual stream            [LP][LDX]
    float* Ws = Xs + LP * LDX;                         // q|k|v (GLOBAL rows), FFN hidden, staging [LPF][LDW]
    float* Us = Ws + LPF * LDW;                        // identity only: LN output   [LP][LDX]
    float* AW = Ws + R0 * LDW;                         // this slice's rows of the attention tile

    int ps = 0;
    DTQN_PROF(a.prof, ps++);
    // ---------------- window gather + embedding ----------------
    int ep, st;
    if (a.ep_len != nullptr) {
        // every workgroup of sequence b (three passes, row slices) evaluates the same counter-based draw; one of them
        // leaves it for the backward kernel
        replay_draw(a.ep_len, a.s_n_valid, a.s_exclude, net.ctx_len, a.s_seed, a.draw_step >= 0 ? (uint32_t)a.draw_step : (uint32_t)a.step_counter[1], b, ep, st);
        if (which == 0 && slice == 0 && t.tid == 0) { a.ep_out[b] = ep; a.start_out[b] = st; }
    } else {
        ep = a.ep_idx != nullptr ? a.ep_idx[b] : b;
        st = a.start != nullptr ? a.start[b] : 0;
    }
    const int row0 = st + (which > 0 ? 1 : 0) + R0;
    const float* obs_rows = a.obs + (size_t)ep * a.obs_ep_stride + (size_t)row0 * O;
    const uint8_t* act_rows = a.actions != nullptr ? a.actions + (size_t)ep * a.act_ep_stride + row0 : nullptr;
    const int KE = net.ke, KEP = net.kep;
    const float* __restrict__ We = theta + net.off_obs_w;
    const float* __restrict__ be = theta + net.off_obs_b;
    const float* __restrict__ pos = theta + net.off_pos + (size_t)R0 * D;
    if (!net.discrete && KE <= 8) {
        // continuous observations: every token row is a handful of floats read straight from the replay
        // window; a single pass, no LDS staging, no barrier
        for (int idx = t.tid; idx < LP * D; idx += NT) {
            const int r = idx / D, d = idx - r * D;
            float v = 0.f;
            if (r < n) {
                if (d < adim) {
                    if (single) v = theta[net.off_act_emb + (int)act_rows[0] * adim + d];
                    else if (R0 + r > 0) v = theta[net.off_act_emb + (int)act_rows[r - 1] * adim + d];
                } else {
                    const float* w = We + (size_t)(d - adim) * KE;
                    const float* e = obs_rows + (size_t)r * O;
                    float acc = be[d - adim];
                    for (int k = 0; k < KE; ++k) acc = fmaf(e[k], w[k], acc);
                    v = acc;
                }
                v += pos[r * D + d];
                v = drop_apply(dr, DROP_EMB, 0, (uint32_t)((R0 + r) * D + d), v);
            }
            Xs[r * LDX + d] = v;
            if (TRAIN) rf(rec, net.ao_x0, D)[idx] = v;
        }
        if (TRAIN)
            for (int idx = t.tid; idx < LP * KEP; idx += NT) {
                const int r = idx / KEP, k = idx - r * KEP;
                rf(rec, net.ao_ein, KEP)[idx] = (r < n && k < KE) ? obs_rows[(size_t)r * O + k] : 0.f;
            }
    } else {
        float* ein = Ws;                               // [LP][KEP] embedding-linear input (wavefront-level gather)
        for (int idx = t.tid; idx < LP * KEP; idx += NT) {
            const int r = idx / KEP, k = idx - r * KEP;
            float v = 0.f;
            if (r < n && k < KE) {
                if (net.discrete) {
                    const int j = k / net.embed_per_obs, c = k - j * net.embed_per_obs;
                    int tok = (int)obs_rows[(size_t)r * O + j];
                    tok = tok < 0 ? 0 : (tok >= net.vocab ? net.vocab - 1 : tok);
                    v = theta[net.off_obs_tab + tok * net.embed_per_obs + c];
                } else {
                    v = obs_rows[(size_t)r * O + k];
                }
            }
            ein[idx] = v;
            if (TRAIN) rf(rec, net.ao_ein, KEP)[idx] = v;
        }
        __syncthreads();
        for (int idx = t.tid; idx < LP * D; idx += NT) {
            const int r = idx / D, d = idx - r * D;
            float v = 0.f;
            if (r < n) {
                if (d < adim) {
                    // previous-action embedding rolled right by one, row 0 zeroed unless n == 1 (dtqn.py:184-192)
                    if (single) v = theta[net.off_act_emb + (int)act_rows[0] * adim + d];
                    else if (R0 + r > 0) v = theta[net.off_act_emb + (int)act_rows[r - 1] * adim + d];
                } else {
                    const float* w = We + (size_t)(d - adim) * KE;
                    const float* e = ein + r * KEP;
                    float acc = be[d - adim];
                    for (int k = 0; k < KE; ++k) acc = fmaf(e[k], w[k], acc);
                    v = acc;
                }
                v += pos[r * D + d];
                v = drop_apply(dr, DROP_EMB, 0, (uint32_t)((R0 + r) * D + d), v);
            }
            Xs[r * LDX + d] = v;
            if (TRAIN) rf(rec, net.ao_x0, D)[idx] = v;
        }
    }
    DTQN_PROF(a.prof, ps++);   // embed done; Xs is published by the barrier that opens layer 0

    // ---------------- transformer layers ----------------
    // Stage discipline (training pass):
    //  * every GEMM stage fetches the weight fragment of its first work item BEFORE the barrier that
    //    publishes its input and RETIRES it right after the barrier, before any store is issued;
    //  * epilogues write LDS only; what the backward needs is copied LDS -> global as coalesced 16 B/lane
    //    stores at the START of the next stage (or straight from the LayerNorm registers), so store
    //    acknowledgements overlap that stage's MFMAs instead of sitting in front of its first wait;
    //  * ReLU patterns are saved as wave ballots (64 bits per accumulator register), not as tensors.
    constexpr int MG2 = pick_mg(D / 16, MT, NW);
    using Own = Owned<D, MT, MG2, NW>;                 // fixed ownership of the FFN-2 output tile
    for (int l = 0; l < net.num_layers; ++l) {
        const float* __restrict__ th = layer_theta(net, theta, l);
        float* lrec = TRAIN ? rec + net.ao_layer0 + (size_t)l * net.act_layer_stride : nullptr;
        const float* src = Xs;
        StageXwT<D, MT, pick_mg(3 * D / 16, MT, NW), NW, 3 * D / 16> g_qkv;
        g_qkv.prefetch(th + net.lo_in_w, D, t, th + net.lo_in_b);
        __syncthreads();                               // residual stream of the previous stage visible
        if (ident) {   // x_norm1 = LN1(x)  (transformer.py:87)
            layernorm_rows<D, NW, LP, TRAIN>(Xs, Us, LDX, LP, th + net.lo_ln1_w, th + net.lo_ln1_b, rf(lrec, net.al_st1, 2), t,
                                  nullptr, rf(lrec, net.al_u1, D));
            __syncthreads();
            src = Us;
        }
        g_qkv.retire();
        if (TRAIN && !ident) tile_store<NW>(src, LDX, rf(lrec, net.al_u1, D), LP, D, t);
        // packed in-projection: qkv = u W_in^T + b_in
        {
            g_qkv.run(src, LDX, t, [&](int r, int c, float v) { AW[r * LDW + c] = v; });   // bias added by the stage
        }
        StageXwT<D, MT, pick_mg(D / 16, MT, NW), NW, D / 16> g_out;
        g_out.prefetch(th + net.lo_out_w, D, t, th + net.lo_out_b);       // in flight during attention
        __syncthreads();
        DTQN_PROF(a.prof, ps++);   // qkv done
        if (TRAIN) {                         // q|k|v -> record before attention overwrites q
            tile_store<NW>(AW, LDW, rf(lrec, net.al_qkv, 3 * D), LP, 3 * D, t);
            __syncthreads();
        }
        if (RS == 2)                                   // K | V of the lower rows: slice 0 -> slice 1 (16-byte write-through stores)
            kv_handover<NW, RS>(Ws, LDW, D, LP, slice, a.xch + ((size_t)seq * net.num_layers + l) * kv_xch_floats(RS, LP, D),
                                a.xflags + ((size_t)seq * net.num_layers + l) * kv_xch_flags(RS), t);
        attention_forward<HD, NW, (HD >= kAttnMfmaMinHeadDim) || (RS == 2 && DTQN_SPLIT_ATTN_MFMA)>(Ws, LDW, D, H, LP, nfull, TRAIN ? lrec + net.al_lse : nullptr, t, R0, LPF, dr, l);
        __syncthreads();
        DTQN_PROF(a.prof, ps++);   // attention done
        g_out.retire();
        if (TRAIN) tile_store<NW>(AW, LDW, rf(lrec, net.al_o, D), LP, D, t);
        // out-projection, ReLU, residual gate:  x <- x + relu(o W_o^T + b_o)   (transformer.py:72 / :96)
        {
            float* m_g = mf(lrec, net.al_m1, D / 16);
            g_out.run(AW, LDW, t, [&](int r, int c, float v) {
                const float y = fmaxf(v, 0.f);
                if (TRAIN) ballot_store(m_g, D / 16, r, c, y > 0.f, t.lane);
                if (gru) Ws[r * LDW + D + c] = y;          // y tile for the GRU gate (k columns are free now)
                else Xs[r * LDX + c] += y;                 // ResGate: x + y  (gates.py:40-41)
            });
        }
        if (gru) {                                         // x <- GRUGate(x, y)  (gates.py:26-31)
            __syncthreads();
            gru_gate_forward<D, MT, NW>(Xs, LDX, Ws, LDW, theta + net.off_gate_attn, net, TRAIN ? lrec + net.al_gate1 : nullptr, t, LPF, R0);
        }
        const float* __restrict__ W1 = th + net.lo_f1_w;
        const float* __restrict__ b1 = th + net.lo_f1_b;
        const float* __restrict__ W2 = th + net.lo_f2_w;
        StageXwT<D, MT, pick_mg(NC / 16, MT, NW), NW, NC / 16> g_f1;
        g_f1.prefetch(W1, D, t, b1);                   // in flight during LN1
        __syncthreads();
        DTQN_PROF(a.prof, ps++);   // out-proj done
        if (!ident) {  // x = LN1(x); s1 (input) and u2 (output) go to the record from the LN registers
            layernorm_rows<D, NW, LP, TRAIN>(Xs, Xs, LDX, LP, th + net.lo_ln1_w, th + net.lo_ln1_b, rf(lrec, net.al_st1, 2), t,
                                  rf(lrec, net.al_s1, D), rf(lrec, net.al_u2, D));
            src = Xs;
        } else {       // x_norm2 = LN2(x)
            layernorm_rows<D, NW, LP, TRAIN>(Xs, Us, LDX, LP, th + net.lo_ln2_w, th + net.lo_ln2_b, rf(lrec, net.al_st2, 2), t,
                                  rf(lrec, net.al_s1, D), rf(lrec, net.al_u2, D));
            src = Us;
        }
        __syncthreads();
        DTQN_PROF(a.prof, ps++);   // LN1 done
        // FFN D -> 4D -> D in hidden-column passes of NC; the second GEMM accumulates in registers
        {
            f32x4 facc[Own::PER_WAVE][MG2];
#pragma unroll
            for (int q = 0; q < Own::PER_WAVE; ++q)
#pragma unroll
                for (int m = 0; m < MG2; ++m) facc[q][m] = zero4();
            float b2v[Own::PER_WAVE];                  // FFN-2 bias of the owned columns: loaded now, used after both chunks
#pragma unroll
            for (int q = 0; q < Own::PER_WAVE; ++q)
                b2v[q] = Own::valid_fast(t.wave, q) ? (th + net.lo_f2_b)[Own::nt(t.wave, q) * 16 + t.i] : 0.f;
            float4 w2f[2][NC / 16];                    // this wave's FFN-2 weight fragments
            float* mh_g = mf(lrec, net.al_mh, 4 * D / 16);
#pragma clang loop unroll_count(D >= 128 ? 1 : 4)
            for (int c0 = 0; c0 < 4 * D; c0 += NC) {       // unrolled (D <= 64): exact s_waitcnt counts across the chunk boundary
                // the second GEMM's first weight fragment does not depend on the hidden: in flight during the first GEMM
                if (Own::valid_fast(t.wave, 0))
                    frag_xwT_fetch<NC>(w2f[0], W2 + (size_t)(Own::nt(t.wave, 0) * 16 + t.i) * 4 * D + c0, t);
                g_f1.retire();
                g_f1.run(src, LDX, t, [&](int r, int c, float v) {
                    const float hv = fmaxf(v, 0.f);
                    if (TRAIN) ballot_store(mh_g, 4 * D / 16, r, c0 + c, hv > 0.f, t.lane);
                    Ws[r * LDW + c] = hv;
                });
                // ... and the next chunk's first W1 fragment is in flight during the second GEMM
                if (c0 + NC < 4 * D) g_f1.prefetch(W1 + (size_t)(c0 + NC) * D, D, t, b1 + c0 + NC);
                __syncthreads();                       // hidden chunk visible
                if (Own::valid_fast(t.wave, 0)) {
#pragma unroll
                    for (int s = 0; s < NC / 16; ++s) retire4(w2f[0][s]);
                }
                if (TRAIN) tile_store<NW>(Ws, LDW, rf(lrec, net.al_h, 4 * D) + c0, LP, NC, t, 4 * D);
#pragma unroll
                for (int q = 0; q < Own::PER_WAVE; ++q) {
                    if (q + 1 < Own::PER_WAVE && Own::valid_fast(t.wave, q + 1))
                        frag_xwT_fetch<NC>(w2f[(q + 1) & 1], W2 + (size_t)(Own::nt(t.wave, q + 1) * 16 + t.i) * 4 * D + c0, t);
                    if (Own::valid_fast(t.wave, q))
                        frag_xwT_mma<NC, MG2>(Ws + Own::mg(t.wave, q) * MG2 * 16 * LDW, LDW, w2f[q & 1], t, facc[q]);
                }
                __syncthreads();                       // everyone is done reading this chunk of the hidden
            }
            float* m_g = mf(lrec, net.al_m2, D / 16);
#pragma unroll
            for (int q = 0; q < Own::PER_WAVE; ++q) {
                if (Own::valid_fast(t.wave, q)) {
                    const int c = Own::nt(t.wave, q) * 16 + t.i;
#pragma unroll
                    for (int m = 0; m < MG2; ++m)
#pragma unroll
                        for (int r4 = 0; r4 < 4; ++r4) {
                            const int r = (Own::mg(t.wave, q) * MG2 + m) * 16 + t.kq * 4 + r4;
                            const float y = fmaxf(drop_apply(dr, DROP_FFN, l, (uint32_t)((R0 + r) * D + c), facc[q][m][r4] + b2v[q]), 0.f);
                            if (TRAIN) ballot_store(m_g, D / 16, r, c, y > 0.f, t.lane);
                            if (gru) Ws[r * LDW + D + c] = y;
                            else Xs[r * LDX + c] += y;
                        }
                }
            }
        }
        if (gru) {
            __syncthreads();
            gru_gate_forward<D, MT, NW>(Xs, LDX, Ws, LDW, theta + net.off_gate_mlp, net, TRAIN ? lrec + net.al_gate2 : nullptr, t, LPF, R0);
        }
        DTQN_PROF(a.prof, ps++);   // FFN done
        __syncthreads();
        if (!ident) {  // x = LN2(x); s2 from the LN registers
            layernorm_rows<D, NW, LP, TRAIN>(Xs, Xs, LDX, LP, th + net.lo_ln2_w, th + net.lo_ln2_b, rf(lrec, net.al_st2, 2), t,
                                  rf(lrec, net.al_s2, D), nullptr);
        } else if (TRAIN) {
            tile_store<NW>(Xs, LDX, rf(lrec, net.al_s2, D), LP, D, t);
        }
        // the residual stream is published by the barrier that opens the next layer / the head
    }

    // ---------------- Q head: Linear(D,D) -> ReLU -> Linear(D,A)  (dtqn.py:149-153,216) ----------------
    StageXwT<D, MT, pick_mg(D / 16, MT, NW), NW, D / 16> g_head;
    g_head.prefetch(theta + net.off_head1_w, D, t, theta + net.off_head1_b);
    __syncthreads();
    DTQN_PROF(a.prof, ps++);       // layers done
    g_head.retire();
    if (TRAIN) tile_store<NW>(Xs, LDX, rf(rec, net.ao_xf, D), LP, D, t);
    {
        g_head.run(Xs, LDX, t, [&](int r, int c, float v) { Ws[r * LDW + c] = fmaxf(v, 0.f); });
    }
    __syncthreads();
    if (TRAIN) tile_store<NW>(Ws, LDW, rf(rec, net.ao_hh, D), LP, D, t);
    {
        const float* __restrict__ W2 = theta + net.off_head2_w;
        const float* __restrict__ b2 = theta + net.off_head2_b;
        float* q = a.q_out + (size_t)which * a.q_which_stride + (size_t)b * a.q_seq_stride + (size_t)R0 * a.q_row_stride;
        for (int idx = t.tid; idx < (n < LP ? n : LP) * A; idx += NT) {
            const int r = idx / A, ac = idx - r * A;
            const float* hrow = Ws + r * LDW;
            const float* w = W2 + (size_t)ac * D;
            float acc = b2[ac];
#pragma unroll 8
            for (int k = 0; k < D; k += 4) {
                const float4 hv = ld4(hrow + k), wv = ld4(w + k);
                acc = fmaf(hv.x, wv.x, acc); acc = fmaf(hv.y, wv.y, acc); acc = fmaf(hv.z, wv.z, acc); acc = fmaf(hv.w, wv.w, acc);
            }
            q[r * a.q_row_stride + ac] = acc;
            // the actor only needs Q[:, -1] (dtqn.py:103): written straight into host memory, no copy enqueued behind the kernel
            if (a.q_last_host != nullptr && R0 + r == (a.last_rows != nullptr ? a.last_rows[seq] - 1 : nfull - 1)) a.q_last_host[seq * A + ac] = acc;
        }
    }
    DTQN_PROF(a.prof, ps++);       // end
}


// ---------------------------------------------------------------------------------------------------------------------
// Weights-through-LDS forward (dtqn_wl.hpp): residual gate, D <= 64.  Same stages, records and results as forward_body;
// every GEMM stage reads its weight tile (and bias / LayerNorm vectors) from LDS, where the tile was parked one stage
// earlier:
//   arena region A: W_in rows [0, 2D)  ->  FFN-1 chunk 0  ->  FFN-1 chunk 1  ->  next layer's W_in (or the head's W_1)
//   arena region B: W_in rows [2D, 3D) | W_out  ->  FFN-2 chunk 0  ->  FFN-2 chunk 1  ->  next layer's W_in
// A tile is loaded global -> registers at the start of the stage BEFORE the one that frees its region (so the L2 round
// trip overlaps that stage's MFMAs), written to LDS right after the barrier that frees the region, and published by the
// next barrier.  Per-layer vectors (13 D floats) sit in one of two parameter blocks, alternating by layer parity.
// ---------------------------------------------------------------------------------------------------------------------
// PAD: width-padded network (DtqnNet.d_real > 0): LayerNorm statistics over the d_real real columns, softmax scale of the real head width
template <int D, int MT, int HD, int NW, int RS, bool TRAIN, bool PAD = false>
__device__ __forceinline__ void forward_body_wl(const FwdArgs& a) {
    static_assert(RS == 1 || RS == 2 || RS == 4, "one, two or four row slices");
    static_assert(D <= 64, "the weight arena is sized for D <= 64");
    constexpr int NT = NW * 64;
    constexpr int LP = MT * 16;
    constexpr int LPF = LP * RS;
    constexpr int LDX = D + 4, LDW = 3 * D + 4;
    constexpr int NC = 2 * D;                      // FFN hidden columns per pass (two passes)
    constexpr int LWD = D + 4, LWC = NC + 4;       // leading dims of [.][D] and [D][NC] weight tiles in the arena
    constexpr int OFF_B = 2 * D * LWD, OFF_WO = 3 * D * LWD;
    // offsets inside a layer's parameter block (dtqn_layout.cpp: ln1 w,b | ln2 w,b | b_in | b_out | b_1 | b_2)
    constexpr int P_LN1W = 0, P_LN1B = D, P_LN2W = 2 * D, P_LN2B = 3 * D, P_INB = 4 * D, P_OUTB = 7 * D, P_F1B = 8 * D, P_F2B = 12 * D;
    constexpr int PSN = wl_small_floats(D);
    static_assert(PSN / 4 <= NT, "one float4 of the parameter block per thread");
    const DtqnNet& net = a.net;
    const Thr t = make_thr();
    int seq, slice;                                // slice 0 (the producer) first, a sequence's slices on one XCD (slice_block_map)
    slice_block_map((int)blockIdx.x - a.block0, a.nseq, RS, seq, slice);
    const int R0 = slice * LP;
    const int which = a.pass0 + seq / a.batch;
    const int b = seq - (which - a.pass0) * a.batch;
    const Drop dr = drop_off();       // dropout > 0 is served by the register-direct stages (the host never picks this kernel then)
    const float* __restrict__ theta = which == 2 ? a.theta_b : a.theta_a;
    const int nfull = a.n, H = net.num_heads, O = net.obs_dim, adim = net.action_dim, A = net.num_actions;
    const int n = nfull - R0;
    const bool single = (a.last_rows != nullptr ? a.last_rows[seq] : nfull) == 1;      // see forward_body
    const int dreal = PAD ? net.d_real : D;          // LayerNorm width
    const bool ident = RS > 1 ? false : net.identity != 0;
    float* rec = TRAIN ? a.act + (size_t)b * net.act_stride : nullptr;
    auto rf = [&](float* base, int off, int w) -> float* { return TRAIN ? base + off + (size_t)R0 * w : nullptr; };
    auto mf = [&](float* base, int off, int ctiles) -> float* { return TRAIN ? base + off + (size_t)(R0 / 16) * ctiles * 8 : nullptr; };

    float* Xs = reinterpret_cast<float*>(dtqn_smem);   // residual stream            [LP][LDX]
    float* Ws = Xs + LP * LDX;                         // q|k|v (GLOBAL rows), FFN hidden, staging [LPF][LDW]
    float* Us = Ws + LPF * LDW;                        // identity only: LN output   [LP][LDX]
    float* Ps = Us + (net.identity != 0 ? LP * LDX : 0);   // parameter blocks       [2][13 D]
    float* Ar = Ps + 2 * PSN;                          // weight arena               [4 D (D + 4)]
    float* AW = Ws + R0 * LDW;

    int ps = 0;
    DTQN_PROF(a.prof, ps++);
    // layer 0's W_in and parameter block go in flight before the window gather
    TileRegs<NW, 3 * D, D> tw_in;
    TileRegs<NW, D, D> tw_hd;                          // the head's first matrix takes W_in's place after the last layer
    float4 ps_reg = make_float4(0.f, 0.f, 0.f, 0.f);
    {
        const float* __restrict__ th0 = layer_theta(net, theta, 0);
        tw_in.load(th0 + net.lo_in_w, D, t);
        if (t.tid < PSN / 4) ps_reg = ld4(th0 + net.lo_ln1_w + 4 * t.tid);
    }

    // ---------------- window gather + embedding (as forward_body) ----------------
    // Continuous observations: a thread's elements of the stream share their column (NT is a multiple of D), so the operands that do
    // NOT depend on the window -- the KE weights and the bias of that column, the position entries of the thread's rows -- go in flight
    // in front of the window draw (whose episode-length load is a round trip of its own), and the rows of all of the thread's
    // elements are fetched together behind it: draw -> rows, two dependent trips, where the plain loop (below, kept for the other
    // shapes) made one trip per element -- 8 at a 64-row tile (9.5 us of the 62 us pass at BASELINE config 2's shapes).
    constexpr int EIT = (LP * D + NT - 1) / NT;      // elements of the [LP][D] stream per thread
    constexpr bool HOIST = kOptHoist && NT % D == 0 && (LP * D) % NT == 0 && EIT <= 8;
    constexpr int HN = HOIST ? EIT : 1, RSTEP = NT / D;
    float hb = 0.f, hp[HN], hw[8];
    const bool hoisted = HOIST && !net.discrete && net.ke <= 8;
    const int hd_ = t.tid % D, hr0 = t.tid / D;
    if (hoisted) {
        const int dd = hd_ >= adim ? hd_ - adim : 0;
        hb = theta[net.off_obs_b + dd];
#pragma unroll
        for (int j = 0; j < 8; ++j) hw[j] = theta[net.off_obs_w + (size_t)dd * net.ke + (j < net.ke ? j : 0)];
#pragma unroll
        for (int k = 0; k < HN; ++k) {
            const int prow = R0 + hr0 + k * RSTEP < net.ctx_len ? R0 + hr0 + k * RSTEP : net.ctx_len - 1;   // rows past the context are never used
            hp[k] = theta[net.off_pos + (size_t)prow * D + hd_];
        }
    }
    int ep, st;
    if (a.ep_len != nullptr) {
        replay_draw(a.ep_len, a.s_n_valid, a.s_exclude, net.ctx_len, a.s_seed, a.draw_step >= 0 ? (uint32_t)a.draw_step : (uint32_t)a.step_counter[1], b, ep, st);
        if (which == 0 && slice == 0 && t.tid == 0) { a.ep_out[b] = ep; a.start_out[b] = st; }
    } else {
        ep = a.ep_idx != nullptr ? a.ep_idx[b] : b;
        st = a.start != nullptr ? a.start[b] : 0;
    }
    const int row0 = st + (which > 0 ? 1 : 0) + R0;
    const float* obs_rows = a.obs + (size_t)ep * a.obs_ep_stride + (size_t)row0 * O;
    const uint8_t* act_rows = a.actions != nullptr ? a.actions + (size_t)ep * a.act_ep_stride + row0 : nullptr;
    const int KE = net.ke, KEP = net.kep;
    const float* __restrict__ We = theta + net.off_obs_w;
    const float* __restrict__ be = theta + net.off_obs_b;
    const float* __restrict__ pos = theta + net.off_pos + (size_t)R0 * D;
    if (hoisted) {
        // same arithmetic as the plain loop below (same fmaf chain per element), operands already in registers
        float ev[HN][8];
#pragma unroll
        for (int kk = 0; kk < HN; ++kk) {              // all rows of this thread's elements in flight together
            const int r = hr0 + kk * RSTEP;
            // unconditional loads (a load under a branch hides the count of outstanding loads from the compiler): rows past the live
            // ones read row 0 of the episode, which always exists, and are not used
            const float* e = r < n ? obs_rows + (size_t)r * O : a.obs + (size_t)ep * a.obs_ep_stride;
#pragma unroll
            for (int k = 0; k < 8; ++k) ev[kk][k] = e[k < KE ? k : 0];
        }
#pragma unroll
        for (int kk = 0; kk < HN; ++kk) {
            const int r = hr0 + kk * RSTEP, d = hd_, idx = r * D + d;
            float v = 0.f;
            if (r < n) {
                if (d < adim) {
                    if (single) v = theta[net.off_act_emb + (int)act_rows[0] * adim + d];
                    else if (R0 + r > 0) v = theta[net.off_act_emb + (int)act_rows[r - 1] * adim + d];
                } else {
                    float acc = hb;
#pragma unroll
                    for (int k = 0; k < 8; ++k)
                        if (k < KE) acc = fmaf(ev[kk][k], hw[k], acc);
                    v = acc;
                }
                v += hp[kk];
                v = drop_apply(dr, DROP_EMB, 0, (uint32_t)((R0 + r) * D + d), v);
            }
            Xs[r * LDX + d] = v;
            if (TRAIN) rf(rec, net.ao_x0, D)[idx] = v;
        }
        if (TRAIN)
            for (int idx = t.tid; idx < LP * KEP; idx += NT) {
                const int r = idx / KEP, k = idx - r * KEP;
                rf(rec, net.ao_ein, KEP)[idx] = (r < n && k < KE) ? obs_rows[(size_t)r * O + k] : 0.f;
            }
    } else if (!net.discrete && KE <= 8) {
        for (int idx = t.tid; idx < LP * D; idx += NT) {
            const int r = idx / D, d = idx - r * D;
            float v = 0.f;
            if (r < n) {
                if (d < adim) {
                    if (single) v = theta[net.off_act_emb + (int)act_rows[0] * adim + d];
                    else if (R0 + r > 0) v = theta[net.off_act_emb + (int)act_rows[r - 1] * adim + d];
                } else {
                    const float* w = We + (size_t)(d - adim) * KE;
                    const float* e = obs_rows + (size_t)r * O;
                    float acc = be[d - adim];
                    for (int k = 0; k < KE; ++k) acc = fmaf(e[k], w[k], acc);
                    v = acc;
                }
                v += pos[r * D + d];
                v = drop_apply(dr, DROP_EMB, 0, (uint32_t)((R0 + r) * D + d), v);
            }
            Xs[r * LDX + d] = v;
            if (TRAIN) rf(rec, net.ao_x0, D)[idx] = v;
        }
        if (TRAIN)
            for (int idx = t.tid; idx < LP * KEP; idx += NT) {
                const int r = idx / KEP, k = idx - r * KEP;
                rf(rec, net.ao_ein, KEP)[idx] = (r < n && k < KE) ? obs_rows[(size_t)r * O + k] : 0.f;
            }
    } else {
        float* ein = Ws;
        for (int idx = t.tid; idx < LP * KEP; idx += NT) {
            const int r = idx / KEP, k = idx - r * KEP;
            float v = 0.f;
            if (r < n && k < KE) {
                if (net.discrete) {
                    const int j = k / net.embed_per_obs, c = k - j * net.embed_per_obs;
                    int tok = (int)obs_rows[(size_t)r * O + j];
                    tok = tok < 0 ? 0 : (tok >= net.vocab ? net.vocab - 1 : tok);
                    v = theta[net.off_obs_tab + tok * net.embed_per_obs + c];
                } else {
                    v = obs_rows[(size_t)r * O + k];
                }
            }
            ein[idx] = v;
            if (TRAIN) rf(rec, net.ao_ein, KEP)[idx] = v;
        }
        __syncthreads();
        for (int idx = t.tid; idx < LP * D; idx += NT) {
            const int r = idx / D, d = idx - r * D;
            float v = 0.f;
            if (r < n) {
                if (d < adim) {
                    if (single) v = theta[net.off_act_emb + (int)act_rows[0] * adim + d];
                    else if (R0 + r > 0) v = theta[net.off_act_emb + (int)act_rows[r - 1] * adim + d];
                } else {
                    const float* w = We + (size_t)(d - adim) * KE;
                    const float* e = ein + r * KEP;
                    float acc = be[d - adim];
                    for (int k = 0; k < KE; ++k) acc = fmaf(e[k], w[k], acc);
                    v = acc;
                }
                v += pos[r * D + d];
                v = drop_apply(dr, DROP_EMB, 0, (uint32_t)((R0 + r) * D + d), v);
            }
            Xs[r * LDX + d] = v;
            if (TRAIN) rf(rec, net.ao_x0, D)[idx] = v;
        }
    }
    tw_in.to_lds(Ar, LWD, t);
    if (t.tid < PSN / 4) st4(Ps + 4 * t.tid, ps_reg);
    DTQN_PROF(a.prof, ps++);   // embed done

    // ---------------- transformer layers ----------------
    constexpr int MG2 = pick_mg(D / 16, MT, NW);
    using Own = Owned<D, MT, MG2, NW>;
    constexpr bool SPLITK = kOptSplitKFwd && Own::ITEMS * 2 == NW && Own::PER_WAVE == 1 && MG2 == 1 && NC % 32 == 0;
    using GQkv = StageXwL<D, MT, pick_mg(3 * D / 16, MT, NW), NW, 3 * D / 16>;
    using GOut = StageXwL<D, MT, pick_mg(D / 16, MT, NW), NW, D / 16>;
    using GF1 = StageXwL<D, MT, pick_mg(NC / 16, MT, NW), NW, NC / 16>;
    for (int l = 0; l < net.num_layers; ++l) {
        const float* __restrict__ th = layer_theta(net, theta, l);
        float* lrec = TRAIN ? rec + net.ao_layer0 + (size_t)l * net.act_layer_stride : nullptr;
        const float* sm = Ps + (l & 1) * PSN;            // this layer's vectors
        const float* src = Xs;
        TileRegs<NW, D, D> tw_o;
        tw_o.load(th + net.lo_out_w, D, t);              // in flight during the in-projection
        __syncthreads();                                 // (a) residual stream, W_in and the parameter block visible
        if (ident) {   // x_norm1 = LN1(x)  (transformer.py:87)
            layernorm_rows<D, NW, LP, TRAIN, PAD, kOptFwdWT>(Xs, Us, LDX, LP, sm + P_LN1W, sm + P_LN1B, rf(lrec, net.al_st1, 2), t,
                                  nullptr, rf(lrec, net.al_u1, D), 0, dreal);
            __syncthreads();
            src = Us;
        }
        if (TRAIN && !ident) rec_tile_store<NW, kOptFwdWT>(src, LDX, rf(lrec, net.al_u1, D), LP, D, t);
        GQkv::run(src, LDX, Ar, sm + P_INB, t, [&](int r, int c, float v) { AW[r * LDW + c] = v; });
        tw_o.to_lds(Ar + OFF_WO, LWD, t);                // behind W_in's last row: free since the previous layer's FFN
        TileRegs<NW, NC, D> tw_1;
        tw_1.load(th + net.lo_f1_w, D, t);               // FFN-1 chunk 0, in flight during attention
        __syncthreads();                                 // (b) q | k | v visible; region A free
        DTQN_PROF(a.prof, ps++);   // qkv done
        // K | V of the rows below this slice.  The hand-over goes FIRST: the slices above wait for it, and the sender's drain of its
        // vector-memory queue in front of the flags would otherwise also sit out the acknowledgements of the q | k | v record's
        // write-through stores (a microsecond on the training pass's critical path); the record leaves behind the receive, whose loads
        // must not queue behind those stores either (vmcnt counts both, in order).
        if (RS > 1) {
            float* xb = a.xch + ((size_t)seq * net.num_layers + l) * kv_xch_floats(RS, LP, D);
            int32_t* xf = a.xflags + ((size_t)seq * net.num_layers + l) * kv_xch_flags(RS);
            kv_send<NW, RS>(Ws, LDW, D, LP, slice, xb, xf, t);
            kv_recv<NW, RS>(Ws, LDW, D, LP, slice, xb, xf, t);
        }
        if (TRAIN) {                                     // q | k | v -> record before attention overwrites q
            rec_tile_store<NW, kOptFwdWT>(AW, LDW, rf(lrec, net.al_qkv, 3 * D), LP, 3 * D, t);
            __syncthreads();
        }
        if constexpr (PAD)       // heads zero-padded to HD columns: the scale is the real head width's
            attention_forward<HD, NW, (HD >= kAttnMfmaMinHeadDim) || (RS >= 2 && DTQN_SPLIT_ATTN_MFMA)>(Ws, LDW, D, H, LP, nfull, TRAIN ? lrec + net.al_lse : nullptr, t, R0, LPF, dr, l, 0, (float)net.hd_real);
        else
            attention_forward<HD, NW, (HD >= kAttnMfmaMinHeadDim) || (RS >= 2 && DTQN_SPLIT_ATTN_MFMA)>(Ws, LDW, D, H, LP, nfull, TRAIN ? lrec + net.al_lse : nullptr, t, R0, LPF, dr, l);
        tw_1.to_lds(Ar, LWD, t);
        TileRegs<NW, D, NC> tw_2;
        tw_2.load(th + net.lo_f2_w, 4 * D, t);           // FFN-2 chunk 0 (columns [0, NC) of W_2), in flight during the out-projection
        __syncthreads();                                 // (c) attention output visible
        DTQN_PROF(a.prof, ps++);   // attention done
        if (TRAIN) rec_tile_store<NW, kOptFwdWT>(AW, LDW, rf(lrec, net.al_o, D), LP, D, t);
        {   // out-projection, ReLU, residual gate:  x <- x + relu(o W_o^T + b_o)   (transformer.py:72 / :96)
            float* m_g = mf(lrec, net.al_m1, D / 16);
            GOut::run(AW, LDW, Ar + OFF_WO, sm + P_OUTB, t, [&](int r, int c, float v) {
                const float y = fmaxf(v, 0.f);
                if (TRAIN) ballot_store(m_g, D / 16, r, c, y > 0.f, t.lane);
                Xs[r * LDX + c] += y;
            });
        }
        __syncthreads();                                 // (d) stream updated; region B free
        DTQN_PROF(a.prof, ps++);   // out-proj done
        tw_2.to_lds(Ar + OFF_B, LWC, t);
        tw_1.load(th + net.lo_f1_w + (size_t)NC * D, D, t);   // FFN-1 chunk 1, in flight during LayerNorm + FFN chunk 0
        if (!ident) {  // x = LN1(x)
            layernorm_rows<D, NW, LP, TRAIN, PAD, kOptFwdWT>(Xs, Xs, LDX, LP, sm + P_LN1W, sm + P_LN1B, rf(lrec, net.al_st1, 2), t,
                                  rf(lrec, net.al_s1, D), rf(lrec, net.al_u2, D), 0, dreal);
            src = Xs;
        } else {       // x_norm2 = LN2(x)
            layernorm_rows<D, NW, LP, TRAIN, PAD, kOptFwdWT>(Xs, Us, LDX, LP, sm + P_LN2W, sm + P_LN2B, rf(lrec, net.al_st2, 2), t,
                                  rf(lrec, net.al_s1, D), rf(lrec, net.al_u2, D), 0, dreal);
            src = Us;
        }
        __syncthreads();                                 // (e) LayerNorm output and FFN-2 chunk 0 visible
        DTQN_PROF(a.prof, ps++);   // LN1 done
        // FFN D -> 4D -> D in two hidden-column passes of NC; the second GEMM accumulates in registers
        f32x4 facc[Own::PER_WAVE][MG2];
#pragma unroll
        for (int q = 0; q < Own::PER_WAVE; ++q)
#pragma unroll
            for (int m = 0; m < MG2; ++m) facc[q][m] = zero4();
        float* mh_g = mf(lrec, net.al_mh, 4 * D / 16);
        const bool more = l + 1 < net.num_layers;
#pragma unroll
        for (int c0 = 0; c0 < 4 * D; c0 += NC) {
            GF1::run(src, LDX, Ar, sm + P_F1B + c0, t, [&](int r, int c, float v) {
                const float hv = fmaxf(v, 0.f);
                if (TRAIN) ballot_store(mh_g, 4 * D / 16, r, c0 + c, hv > 0.f, t.lane);
                Ws[r * LDW + c] = hv;
            });
            __syncthreads();                             // (f) / (h) hidden chunk visible; region A free
            if (c0 == 0) {
                tw_1.to_lds(Ar, LWD, t);                 // FFN-1 chunk 1
                tw_2.load(th + net.lo_f2_w + NC, 4 * D, t);   // FFN-2 chunk 1, in flight during FFN-2 chunk 0
            }
            if (TRAIN) rec_tile_store<NW, kOptFwdWT>(Ws, LDW, rf(lrec, net.al_h, 4 * D) + c0, LP, NC, t, 4 * D);
            if constexpr (SPLITK) {
                // four column tiles, eight waves: wave w takes tile w % 4 and half w / 4 of the chunk's hidden columns (summed below)
                const int it = t.wave % Own::ITEMS, kh = t.wave / Own::ITEMS;
                frag_xwl_mma<NC / 2, 1>(Ws + kh * (NC / 2), LDW, Ar + OFF_B + (it * 16 + t.i) * LWC + kh * (NC / 2), t, facc[0]);
            } else {
#pragma unroll
                for (int q = 0; q < Own::PER_WAVE; ++q)
                    if (Own::valid_fast(t.wave, q))
                        frag_xwl_mma<NC, MG2>(Ws + Own::mg(t.wave, q) * MG2 * 16 * LDW, LDW,
                                              Ar + OFF_B + (Own::nt(t.wave, q) * 16 + t.i) * LWC, t, facc[q]);
            }
            if (c0 == 0) {
                __syncthreads();                         // (g) chunk 0 of the hidden and of W_2 consumed; FFN-1 chunk 1 visible
                tw_2.to_lds(Ar + OFF_B, LWC, t);
                // what the stage after this layer needs goes in flight now: the next layer's W_in and vectors, or the head's
                if (more) {
                    const float* __restrict__ thn = layer_theta(net, theta, l + 1);
                    tw_in.load(thn + net.lo_in_w, D, t);
                    if (t.tid < PSN / 4) ps_reg = ld4(thn + net.lo_ln1_w + 4 * t.tid);
                } else {
                    tw_hd.load(theta + net.off_head1_w, D, t);
                    if (t.tid < D / 4) ps_reg = ld4(theta + net.off_head1_b + 4 * t.tid);
                }
            }
        }
        if constexpr (SPLITK) {
            // the upper four waves hand their half of the sums to the lower four through region A of the arena (free since barrier (h))
            if (t.wave >= Own::ITEMS) {
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) Ar[((t.wave - Own::ITEMS) * 4 + r4) * 64 + t.lane] = facc[0][0][r4];
            }
            __syncthreads();
            if (t.wave < Own::ITEMS) {
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) facc[0][0][r4] += Ar[(t.wave * 4 + r4) * 64 + t.lane];
            }
        }
        {
            float* m_g = mf(lrec, net.al_m2, D / 16);
#pragma unroll
            for (int q = 0; q < Own::PER_WAVE; ++q) {
                if (Own::valid_fast(t.wave, q)) {
                    const int c = Own::nt(t.wave, q) * 16 + t.i;
                    const float b2 = sm[P_F2B + c];
#pragma unroll
                    for (int m = 0; m < MG2; ++m)
#pragma unroll
                        for (int r4 = 0; r4 < 4; ++r4) {
                            const int r = (Own::mg(t.wave, q) * MG2 + m) * 16 + t.kq * 4 + r4;
                            const float y = fmaxf(drop_apply(dr, DROP_FFN, l, (uint32_t)((R0 + r) * D + c), facc[q][m][r4] + b2), 0.f);
                            if (TRAIN) ballot_store(m_g, D / 16, r, c, y > 0.f, t.lane);
                            Xs[r * LDX + c] += y;
                        }
                }
            }
        }
        __syncthreads();                                 // (i) stream updated; the whole arena is free
        DTQN_PROF(a.prof, ps++);   // FFN done
        if (more) {
            tw_in.to_lds(Ar, LWD, t);
            if (t.tid < PSN / 4) st4(Ps + ((l + 1) & 1) * PSN + 4 * t.tid, ps_reg);
        } else {
            tw_hd.to_lds(Ar, LWD, t);
            if (t.tid < D / 4) st4(Ps + ((l + 1) & 1) * PSN + 4 * t.tid, ps_reg);
        }
        if (!ident) {  // x = LN2(x)
            layernorm_rows<D, NW, LP, TRAIN, PAD, kOptFwdWT>(Xs, Xs, LDX, LP, sm + P_LN2W, sm + P_LN2B, rf(lrec, net.al_st2, 2), t,
                                  rf(lrec, net.al_s2, D), nullptr, 0, dreal);
        } else if (TRAIN) {
            rec_tile_store<NW, kOptFwdWT>(Xs, LDX, rf(lrec, net.al_s2, D), LP, D, t);
        }
    }

    // ---------------- Q head: Linear(D,D) -> ReLU -> Linear(D,A)  (dtqn.py:149-153,216) ----------------
    __syncthreads();
    DTQN_PROF(a.prof, ps++);       // layers done
    if (TRAIN) rec_tile_store<NW, kOptFwdWT>(Xs, LDX, rf(rec, net.ao_xf, D), LP, D, t);
    GOut::run(Xs, LDX, Ar, Ps + (net.num_layers & 1) * PSN, t, [&](int r, int c, float v) { Ws[r * LDW + c] = fmaxf(v, 0.f); });
    __syncthreads();
    if (TRAIN) rec_tile_store<NW, kOptFwdWT>(Ws, LDW, rf(rec, net.ao_hh, D), LP, D, t);
    {
        float* q = a.q_out + (size_t)which * a.q_which_stride + (size_t)b * a.q_seq_stride + (size_t)R0 * a.q_row_stride;
        auto q_rows = [&](const float* __restrict__ W2, const float* __restrict__ b2) {
            for (int idx = t.tid; idx < (n < LP ? n : LP) * A; idx += NT) {
                const int r = idx / A, ac = idx - r * A;
                const float* hrow = Ws + r * LDW;
                const float* w = W2 + (size_t)ac * D;
                float acc = b2[ac];
#pragma unroll 8
                for (int k = 0; k < D; k += 4) {
                    const float4 hv = ld4(hrow + k), wv = ld4(w + k);
                    acc = fmaf(hv.x, wv.x, acc); acc = fmaf(hv.y, wv.y, acc); acc = fmaf(hv.z, wv.z, acc); acc = fmaf(hv.w, wv.w, acc);
                }
                q[r * a.q_row_stride + ac] = acc;
                if (a.q_last_host != nullptr && R0 + r == (a.last_rows != nullptr ? a.last_rows[seq] - 1 : nfull - 1)) a.q_last_host[seq * A + ac] = acc;
            }
        };
        q_rows(theta + net.off_head2_w, theta + net.off_head2_b);
    }
    DTQN_PROF(a.prof, ps++);       // end
}

// WL: weights through LDS (forward_body_wl; residual gate, D <= 64, chosen by the host when the arena fits).
// DROP: keep-mask code compiled in (register-direct stages only).
// PAD: width-padded network (forward_body_wl only)
template <int D, int MT, int HD, int NW, bool GRU, int RS, bool WL, bool DROP, bool PAD = false>
__global__ __launch_bounds__(NW * 64) void dtqn_forward_kernel(FwdArgs a) {
    static_assert(!(WL && DROP), "dropout runs on the register-direct stages");
    static_assert(!PAD || WL, "width padding runs on forward_body_wl");
    int seq_, slice_;
    slice_block_map((int)blockIdx.x - a.block0, a.nseq, RS, seq_, slice_);
    const int which = a.pass0 + seq_ / a.batch;                               // workgroup-uniform
    if constexpr (WL) {
        if (a.act != nullptr && which == 0) forward_body_wl<D, MT, HD, NW, RS, true, PAD>(a);
        else forward_body_wl<D, MT, HD, NW, RS, false, PAD>(a);
    } else {
        if (a.act != nullptr && which == 0) forward_body<D, MT, HD, NW, GRU, RS, true, DROP>(a);
        else forward_body<D, MT, HD, NW, GRU, RS, false, DROP>(a);
    }
}

// Weights through LDS: residual gate, D <= 64, parameter-block layout as dtqn_layout.cpp writes it, arena fits.
// DTQN_WL=0 in the environment keeps the register-direct stages (A/B runs).
inline bool fwd_wl_ok(const DtqnNet* net) {
    const int D = net->d_model;
    if (net->tiled || net->gate != DTQN_GATE_RES || D > 64) return false;
    const char* e = getenv("DTQN_WL");
    if (e != nullptr && e[0] == '0') return false;
    const int b = net->lo_ln1_w;
    return net->lo_ln1_b - b == D && net->lo_ln2_w - b == 2 * D && net->lo_ln2_b - b == 3 * D && net->lo_in_b - b == 4 * D &&
           net->lo_out_b - b == 7 * D && net->lo_f1_b - b == 8 * D && net->lo_f2_b - b == 12 * D;
}
// lp_rows: rows of the [LP][.] tiles of the launch (0 = the network's padded context)
inline size_t fwd_lds_bytes(const DtqnNet* net, bool wl = false) {
    const int LP = net->lp, D = net->d_model;
    size_t fl = (size_t)LP * (D + 4) + (size_t)LP * (3 * D + 4);
    if (net->identity) fl += (size_t)LP * (D + 4);
    if (wl) fl += 2 * (size_t)wl_small_floats(D) + (size_t)wl_arena_floats(D);
    return fl * sizeof(float);
}

template <int D, int MT, int HD, int NW, bool GRU, int RS, bool WL, bool DROP, bool PAD = false>
int launch_fwd4(const FwdArgs& a, int nseq, hipStream_t stream) {
    const size_t lds = fwd_lds_bytes(&a.net, WL);
    static size_t attr_lds[kMaxDevices] = {};    // per instantiation and device
    raise_lds_limit(reinterpret_cast<const void*>(&dtqn_forward_kernel<D, MT, HD, NW, GRU, RS, WL, DROP, PAD>), lds, attr_lds);
    (void)hipGetLastError();   // drop stale errors left by other users of the runtime
    hipLaunchKernelGGL((dtqn_forward_kernel<D, MT, HD, NW, GRU, RS, WL, DROP, PAD>), dim3(nseq * RS), dim3(NW * 64), lds, stream, a);
    return hipGetLastError() == hipSuccess ? DTQN_OK : DTQN_ERR_LAUNCH;
}
// The shapes that exist as weights-through-LDS kernels of d_model 64 / a 64-row context only (dtqn_limits.h, dtqn_ws_lite: head width 32
// and the width-padded networks): rs = 4 (four 16-row slices: the TD update's passes, the latency-mode actor) or 1 (one workgroup per
// sequence: inference at any batch).  Everything else of such a network runs on its row-block twin (dtqn_td_prefers_tiled).
template <int HD, bool PAD>
int launch_fwd_lite(const FwdArgs& a, int nseq, int rs, hipStream_t stream) {
    if (a.drop_thresh != 0u || !fwd_wl_ok(&a.net) || fwd_lds_bytes(&a.net, true) > 160 * 1024 || a.net.lp != 64) return DTQN_ERR_CONFIG;
    if (rs == 4) return launch_fwd4<64, 1, HD, 8, false, 4, true, false, PAD>(a, nseq, stream);
    if (rs == 1) return launch_fwd4<64, 4, HD, 8, false, 1, true, false, PAD>(a, nseq, stream);
    return DTQN_ERR_CONFIG;
}
template <int D, int MT, int HD, int NW, bool GRU, int RS>
int launch_fwd2(const FwdArgs& a, int nseq, hipStream_t stream) {
    if constexpr (RS == 4) {      // four 16-row slices: the weights-through-LDS body only (residual gate, D <= 64, no dropout)
        static_assert(!GRU && D <= 64, "four row slices run on forward_body_wl");
        if (a.drop_thresh != 0u || !fwd_wl_ok(&a.net) || fwd_lds_bytes(&a.net, true) > 160 * 1024) return DTQN_ERR_CONFIG;
        return launch_fwd4<D, MT, HD, NW, GRU, RS, true, false>(a, nseq, stream);
    } else {
        if (a.drop_thresh != 0u) return launch_fwd4<D, MT, HD, NW, GRU, RS, false, true>(a, nseq, stream);
        if constexpr (!GRU && D <= 64) {
            if (fwd_wl_ok(&a.net) && fwd_lds_bytes(&a.net, true) <= 160 * 1024) return launch_fwd4<D, MT, HD, NW, GRU, RS, true, false>(a, nseq, stream);
        }
        return launch_fwd4<D, MT, HD, NW, GRU, RS, false, false>(a, nseq, stream);
    }
}
template <int D, int MT, int HD, int NW>
int launch_fwd(const FwdArgs& a, int nseq, hipStream_t stream) {
    if (a.net.gate == DTQN_GATE_GRU) {
        if constexpr (D <= 64) return launch_fwd2<D, MT, HD, NW, true, 1>(a, nseq, stream);
        else return DTQN_ERR_CONFIG;
    }
    return launch_fwd2<D, MT, HD, NW, false, 1>(a, nseq, stream);
}


// FwdArgs of a TD-update forward, passes [pass0 ..) (defined in dtqn_forward.hip)
void td_forward_args(const DtqnNet* net, const DtqnReplay* rp, const DtqnTd* td, int pass0, int draw_step, FwdArgs* out);

// The instantiations are spread over several translation units (dtqn_forward_inst*.hip) so that they compile in
// parallel; the dispatcher (dtqn_forward.hip) only sees these declarations.
#define DTQN_FWD_GROUP_A(X) X(64, 4, 8, 4) X(64, 4, 8, 8) X(64, 4, 8, 16) X(64, 4, 16, 8)
#define DTQN_FWD_GROUP_B(X) X(128, 4, 16, 4) X(128, 4, 16, 8) X(128, 2, 16, 8) X(128, 1, 16, 8)
#define DTQN_FWD_GROUP_C(X) X(64, 2, 8, 8) X(64, 1, 8, 8) X(64, 2, 16, 8) X(64, 1, 16, 8) X(16, 1, 8, 4) X(16, 1, 8, 8) X(32, 2, 8, 4) X(32, 1, 16, 4)
// row-split (two workgroups per sequence) instantiations: (D, MT, HD, NW, GRU)
#define DTQN_FWD_GROUP_D(X) X(64, 2, 8, 8, true, 2) X(64, 2, 16, 8, true, 2) X(64, 2, 8, 8, false, 2) X(64, 2, 16, 8, false, 2) X(128, 2, 16, 8, false, 2) \
    X(64, 1, 8, 8, false, 4) X(64, 1, 16, 8, false, 4)
// the four-slice / whole-tile pairs of dtqn_ws_lite shapes: (HD, PAD)   (dtqn_forward_inste.hip)
#define DTQN_FWD_GROUP_E(X) X(8, true) X(16, true) X(32, true) X(32, false)
#define DTQN_FWDL_DECL(hd, pad) extern template int launch_fwd_lite<hd, pad>(const FwdArgs&, int, int, hipStream_t);
#define DTQN_FWDL_DEF(hd, pad) template int launch_fwd_lite<hd, pad>(const FwdArgs&, int, int, hipStream_t);
#define DTQN_FWD_DECL(d, mt, hd, nw) extern template int launch_fwd<d, mt, hd, nw>(const FwdArgs&, int, hipStream_t);
#define DTQN_FWD_DEF(d, mt, hd, nw) template int launch_fwd<d, mt, hd, nw>(const FwdArgs&, int, hipStream_t);
#define DTQN_FWD2_DECL(d, mt, hd, nw, gru, rs) extern template int launch_fwd2<d, mt, hd, nw, gru, rs>(const FwdArgs&, int, hipStream_t);
#define DTQN_FWD2_DEF(d, mt, hd, nw, gru, rs) template int launch_fwd2<d, mt, hd, nw, gru, rs>(const FwdArgs&, int, hipStream_t);

}  // namespace dtqn
