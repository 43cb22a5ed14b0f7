// DTQN forward for gfx950: embed -> +pos -> NL x (causal MHA, gate, LN, FFN, gate, LN) -> Q head,
// one workgroup per sequence, the whole [LP x D] context tile resident in LDS.
//
// Replaces DTQN.forward (dtqn/networks/dtqn.py:158-218), TransformerLayer.forward /
// TransformerIdentityLayer.forward (transformer.py:63-78,86-101), ObservationEmbeddingRepresentation
// (representations.py:17-23), and -- for the TD update -- the window gather of
// ReplayBuffer.sample (replay_buffer.py:160-167): the workgroup reads its (episode, start) window
// straight out of the device-resident replay arrays.
#include "dtqn_device.hpp"

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
    float* q_out;               // which-major
    long long q_which_stride, q_seq_stride;
    int q_row_stride;
    float* act;                 // nullptr: inference; else activation records for which == 0
};

__device__ __forceinline__ int lds_ldx(int D) { return D + 4; }
__device__ __forceinline__ int lds_ldw(int D) { return 3 * D + 4; }

template <int D, int MT, int HD>
__global__ __launch_bounds__(DTQN_THREADS) void dtqn_forward_kernel(FwdArgs a) {
    constexpr int LP = MT * 16;
    constexpr int LDX = D + 4, LDW = 3 * D + 4;
    constexpr int NC = 2 * D;                      // FFN hidden columns per pass
    const DtqnNet& net = a.net;
    const Thr t = make_thr();
    const int which = (int)blockIdx.x / a.batch;
    const int b = (int)blockIdx.x - which * a.batch;
    const float* __restrict__ theta = which == 2 ? a.theta_b : a.theta_a;
    const int n = a.n, H = net.num_heads, O = net.obs_dim, adim = net.action_dim, A = net.num_actions;
    const bool ident = net.identity != 0;
    float* rec = (a.act != nullptr && which == 0) ? a.act + (size_t)b * net.act_stride : nullptr;

    float* Xs = reinterpret_cast<float*>(dtqn_smem);   // residual stream            [LP][LDX]
    float* Ws = Xs + LP * LDX;                         // q|k|v, FFN hidden, staging [LP][LDW]
    float* Us = Ws + LP * LDW;                         // identity only: LN output   [LP][LDX]

    // ---------------- window gather + embedding ----------------
    const int ep = a.ep_idx != nullptr ? a.ep_idx[b] : b;
    const int row0 = (a.start != nullptr ? a.start[b] : 0) + (which > 0 ? 1 : 0);
    const float* obs_rows = a.obs + (size_t)ep * a.obs_ep_stride + (size_t)row0 * O;
    const uint8_t* act_rows = a.actions != nullptr ? a.actions + (size_t)ep * a.act_ep_stride + row0 : nullptr;
    const int KE = net.ke, KEP = net.kep;
    float* ein = Ws;                                   // [LP][KEP] embedding-linear input
    for (int idx = t.tid; idx < LP * KEP; idx += DTQN_THREADS) {
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
        if (rec != nullptr) rec[net.ao_ein + idx] = v;
    }
    __syncthreads();
    {
        const float* __restrict__ We = theta + net.off_obs_w;
        const float* __restrict__ be = theta + net.off_obs_b;
        const float* __restrict__ pos = theta + net.off_pos;
        for (int idx = t.tid; idx < LP * D; idx += DTQN_THREADS) {
            const int r = idx / D, d = idx - r * D;
            float v = 0.f;
            if (r < n) {
                if (d < adim) {
                    // previous-action embedding rolled right by one, row 0 zeroed unless n == 1 (dtqn.py:184-192)
                    if (n == 1) v = theta[net.off_act_emb + (int)act_rows[0] * adim + d];
                    else if (r > 0) v = theta[net.off_act_emb + (int)act_rows[r - 1] * adim + d];
                } else {
                    const float* w = We + (size_t)(d - adim) * KE;
                    const float* e = ein + r * KEP;
                    float acc = be[d - adim];
                    for (int k = 0; k < KE; ++k) acc = fmaf(e[k], w[k], acc);
                    v = acc;
                }
                v += pos[r * D + d];
            }
            Xs[r * LDX + d] = v;
            if (rec != nullptr) rec[net.ao_x0 + idx] = v;
        }
    }
    __syncthreads();

    // ---------------- transformer layers ----------------
    for (int l = 0; l < net.num_layers; ++l) {
        const float* __restrict__ th = layer_theta(net, theta, l);
        float* lrec = rec != nullptr ? rec + net.ao_layer0 + (size_t)l * net.act_layer_stride : nullptr;
        const float* src = Xs;
        if (ident) {   // x_norm1 = LN1(x)  (transformer.py:87)
            layernorm_rows<D>(Xs, Us, LDX, LP, th + net.lo_ln1_w, th + net.lo_ln1_b, lrec ? lrec + net.al_st1 : nullptr, t);
            __syncthreads();
            src = Us;
        }
        if (lrec != nullptr) tile_store(src, LDX, lrec + net.al_u1, LP, D, t);
        // packed in-projection: qkv = u W_in^T + b_in
        {
            const float* __restrict__ bin = th + net.lo_in_b;
            float* qkv_g = lrec ? lrec + net.al_qkv : nullptr;
            gemm_xwT<D, MT>(src, LDX, th + net.lo_in_w, D, 3 * D, t, [&](int r, int c, float v) {
                v += bin[c];
                Ws[r * LDW + c] = v;
                if (qkv_g != nullptr) qkv_g[r * 3 * D + c] = v;
            });
        }
        __syncthreads();
        attention_forward<HD>(Ws, LDW, D, H, LP, n, lrec ? lrec + net.al_lse : nullptr, t);
        __syncthreads();
        if (lrec != nullptr) tile_store(Ws, LDW, lrec + net.al_o, LP, D, t);
        // out-projection, ReLU, residual gate:  x <- x + relu(o W_o^T + b_o)   (transformer.py:72 / :96)
        {
            const float* __restrict__ bo = th + net.lo_out_b;
            float* y_g = lrec ? lrec + net.al_y1 : nullptr;
            float* s_g = lrec ? lrec + net.al_s1 : nullptr;
            gemm_xwT<D, MT>(Ws, LDW, th + net.lo_out_w, D, D, t, [&](int r, int c, float v) {
                const float y = fmaxf(v + bo[c], 0.f);
                const float s = Xs[r * LDX + c] + y;
                Xs[r * LDX + c] = s;
                if (y_g != nullptr) { y_g[r * D + c] = y; s_g[r * D + c] = s; }
            });
        }
        __syncthreads();
        if (!ident) {  // x = LN1(x)
            layernorm_rows<D>(Xs, Xs, LDX, LP, th + net.lo_ln1_w, th + net.lo_ln1_b, lrec ? lrec + net.al_st1 : nullptr, t);
            src = Xs;
        } else {       // x_norm2 = LN2(x)
            layernorm_rows<D>(Xs, Us, LDX, LP, th + net.lo_ln2_w, th + net.lo_ln2_b, lrec ? lrec + net.al_st2 : nullptr, t);
            src = Us;
        }
        __syncthreads();
        if (lrec != nullptr) tile_store(src, LDX, lrec + net.al_u2, LP, D, t);
        // FFN D -> 4D -> D in hidden-column passes of NC; the second GEMM accumulates in registers
        {
            constexpr int NTW = (D / 16 + DTQN_WAVES - 1) / DTQN_WAVES;   // output n-tiles per wave
            f32x4 facc[NTW][MT];
#pragma unroll
            for (int q = 0; q < NTW; ++q)
#pragma unroll
                for (int m = 0; m < MT; ++m) facc[q][m] = zero4();
            const float* __restrict__ W1 = th + net.lo_f1_w;
            const float* __restrict__ b1 = th + net.lo_f1_b;
            const float* __restrict__ W2 = th + net.lo_f2_w;
            float* h_g = lrec ? lrec + net.al_h : nullptr;
            for (int c0 = 0; c0 < 4 * D; c0 += NC) {
                gemm_xwT<D, MT>(src, LDX, W1 + (size_t)c0 * D, D, NC, t, [&](int r, int c, float v) {
                    const float hv = fmaxf(v + b1[c0 + c], 0.f);
                    Ws[r * LDW + c] = hv;
                    if (h_g != nullptr) h_g[r * 4 * D + c0 + c] = hv;
                });
                __syncthreads();
#pragma unroll
                for (int q = 0; q < NTW; ++q) {
                    const int nt = t.wave + q * DTQN_WAVES;
                    if (nt * 16 < D) mma_xwT_tile<NC, MT>(Ws, LDW, W2 + (size_t)(nt * 16 + t.i) * 4 * D + c0, t, facc[q]);
                }
                __syncthreads();
            }
            const float* __restrict__ b2 = th + net.lo_f2_b;
            float* y_g = lrec ? lrec + net.al_y2 : nullptr;
            float* s_g = lrec ? lrec + net.al_s2 : nullptr;
#pragma unroll
            for (int q = 0; q < NTW; ++q) {
                const int nt = t.wave + q * DTQN_WAVES;
                if (nt * 16 < D) {
                    const int c = nt * 16 + t.i;
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int r4 = 0; r4 < 4; ++r4) {
                            const int r = m * 16 + t.kq * 4 + r4;
                            const float y = fmaxf(facc[q][m][r4] + b2[c], 0.f);
                            const float s = Xs[r * LDX + c] + y;
                            Xs[r * LDX + c] = s;
                            if (y_g != nullptr) { y_g[r * D + c] = y; s_g[r * D + c] = s; }
                        }
                }
            }
        }
        __syncthreads();
        if (!ident) {  // x = LN2(x)
            layernorm_rows<D>(Xs, Xs, LDX, LP, th + net.lo_ln2_w, th + net.lo_ln2_b, lrec ? lrec + net.al_st2 : nullptr, t);
            __syncthreads();
        }
    }

    // ---------------- Q head: Linear(D,D) -> ReLU -> Linear(D,A)  (dtqn.py:149-153,216) ----------------
    if (rec != nullptr) tile_store(Xs, LDX, rec + net.ao_xf, LP, D, t);
    {
        const float* __restrict__ bh = theta + net.off_head1_b;
        float* hh_g = rec ? rec + net.ao_hh : nullptr;
        gemm_xwT<D, MT>(Xs, LDX, theta + net.off_head1_w, D, D, t, [&](int r, int c, float v) {
            const float hv = fmaxf(v + bh[c], 0.f);
            Ws[r * LDW + c] = hv;
            if (hh_g != nullptr) hh_g[r * D + c] = hv;
        });
    }
    __syncthreads();
    {
        const float* __restrict__ W2 = theta + net.off_head2_w;
        const float* __restrict__ b2 = theta + net.off_head2_b;
        float* q = a.q_out + (size_t)which * a.q_which_stride + (size_t)b * a.q_seq_stride;
        for (int idx = t.tid; idx < n * A; idx += DTQN_THREADS) {
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
        }
    }
}

static size_t fwd_lds_bytes(const DtqnNet* net) {
    const int LP = net->lp, D = net->d_model;
    size_t fl = (size_t)LP * (D + 4) + (size_t)LP * (3 * D + 4);
    if (net->identity) fl += (size_t)LP * (D + 4);
    return fl * sizeof(float);
}

template <int D, int MT, int HD>
static int launch_fwd(const FwdArgs& a, int nblocks, hipStream_t stream) {
    const size_t lds = fwd_lds_bytes(&a.net);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&dtqn_forward_kernel<D, MT, HD>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipGetLastError();   // drop stale errors left by other users of the runtime
    hipLaunchKernelGGL((dtqn_forward_kernel<D, MT, HD>), dim3(nblocks), dim3(DTQN_THREADS), lds, stream, a);
    return hipGetLastError() == hipSuccess ? DTQN_OK : DTQN_ERR_LAUNCH;
}

static int dispatch_fwd(const FwdArgs& a, int nblocks, hipStream_t stream) {
    const int D = a.net.d_model, MT = a.net.lp / 16, HD = a.net.head_dim;
#define DTQN_FWD_CASE(d, mt, hd) \
    if (D == d && MT == mt && HD == hd) return launch_fwd<d, mt, hd>(a, nblocks, stream);
    DTQN_FWD_CASE(64, 4, 8)
    DTQN_FWD_CASE(128, 4, 16)
    DTQN_FWD_CASE(64, 4, 16)
    DTQN_FWD_CASE(16, 1, 8)
    DTQN_FWD_CASE(32, 2, 8)
    DTQN_FWD_CASE(32, 1, 16)
#undef DTQN_FWD_CASE
    return DTQN_ERR_CONFIG;
}

}  // namespace dtqn

using namespace dtqn;

extern "C" int dtqn_lds_bytes_forward(const DtqnNet* net, int /*training*/) {
    if (!net) return 0;
    const size_t b = fwd_lds_bytes(net);
    return b <= 160 * 1024 ? (int)b : 0;
}

extern "C" int dtqn_forward(const DtqnNet* net, const float* theta, const float* obs, const uint8_t* actions,
                            int batch, int n, float* q_out, void* stream) {
    if (!net || !theta || !obs || !q_out || batch < 1) return DTQN_ERR_ARG;
    if (n < 1 || n > net->ctx_len) return DTQN_ERR_ARG;                 // dtqn.py:170-173
    if (net->action_dim > 0 && !actions) return DTQN_ERR_ARG;
    if (net->gate != DTQN_GATE_RES) return DTQN_ERR_CONFIG;
    FwdArgs a;
    a.net = *net;
    a.theta_a = theta; a.theta_b = theta;
    a.obs = obs; a.actions = actions;
    a.obs_ep_stride = (long long)n * net->obs_dim;
    a.act_ep_stride = n;
    a.ep_idx = nullptr; a.start = nullptr;
    a.n = n; a.batch = batch;
    a.q_out = q_out;
    a.q_which_stride = 0;
    a.q_seq_stride = (long long)n * net->num_actions;
    a.q_row_stride = net->num_actions;
    a.act = nullptr;
    return dispatch_fwd(a, batch, (hipStream_t)stream);
}

extern "C" int dtqn_td_forward(const DtqnNet* net, const DtqnReplay* rp, const DtqnTd* td, void* stream) {
    if (!net || !rp || !td || td->batch < 1) return DTQN_ERR_ARG;
    if (rp->obs_dim != net->obs_dim || rp->max_steps < net->ctx_len) return DTQN_ERR_ARG;
    if (net->gate != DTQN_GATE_RES) return DTQN_ERR_CONFIG;
    FwdArgs a;
    a.net = *net;
    a.theta_a = td->theta_pol; a.theta_b = td->theta_tgt;
    a.obs = rp->obs; a.actions = rp->actions;
    a.obs_ep_stride = (long long)(rp->max_steps + 1) * rp->obs_dim;
    a.act_ep_stride = rp->max_steps + 1;
    a.ep_idx = td->ep_idx; a.start = td->start;
    a.n = net->ctx_len; a.batch = td->batch;
    a.q_out = td->q3;
    a.q_which_stride = (long long)td->batch * net->lp * net->ap;
    a.q_seq_stride = (long long)net->lp * net->ap;
    a.q_row_stride = net->ap;
    a.act = td->act;
    return dispatch_fwd(a, 3 * td->batch, (hipStream_t)stream);
}
