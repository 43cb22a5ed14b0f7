// Row-block tiled DTQN forward for contexts / widths that do not fit one workgroup's LDS
// (BASELINE configs 4 and 5: L = 128 / 256, D = 128 / 256).  Same arithmetic as dtqn_forward.hip
// (DTQN.forward, dtqn/networks/dtqn.py:158-218), different decomposition:
//   * tensors live in a global workspace, token-major [S][LPB][cols], LPB = L rounded up to 64;
//   * every projection is a matrix-core GEMM over 64-row blocks (the block's input tile staged in LDS,
//     128 output columns per workgroup, K walked in chunks of D with register accumulation);
//   * attention runs per (sequence, head) with that head's q | k | v held in LDS;
//   * LayerNorm is a row-wise kernel.
// Forward only (inference / actor / parity of Q-values); the training path of these configs is not built.
#include "dtqn_device.hpp"

namespace dtqn {

constexpr int TNW = 8;                 // waves per workgroup of the tiled kernels
constexpr int TNT = TNW * 64;
constexpr int TROWS = 64;              // rows per block

struct TlCommon {
    DtqnNet net;
    int S;                             // sequences
    int n;                             // real rows per sequence
    int lpb;                           // padded rows per sequence
};

// ---- embedding + position ---------------------------------------------------------------------
struct TlEmbedArgs {
    TlCommon c;
    const float* theta;
    const float* obs;                  // [S][n][O]
    const uint8_t* actions;            // [S][n] or nullptr
    float* X;                          // [S][lpb][D]
};
__global__ __launch_bounds__(TNT) void tl_embed_kernel(TlEmbedArgs a) {
    const DtqnNet& net = a.c.net;
    const int D = net.d_model, O = net.obs_dim, adim = net.action_dim, KE = net.ke, n = a.c.n;
    const int s = (int)blockIdx.x / (a.c.lpb / TROWS), rb = (int)blockIdx.x % (a.c.lpb / TROWS);
    const float* __restrict__ theta = a.theta;
    const float* obs_rows = a.obs + (size_t)s * n * O;
    const uint8_t* act_rows = a.actions != nullptr ? a.actions + (size_t)s * n : nullptr;
    float* xo = a.X + ((size_t)s * a.c.lpb + rb * TROWS) * D;
    for (int idx = (int)threadIdx.x; idx < TROWS * D; idx += TNT) {
        const int rl = idx / D, d = idx - rl * D, r = rb * TROWS + rl;
        float v = 0.f;
        if (r < n) {
            if (d < adim) {
                if (n == 1) v = theta[net.off_act_emb + (int)act_rows[0] * adim + d];
                else if (r > 0) v = theta[net.off_act_emb + (int)act_rows[r - 1] * adim + d];
            } else {
                const float* w = theta + net.off_obs_w + (size_t)(d - adim) * KE;
                float acc = theta[net.off_obs_b + d - adim];
                if (net.discrete) {
                    for (int j = 0; j < O; ++j) {
                        int tok = (int)obs_rows[(size_t)r * O + j];
                        tok = tok < 0 ? 0 : (tok >= net.vocab ? net.vocab - 1 : tok);
                        const float* e = theta + net.off_obs_tab + tok * net.embed_per_obs;
                        for (int cdim = 0; cdim < net.embed_per_obs; ++cdim) acc = fmaf(e[cdim], w[j * net.embed_per_obs + cdim], acc);
                    }
                } else {
                    for (int k = 0; k < KE; ++k) acc = fmaf(obs_rows[(size_t)r * O + k], w[k], acc);
                }
                v = acc;
            }
            v += theta[net.off_pos + r * D + d];
        }
        xo[idx] = v;
    }
}

// ---- linear: OUT[rows][N] (op)= IN[rows][K] * W[N][K]^T + b ------------------------------------------
//   mode 0: OUT = acc + b      mode 1: OUT = relu(acc + b)      mode 2: OUT += relu(acc + b)   (residual gate)
struct TlLinearArgs {
    const float* in;   int ldi;        // input tensor, row stride (floats); rows are global row indices
    const float* W;    int K, N;       // W [N][K]
    const float* bias;
    float* out;        int ldo;
    int mode;
};
template <int D>
__global__ __launch_bounds__(TNT) void tl_linear_kernel(TlLinearArgs a) {
    constexpr int LDT = D + 4;
    float* Xt = reinterpret_cast<float*>(dtqn_smem);                   // [64][LDT] input tile of the current K chunk
    const Thr t = make_thr();
    const size_t row0 = (size_t)blockIdx.x * TROWS;
    const int ntile = (int)blockIdx.y * TNW + t.wave;                 // this wave's 16-column output tile
    const int col = ntile * 16 + t.i;
    const bool live = col < a.N;
    f32x4 acc[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) acc[m] = zero4();
    float4 bf[2][D / 16];
    const float* wrow = a.W + (size_t)(live ? col : 0) * a.K;
    frag_xwT_fetch<D>(bf[0], wrow, t);
    const int nchunks = a.K / D;
    for (int kc = 0; kc < nchunks; ++kc) {
        __syncthreads();                                              // previous chunk's tile fully consumed
        for (int idx = t.tid; idx < TROWS * (D / 4); idx += TNT) {
            const int r = idx / (D / 4), c = (idx - r * (D / 4)) * 4;
            st4(Xt + r * LDT + c, ld4(a.in + (row0 + r) * a.ldi + (size_t)kc * D + c));
        }
        if (kc + 1 < nchunks) frag_xwT_fetch<D>(bf[(kc + 1) & 1], wrow + (size_t)(kc + 1) * D, t);
        __syncthreads();
        if (kc & 1) frag_xwT_mma<D, 4>(Xt, LDT, bf[1], t, acc);
        else frag_xwT_mma<D, 4>(Xt, LDT, bf[0], t, acc);
    }
    if (live) {
        const float b = a.bias != nullptr ? a.bias[col] : 0.f;
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                float* op = a.out + (row0 + m * 16 + t.kq * 4 + r4) * a.ldo + col;
                const float v = acc[m][r4] + b;
                if (a.mode == 0) *op = v;
                else if (a.mode == 1) *op = fmaxf(v, 0.f);
                else *op += fmaxf(v, 0.f);
            }
    }
}

// ---- attention per (sequence, head) -------------------------------------------------------------------
struct TlAttnArgs {
    const float* qkv;                  // [S][lpb][3D]
    float* o;                          // [S][lpb][D]
    int D, lpb, n;
};
template <int HD>
__global__ __launch_bounds__(256) void tl_attn_kernel(TlAttnArgs a) {
    constexpr int LDH = 3 * HD + 4;
    float* T = reinterpret_cast<float*>(dtqn_smem);                    // [lpb][q | k | v] of this head
    const Thr t = make_thr();
    const int s = (int)blockIdx.x, h = (int)blockIdx.y;
    const float* src = a.qkv + (size_t)s * a.lpb * 3 * a.D;
    for (int idx = t.tid; idx < a.lpb * 3 * (HD / 4); idx += 256) {
        const int r = idx / (3 * (HD / 4)), rem = idx - r * (3 * (HD / 4));
        const int which = rem / (HD / 4), c = (rem - which * (HD / 4)) * 4;
        st4(T + r * LDH + which * HD + c, ld4(src + (size_t)r * 3 * a.D + which * a.D + h * HD + c));
    }
    __syncthreads();
    attention_forward<HD, 4>(T, LDH, HD, 1, a.lpb, a.n, nullptr, t);   // one head: "D" = HD, H = 1
    __syncthreads();
    float* dst = a.o + (size_t)s * a.lpb * a.D + h * HD;
    for (int idx = t.tid; idx < a.lpb * (HD / 4); idx += 256) {
        const int r = idx / (HD / 4), c = (idx - r * (HD / 4)) * 4;
        st4(dst + (size_t)r * a.D + c, ld4(T + r * LDH + c));
    }
}

// ---- LayerNorm over 64-row blocks -----------------------------------------------------------------------
struct TlLnArgs {
    const float* src;
    float* dst;
    const float* gamma;
    const float* beta;
};
template <int D>
__global__ __launch_bounds__(TNT) void tl_layernorm_kernel(TlLnArgs a) {
    const Thr t = make_thr();
    const size_t row0 = (size_t)blockIdx.x * TROWS;
    layernorm_rows<D, TNW>(a.src + row0 * D, a.dst + row0 * D, D, TROWS, a.gamma, a.beta, nullptr, t);
}

// ---- Q = HH W2^T + b2 -----------------------------------------------------------------------------------
struct TlQArgs {
    const float* hh;                   // [S][lpb][D]
    const float* W2;
    const float* b2;
    float* q;                          // [S][n][A]
    int D, A, lpb, n, S;
};
__global__ __launch_bounds__(256) void tl_qhead_kernel(TlQArgs a) {
    const int total = a.S * a.n * a.A;
    for (int idx = (int)(blockIdx.x * 256 + threadIdx.x); idx < total; idx += (int)gridDim.x * 256) {
        const int ac = idx % a.A, r = (idx / a.A) % a.n, s = idx / (a.A * a.n);
        const float* hrow = a.hh + ((size_t)s * a.lpb + r) * a.D;
        const float* w = a.W2 + (size_t)ac * a.D;
        float acc = a.b2[ac];
        for (int k = 0; k < a.D; k += 4) {
            const float4 hv = ld4(hrow + k), wv = ld4(w + k);
            acc = fmaf(hv.x, wv.x, acc); acc = fmaf(hv.y, wv.y, acc); acc = fmaf(hv.z, wv.z, acc); acc = fmaf(hv.w, wv.w, acc);
        }
        a.q[idx] = acc;
    }
}

// ---- host orchestration -------------------------------------------------------------------------------------
template <int D>
static int launch_linear(const TlLinearArgs& a, int row_blocks, hipStream_t stream) {
    const size_t lds = (size_t)TROWS * (D + 4) * sizeof(float);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tl_linear_kernel<D>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipGetLastError();
    hipLaunchKernelGGL((tl_linear_kernel<D>), dim3(row_blocks, (a.N + 16 * TNW - 1) / (16 * TNW)), dim3(TNT), lds, stream, a);
    return hipGetLastError() == hipSuccess ? DTQN_OK : DTQN_ERR_LAUNCH;
}
template <int D>
static int launch_ln(const TlLnArgs& a, int row_blocks, hipStream_t stream) {
    (void)hipGetLastError();
    hipLaunchKernelGGL((tl_layernorm_kernel<D>), dim3(row_blocks), dim3(TNT), 0, stream, a);
    return hipGetLastError() == hipSuccess ? DTQN_OK : DTQN_ERR_LAUNCH;
}
template <int HD>
static int launch_attn(const TlAttnArgs& a, int S, int H, hipStream_t stream) {
    const size_t lds = (size_t)a.lpb * (3 * HD + 4) * sizeof(float);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tl_attn_kernel<HD>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipGetLastError();
    hipLaunchKernelGGL((tl_attn_kernel<HD>), dim3(S, H), dim3(256), lds, stream, a);
    return hipGetLastError() == hipSuccess ? DTQN_OK : DTQN_ERR_LAUNCH;
}

template <int D>
static int forward_tiled(const DtqnNet& net, const float* theta, const float* obs, const uint8_t* actions, int S, int n,
                         float* q_out, float* ws, hipStream_t stream) {
    const int lpb = net.lp, H = net.num_heads, HD = net.head_dim;
    const size_t R = (size_t)S * lpb;
    const int RB = (int)(R / TROWS);
    float* X = ws;
    float* U = X + R * D;
    float* QKV = U + R * D;
    float* O = QKV + R * 3 * D;
    float* HID = O + R * D;
    int rc;
    {
        TlEmbedArgs e;
        e.c.net = net; e.c.S = S; e.c.n = n; e.c.lpb = lpb;
        e.theta = theta; e.obs = obs; e.actions = actions; e.X = X;
        (void)hipGetLastError();
        hipLaunchKernelGGL(tl_embed_kernel, dim3(RB), dim3(TNT), 0, stream, e);
        if (hipGetLastError() != hipSuccess) return DTQN_ERR_LAUNCH;
    }
    const bool ident = net.identity != 0;
    for (int l = 0; l < net.num_layers; ++l) {
        const float* th = theta + net.off_layer0 + (size_t)l * net.layer_stride;
        const float* src = X;
        if (ident) {
            TlLnArgs ln{X, U, th + net.lo_ln1_w, th + net.lo_ln1_b};
            if ((rc = launch_ln<D>(ln, RB, stream)) != DTQN_OK) return rc;
            src = U;
        }
        TlLinearArgs qkv{src, D, th + net.lo_in_w, D, 3 * D, th + net.lo_in_b, QKV, 3 * D, 0};
        if ((rc = launch_linear<D>(qkv, RB, stream)) != DTQN_OK) return rc;
        TlAttnArgs at{QKV, O, D, lpb, n};
        if (HD == 8) rc = launch_attn<8>(at, S, H, stream);
        else if (HD == 16) rc = launch_attn<16>(at, S, H, stream);
        else if (HD == 32) rc = launch_attn<32>(at, S, H, stream);
        else rc = DTQN_ERR_CONFIG;
        if (rc != DTQN_OK) return rc;
        TlLinearArgs outp{O, D, th + net.lo_out_w, D, D, th + net.lo_out_b, X, D, 2};          // x += relu(o W_o^T + b)
        if ((rc = launch_linear<D>(outp, RB, stream)) != DTQN_OK) return rc;
        if (!ident) {
            TlLnArgs ln{X, X, th + net.lo_ln1_w, th + net.lo_ln1_b};
            if ((rc = launch_ln<D>(ln, RB, stream)) != DTQN_OK) return rc;
            src = X;
        } else {
            TlLnArgs ln{X, U, th + net.lo_ln2_w, th + net.lo_ln2_b};
            if ((rc = launch_ln<D>(ln, RB, stream)) != DTQN_OK) return rc;
            src = U;
        }
        TlLinearArgs f1{src, D, th + net.lo_f1_w, D, 4 * D, th + net.lo_f1_b, HID, 4 * D, 1};
        if ((rc = launch_linear<D>(f1, RB, stream)) != DTQN_OK) return rc;
        TlLinearArgs f2{HID, 4 * D, th + net.lo_f2_w, 4 * D, D, th + net.lo_f2_b, X, D, 2};
        if ((rc = launch_linear<D>(f2, RB, stream)) != DTQN_OK) return rc;
        if (!ident) {
            TlLnArgs ln{X, X, th + net.lo_ln2_w, th + net.lo_ln2_b};
            if ((rc = launch_ln<D>(ln, RB, stream)) != DTQN_OK) return rc;
        }
    }
    TlLinearArgs h1{X, D, theta + net.off_head1_w, D, D, theta + net.off_head1_b, U, D, 1};
    if ((rc = launch_linear<D>(h1, RB, stream)) != DTQN_OK) return rc;
    TlQArgs qa{U, theta + net.off_head2_w, theta + net.off_head2_b, q_out, D, net.num_actions, lpb, n, S};
    const int total = S * n * net.num_actions;
    (void)hipGetLastError();
    hipLaunchKernelGGL(tl_qhead_kernel, dim3((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048), dim3(256), 0, stream, qa);
    return hipGetLastError() == hipSuccess ? DTQN_OK : DTQN_ERR_LAUNCH;
}

}  // namespace dtqn

using namespace dtqn;

extern "C" int dtqn_forward_workspace_floats(const DtqnNet* net, int batch) {
    if (!net || batch < 1) return 0;
    const long long R = (long long)batch * net->lp;
    const long long fl = R * net->d_model * 10;       // X, U, QKV (3), O, HID (4)
    return fl < 0x7fffffffLL ? (int)fl : 0;
}

extern "C" int dtqn_forward_tiled(const DtqnNet* net, const float* theta, const float* obs, const uint8_t* actions,
                                  int batch, int n, float* q_out, float* workspace, void* stream) {
    if (!net || !theta || !obs || !q_out || !workspace || batch < 1) return DTQN_ERR_ARG;
    if (n < 1 || n > net->ctx_len) return DTQN_ERR_ARG;                 // dtqn.py:170-173
    if (net->action_dim > 0 && !actions) return DTQN_ERR_ARG;
    if (!net->tiled || net->gate != DTQN_GATE_RES) return DTQN_ERR_CONFIG;
    hipStream_t s = (hipStream_t)stream;
    switch (net->d_model) {
        case 64: return forward_tiled<64>(*net, theta, obs, actions, batch, n, q_out, workspace, s);
        case 128: return forward_tiled<128>(*net, theta, obs, actions, batch, n, q_out, workspace, s);
        case 256: return forward_tiled<256>(*net, theta, obs, actions, batch, n, q_out, workspace, s);
        default: return DTQN_ERR_CONFIG;
    }
}
