// Image observation embedding of DTQN on gfx950: five 3x3 convolutions + Flatten + Linear (dtqn/networks/representations.py:77-130,
// reached from dtqn/networks/dtqn.py:71-77 when obs_dim is a (C, H, W) tuple; MiniHack pixel crops, envs/mini_hack.py:18-76).
//
//   Conv(C, 64, s2) ReLU  Conv(64, 64) ReLU  Conv(64, 64, s2) ReLU  Conv(64, 128) ReLU  Conv(128, 128, s2) ReLU  Flatten  Linear(128 h5 w5, D)
//
// Implicit GEMMs on the exact-f32 matrix core (v_mfma_f32_16x16x4_f32), no im2col buffer:
//   * activations are NHWC ([token][pixel][channel]) so that the channel axis -- the contraction of every tap -- is contiguous;
//   * a workgroup owns 64 output pixels x all output channels; for each of the 9 taps the 64 shifted input rows go through LDS once
//     and the tap's [Cout][Cin] weight slab comes from a per-update transposed copy (`wprep`, written by img_prep_kernel from the
//     reference-layout parameters in theta), read as 16-byte fragments straight out of L2;
//   * the first convolution (C <= 3) contracts its 9 C inputs in one step (the reference layout [Cout][C][3][3] IS that GEMM's
//     weight matrix), gathering uint8 pixels from the replay;
//   * Flatten + Linear is the same kernel with one "tap" per spatial position of the last feature map and tokens as output rows;
//   * data gradients are the same kernel with the transposed slabs and the scatter pattern of the stride turned into a gather;
//   * weight gradients contract over pixels (dY^T X per tap) with both operands loaded 16 B / lane along their channel axes,
//     split over pixel ranges into partials that a second kernel sums in a fixed order (deterministic: no atomics) and writes
//     back in the reference layout.
// Bound: f32 MFMA (2 * 9 * Cin * Cout FLOP per output pixel; 787 MFLOP per 144 x 144 token forward).
#include "dtqn_device.hpp"
#include "dtqn_frag16.hpp"

namespace dtqn {

constexpr int IT = 256, IW = 4;          // threads / waves per workgroup
constexpr int IROWS = 64;                // output rows (pixels or tokens) per workgroup

struct ImgLayer {
    int cin, cout, stride, hi, wi, ho, wo;
};
__host__ __device__ inline ImgLayer img_layer(const DtqnNet& n, int l) {
    switch (l) {
        case 0: return {n.img_c, 64, 2, n.img_h, n.img_w, n.img_h1, n.img_w1};
        case 1: return {64, 64, 1, n.img_h1, n.img_w1, n.img_h1, n.img_w1};
        case 2: return {64, 64, 2, n.img_h1, n.img_w1, n.img_h3, n.img_w3};
        case 3: return {64, 128, 1, n.img_h3, n.img_w3, n.img_h3, n.img_w3};
        default: return {128, 128, 2, n.img_h3, n.img_w3, n.img_h5, n.img_w5};
    }
}
static inline int img_off_w(const DtqnNet& n, int l) { return l == 0 ? n.off_cw0 : l == 1 ? n.off_cw1 : l == 2 ? n.off_cw2 : l == 3 ? n.off_cw3 : n.off_cw4; }
static inline int img_off_b(const DtqnNet& n, int l) { return l == 0 ? n.off_cb0 : l == 1 ? n.off_cb1 : l == 2 ? n.off_cb2 : l == 3 ? n.off_cb3 : n.off_cb4; }

// ---- wprep layout (floats) -----------------------------------------------------------------------------------------
struct ImgPrepMap {
    long long w0;            // [64][K1]                      first convolution, zero padded
    long long fwd[5];        // l = 1..4: [9][cout][cin]      forward slabs
    long long dgr[5];        // l = 1..4: [9][cin][cout]      data-gradient slabs
    long long lin_f;         // [P][D][128]                   Linear forward, one slab per spatial position
    long long lin_d;         // [P][128][D]
    long long total;
};
static inline ImgPrepMap img_prep_map(const DtqnNet& n) {
    ImgPrepMap m;
    long long pos = 0;
    m.w0 = pos; pos += 64LL * n.img_k1;
    m.fwd[0] = m.dgr[0] = -1;
    for (int l = 1; l < 5; ++l) {
        const ImgLayer L = img_layer(n, l);
        m.fwd[l] = pos; pos += 9LL * L.cout * L.cin;
        m.dgr[l] = pos; pos += 9LL * L.cin * L.cout;
    }
    const long long P = (long long)n.img_h5 * n.img_w5, DO = n.d_model;
    m.lin_f = pos; pos += P * DO * 128;
    m.lin_d = pos; pos += P * 128 * DO;
    m.total = pos;
    return m;
}
struct ImgActMap {
    long long y[5];
    long long total;
};
static inline ImgActMap img_act_map(const DtqnNet& n, long long tokens) {
    ImgActMap m;
    long long pos = 0;
    for (int l = 0; l < 5; ++l) {
        const ImgLayer L = img_layer(n, l);
        m.y[l] = pos;
        pos += tokens * L.ho * L.wo * L.cout;
    }
    m.total = pos;
    return m;
}

// ---- theta -> wprep ------------------------------------------------------------------------------------------------
struct ImgPrepArgs {
    const float* theta;
    float* wprep;
    long long dst;           // offset of this piece in wprep
    int src;                 // offset of the source tensor in theta
    int kind;                // 0: first conv (pad K), 1: forward slabs, 2: dgrad slabs, 3: linear forward, 4: linear dgrad
    int cin, cout, k1, P, DO;
    long long n;
};
__global__ __launch_bounds__(IT) void img_prep_kernel(ImgPrepArgs a) {
    const long long idx = (long long)blockIdx.x * IT + threadIdx.x;
    if (idx >= a.n) return;
    const float* W = a.theta + a.src;
    float v;
    if (a.kind == 0) {                       // dst [co][k1]  <-  W[co][cin * 9] (k = ci * 9 + ky * 3 + kx), zero padded
        const int co = (int)(idx / a.k1), k = (int)(idx % a.k1);
        v = k < a.cin * 9 ? W[(size_t)co * a.cin * 9 + k] : 0.f;
    } else if (a.kind == 1) {                // dst [tap][co][ci]  <-  W[co][ci][tap]
        const int ci = (int)(idx % a.cin), co = (int)((idx / a.cin) % a.cout), tap = (int)(idx / ((long long)a.cin * a.cout));
        v = W[((size_t)co * a.cin + ci) * 9 + tap];
    } else if (a.kind == 2) {                // dst [tap][ci][co]
        const int co = (int)(idx % a.cout), ci = (int)((idx / a.cout) % a.cin), tap = (int)(idx / ((long long)a.cin * a.cout));
        v = W[((size_t)co * a.cin + ci) * 9 + tap];
    } else if (a.kind == 3) {                // dst [p][d][c]  <-  W_e[d][c * P + p]   (Flatten of NCHW, representations.py:128)
        const int c = (int)(idx % 128), d = (int)((idx / 128) % a.DO), p = (int)(idx / (128LL * a.DO));
        v = W[(size_t)d * 128 * a.P + (size_t)c * a.P + p];
    } else {                                 // dst [p][c][d]
        const int d = (int)(idx % a.DO), c = (int)((idx / a.DO) % 128), p = (int)(idx / (128LL * a.DO));
        v = W[(size_t)d * 128 * a.P + (size_t)c * a.P + p];
    }
    a.wprep[a.dst + idx] = v;
}

// ---- the implicit GEMM ----------------------------------------------------------------------------------------------
enum { IMG_CONV_FWD = 0, IMG_CONV_DGRAD = 1, IMG_LIN_FWD = 2, IMG_LIN_DGRAD = 3, IMG_CONV1_FWD = 4 };
struct ImgGemmArgs {
    const float* in;           // CONV_FWD: input map; CONV_DGRAD: dL/d(output map); LIN_FWD: last feature map; LIN_DGRAD: dxemb base
    const float* mask;         // CONV_DGRAD: the convolution's own (post-ReLU) output: gradient passes where it is > 0
    const uint8_t* img;        // CONV1_FWD
    const int32_t* img_index;
    const float* W;            // this layer's slabs in wprep
    const float* bias;
    float* out;
    float* out1;               // LIN_FWD: optional second destination
    const int32_t *dst0, *dst1, *dsrc;
    int tokens, hi, wi, ho, wo, stride, C, taps, tiles_per_tok;
    // CONV_DGRAD with stride 2: the input pixels of a workgroup share their row / column parity, so that only the taps that can reach
    // them are walked (1, 2, 2 or 4 of the 9: an even row is reached by ky = 1 only, an odd one by ky = 0 and 2)
    int cls_tiles[4], cls_first[4];          // tiles of parity class c = 2 * (iy & 1) + (ix & 1) and the first tile index of the class
};
// K: contraction per tap, N: output columns.  Wave w owns column tiles w, w + 4, ... (N / 64 of them) x four 16-row tiles.
// DB: two LDS tiles, the next tap's rows and fragments in flight during this tap's MFMAs (one barrier per tap).  It pays where the
// fragments are long (K or N = 128 forward, the Linear: -9 ... -42 % per launch) and costs occupancy where they are short or the
// gather is masked (64 x 64 forward +5 %, data gradients +14 ... +43 %): measured per instantiation, selected in img_gemm().
template <int K, int N, int MODE, bool DB>
__global__ __launch_bounds__(IT) void img_gemm_kernel(ImgGemmArgs a) {
    constexpr int LDA = K + 4, NCT = N / 64;
    static_assert(K % 16 == 0 && N % 64 == 0, "fragment shapes");
    float* As = reinterpret_cast<float*>(dtqn_smem);           // [64][LDA]
    const Thr t = make_thr();
    int tok0 = 0, tile = 0, tap_lo = 0, tap_hi = a.taps;
    if (MODE == IMG_LIN_FWD) {
        tok0 = (int)blockIdx.x * IROWS;
    } else if (MODE == IMG_LIN_DGRAD) {
        tap_lo = (int)blockIdx.x % a.taps; tap_hi = tap_lo + 1;
        tok0 = ((int)blockIdx.x / a.taps) * IROWS;
    } else {
        tok0 = (int)blockIdx.x / a.tiles_per_tok;
        tile = (int)blockIdx.x - tok0 * a.tiles_per_tok;
    }
    // rows of this workgroup: output pixels of token tok0 (convolutions; input pixels for the data gradient) or tokens (linear)
    int npix = MODE == IMG_CONV_DGRAD ? a.hi * a.wi : a.ho * a.wo;
    int py = 0, px = 0, ch = 0, cw = 0;          // stride-2 data gradient: parity class of this workgroup's pixels and its extent
    const bool classes = MODE == IMG_CONV_DGRAD && a.stride == 2;
    if (classes) {
        int c = 3;
        while (c > 0 && tile < a.cls_first[c]) --c;
        tile -= a.cls_first[c];
        py = c >> 1; px = c & 1;
        ch = (a.hi - py + 1) / 2; cw = (a.wi - px + 1) / 2;
        npix = ch * cw;
    }
    f32x4 acc[NCT][4];
#pragma unroll
    for (int c = 0; c < NCT; ++c)
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[c][m] = zero4();
    auto tap_live = [&](int tap) { return !(classes && ((((py + 1 - tap / 3) & 1) != 0) || (((px + 1 - tap % 3) & 1) != 0))); };
    if (MODE == IMG_CONV1_FWD) {
        // one contraction step of K = 9 C (padded): uint8 pixels gathered in the reference's (ci, ky, kx) order
        float4 bf[NCT][K / 16];
#pragma unroll
        for (int c = 0; c < NCT; ++c) frag16_fetch<K>(bf[c], a.W + (size_t)((t.wave + c * IW) * 16 + t.i) * K, t);
        for (int idx = t.tid; idx < IROWS * K; idx += IT) {
            const int r = idx / K, k = idx - r * K, p = tile * IROWS + r;
            float v = 0.f;
            if (p < npix && k < 9 * a.C) {
                const int ci = k / 9, kk = k - ci * 9, qy = kk / 3, qx = kk - qy * 3;
                const int oy = p / a.wo, ox = p - oy * a.wo, iy = oy * a.stride + qy - 1, ix = ox * a.stride + qx - 1;
                if (iy >= 0 && iy < a.hi && ix >= 0 && ix < a.wi)
                    v = (float)a.img[(size_t)a.img_index[tok0] * a.C * a.hi * a.wi + ((size_t)ci * a.hi + iy) * a.wi + ix];
            }
            As[r * LDA + k] = v;
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < NCT; ++c) frag16_mma<K, 4>(As, LDA, bf[c], t, acc[c]);
    } else {
        // Two LDS tiles: the shifted input rows of tap t + 1 (and its weight fragments) are in flight as global loads while tap t
        // multiplies; they are dropped into the other tile behind the MFMAs -- one barrier per tap.
        constexpr int NV = IROWS * (K / 4) / IT;                       // float4 pieces of the 64 x K tile per thread
        static_assert(IROWS * (K / 4) % IT == 0, "tile pieces per thread");
        float* As1 = As + IROWS * LDA;
        auto gather = [&](int tap, int idx) -> float4 {
            const int r = idx / (K / 4), c4 = (idx - r * (K / 4)) * 4;
            const int ky = tap / 3, kx = tap - ky * 3;
            const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
            if (MODE == IMG_CONV_FWD) {
                const int p = tile * IROWS + r;
                if (p >= npix) return z;
                const int oy = p / a.wo, ox = p - oy * a.wo, iy = oy * a.stride + ky - 1, ix = ox * a.stride + kx - 1;
                if (iy < 0 || iy >= a.hi || ix < 0 || ix >= a.wi) return z;
                return ld4(a.in + (((size_t)tok0 * a.hi + iy) * a.wi + ix) * K + c4);
            } else if (MODE == IMG_CONV_DGRAD) {
                const int q = tile * IROWS + r;
                if (q >= npix) return z;
                const int iy = classes ? 2 * (q / cw) + py : q / a.wi, ix = classes ? 2 * (q % cw) + px : q - (q / a.wi) * a.wi;
                const int ty = iy + 1 - ky, tx = ix + 1 - kx;
                if (ty < 0 || tx < 0 || ty % a.stride != 0 || tx % a.stride != 0) return z;
                const int oy = ty / a.stride, ox = tx / a.stride;
                if (oy >= a.ho || ox >= a.wo) return z;
                const size_t at = (((size_t)tok0 * a.ho + oy) * a.wo + ox) * K + c4;
                const float4 g = ld4(a.in + at), y = ld4(a.mask + at);
                return make_float4(y.x > 0.f ? g.x : 0.f, y.y > 0.f ? g.y : 0.f, y.z > 0.f ? g.z : 0.f, y.w > 0.f ? g.w : 0.f);
            } else if (MODE == IMG_LIN_FWD) {
                const int tk = tok0 + r;
                return tk < a.tokens ? ld4(a.in + ((size_t)tk * a.taps + tap) * K + c4) : z;
            } else {   // IMG_LIN_DGRAD
                const int tk = tok0 + r;
                return (tk < a.tokens && a.dsrc[tk] >= 0) ? ld4(a.in + (size_t)a.dsrc[tk] + c4) : z;
            }
        };
        auto next_live = [&](int tap) { while (tap < tap_hi && !tap_live(tap)) ++tap; return tap; };
        if constexpr (!DB) {
            for (int tap = tap_lo; tap < tap_hi; ++tap) {
                if (!tap_live(tap)) continue;
                float4 bf1[NCT][K / 16];                                // the tap's weight fragments go in flight before the staging barrier
#pragma unroll
                for (int c = 0; c < NCT; ++c) frag16_fetch<K>(bf1[c], a.W + (size_t)tap * N * K + (size_t)((t.wave + c * IW) * 16 + t.i) * K, t);
                __syncthreads();                                        // previous tap's tile consumed
                for (int idx = t.tid; idx < IROWS * (K / 4); idx += IT) {
                    const int r = idx / (K / 4), c4 = (idx - r * (K / 4)) * 4;
                    st4(As + r * LDA + c4, gather(tap, idx));
                }
                __syncthreads();
#pragma unroll
                for (int c = 0; c < NCT; ++c) frag16_mma<K, 4>(As, LDA, bf1[c], t, acc[c]);
            }
        } else {
        float4 av[NV], bf[2][NCT][K / 16];
        int tap = next_live(tap_lo), buf = 0;
        if (tap < tap_hi) {
#pragma unroll
            for (int c = 0; c < NCT; ++c) frag16_fetch<K>(bf[0][c], a.W + (size_t)tap * N * K + (size_t)((t.wave + c * IW) * 16 + t.i) * K, t);
#pragma unroll
            for (int v = 0; v < NV; ++v) av[v] = gather(tap, t.tid + v * IT);
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const int idx = t.tid + v * IT, r = idx / (K / 4), c4 = (idx - r * (K / 4)) * 4;
                st4(As + r * LDA + c4, av[v]);
            }
        }
        __syncthreads();
        while (tap < tap_hi) {
            const int nxt = next_live(tap + 1);
            float* cur = buf ? As1 : As;
            float* oth = buf ? As : As1;
            if (nxt < tap_hi) {
                if (buf == 0) {
#pragma unroll
                    for (int c = 0; c < NCT; ++c) frag16_fetch<K>(bf[1][c], a.W + (size_t)nxt * N * K + (size_t)((t.wave + c * IW) * 16 + t.i) * K, t);
                } else {
#pragma unroll
                    for (int c = 0; c < NCT; ++c) frag16_fetch<K>(bf[0][c], a.W + (size_t)nxt * N * K + (size_t)((t.wave + c * IW) * 16 + t.i) * K, t);
                }
#pragma unroll
                for (int v = 0; v < NV; ++v) av[v] = gather(nxt, t.tid + v * IT);
            }
            if (buf == 0) {
#pragma unroll
                for (int c = 0; c < NCT; ++c) frag16_mma<K, 4>(cur, LDA, bf[0][c], t, acc[c]);
            } else {
#pragma unroll
                for (int c = 0; c < NCT; ++c) frag16_mma<K, 4>(cur, LDA, bf[1][c], t, acc[c]);
            }
            if (nxt < tap_hi) {
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    const int idx = t.tid + v * IT, r = idx / (K / 4), c4 = (idx - r * (K / 4)) * 4;
                    st4(oth + r * LDA + c4, av[v]);
                }
            }
            __syncthreads();                     // tile of the next tap published; this tap's tile free again
            tap = nxt;
            buf ^= 1;
        }
        }
    }
    // epilogue: lane (i, kq) holds column ct * 16 + i of rows m * 16 + kq * 4 + r
#pragma unroll
    for (int c = 0; c < NCT; ++c) {
        const int col = (t.wave + c * IW) * 16 + t.i;
        const float bv = (MODE == IMG_CONV_DGRAD || MODE == IMG_LIN_DGRAD) ? 0.f : a.bias[col];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const int r = m * 16 + t.kq * 4 + r4;
                const float v = acc[c][m][r4] + bv;
                if (MODE == IMG_CONV_FWD || MODE == IMG_CONV1_FWD) {
                    const int p = tile * IROWS + r;
                    if (p < npix) a.out[((size_t)tok0 * npix + p) * N + col] = fmaxf(v, 0.f);
                } else if (MODE == IMG_CONV_DGRAD) {
                    const int q = tile * IROWS + r;
                    if (q < npix) {
                        const int pix = classes ? (2 * (q / cw) + py) * a.wi + 2 * (q % cw) + px : q;
                        a.out[((size_t)tok0 * a.hi * a.wi + pix) * N + col] = v;
                    }
                } else if (MODE == IMG_LIN_FWD) {
                    const int tk = tok0 + r;
                    if (tk < a.tokens) {
                        if (a.dst0[tk] >= 0) a.out[(size_t)a.dst0[tk] * N + col] = v;
                        if (a.out1 != nullptr && a.dst1[tk] >= 0) a.out1[(size_t)a.dst1[tk] * N + col] = v;
                    }
                } else {
                    const int tk = tok0 + r;
                    if (tk < a.tokens) a.out[((size_t)tk * a.taps + tap_lo) * N + col] = v;
                }
            }
    }
}

// ---- weight gradients: partial[split][tap][n][k] = sum over the split's rows of dY[row][n] * X_tap[row][k] ------------------
enum { IMG_WG_CONV = 0, IMG_WG_LIN = 1, IMG_WG_CONV1 = 2 };
struct ImgWgradArgs {
    const float* dy;           // CONV / CONV1: dL/d(output map) [tok][ho wo][N];  LIN: dxemb base
    const float* y;            // CONV / CONV1: the layer's output (ReLU mask)
    const float* x;            // CONV: input map [tok][hi wi][K];  LIN: last feature map [tok][P][128]
    const uint8_t* img;
    const int32_t* img_index;
    const int32_t* dsrc;
    float* part;               // [splits][taps][N][Kp]
    float* bpart;              // [splits][N]
    float* direct;             // LIN (one split): the gradient of W_e itself, written in the reference layout [d][c * P + p]
    int tokens, hi, wi, ho, wo, stride, C, taps, N, K, Kp, splits;
    long long rows;            // tokens * ho * wo (convolutions) or tokens (linear)
};
// One workgroup = one 64 x 64 tile of one tap's dW over one row range.  Lane (i, kq) carries 4 consecutive dY columns and 4
// consecutive X columns of row 4 * step + kq; the 16 x 16 x 4 products tile the 64 x 64 block as in dtqn_wgrad_kernel.
template <int MODE>
__global__ __launch_bounds__(IT) void img_wgrad_kernel(ImgWgradArgs a) {
    const Thr t = make_thr();
    const int tiles_n = a.N / 64, tiles_k = (a.Kp + 63) / 64;
    int id = (int)blockIdx.x;
    const int bk = id % tiles_k; id /= tiles_k;
    const int bn = id % tiles_n; id /= tiles_n;
    const int tap = id % a.taps, split = id / a.taps;
    const int ycol = bn * 64 + 4 * t.i, xcol = bk * 64 + 4 * t.i;
    const bool xok = xcol < a.Kp;
    const long long per = ((a.rows + a.splits - 1) / a.splits + 15) / 16 * 16;
    const long long r_lo = (long long)split * per, r_hi = r_lo + per < a.rows ? r_lo + per : a.rows;
    const int ky = tap / 3, kx = tap - ky * 3, npix = a.ho * a.wo;
    f32x4 acc[4][4];
#pragma unroll
    for (int cn = 0; cn < 4; ++cn)
#pragma unroll
        for (int ck = 0; ck < 4; ++ck) acc[cn][ck] = zero4();
    float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    auto load = [&](long long row, float4& av, float4& bv) {
        av = z4; bv = z4;
        if (row >= r_hi) return;
        if (MODE == IMG_WG_LIN) {
            const int tk = (int)row;
            if (a.dsrc[tk] < 0) return;
            av = ld4(a.dy + (size_t)a.dsrc[tk] + ycol);
            if (xok) bv = ld4(a.x + ((size_t)tk * a.taps + tap) * a.K + xcol);
            return;
        }
        const int tk = (int)(row / npix), p = (int)(row - (long long)tk * npix);
        const size_t at = (size_t)row * a.N + ycol;
        const float4 g = ld4(a.dy + at), y = ld4(a.y + at);
        av = make_float4(y.x > 0.f ? g.x : 0.f, y.y > 0.f ? g.y : 0.f, y.z > 0.f ? g.z : 0.f, y.w > 0.f ? g.w : 0.f);
        const int oy = p / a.wo, ox = p - oy * a.wo;
        if (MODE == IMG_WG_CONV) {
            const int iy = oy * a.stride + ky - 1, ix = ox * a.stride + kx - 1;
            if (xok && iy >= 0 && iy < a.hi && ix >= 0 && ix < a.wi) bv = ld4(a.x + (((size_t)tk * a.hi + iy) * a.wi + ix) * a.K + xcol);
        } else {   // first convolution: X column k = ci * 9 + qy * 3 + qx, gathered from the uint8 image
            float xv[4] = {0.f, 0.f, 0.f, 0.f};
            const uint8_t* im = a.img + (size_t)a.img_index[tk] * a.C * a.hi * a.wi;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int k = xcol + c;
                if (k < 9 * a.C) {
                    const int ci = k / 9, kk = k - ci * 9, qy = kk / 3, qx = kk - qy * 3;
                    const int iy = oy * a.stride + qy - 1, ix = ox * a.stride + qx - 1;
                    if (iy >= 0 && iy < a.hi && ix >= 0 && ix < a.wi) xv[c] = (float)im[((size_t)ci * a.hi + iy) * a.wi + ix];
                }
            }
            bv = make_float4(xv[0], xv[1], xv[2], xv[3]);
        }
    };
    // units of 16 rows dealt to the 4 waves, one unit fetched ahead of the one that multiplies
    float4 av[2][4], bv[2][4];
    const long long units = (r_hi - r_lo + 15) / 16;
    auto unit_load = [&](long long u, float4 (&a4)[4], float4 (&b4)[4]) {
#pragma unroll
        for (int k = 0; k < 4; ++k) load(r_lo + u * 16 + 4 * k + t.kq, a4[k], b4[k]);
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
    long long u = t.wave;
    if (u < units) unit_load(u, av[0], bv[0]);
    for (; u < units; u += 2 * IW) {
        if (u + IW < units) unit_load(u + IW, av[1], bv[1]);
        unit_mma(av[0], bv[0]);
        if (u + IW < units) {
            if (u + 2 * IW < units) unit_load(u + 2 * IW, av[0], bv[0]);
            unit_mma(av[1], bv[1]);
        }
    }
    bsum.x += __shfl_xor(bsum.x, 16); bsum.y += __shfl_xor(bsum.y, 16); bsum.z += __shfl_xor(bsum.z, 16); bsum.w += __shfl_xor(bsum.w, 16);
    bsum.x += __shfl_xor(bsum.x, 32); bsum.y += __shfl_xor(bsum.y, 32); bsum.z += __shfl_xor(bsum.z, 32); bsum.w += __shfl_xor(bsum.w, 32);
    // cross-wave sum through LDS (fixed order), then coalesced rows of the partial
    constexpr int SLD = 68;
    float* slab = reinterpret_cast<float*>(dtqn_smem) + (size_t)t.wave * 65 * SLD;
#pragma unroll
    for (int cn = 0; cn < 4; ++cn)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            st4(slab + (4 * (t.kq * 4 + r) + cn) * SLD + 4 * t.i, make_float4(acc[cn][0][r], acc[cn][1][r], acc[cn][2][r], acc[cn][3][r]));
    if (t.kq == 0) st4(slab + 64 * SLD + 4 * t.i, bsum);
    __syncthreads();
    const float* s0 = reinterpret_cast<const float*>(dtqn_smem);
    float* out = a.part + (((size_t)split * a.taps + tap) * a.N) * a.Kp;
    for (int idx = t.tid; idx < 64 * 16; idx += IT) {
        const int nl = idx >> 4, k4 = (idx & 15) * 4, n = bn * 64 + nl, k = bk * 64 + k4;
        if (k < a.Kp) {
            float4 v = ld4(s0 + nl * SLD + k4);
#pragma unroll
            for (int wv = 1; wv < IW; ++wv) {
                const float4 x = ld4(s0 + (size_t)wv * 65 * SLD + nl * SLD + k4);
                v.x += x.x; v.y += x.y; v.z += x.z; v.w += x.w;
            }
            if (MODE == IMG_WG_LIN && a.direct != nullptr) {
                const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int c = 0; c < 4; ++c) a.direct[((size_t)n * a.K + k + c) * a.taps + tap] = vv[c];
            } else {
                st4(out + (size_t)n * a.Kp + k, v);
            }
        }
    }
    if (tap == 0 && bk == 0 && t.tid < 64) {
        float v = 0.f;
#pragma unroll
        for (int wv = 0; wv < IW; ++wv) v += s0[(size_t)wv * 65 * SLD + 64 * SLD + t.tid];
        a.bpart[(size_t)split * a.N + bn * 64 + t.tid] = v;
    }
}

// ---- weight gradients of the 64- / 128-channel convolutions: operands through LDS, taps dealt to the waves ------------------------
// The per-tap kernel above reads dY, the ReLU mask and the shifted input once PER TAP from global memory (27 map reads per layer:
// HBM-bound, 9.9 ms per launch at 1632 tokens of 72 x 72).  Here a workgroup walks 8 x 8 output patches: the patch's masked dY
// [64 pixels][64 channels] and the (7 s + 3)^2 input halo it touches are staged in LDS ONCE, and the nine taps are dealt to the four
// waves (wave w owns taps w, w + 4, w + 8): every wave contracts all 64 pixels of the patch for its taps out of LDS, keeping its
// 64 x 64 tap tiles in registers across the whole patch range.  No cross-wave reduction; one partial per (split, tap).
struct ImgWgradLdsArgs {
    const float* dy;           // dL/d(output map) [tok][ho wo][N]
    const float* y;            // the layer's output (ReLU mask)
    const float* x;            // input map [tok][hi wi][K]
    float* part;               // [splits][9][N][K]
    float* bpart;              // [splits][N]
    int tokens, hi, wi, ho, wo, N, K, splits, py_n, px_n;      // py_n x px_n patches per token
    long long patches;
};
template <int S>
__global__ __launch_bounds__(IT) void img_wgrad_lds_kernel(ImgWgradLdsArgs a) {
    constexpr int HW = 7 * S + 3, LDT = 68;
    float* dYs = reinterpret_cast<float*>(dtqn_smem);          // [64][LDT]
    float* Xs = dYs + 64 * LDT;                                // [HW * HW][LDT]
    const Thr t = make_thr();
    const int tiles_k = a.K / 64, tiles_n = a.N / 64;
    int id = (int)blockIdx.x;
    const int bk = id % tiles_k; id /= tiles_k;
    const int bn = id % tiles_n; id /= tiles_n;
    const int split = id;
    const long long per = (a.patches + a.splits - 1) / a.splits;
    const long long p_lo = (long long)split * per, p_hi = p_lo + per < a.patches ? p_lo + per : a.patches;
    const int ppt = a.py_n * a.px_n;
    // work units of this wave: unit u = (tap u / 2, half u % 2 of the 64 dY columns: cn in {2 h, 2 h + 1}); 18 units dealt round-robin
    // -> 5, 5, 4, 4 per wave (whole taps would be 3, 2, 2, 2: the busiest wave sets the pace)
    constexpr int NU = 5;
    int toff[NU], uh[NU], utap[NU];
    bool ulive[NU];
#pragma unroll
    for (int j = 0; j < NU; ++j) {
        const int u = t.wave + 4 * j;
        ulive[j] = u < 18;
        utap[j] = u >> 1; uh[j] = u & 1;
        toff[j] = ((utap[j] / 3) * HW + utap[j] % 3) * LDT;
    }
    f32x4 acc[NU][2][4];
#pragma unroll
    for (int j = 0; j < NU; ++j)
#pragma unroll
        for (int cn = 0; cn < 2; ++cn)
#pragma unroll
            for (int ck = 0; ck < 4; ++ck) acc[j][cn][ck] = zero4();
    float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    for (long long p = p_lo; p < p_hi; ++p) {
        const int tk = (int)(p / ppt), pp = (int)(p - (long long)tk * ppt);
        const int oy0 = (pp / a.px_n) * 8, ox0 = (pp % a.px_n) * 8;
        __syncthreads();                                        // previous patch consumed
        for (int idx = t.tid; idx < 64 * 16; idx += IT) {
            const int r = idx >> 4, c4 = (idx & 15) * 4, oy = oy0 + (r >> 3), ox = ox0 + (r & 7);
            float4 v = z4;
            if (oy < a.ho && ox < a.wo) {
                const size_t at = (((size_t)tk * a.ho + oy) * a.wo + ox) * a.N + bn * 64 + c4;
                const float4 g = ld4(a.dy + at), yv = ld4(a.y + at);
                v = make_float4(yv.x > 0.f ? g.x : 0.f, yv.y > 0.f ? g.y : 0.f, yv.z > 0.f ? g.z : 0.f, yv.w > 0.f ? g.w : 0.f);
            }
            st4(dYs + r * LDT + c4, v);
        }
        for (int idx = t.tid; idx < HW * HW * 16; idx += IT) {
            const int r = idx >> 4, c4 = (idx & 15) * 4, iy = oy0 * S - 1 + r / HW, ix = ox0 * S - 1 + r % HW;
            float4 v = z4;
            if (iy >= 0 && iy < a.hi && ix >= 0 && ix < a.wi) v = ld4(a.x + (((size_t)tk * a.hi + iy) * a.wi + ix) * a.K + bk * 64 + c4);
            st4(Xs + r * LDT + c4, v);
        }
        __syncthreads();
#pragma unroll 4
        for (int step = 0; step < 16; ++step) {
            const int pl = 4 * step + t.kq, oy = pl >> 3, ox = pl & 7;
            const float4 a4 = ld4(dYs + pl * LDT + 4 * t.i);
            const float aa[4] = {a4.x, a4.y, a4.z, a4.w};
            if (t.wave == 0) { bsum.x += a4.x; bsum.y += a4.y; bsum.z += a4.z; bsum.w += a4.w; }
            const float* xb = Xs + ((oy * S) * HW + ox * S) * LDT + 4 * t.i;
#pragma unroll
            for (int j = 0; j < NU; ++j) {
                if (!ulive[j]) continue;
                const float4 b4 = ld4(xb + toff[j]);
                const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
                const float a0 = uh[j] ? aa[2] : aa[0], a1 = uh[j] ? aa[3] : aa[1];
#pragma unroll
                for (int ck = 0; ck < 4; ++ck) {
                    acc[j][0][ck] = mfma16(a0, bb[ck], acc[j][0][ck]);
                    acc[j][1][ck] = mfma16(a1, bb[ck], acc[j][1][ck]);
                }
            }
        }
    }
    // acc[j][c][ck][r]: n = 4 * (kq * 4 + r) + 2 h + c, k = 4 * i + ck of tap utap[j]
#pragma unroll
    for (int j = 0; j < NU; ++j) {
        if (!ulive[j]) continue;
        float* out = a.part + ((((size_t)split * 9 + utap[j]) * a.N + bn * 64) * a.K) + bk * 64;
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                st4(out + (size_t)(4 * (t.kq * 4 + r) + 2 * uh[j] + c) * a.K + 4 * t.i,
                    make_float4(acc[j][c][0][r], acc[j][c][1][r], acc[j][c][2][r], acc[j][c][3][r]));
    }
    if (t.wave == 0 && bk == 0) {
        bsum.x += __shfl_xor(bsum.x, 16); bsum.y += __shfl_xor(bsum.y, 16); bsum.z += __shfl_xor(bsum.z, 16); bsum.w += __shfl_xor(bsum.w, 16);
        bsum.x += __shfl_xor(bsum.x, 32); bsum.y += __shfl_xor(bsum.y, 32); bsum.z += __shfl_xor(bsum.z, 32); bsum.w += __shfl_xor(bsum.w, 32);
        if (t.kq == 0) st4(a.bpart + (size_t)split * a.N + bn * 64 + 4 * t.i, bsum);
    }
}

// partial sums -> flat gradient, back in the reference layout
struct ImgReduceArgs {
    const float* part;
    const float* bpart;
    float* grad;
    int off_w, off_b, kind;    // kind 0: first conv, 1: conv [co][ci][tap], 2: linear [d][c * P + p]
    int taps, N, K, Kp, splits, C;
    long long n;
};
__global__ __launch_bounds__(IT) void img_reduce_kernel(ImgReduceArgs a) {
    // 64 elements per workgroup, the splits of an element summed by 4 threads (every 4th split each) and folded in a fixed order
    // through LDS: a single thread per element walked up to 768 partials in one dependent chain (1.5 ms for the 64 x 64 layers)
    __shared__ float red[IT];
    const int el = (int)threadIdx.x & 63, q = (int)threadIdx.x >> 6;
    const long long idx = (long long)blockIdx.x * 64 + el;
    if (blockIdx.x * 64 + threadIdx.x < (unsigned)a.N && threadIdx.x < 64) {
        float v = 0.f;
        for (int s = 0; s < a.splits; ++s) v += a.bpart[(size_t)s * a.N + idx];
        a.grad[a.off_b + idx] = v;
    }
    int tap = 0, n = 0, k = 0;
    size_t dst = 0;
    const bool live = idx < a.n;
    if (live) {
        if (a.kind == 0) {             // idx over [co][9 C]
            n = (int)(idx / (9 * a.C)); k = (int)(idx % (9 * a.C)); tap = 0;
            dst = (size_t)idx;
        } else {                       // idx over the partial's [tap][n][k] (reads coalesced over the splits); one scattered write:
            k = (int)(idx % a.K); n = (int)((idx / a.K) % a.N); tap = (int)(idx / ((long long)a.K * a.N));
            dst = a.kind == 1 ? ((size_t)n * a.K + k) * 9 + tap                      // conv  [co][ci][ky][kx]
                              : ((size_t)n * a.K + k) * a.taps + tap;                 // W_e   [d][c * P + p]
        }
    }
    float v = 0.f;
    if (live)
        for (int s = q; s < a.splits; s += 4) v += a.part[(((size_t)s * a.taps + tap) * a.N + n) * a.Kp + k];
    red[threadIdx.x] = v;
    __syncthreads();
    if (live && q == 0) a.grad[a.off_w + dst] = ((red[el] + red[64 + el]) + red[128 + el]) + red[192 + el];
}

// token lists of a TD update
struct ImgListArgs {
    const int32_t *ep_idx, *start;
    int32_t *pol_index, *pol_dst0, *pol_dst1, *pol_dsrc, *tgt_index, *tgt_dst0;
    int batch, L, lpb, T1, D, a_dim;
    long long grd_stride, go_dx0;
};
__global__ __launch_bounds__(IT) void img_lists_kernel(ImgListArgs a) {
    const int idx = (int)blockIdx.x * IT + threadIdx.x;
    const int L1 = a.L + 1;
    if (idx < a.batch * L1) {
        const int b = idx / L1, r = idx - b * L1;
        a.pol_index[idx] = a.ep_idx[b] * a.T1 + a.start[b] + r;
        // row r of the window: position r of policy(o) (pass 0) and position r - 1 of policy(o') (pass 1)
        a.pol_dst0[idx] = r < a.L ? (0 * a.batch + b) * a.lpb + r : -1;
        a.pol_dst1[idx] = r >= 1 ? (1 * a.batch + b) * a.lpb + (r - 1) : -1;
        a.pol_dsrc[idx] = r < a.L ? (int32_t)((long long)b * a.grd_stride + a.go_dx0 + (long long)r * a.D + a.a_dim) : -1;
    }
    if (idx < a.batch * a.L) {
        const int b = idx / a.L, r = idx - b * a.L;
        a.tgt_index[idx] = a.ep_idx[b] * a.T1 + a.start[b] + r + 1;
        a.tgt_dst0[idx] = (2 * a.batch + b) * a.lpb + r;
    }
}

// ---- host side ------------------------------------------------------------------------------------------------------
#define IMG_LAUNCH(kernel, grid, lds, stream, args)                                        \
    do {                                                                                   \
        (void)hipGetLastError();                                                           \
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(IT), lds, stream, args);               \
        if (hipGetLastError() != hipSuccess) return DTQN_ERR_LAUNCH;                       \
    } while (0)

template <int K, int N, int MODE>
static int img_gemm(const ImgGemmArgs& a, int grid, hipStream_t s) {
    constexpr bool DB = MODE == IMG_LIN_FWD || (MODE == IMG_CONV_FWD && (K == 128 || N == 128));
    const size_t lds = (size_t)(DB ? 2 : 1) * IROWS * (K + 4) * sizeof(float);
    static size_t attr_lds[kMaxDevices] = {};      // per instantiation and device
    raise_lds_limit(reinterpret_cast<const void*>(&img_gemm_kernel<K, N, MODE, DB>), lds, attr_lds);
    IMG_LAUNCH((img_gemm_kernel<K, N, MODE, DB>), grid, lds, s, a);
    return DTQN_OK;
}

static int img_conv_fwd(const DtqnNet& net, int l, const float* in, const float* W, const float* bias, float* out, int tokens, hipStream_t s) {
    const ImgLayer L = img_layer(net, l);
    ImgGemmArgs a = {};
    a.in = in; a.W = W; a.bias = bias; a.out = out;
    a.tokens = tokens; a.hi = L.hi; a.wi = L.wi; a.ho = L.ho; a.wo = L.wo; a.stride = L.stride; a.taps = 9;
    a.tiles_per_tok = (L.ho * L.wo + IROWS - 1) / IROWS;
    const int grid = tokens * a.tiles_per_tok;
    if (L.cin == 64 && L.cout == 64) return img_gemm<64, 64, IMG_CONV_FWD>(a, grid, s);
    if (L.cin == 64 && L.cout == 128) return img_gemm<64, 128, IMG_CONV_FWD>(a, grid, s);
    return img_gemm<128, 128, IMG_CONV_FWD>(a, grid, s);
}
static int img_conv_dgrad(const DtqnNet& net, int l, const float* dy, const float* y, const float* Wd, float* dx, int tokens, hipStream_t s) {
    const ImgLayer L = img_layer(net, l);
    ImgGemmArgs a = {};
    a.in = dy; a.mask = y; a.W = Wd; a.out = dx;
    a.tokens = tokens; a.hi = L.hi; a.wi = L.wi; a.ho = L.ho; a.wo = L.wo; a.stride = L.stride; a.taps = 9;
    a.tiles_per_tok = (L.hi * L.wi + IROWS - 1) / IROWS;
    if (L.stride == 2) {
        int first = 0;
        for (int c = 0; c < 4; ++c) {
            const int ch = (L.hi - (c >> 1) + 1) / 2, cw = (L.wi - (c & 1) + 1) / 2;
            a.cls_first[c] = first;
            a.cls_tiles[c] = (ch * cw + IROWS - 1) / IROWS;
            first += a.cls_tiles[c];
        }
        a.tiles_per_tok = first;
    }
    const int grid = tokens * a.tiles_per_tok;
    // contraction over the convolution's output channels, columns = its input channels
    if (L.cin == 64 && L.cout == 64) return img_gemm<64, 64, IMG_CONV_DGRAD>(a, grid, s);
    if (L.cin == 64 && L.cout == 128) return img_gemm<128, 64, IMG_CONV_DGRAD>(a, grid, s);
    return img_gemm<128, 128, IMG_CONV_DGRAD>(a, grid, s);
}

static int img_splits(long long rows, int tile_groups) {
    // enough workgroups for a few rounds of the 256 CUs, at least 4 K rows each
    long long want = 2048 / (tile_groups > 0 ? tile_groups : 1);
    if (want < 1) want = 1;
    long long cap = rows / 4096;
    if (cap < 1) cap = 1;
    return (int)(want < cap ? want : cap);
}

}  // namespace dtqn

using namespace dtqn;

extern "C" int dtqn_img_prep_floats(const DtqnNet* net) {
    if (!net || net->img_c <= 0) return 0;
    const long long n = img_prep_map(*net).total;
    return n < 0x7fffffffLL ? (int)n : 0;
}
extern "C" long long dtqn_img_act_floats(const DtqnNet* net, int tokens) {
    if (!net || net->img_c <= 0 || tokens < 1) return 0;
    return img_act_map(*net, tokens).total;
}
extern "C" long long dtqn_img_gact_floats(const DtqnNet* net, int tokens) {
    if (!net || net->img_c <= 0 || tokens < 1) return 0;
    long long mx = 0;
    for (int l = 0; l < 5; ++l) {
        const ImgLayer L = img_layer(*net, l);
        const long long n = (long long)tokens * L.ho * L.wo * L.cout;
        if (n > mx) mx = n;
    }
    return 2 * mx;
}
static void img_wgrad_plan(const DtqnNet& net, int which, long long& part_floats, int& splits, long long rows) {
    // which 0: first convolution (per-tap kernel, one "tap"), 1..4: LDS-staged kernel over 8 x 8 patches, 5: linear (per-tap kernel)
    int taps, N, Kp;
    if (which == 5) { taps = net.img_h5 * net.img_w5; N = net.d_model; Kp = 128; }
    else { const ImgLayer L = img_layer(net, which); taps = which == 0 ? 1 : 9; N = L.cout; Kp = which == 0 ? net.img_k1 : L.cin; }
    if (which >= 1 && which <= 4) {
        const ImgLayer L = img_layer(net, which);
        const long long tokens = rows / ((long long)L.ho * L.wo);
        const long long patches = tokens * ((L.ho + 7) / 8) * ((L.wo + 7) / 8);
        long long want = 512 / ((N / 64) * (Kp / 64)), cap = patches / 4;       // two workgroups per CU
        if (cap < 1) cap = 1;
        splits = (int)(want < cap ? want : cap);
    } else {
        const int groups = taps * (N / 64) * ((Kp + 63) / 64);
        splits = which == 5 ? 1 : img_splits(rows, groups);
    }
    part_floats = (long long)splits * taps * N * Kp;
}
extern "C" long long dtqn_img_wpart_floats(const DtqnNet* net) {
    if (!net || net->img_c <= 0) return 0;
    // sized for the largest plan at any token count: splits <= 2048 / groups, so splits * groups <= 2048 tiles of 64 x 64 (+ biases)
    long long mx = 0;
    for (int w = 0; w < 6; ++w) {
        long long pf; int sp;
        img_wgrad_plan(*net, w, pf, sp, 1LL << 40);
        if (pf > mx) mx = pf;
    }
    return mx + 2048LL * 256;      // + the bias partials ([splits][N], splits <= 2048)
}

extern "C" int dtqn_img_prep(const DtqnNet* net, const float* theta, float* wprep, void* stream) {
    if (!net || net->img_c <= 0 || !theta || !wprep) return DTQN_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const ImgPrepMap m = img_prep_map(*net);
    auto run = [&](int kind, long long dst, int src, int cin, int cout, long long n) -> int {
        ImgPrepArgs a;
        a.theta = theta; a.wprep = wprep; a.dst = dst; a.src = src; a.kind = kind; a.cin = cin; a.cout = cout;
        a.k1 = net->img_k1; a.P = net->img_h5 * net->img_w5; a.DO = net->d_model; a.n = n;
        IMG_LAUNCH(img_prep_kernel, (unsigned)((n + IT - 1) / IT), 0, s, a);
        return DTQN_OK;
    };
    int rc;
    if ((rc = run(0, m.w0, net->off_cw0, net->img_c, 64, 64LL * net->img_k1)) != DTQN_OK) return rc;
    for (int l = 1; l < 5; ++l) {
        const ImgLayer L = img_layer(*net, l);
        if ((rc = run(1, m.fwd[l], img_off_w(*net, l), L.cin, L.cout, 9LL * L.cin * L.cout)) != DTQN_OK) return rc;
        if ((rc = run(2, m.dgr[l], img_off_w(*net, l), L.cin, L.cout, 9LL * L.cin * L.cout)) != DTQN_OK) return rc;
    }
    const long long P = (long long)net->img_h5 * net->img_w5;
    if ((rc = run(3, m.lin_f, net->off_obs_w, 0, 0, P * net->d_model * 128)) != DTQN_OK) return rc;
    return run(4, m.lin_d, net->off_obs_w, 0, 0, P * net->d_model * 128);
}

extern "C" int dtqn_img_encode(const DtqnNet* net, const float* theta, const float* wprep, const uint8_t* images_u8, const int32_t* img_index,
                               int tokens, float* act, float* out0, const int32_t* dst0, float* out1, const int32_t* dst1, void* stream) {
    if (!net || net->img_c <= 0 || !theta || !wprep || !images_u8 || !img_index || !act || !out0 || !dst0 || tokens < 1) return DTQN_ERR_ARG;
    if (out1 != nullptr && !dst1) return DTQN_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const ImgPrepMap pm = img_prep_map(*net);
    const ImgActMap am = img_act_map(*net, tokens);
    int rc;
    {
        const ImgLayer L = img_layer(*net, 0);
        ImgGemmArgs a = {};
        a.img = images_u8; a.img_index = img_index; a.W = wprep + pm.w0; a.bias = theta + net->off_cb0; a.out = act + am.y[0];
        a.tokens = tokens; a.hi = L.hi; a.wi = L.wi; a.ho = L.ho; a.wo = L.wo; a.stride = L.stride; a.C = net->img_c; a.taps = 1;
        a.tiles_per_tok = (L.ho * L.wo + IROWS - 1) / IROWS;
        const int grid = tokens * a.tiles_per_tok;
        if (net->img_k1 == 16) rc = img_gemm<16, 64, IMG_CONV1_FWD>(a, grid, s);
        else rc = img_gemm<32, 64, IMG_CONV1_FWD>(a, grid, s);
        if (rc != DTQN_OK) return rc;
    }
    for (int l = 1; l < 5; ++l)
        if ((rc = img_conv_fwd(*net, l, act + am.y[l - 1], wprep + pm.fwd[l], theta + img_off_b(*net, l), act + am.y[l], tokens, s)) != DTQN_OK) return rc;
    ImgGemmArgs a = {};
    a.in = act + am.y[4]; a.W = wprep + pm.lin_f; a.bias = theta + net->off_obs_b; a.out = out0; a.out1 = out1; a.dst0 = dst0; a.dst1 = dst1;
    a.tokens = tokens; a.taps = net->img_h5 * net->img_w5;
    const int grid = (tokens + IROWS - 1) / IROWS;
    switch (net->d_model) {
        case 64: return img_gemm<128, 64, IMG_LIN_FWD>(a, grid, s);
        case 128: return img_gemm<128, 128, IMG_LIN_FWD>(a, grid, s);
        case 256: return img_gemm<128, 256, IMG_LIN_FWD>(a, grid, s);
        default: return DTQN_ERR_CONFIG;
    }
}

extern "C" int dtqn_img_backward(const DtqnNet* net, const float* theta, const float* wprep, const uint8_t* images_u8, const int32_t* img_index,
                                 int tokens, const float* act, const float* dxemb_base, const int32_t* dsrc, float* gact, float* wpart,
                                 float* grad_out, void* stream) {
    if (!net || net->img_c <= 0 || !theta || !wprep || !images_u8 || !img_index || !act || !dxemb_base || !dsrc || !gact || !wpart || !grad_out ||
        tokens < 1)
        return DTQN_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const ImgPrepMap pm = img_prep_map(*net);
    const ImgActMap am = img_act_map(*net, tokens);
    const long long half = dtqn_img_gact_floats(net, tokens) / 2;
    float* gA = gact;
    float* gB = gact + half;
    const int P = net->img_h5 * net->img_w5, DO = net->d_model;
    int rc;
    // weight-gradient pass of one layer: partials, then the reduce into grad_out
    auto wgrad = [&](int which, const float* dy, const float* y, const float* x) -> int {
        ImgWgradArgs a = {};
        long long pf;
        int splits;
        a.dy = dy; a.y = y; a.x = x; a.img = images_u8; a.img_index = img_index; a.dsrc = dsrc; a.tokens = tokens;
        int off_w, off_b, kind;
        if (which == 5) {
            a.taps = P; a.N = DO; a.K = 128; a.Kp = 128; a.rows = tokens; a.ho = a.wo = 1;
            a.direct = grad_out + net->off_obs_w;          // one split: no partial, no reduce pass over 2.65 M elements
            off_w = net->off_obs_w; off_b = net->off_obs_b; kind = 2;
        } else {
            const ImgLayer L = img_layer(*net, which);
            a.hi = L.hi; a.wi = L.wi; a.ho = L.ho; a.wo = L.wo; a.stride = L.stride; a.C = net->img_c;
            a.taps = which == 0 ? 1 : 9; a.N = L.cout; a.K = which == 0 ? 9 * net->img_c : L.cin; a.Kp = which == 0 ? net->img_k1 : L.cin;
            a.rows = (long long)tokens * L.ho * L.wo;
            off_w = img_off_w(*net, which); off_b = img_off_b(*net, which); kind = which == 0 ? 0 : 1;
        }
        img_wgrad_plan(*net, which, pf, splits, a.rows);
        a.splits = splits;
        a.part = wpart;
        a.bpart = wpart + pf;
        if (which >= 1 && which <= 4) {
            const ImgLayer L = img_layer(*net, which);
            ImgWgradLdsArgs w = {};
            w.dy = dy; w.y = y; w.x = x; w.part = wpart; w.bpart = wpart + pf;
            w.tokens = tokens; w.hi = L.hi; w.wi = L.wi; w.ho = L.ho; w.wo = L.wo; w.N = L.cout; w.K = L.cin; w.splits = splits;
            w.py_n = (L.ho + 7) / 8; w.px_n = (L.wo + 7) / 8;
            w.patches = (long long)tokens * w.py_n * w.px_n;
            const int hw = 7 * L.stride + 3;
            const size_t lds2 = ((size_t)64 + (size_t)hw * hw) * 68 * sizeof(float);
            const int grid2 = splits * (L.cout / 64) * (L.cin / 64);
            static size_t attr2[2][kMaxDevices] = {};
            if (L.stride == 1) {
                raise_lds_limit(reinterpret_cast<const void*>(&img_wgrad_lds_kernel<1>), lds2, attr2[0]);
                IMG_LAUNCH((img_wgrad_lds_kernel<1>), grid2, lds2, s, w);
            } else {
                raise_lds_limit(reinterpret_cast<const void*>(&img_wgrad_lds_kernel<2>), lds2, attr2[1]);
                IMG_LAUNCH((img_wgrad_lds_kernel<2>), grid2, lds2, s, w);
            }
        } else {
        const int grid = splits * a.taps * (a.N / 64) * ((a.Kp + 63) / 64);
        const size_t lds = (size_t)IW * 65 * 68 * sizeof(float);
        static size_t attr_lds[3][kMaxDevices] = {};
        if (which == 5) {
            raise_lds_limit(reinterpret_cast<const void*>(&img_wgrad_kernel<IMG_WG_LIN>), lds, attr_lds[0]);
            IMG_LAUNCH((img_wgrad_kernel<IMG_WG_LIN>), grid, lds, s, a);
        } else if (which == 0) {
            raise_lds_limit(reinterpret_cast<const void*>(&img_wgrad_kernel<IMG_WG_CONV1>), lds, attr_lds[1]);
            IMG_LAUNCH((img_wgrad_kernel<IMG_WG_CONV1>), grid, lds, s, a);
        } else {
            raise_lds_limit(reinterpret_cast<const void*>(&img_wgrad_kernel<IMG_WG_CONV>), lds, attr_lds[2]);
            IMG_LAUNCH((img_wgrad_kernel<IMG_WG_CONV>), grid, lds, s, a);
        }
        }
        ImgReduceArgs r = {};
        r.part = wpart; r.bpart = wpart + pf; r.grad = grad_out; r.off_w = off_w; r.off_b = off_b; r.kind = kind;
        r.taps = a.taps; r.N = a.N; r.K = a.K; r.Kp = a.Kp; r.splits = splits; r.C = net->img_c;
        r.n = which == 5 ? 0 : (long long)a.N * a.K * a.taps;      // (linear: weights already in place, biases only)
        const long long nel = r.n > a.N ? r.n : a.N;
        IMG_LAUNCH(img_reduce_kernel, (unsigned)((nel + 63) / 64), 0, s, r);
        return DTQN_OK;
    };
    // Linear: dfeat = dxemb W_e (per spatial position), dW_e, db_e
    {
        ImgGemmArgs a = {};
        a.in = dxemb_base; a.dsrc = dsrc; a.W = wprep + pm.lin_d; a.out = gA; a.tokens = tokens; a.taps = P;
        const int grid = P * ((tokens + IROWS - 1) / IROWS);
        switch (DO) {
            case 64: rc = img_gemm<64, 128, IMG_LIN_DGRAD>(a, grid, s); break;
            case 128: rc = img_gemm<128, 128, IMG_LIN_DGRAD>(a, grid, s); break;
            case 256: rc = img_gemm<256, 128, IMG_LIN_DGRAD>(a, grid, s); break;
            default: return DTQN_ERR_CONFIG;
        }
        if (rc != DTQN_OK) return rc;
        if ((rc = wgrad(5, dxemb_base, nullptr, act + am.y[4])) != DTQN_OK) return rc;
    }
    // convolutions 5..2: weight gradients from (masked dY, input map), then the data gradient for the layer below
    float* gy = gA;
    float* gx = gB;
    for (int l = 4; l >= 1; --l) {
        if ((rc = wgrad(l, gy, act + am.y[l], act + am.y[l - 1])) != DTQN_OK) return rc;
        if ((rc = img_conv_dgrad(*net, l, gy, act + am.y[l], wprep + pm.dgr[l], gx, tokens, s)) != DTQN_OK) return rc;
        float* tmp = gy; gy = gx; gx = tmp;
    }
    return wgrad(0, gy, act + am.y[0], nullptr);
}

extern "C" int dtqn_img_td_lists(const DtqnNet* net, const DtqnReplay* rp, const DtqnTd* td, int32_t* pol_index, int32_t* pol_dst0, int32_t* pol_dst1,
                                 int32_t* pol_dsrc, int32_t* tgt_index, int32_t* tgt_dst0, void* stream) {
    if (!net || net->img_c <= 0 || !rp || !td || !pol_index || !pol_dst0 || !pol_dst1 || !pol_dsrc || !tgt_index || !tgt_dst0) return DTQN_ERR_ARG;
    if ((long long)td->batch * net->grd_stride >= 0x7fffffffLL) return DTQN_ERR_CONFIG;      // dsrc holds float offsets as int32
    ImgListArgs a;
    a.ep_idx = td->ep_idx; a.start = td->start;
    a.pol_index = pol_index; a.pol_dst0 = pol_dst0; a.pol_dst1 = pol_dst1; a.pol_dsrc = pol_dsrc; a.tgt_index = tgt_index; a.tgt_dst0 = tgt_dst0;
    a.batch = td->batch; a.L = net->ctx_len; a.lpb = net->lp; a.T1 = rp->max_steps + 1; a.D = net->d_model; a.a_dim = net->action_dim;
    a.grd_stride = net->grd_stride; a.go_dx0 = net->go_dx0;
    const int n = td->batch * (net->ctx_len + 1);
    IMG_LAUNCH(img_lists_kernel, (n + IT - 1) / IT, 0, (hipStream_t)stream, a);
    return DTQN_OK;
}
