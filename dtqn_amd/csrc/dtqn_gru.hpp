// GTrXL GRU gate (dtqn/networks/gates.py:5-31) on one [LP x D] LDS tile, forward and backward.
//   z = sigmoid(W_z y + b_z + U_z x),  r = sigmoid(W_r y + U_r x),  h = tanh(W_g y + U_g (r * x)),
//   out = (1 - z) * x + z * h            x = residual stream, y = relu(sub-layer output)
// The reference passes ONE attn_gate and ONE mlp_gate instance to every layer (dtqn.py:107-131), so the
// six matrices of a gate are shared across layers; their gradients are summed over layers by the
// multi-layer jobs of dtqn_wgrad_kernel.
#pragma once
#include "dtqn_device.hpp"

namespace dtqn {

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }

// Forward.  Xs: stream x (in) -> gate output (out).  Ws: wide LDS tile whose columns [D, 2D) hold y; columns
// [0, D) and [2D, 3D) are scratch (z and r*x).  grec: this gate's record (z, r, h~, r*x, x, y) or nullptr.
// Caller: barrier before (x and y visible) and after (Xs updated).
template <int D, int MT, int NW>
__device__ __forceinline__ void gru_gate_forward(float* Xs, int ldx, float* Ws, int ldw, const float* __restrict__ gw,
                                                 const DtqnNet& net, float* __restrict__ grec, const Thr& t,
                                                 int lpf = MT * 16, int r0 = 0) {
    // row slices: the record tensors are [lpf][D] (whole sequence), this workgroup's LP rows start at row r0
    constexpr int LP = MT * 16;
    constexpr int MG = pick_mg(D / 16, MT, NW);
    using Own = Owned<D, MT, MG, NW>;
    const size_t TS = (size_t)lpf * D;          // one record tensor
    if (grec != nullptr) grec += (size_t)r0 * D;
    float* Z = Ws;
    const float* Y = Ws + D;
    float* RX = Ws + 2 * D;
    if (grec != nullptr) {
        tile_store<NW>(Xs, ldx, grec + 4 * TS, LP, D, t);
        tile_store<NW>(Y, ldw, grec + 5 * TS, LP, D, t);
    }
    const float* __restrict__ Wz = gw + net.go_w_z;
    const float* __restrict__ Uz = gw + net.go_u_z;
    const float* __restrict__ Wr = gw + net.go_w_r;
    const float* __restrict__ Ur = gw + net.go_u_r;
    const float* __restrict__ bz = gw + net.go_b_z;
#pragma unroll
    for (int q = 0; q < Own::PER_WAVE; ++q) {
        if (Own::valid(t.wave, q)) {
            const int nt = Own::nt(t.wave, q), mg = Own::mg(t.wave, q);
            const size_t wrow = (size_t)(nt * 16 + t.i) * D;
            f32x4 az[MG], ar[MG];
#pragma unroll
            for (int m = 0; m < MG; ++m) { az[m] = zero4(); ar[m] = zero4(); }
            float4 f0[D / 16], f1[D / 16];
            frag_xwT_fetch<D>(f0, Wz + wrow, t);
            frag_xwT_fetch<D>(f1, Uz + wrow, t);
            frag_xwT_mma<D, MG>(Y + mg * MG * 16 * ldw, ldw, f0, t, az);
            frag_xwT_fetch<D>(f0, Wr + wrow, t);
            frag_xwT_mma<D, MG>(Xs + mg * MG * 16 * ldx, ldx, f1, t, az);
            frag_xwT_fetch<D>(f1, Ur + wrow, t);
            frag_xwT_mma<D, MG>(Y + mg * MG * 16 * ldw, ldw, f0, t, ar);
            frag_xwT_mma<D, MG>(Xs + mg * MG * 16 * ldx, ldx, f1, t, ar);
            const int c = nt * 16 + t.i;
            const float bzc = bz[c];
#pragma unroll
            for (int m = 0; m < MG; ++m)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const int r = (mg * MG + m) * 16 + t.kq * 4 + r4;
                    const float z = sigmoidf_(az[m][r4] + bzc), rr = sigmoidf_(ar[m][r4]);
                    const float rx = rr * Xs[r * ldx + c];
                    Z[r * ldw + c] = z;
                    RX[r * ldw + c] = rx;
                    if (grec != nullptr) {
                        grec[0 * TS + r * D + c] = z;
                        grec[1 * TS + r * D + c] = rr;
                        grec[3 * TS + r * D + c] = rx;
                    }
                }
        }
    }
    __syncthreads();
    const float* __restrict__ Wg = gw + net.go_w_g;
    const float* __restrict__ Ug = gw + net.go_u_g;
#pragma unroll
    for (int q = 0; q < Own::PER_WAVE; ++q) {
        if (Own::valid(t.wave, q)) {
            const int nt = Own::nt(t.wave, q), mg = Own::mg(t.wave, q);
            const size_t wrow = (size_t)(nt * 16 + t.i) * D;
            f32x4 ah[MG];
#pragma unroll
            for (int m = 0; m < MG; ++m) ah[m] = zero4();
            float4 f0[D / 16], f1[D / 16];
            frag_xwT_fetch<D>(f0, Wg + wrow, t);
            frag_xwT_fetch<D>(f1, Ug + wrow, t);
            frag_xwT_mma<D, MG>(Y + mg * MG * 16 * ldw, ldw, f0, t, ah);
            frag_xwT_mma<D, MG>(RX + mg * MG * 16 * ldw, ldw, f1, t, ah);
            const int c = nt * 16 + t.i;
#pragma unroll
            for (int m = 0; m < MG; ++m)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const int r = (mg * MG + m) * 16 + t.kq * 4 + r4;
                    const float hc = tanhf(ah[m][r4]);
                    const float z = Z[r * ldw + c], x = Xs[r * ldx + c];
                    Xs[r * ldx + c] = (1.0f - z) * x + z * hc;
                    if (grec != nullptr) grec[2 * TS + r * D + c] = hc;
                }
        }
    }
}

// Backward.  DX: dL/d(out) (in) -> dL/dx (out, the skip path of the stream).  T2: receives dL/dy.
// W5: wide LDS scratch with room for five [LP][D] tiles.  grec: the gate's saved record,
// ggrd: its gradient record (dz_pre, dr_pre, dh_pre).  Caller: barrier before and after.
template <int D, int MT, int NW>
__device__ __forceinline__ void gru_gate_backward(float* DX, float* T2, int ldx, float* W5, int ld5,
                                                  const float* __restrict__ gw, const DtqnNet& net,
                                                  const float* __restrict__ grec, float* __restrict__ ggrd, const Thr& t,
                                                  int lpf = MT * 16, int r0 = 0) {
    constexpr int LP = MT * 16;
    constexpr int NT = NW * 64;
    const size_t TS = (size_t)lpf * D;          // one record tensor; this workgroup's rows start at r0
    grec += (size_t)r0 * D;
    ggrd += (size_t)r0 * D;
    constexpr int MG = pick_mg(D / 16, MT, NW);
    using Own = Owned<D, MT, MG, NW>;
    float* A = W5;             // dz_pre
    float* Bt = W5 + D;        // dh_pre
    float* C = W5 + 2 * D;     // dr_pre
    float* XT = W5 + 3 * D;    // x
    float* RT = W5 + 4 * D;    // r
    for (int idx = t.tid; idx < LP * D; idx += NT) {
        const int r = idx / D, c = idx - r * D;
        const float z = grec[0 * TS + idx], rr = grec[1 * TS + idx], hc = grec[2 * TS + idx], x = grec[4 * TS + idx];
        const float g = DX[r * ldx + c];
        const float dzp = g * (hc - x) * z * (1.0f - z);
        const float dhp = g * z * (1.0f - hc * hc);
        A[r * ld5 + c] = dzp;
        Bt[r * ld5 + c] = dhp;
        XT[r * ld5 + c] = x;
        RT[r * ld5 + c] = rr;
        DX[r * ldx + c] = g * (1.0f - z);
        ggrd[0 * TS + idx] = dzp;
        ggrd[2 * TS + idx] = dhp;
    }
    __syncthreads();
    // d(r*x) = dh_pre U_g ;  dr_pre = d(r*x) * x * r(1-r) ;  dx += d(r*x) * r
    {
        StageDyW<D, MT, MG, NW, D / 16> g;
        g.prefetch(gw + net.go_u_g, D, t);
        g.run(Bt, ld5, t, [&](int r, int c, float v) {
            const float x = XT[r * ld5 + c], rr = RT[r * ld5 + c];
            const float drp = v * x * rr * (1.0f - rr);
            C[r * ld5 + c] = drp;
            DX[r * ldx + c] += v * rr;
            ggrd[1 * TS + r * D + c] = drp;
        });
    }
    __syncthreads();
    // dx += dz_pre U_z + dr_pre U_r ;  dy = dh_pre W_g + dz_pre W_z + dr_pre W_r
    const float* mats[5] = {gw + net.go_u_z, gw + net.go_u_r, gw + net.go_w_g, gw + net.go_w_z, gw + net.go_w_r};
    const float* srcs[5] = {A, C, Bt, A, C};
#pragma unroll
    for (int q = 0; q < Own::PER_WAVE; ++q) {
        if (Own::valid(t.wave, q)) {
            const int nt = Own::nt(t.wave, q), mg = Own::mg(t.wave, q);
            f32x4 ax[MG], ay[MG];
#pragma unroll
            for (int m = 0; m < MG; ++m) { ax[m] = zero4(); ay[m] = zero4(); }
            float f[2][D / 4];
            frag_dyw_fetch<D>(f[0], mats[0] + nt * 16 + t.i, D, t);
#pragma unroll
            for (int p = 0; p < 5; ++p) {
                if (p + 1 < 5) frag_dyw_fetch<D>(f[(p + 1) & 1], mats[p + 1] + nt * 16 + t.i, D, t);
                if (p < 2) frag_dyw_mma<D, MG>(srcs[p] + mg * MG * 16 * ld5, ld5, f[p & 1], t, ax);
                else frag_dyw_mma<D, MG>(srcs[p] + mg * MG * 16 * ld5, ld5, f[p & 1], t, ay);
            }
            const int c = nt * 16 + t.i;
#pragma unroll
            for (int m = 0; m < MG; ++m)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const int r = (mg * MG + m) * 16 + t.kq * 4 + r4;
                    DX[r * ldx + c] += ax[m][r4];
                    T2[r * ldx + c] = ay[m][r4];
                }
        }
    }
}

}  // namespace dtqn
