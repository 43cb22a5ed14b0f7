// Microbenchmark: the attention stages in isolation, one workgroup per "sequence", LDS-resident tiles, whole-sequence
// shapes (LP = 64, n = 50): head_dim 8 (cfg 1: D = 64, 8 heads) and head_dim 16 (cfg 3: D = 128, 8 heads; the
// backward handles a 64-column head group at a time).  VALU vs matrix-core implementations, 8 waves per workgroup.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Idtqn_amd/csrc tools/microbench/attn_bench.hip
#include <cstdio>
#include "dtqn_device.hpp"
#include "dtqn_bwd_device.hpp"

using namespace dtqn;

extern "C" void* dtqn_debug_profile_buffer(void) { return nullptr; }

template <int VARIANT, int HD, int NW>
__global__ __launch_bounds__(NW * 64) void fwd_kernel(float* out, float* lse, int iters, int n) {
    constexpr int LP = 64, H = 8, D = H * HD, LD = 3 * D + 4;
    float* Ws = reinterpret_cast<float*>(dtqn_smem);
    const Thr t = make_thr();
    for (int idx = t.tid; idx < LP * LD; idx += NW * 64) Ws[idx] = 0.01f * (float)((idx * 37 + blockIdx.x * 11) % 97) - 0.5f;
    __syncthreads();
    float* l = lse + (size_t)blockIdx.x * H * LP;
    for (int it = 0; it < iters; ++it) {
        if (VARIANT == 0) attention_forward_valu<HD, NW>(Ws, LD, D, H, LP, n, l, t);
        else attention_forward_mfma<HD, NW>(Ws, LD, D, H, LP, n, l, t);
        __syncthreads();
    }
    out[blockIdx.x * NW * 64 + t.tid] = Ws[t.tid];
}

template <int VARIANT, int HD, int NW>
__global__ __launch_bounds__(NW * 64) void bwd_kernel(float* out, int iters, int n) {
    constexpr int LP = 64, GW = 64, LD5 = 6 * GW + 4;
    float* W5 = reinterpret_cast<float*>(dtqn_smem);
    float* delta_s = W5 + LP * LD5;
    float* lse_s = delta_s + (GW / HD) * LP;
    const Thr t = make_thr();
    for (int idx = t.tid; idx < LP * LD5; idx += NW * 64) W5[idx] = 0.01f * (float)((idx * 37 + blockIdx.x * 11) % 97) - 0.5f;
    for (int idx = t.tid; idx < (GW / HD) * LP; idx += NW * 64) { delta_s[idx] = 0.01f; lse_s[idx] = 3.0f; }
    __syncthreads();
    for (int it = 0; it < iters; ++it) {
        if (VARIANT == 0) attention_backward_group_valu<HD, NW>(W5, LD5, GW, LP, n, delta_s, lse_s, t);
        else attention_backward_group_mfma<HD, NW>(W5, LD5, GW, LP, n, delta_s, lse_s, t);
        __syncthreads();
    }
    out[blockIdx.x * NW * 64 + t.tid] = W5[t.tid];
}

// clock probe: N MFMAs on four independent accumulators, one wave per SIMD (32 cycles issue each on the f32 matrix core)
__global__ __launch_bounds__(256) void mfma_rate_kernel(float* out, int n) {
    f32x4 a0 = zero4(), a1 = zero4(), a2 = zero4(), a3 = zero4();
    const float x = (float)threadIdx.x * 1e-3f, y = 1.0f + x;
    for (int i = 0; i < n; i += 4) {
        a0 = mfma16(x, y, a0); a1 = mfma16(y, x, a1); a2 = mfma16(x, x, a2); a3 = mfma16(y, y, a3);
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
}

template <typename F>
static float time_us(F launch, int reps) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / reps;
}

template <int VARIANT, int HD>
static float fwd_us(float* out, float* lse, int WG, int iters, int n) {
    const size_t lds = 64 * (3 * 8 * HD + 4) * sizeof(float);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&fwd_kernel<VARIANT, HD, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const float t0 = time_us([&] { hipLaunchKernelGGL((fwd_kernel<VARIANT, HD, 8>), dim3(WG), dim3(512), lds, 0, out, lse, 0, n); }, 20);
    const float t1 = time_us([&] { hipLaunchKernelGGL((fwd_kernel<VARIANT, HD, 8>), dim3(WG), dim3(512), lds, 0, out, lse, iters, n); }, 20);
    return (t1 - t0) / iters;
}
template <int VARIANT, int HD>
static float bwd_us(float* out, int WG, int iters, int n) {
    const size_t lds = (64 * (6 * 64 + 4) + 2 * (64 / HD) * 64) * sizeof(float);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&bwd_kernel<VARIANT, HD, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const float t0 = time_us([&] { hipLaunchKernelGGL((bwd_kernel<VARIANT, HD, 8>), dim3(WG), dim3(512), lds, 0, out, 0, n); }, 20);
    const float t1 = time_us([&] { hipLaunchKernelGGL((bwd_kernel<VARIANT, HD, 8>), dim3(WG), dim3(512), lds, 0, out, iters, n); }, 20);
    return (t1 - t0) / iters;
}

int main() {
    const int WG = 32, iters = 200, n = 50;
    float *out, *lse;
    hipMalloc(&out, WG * 1024 * sizeof(float));
    hipMalloc(&lse, WG * 8 * 64 * sizeof(float));
    const int NM = 40000;
    const float m0 = time_us([&] { hipLaunchKernelGGL(mfma_rate_kernel, dim3(WG), dim3(256), 0, 0, out, 0); }, 20);
    const float m1 = time_us([&] { hipLaunchKernelGGL(mfma_rate_kernel, dim3(WG), dim3(256), 0, 0, out, NM); }, 20);
    printf("v_mfma_f32_16x16x4_f32: %.2f ns each per SIMD => %.2f GHz at 32 cycles\n", (m1 - m0) * 1e3f / NM, 32.0f / ((m1 - m0) * 1e3f / NM));
    printf("attention forward   hd=8  (D=64):   VALU %.3f us   MFMA %.3f us\n", fwd_us<0, 8>(out, lse, WG, iters, n), fwd_us<1, 8>(out, lse, WG, iters, n));
    printf("attention forward   hd=16 (D=128):  VALU %.3f us   MFMA %.3f us\n", fwd_us<0, 16>(out, lse, WG, iters, n), fwd_us<1, 16>(out, lse, WG, iters, n));
    printf("attention backward  hd=8  (64-col group):  VALU %.3f us   MFMA %.3f us\n", bwd_us<0, 8>(out, WG, iters, n), bwd_us<1, 8>(out, WG, iters, n));
    printf("attention backward  hd=16 (64-col group):  VALU %.3f us   MFMA %.3f us\n", bwd_us<0, 16>(out, WG, iters, n), bwd_us<1, 16>(out, WG, iters, n));
    return 0;
}
