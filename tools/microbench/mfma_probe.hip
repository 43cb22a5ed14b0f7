// Microbenchmark: what bounds the row-block GEMM loop (tl_ffn_kernel's structure: 8 waves, 32 or 64 rows, a 128-wide contraction chunk
// per phase, A fragments out of LDS, weight fragments out of L2 one phase ahead, two barriers per chunk)?  The same loop with single
// ingredients removed: F_W weight fetches, F_A LDS reads of the A operand, F_BAR barriers, F_EPI the epilogue between the phases.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Idtqn_amd/csrc tools/microbench/mfma_probe.hip -o tools/microbench/mfma_probe
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <hip/hip_runtime.h>
#include "dtqn_device.hpp"
#include "dtqn_frag16.hpp"
using namespace dtqn;
extern "C" void* dtqn_debug_profile_buffer(void) { return nullptr; }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

enum { F_W = 1, F_A = 2, F_BAR = 4, F_EPI = 8, F_FENCE = 16, F_PAIR = 32, F_PACK = 64 };
// A operand from registers instead of LDS (same MFMA count and accumulator pattern as frag16_mma)
template <int K, int MG>
__device__ __forceinline__ void reg_mma(const float4 (&bf)[8], float x, f32x4 (&acc)[MG]) {
#pragma unroll
    for (int s = 0; s < K / 16; ++s) {
        const float b4[4] = {bf[s].x, bf[s].y, bf[s].z, bf[s].w};
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int m = 0; m < MG; ++m) acc[m] = mfma16(x + (float)m, b4[c], acc[m]);
    }
}
template <int MR, int FLAGS, int WPS>
__global__ __launch_bounds__(512, WPS) void probe_kernel(const float* __restrict__ W1, const float* __restrict__ W2, const float* __restrict__ in, float* out, int chunks) {
    constexpr int D = 128, MT = MR / 16, LDX = D + 4, LDH = 132, HID = 512;
    float* Xt = reinterpret_cast<float*>(dtqn_smem);
    float* Hs = Xt + MR * LDX;
    const Thr t = make_thr();
    const int wc = t.wave * 16 + t.i;
    auto fetchA = [&](float4 (&bf)[8], int j) {
        const float* wr = W1 + (size_t)((j & 3) * 128 + wc) * D + t.kq * 4;
        // F_PACK: the same 8 KB of this wave's fragment, laid out fragment-major (q, lane, 4 floats): every load instruction reads 1 KB contiguous
        const float* wpk = W1 + (size_t)((j & 3) * 8 + t.wave) * 2048 + t.lane * 4;
#pragma unroll
        for (int q = 0; q < 8; ++q) bf[q] = (FLAGS & F_W) ? ((FLAGS & F_PACK) ? ld4(wpk + 256 * q) : ld4(wr + 16 * q)) : make_float4(1.f + q, 2.f, 3.f, 4.f);
    };
    auto fetchB = [&](float4 (&bf)[8], int j) {
        const float* wr = W2 + (size_t)wc * HID + (j & 3) * 128 + t.kq * 4;
        const float* wpk = W2 + (size_t)((j & 3) * 8 + t.wave) * 2048 + t.lane * 4;
#pragma unroll
        for (int q = 0; q < 8; ++q) bf[q] = (FLAGS & F_W) ? ((FLAGS & F_PACK) ? ld4(wpk + 256 * q) : ld4(wr + 16 * q)) : make_float4(1.f, 2.f + q, 3.f, 4.f);
    };
    float4 bf0[8], bf1[8];
    fetchA(bf0, 0);
    for (int idx = t.tid; idx < MR * (D / 4); idx += 512) {
        const int r = idx / (D / 4), c = (idx - r * (D / 4)) * 4;
        st4(Xt + r * LDX + c, ld4(in + ((size_t)blockIdx.x * MR + r) * D + c));
    }
    f32x4 accO[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) accO[m] = zero4();
    __syncthreads();
    for (int j = 0; j < chunks; ++j) {
        f32x4 accA[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) accA[m] = zero4();
        fetchB(bf1, j);
        if (FLAGS & F_FENCE) DTQN_SCHED_FENCE();                      // the fetch is ISSUED here, not where the register allocator finds room for it
        if ((FLAGS & F_A) && (FLAGS & F_PAIR) && MT == 4) {
            frag16_mma<128, 2, 8>(Xt, LDX, bf0, t, reinterpret_cast<f32x4(&)[2]>(accA[0]));
            frag16_mma<128, 2, 8>(Xt + 32 * LDX, LDX, bf0, t, reinterpret_cast<f32x4(&)[2]>(accA[MT > 2 ? 2 : 0]));
        } else
        if (FLAGS & F_A) frag16_mma<128, MT, 8>(Xt, LDX, bf0, t, accA);
        else reg_mma<128, MT>(bf0, (float)t.lane, accA);
        if (FLAGS & F_EPI) {
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) Hs[(m * 16 + t.kq * 4 + r4) * LDH + wc] = fmaxf(accA[m][r4], 0.f);
        } else {
#pragma unroll
            for (int m = 0; m < MT; ++m) accO[m][0] += accA[m][1];
        }
        if (FLAGS & F_BAR) __syncthreads();
        fetchA(bf0, j + 1);
        if (FLAGS & F_FENCE) DTQN_SCHED_FENCE();
        if ((FLAGS & F_A) && (FLAGS & F_PAIR) && MT == 4) {
            frag16_mma<128, 2, 8>(Hs, LDH, bf1, t, reinterpret_cast<f32x4(&)[2]>(accO[0]));
            frag16_mma<128, 2, 8>(Hs + 32 * LDH, LDH, bf1, t, reinterpret_cast<f32x4(&)[2]>(accO[MT > 2 ? 2 : 0]));
        } else
        if (FLAGS & F_A) frag16_mma<128, MT, 8>(Hs, LDH, bf1, t, accO);
        else reg_mma<128, MT>(bf1, (float)t.lane, accO);
        if (FLAGS & F_BAR) __syncthreads();
    }
    float v = 0.f;
#pragma unroll
    for (int m = 0; m < MT; ++m) v += accO[m][0] + accO[m][1] + accO[m][2] + accO[m][3];
    out[(size_t)blockIdx.x * 512 + t.tid] = v + bf0[0].x;
}

template <int MR, int FLAGS, int WPS>
static void run(const char* what, int nblocks, int chunks, const float* W1, const float* W2, const float* in, float* out) {
    const size_t lds = (size_t)MR * (132 + 132) * 4;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&probe_kernel<MR, FLAGS, WPS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((probe_kernel<MR, FLAGS, WPS>), dim3(nblocks), dim3(512), lds, 0, W1, W2, in, out, chunks);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, 0));
    const int reps = 20;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((probe_kernel<MR, FLAGS, WPS>), dim3(nblocks), dim3(512), lds, 0, W1, W2, in, out, chunks);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / reps, gflop = (double)nblocks * MR * chunks * 2.0 * 2.0 * 128 * 128 / 1e9;
    printf("%-58s MR=%d wg/CU<=%d blocks=%d chunks=%d: %8.1f us  %6.1f TFLOP/s (%.3f of 157.3)\n", what, MR, WPS / 2, nblocks, chunks, us, gflop / us * 1e3, gflop / us * 1e3 / 157.3);
}

int main() {
    float *W1, *W2, *in, *out;
    const int maxblocks = 4096;
    CK(hipMalloc(&W1, 512 * 128 * 4)); CK(hipMalloc(&W2, 512 * 128 * 4)); CK(hipMalloc(&in, (size_t)maxblocks * 64 * 128 * 4)); CK(hipMalloc(&out, (size_t)maxblocks * 512 * 4));
    std::vector<float> h((size_t)maxblocks * 64 * 128);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 2001) * 1e-3f - 1.0f;
    CK(hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(W1, h.data(), 512 * 128 * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(W2, h.data() + 70000, 512 * 128 * 4, hipMemcpyHostToDevice));
    // config-4 geometry: 1536 workgroups of 32 rows, 4 chunks each; then the same work as 512 workgroups with 12 chunks (no tails)
    printf("--- 1536 x 4 chunks (config 4 forward geometry, 2 workgroups per CU)\n");
    run<32, F_W | F_A | F_BAR | F_EPI, 4>("everything (the tl_ffn loop)", 1536, 4, W1, W2, in, out);
    run<32, F_W | F_A | F_BAR | F_EPI | F_FENCE, 4>("everything + scheduling fence behind the fetch", 1536, 4, W1, W2, in, out);
    run<32, F_W | F_A | F_BAR | F_EPI | F_PACK, 4>("everything, fragment-major weights (1 KB per load)", 1536, 4, W1, W2, in, out);
    run<32, F_W | F_A | F_BAR | F_EPI | F_PACK | F_FENCE, 4>("everything, fragment-major weights + fence", 1536, 4, W1, W2, in, out);
    run<32, F_A | F_BAR | F_EPI, 4>("no weight fetches", 1536, 4, W1, W2, in, out);
    run<32, F_W | F_BAR | F_EPI, 4>("no LDS reads of A", 1536, 4, W1, W2, in, out);
    run<32, F_W | F_A | F_EPI, 4>("no barriers", 1536, 4, W1, W2, in, out);
    run<32, F_W | F_A | F_BAR, 4>("no epilogue (LDS writes)", 1536, 4, W1, W2, in, out);
    run<32, 0, 4>("MFMAs only", 1536, 4, W1, W2, in, out);
    printf("--- 512 x 12 chunks (one round, long workgroups)\n");
    run<32, F_W | F_A | F_BAR | F_EPI, 4>("everything", 512, 12, W1, W2, in, out);
    run<32, F_A | F_BAR | F_EPI, 4>("no weight fetches", 512, 12, W1, W2, in, out);
    run<32, F_W | F_BAR | F_EPI, 4>("no LDS reads of A", 512, 12, W1, W2, in, out);
    run<32, F_W | F_A | F_EPI, 4>("no barriers", 512, 12, W1, W2, in, out);
    run<32, 0, 4>("MFMAs only", 512, 12, W1, W2, in, out);
    printf("--- 64-row workgroups: 768 x 4 chunks, 256 x 12 chunks (one per CU), 512 x 12\n");
    run<64, F_W | F_A | F_BAR | F_EPI, 4>("everything", 768, 4, W1, W2, in, out);
    run<64, F_W | F_A | F_BAR | F_EPI | F_FENCE, 4>("everything + fence", 768, 4, W1, W2, in, out);
    run<64, F_W | F_A | F_BAR | F_EPI | F_PACK, 4>("everything, fragment-major weights", 768, 4, W1, W2, in, out);
    run<64, F_W | F_A | F_BAR | F_EPI | F_PACK | F_FENCE | F_PAIR, 4>("everything, fragment-major weights + fence + pairs", 768, 4, W1, W2, in, out);
    run<64, F_W | F_A | F_BAR | F_EPI | F_FENCE | F_PAIR, 4>("everything + fence + row tiles in pairs", 768, 4, W1, W2, in, out);
    run<64, F_W | F_A | F_BAR | F_EPI | F_PAIR, 4>("everything + row tiles in pairs", 768, 4, W1, W2, in, out);
    run<64, F_W | F_A | F_BAR | F_EPI, 4>("everything", 512, 12, W1, W2, in, out);
    run<64, F_A | F_BAR | F_EPI, 4>("no weight fetches", 512, 12, W1, W2, in, out);
    run<64, F_W | F_BAR | F_EPI, 4>("no LDS reads of A", 512, 12, W1, W2, in, out);
    run<64, F_W | F_A | F_EPI, 4>("no barriers", 512, 12, W1, W2, in, out);
    run<64, 0, 4>("MFMAs only", 512, 12, W1, W2, in, out);
    run<64, F_W | F_A | F_BAR | F_EPI, 2>("everything, 256 registers (1 workgroup per CU)", 256, 24, W1, W2, in, out);
    return 0;
}
