// Microbenchmark: the fused feed-forward block of the row-block path (tl_ffn_kernel, dtqn_tiled.hip) at BASELINE config 4 / 5
// forward shapes, with per-phase clocks (s_memtime at the phase boundaries of every wave's lane 0, kept in LDS and written out at
// the end) -- where a workgroup's time goes between its MFMA phases, epilogues and barriers.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Idtqn_amd/csrc tools/microbench/ffn_bench.hip -o tools/microbench/ffn_bench
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <hip/hip_runtime.h>

__device__ long long* g_stamps;       // [blocks][8 waves][64 slots]
#define TL_NSLOT 64
#ifndef NO_STAMPS
#define TL_STAMP(slot)                                                                                         \
    do {                                                                                                       \
        if ((threadIdx.x & 63) == 0) {                                                                         \
            tl_stamp_lds[(threadIdx.x >> 6) * TL_NSLOT + (slot)] = clock64();                                  \
            if ((slot) == 0) {                                                                                 \
                tl_stamp_lds[(threadIdx.x >> 6) * TL_NSLOT + 60] = wall_clock64();                             \
                tl_stamp_lds[(threadIdx.x >> 6) * TL_NSLOT + 61] = ((long long)__builtin_amdgcn_s_getreg(63508) << 32) | (unsigned)__builtin_amdgcn_s_getreg(63492); \
            }                                                                                                  \
            if ((slot) == 62) tl_stamp_lds[(threadIdx.x >> 6) * TL_NSLOT + 63] = wall_clock64();               \
        }                                                                                                      \
    } while (0)
#define TL_STAMP_BLOCK(blk) const int tl_blk_ = (blk); if ((threadIdx.x & 63) == 0) tl_stamp_lds[(threadIdx.x >> 6) * TL_NSLOT + 59] = wall_clock64()
#define TL_STAMP_FLUSH()                                                                                       \
    do {                                                                                                       \
        if ((threadIdx.x & 63) == 0 && g_stamps != nullptr)                                                    \
            for (int q_ = 0; q_ < TL_NSLOT; ++q_)                                                              \
                g_stamps[((size_t)tl_blk_ * 8 + (threadIdx.x >> 6)) * TL_NSLOT + q_] = tl_stamp_lds[(threadIdx.x >> 6) * TL_NSLOT + q_]; \
    } while (0)
__shared__ long long tl_stamp_lds[8 * TL_NSLOT];
#endif

#include "../../dtqn_amd/csrc/dtqn_tiled.hip"

extern "C" void* dtqn_debug_profile_buffer(void) { return nullptr; }
// symbols the included translation unit expects from the rest of the engine (never called here)
extern "C" int dtqn_td_row_split(const DtqnNet*, int) { return 0; }
extern "C" int dtqn_td_xch_floats(const DtqnNet*, int) { return 0; }
extern "C" int dtqn_td_xch_flags(const DtqnNet*, int) { return 0; }
extern "C" int dtqn_td_wpack(const DtqnNet*, const DtqnTd*, void*) { return 0; }

// tick calibration: N dependent-free MFMAs on one wave per SIMD take 32 N shader cycles
__global__ __launch_bounds__(256) void tick_kernel(long long* out, float* sink, int n) {
    f32x4 a0 = dtqn::zero4(), a1 = dtqn::zero4(), a2 = dtqn::zero4(), a3 = dtqn::zero4();
    const float x = (float)threadIdx.x * 1e-3f, y = 1.0f + x;
    const long long t0 = clock64();
    const long long w0 = wall_clock64();
    for (int i = 0; i < n; i += 4) {
        a0 = dtqn::mfma16(x, y, a0); a1 = dtqn::mfma16(y, x, a1); a2 = dtqn::mfma16(x, x, a2); a3 = dtqn::mfma16(y, y, a3);
    }
    sink[blockIdx.x * 256 + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
    const long long t1 = clock64();
    const long long w1 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = w1 - w0; }
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int D>
static void run(int S, int lpb, int n_save, int reps) {
    const int HID = 4 * D, rpb = lpb / 64;
    const size_t rows = (size_t)S * lpb;
    float *in, *out, *res, *h, *lnout, *st, *W1, *W2, *b1, *b2, *g, *be;
    unsigned long long *mh, *m2;
    CK(hipMalloc(&in, rows * D * 4)); CK(hipMalloc(&out, rows * D * 4)); CK(hipMalloc(&res, rows * D * 4)); CK(hipMalloc(&lnout, rows * D * 4));
    CK(hipMalloc(&h, rows * HID * 4)); CK(hipMalloc(&st, rows * 2 * 4));
    CK(hipMalloc(&mh, rows / 16 * (HID / 16) * 4 * 8)); CK(hipMalloc(&m2, rows / 16 * (D / 16) * 4 * 8));
    CK(hipMalloc(&W1, (size_t)HID * D * 4)); CK(hipMalloc(&W2, (size_t)HID * D * 4)); CK(hipMalloc(&b1, HID * 4)); CK(hipMalloc(&b2, D * 4));
    CK(hipMalloc(&g, D * 4)); CK(hipMalloc(&be, D * 4));
    std::vector<float> hx(rows * D), hw((size_t)HID * D);
    for (size_t i = 0; i < hx.size(); ++i) hx[i] = (float)((i * 2654435761u) % 2001) * 1e-3f - 1.0f;
    for (size_t i = 0; i < hw.size(); ++i) hw[i] = ((float)((i * 40503u) % 2001) * 1e-3f - 1.0f) * 0.05f;
    CK(hipMemcpy(in, hx.data(), hx.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(res, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(W1, hw.data(), hw.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(W2, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(b1, 0, HID * 4)); CK(hipMemset(b2, 0, D * 4)); CK(hipMemset(be, 0, D * 4));
    std::vector<float> ones(D, 1.0f);
    CK(hipMemcpy(g, ones.data(), D * 4, hipMemcpyHostToDevice));
    long long* stamps;
    const int maxblocks = S * rpb * 2;
    CK(hipMalloc(&stamps, (size_t)maxblocks * 8 * TL_NSLOT * 8));
    CK(hipMemset(stamps, 0, (size_t)maxblocks * 8 * TL_NSLOT * 8));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &stamps, sizeof(stamps)));
    dtqn::TlFfnArgs fa = {};
    auto F = [&](float* p, int ld) { return dtqn::Fld{p, (long long)lpb * ld, ld}; };
    fa.in = F(in, D); fa.out = F(out, D); fa.res = fa.in;      // post-LN layer: the residual is the block input (forward_records)
    fa.h = F(h, HID);
    fa.mh = dtqn::Fld{reinterpret_cast<float*>(mh), (long long)lpb / 16 * (HID / 16) * 4 * 2, 0};
    fa.m2 = dtqn::Fld{reinterpret_cast<float*>(m2), (long long)lpb / 16 * (D / 16) * 4 * 2, 0};
    fa.W1a = fa.W1b = W1; fa.W2a = fa.W2b = W2; fa.b1a = fa.b1b = b1; fa.b2a = fa.b2b = b2;
    fa.split = S; fa.rpb = rpb; fa.mode = 2; fa.n_save = n_save;
    fa.ln_out = F(lnout, D); fa.ln_st = F(st, 2); fa.lga = fa.lgb = g; fa.lba = fa.lbb = be;
    fa.drop = dtqn::tl_drop_none(); fa.layer = 0;
    if (getenv("FFN_BENCH_PACKED") != nullptr) {        // the fragment-major fetch path (timing only: the buffers hold the same bytes in another order)
        fa.W1pa = fa.W1pb = W1; fa.W2pa = fa.W2pb = W2;
    }
    hipStream_t stream; CK(hipStreamCreate(&stream));
    for (int i = 0; i < 3; ++i) if (dtqn::launch_ffn<D>(fa, S, stream) != 0) { printf("launch failed\n"); exit(1); }
    CK(hipStreamSynchronize(stream));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, stream));
    for (int i = 0; i < reps; ++i) dtqn::launch_ffn<D>(fa, S, stream);
    CK(hipEventRecord(e1, stream)); CK(hipStreamSynchronize(stream));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    {   // per-launch times (the 1.5-round launches are bimodal: which slots the last half round lands on)
        std::vector<hipEvent_t> ev(reps + 1);
        for (auto& e : ev) CK(hipEventCreate(&e));
        CK(hipEventRecord(ev[0], stream));
        for (int i = 0; i < reps; ++i) { dtqn::launch_ffn<D>(fa, S, stream); CK(hipEventRecord(ev[i + 1], stream)); }
        CK(hipStreamSynchronize(stream));
        printf("  per launch (us):");
        for (int i = 0; i < reps; ++i) { float m; CK(hipEventElapsedTime(&m, ev[i], ev[i + 1])); printf(" %.0f", m * 1e3); }
        printf("\n");
    }
    const double us = ms * 1e3 / reps, gflop = (double)rows * 2.0 * 2.0 * D * HID / 1e9;
    const bool r32 = dtqn::tl_rows32(S * rpb, 256 * (D <= 128 ? 2 : 1), D, "DTQN_ROWS_FFN");
    printf("tl_ffn D=%d S=%d lpb=%d n_save=%d rows/wg=%d: %.1f us, %.1f TFLOP/s (%.3f of 157.3)\n", D, S, lpb, n_save, r32 ? 32 : 64, us, gflop / us * 1e3,
           gflop / us * 1e3 / 157.3);
#ifndef NO_STAMPS
    const int nblocks = S * rpb * (r32 ? 2 : 1);
    std::vector<long long> hs((size_t)nblocks * 8 * TL_NSLOT);
    CK(hipMemcpy(hs.data(), stamps, hs.size() * 8, hipMemcpyDeviceToHost));
    const int NJ = HID / 128, nslots = 2 + 6 * NJ;
    // average phase lengths over all blocks, per wave
    for (int w = 0; w < 8; ++w) {
        std::vector<double> avg(nslots, 0.0);
        for (int b = 0; b < nblocks; ++b) {
            const long long* p = &hs[((size_t)b * 8 + w) * TL_NSLOT];
            for (int q = 1; q < nslots; ++q) avg[q] += (double)(p[q] - p[q - 1]);
        }
        double ph[6] = {0, 0, 0, 0, 0, 0};
        for (int j = 0; j < NJ; ++j)
            for (int k = 0; k < 6; ++k) ph[k] += avg[2 + 6 * j + k];
        double tail = 0, life = 0;
        for (int b = 0; b < nblocks; ++b) {
            const long long* p = &hs[((size_t)b * 8 + w) * TL_NSLOT];
            tail += (double)(p[62] - p[nslots - 1]); life += (double)(p[62] - p[0]);
        }
        if (w < 4) {
            double tt[5] = {0, 0, 0, 0, 0};
            for (int b = 0; b < nblocks; ++b) {
                const long long* p = &hs[((size_t)b * 8 + w) * TL_NSLOT];
                tt[0] += (double)(p[50] - p[nslots - 1]); tt[1] += (double)(p[51] - p[50]); tt[2] += (double)(p[52] - p[51]); tt[3] += (double)(p[53] - p[52]); tt[4] += (double)(p[62] - p[53]);
            }
            printf("  wave %d: tail %.0f = acc->LDS %.0f, barrier %.0f, res loads + sum %.0f, butterflies %.0f, gamma/beta + stores %.0f; lifetime stamp0 -> end %.0f\n", w, tail / nblocks,
                   tt[0] / nblocks, tt[1] / nblocks, tt[2] / nblocks, tt[3] / nblocks, tt[4] / nblocks, life / nblocks);
        }
        printf("  wave %d ticks/chunk: A-mma %.0f  epiA %.0f  bar1 %.0f  h-store %.0f  B-mma %.0f  bar2 %.0f | stage %.0f | loop total %.0f\n", w, ph[0] / NJ / nblocks, ph[1] / NJ / nblocks,
               ph[2] / NJ / nblocks, ph[3] / NJ / nblocks, ph[4] / NJ / nblocks, ph[5] / NJ / nblocks, avg[1] / nblocks,
               (ph[0] + ph[1] + ph[2] + ph[3] + ph[4] + ph[5]) / nblocks);
    }
    {
        // per-CU timeline from the 100 MHz wall clock: CU key = (xcc, se, sh, cu); residency = sum of lifetimes / span
        std::vector<std::vector<std::pair<long long, long long>>> per(8 * 8 * 2 * 16);
        long long w_lo = 1LL << 62, w_hi = 0;
        for (int b = 0; b < nblocks; ++b) {
            const long long* p = &hs[((size_t)b * 8) * TL_NSLOT];
            const unsigned hw = (unsigned)p[61], xcc = (unsigned)(p[61] >> 32) & 7;
            const unsigned cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
            per[((xcc * 8 + se) * 2 + sh) * 16 + cu].push_back({p[59], p[63]});
            if (p[59] < w_lo) w_lo = p[59];
            if (p[63] > w_hi) w_hi = p[63];
        }
        int ncu = 0; double life = 0; size_t mx = 0, mn = 1 << 30;
        for (auto& v : per) if (!v.empty()) { ++ncu; mx = v.size() > mx ? v.size() : mx; mn = v.size() < mn ? v.size() : mn; for (auto& q : v) life += (double)(q.second - q.first); }
        printf("  wall clock: span %.1f us over %d CUs (workgroups per CU: %zu .. %zu); mean lifetime %.1f us; mean resident workgroups per CU %.2f\n", (w_hi - w_lo) / 100.0, ncu, mn, mx,
               life / nblocks / 100.0, life / ((double)(w_hi - w_lo) * ncu));
        // one CU's timeline
        for (auto& v : per) if (v.size() == mx) {
            std::sort(v.begin(), v.end());
            printf("  a CU with %zu blocks (loop top, end of epilogue; us):", mx);
            for (auto& q : v) printf(" (%.1f, %.1f)", (q.first - w_lo) / 100.0, (q.second - w_lo) / 100.0);
            printf("\n");
            break;
        }
    }
#endif
    for (void* p_ : {(void*)in, (void*)out, (void*)res, (void*)h, (void*)lnout, (void*)st, (void*)mh, (void*)m2, (void*)W1, (void*)W2, (void*)stamps, (void*)b1, (void*)b2, (void*)g, (void*)be}) CK(hipFree(p_));
}

int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 20;
    {
        long long* o; float* sink; long long ho[2];
        CK(hipMalloc(&o, 16)); CK(hipMalloc(&sink, 256 * 256 * 4));
        for (int r = 0; r < 2; ++r) {
            hipLaunchKernelGGL(tick_kernel, dim3(256), dim3(256), 0, 0, o, sink, 40000);
            CK(hipMemcpy(ho, o, 16, hipMemcpyDeviceToHost));
            printf("tick calibration: 40000 MFMAs (1.28 M shader cycles at 32 per MFMA) = %lld clock64 ticks, %lld wall_clock64 ticks (100 MHz)\n", ho[0], ho[1]);
        }
    }
    run<128>(384, 128, 128, reps);     // config 4 forward: three passes of 128 sequences, the first saved
    run<128>(1536, 64, 512, reps);     // config 3 forward: 1536 blocks of 64 rows = three rounds of 512
    run<256>(96, 256, 32, reps);       // config 5 forward
    return 0;
}
