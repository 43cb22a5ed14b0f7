// TEST-ONLY host emulation of the small subset of HIP the dtqn_amd kernels use.
//
// The product is built by hipcc for gfx950 (see __graft_entry__.build()).  This header is
// NOT part of the product and is never on hipcc's include path; it exists so that the CPU test
// suite (pytest -m "not gpu", no GPU in the build container) can compile the very same kernel
// sources with the host clang++ and execute their logic -- indexing, LDS layouts, barriers,
// wave64 collectives, MFMA fragment maps -- one workgroup at a time on cooperative fibers.
// It emulates gfx950 semantics the kernels rely on: 64-lane waves, __syncthreads, __shfl*,
// v_mfma_f32_16x16x4_f32 / v_mfma_f32_32x32x2_f32 operand and accumulator lane maps
// (/opt/skills/guides/cdna_hip_programming.md section 3).
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>

#define DTQN_HIPEMU 1
#define DTQN_ASM_KEEP(x) ((void)(x))   /* device-only register keep-alive (dtqn_device.hpp) */
#define DTQN_EXP2(x) exp2f(x)          /* v_exp_f32 (dtqn_device.hpp) */
/* agent-scope atomics of the row-split hand-over (dtqn_device.hpp): blocks may run on different host threads */
template <typename T> static inline T hipemu_agent_load(const T* p) { T v; __atomic_load(const_cast<T*>(p), &v, __ATOMIC_ACQUIRE); return v; }
template <typename T> static inline void hipemu_agent_store(T* p, T v) { __atomic_store(p, &v, __ATOMIC_RELEASE); }
#define DTQN_AGENT_LOAD(p) hipemu_agent_load(p)
#define DTQN_AGENT_STORE(p, v) hipemu_agent_store(p, v)
#define DTQN_SYSTEM_LOAD(p) hipemu_agent_load(p)
#define DTQN_SYSTEM_STORE(p, v) hipemu_agent_store(p, v)
template <typename T> static inline T hipemu_agent_add(T* p, T v) { return __atomic_fetch_add(p, v, __ATOMIC_ACQ_REL); }
#define DTQN_AGENT_ADD(p, v) hipemu_agent_add(p, v)
#define DTQN_SPIN_PAUSE() ((void)0)
#define DTQN_SPIN_PAUSE_LONG() ((void)0)
#define DTQN_WAIT_VMEM() ((void)0)
#define DTQN_SCHED_FENCE() ((void)0)

// ---- qualifiers -------------------------------------------------------------
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ thread_local

// ---- basic types ------------------------------------------------------------
struct dim3 {
    unsigned x, y, z;
    constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct alignas(16) float4 { float x, y, z, w; };
struct alignas(8) float2 { float x, y; };
struct alignas(16) int4 { int x, y, z, w; };
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
/* 16-byte hand-over stores / loads through a buffer descriptor (dtqn_device.hpp) */
struct DtqnRsrc { char* base; };
#define DTQN_XCH_RSRC(ptr, bytes) DtqnRsrc{reinterpret_cast<char*>(const_cast<float*>(ptr))}
static inline void dtqn_xch_store4(DtqnRsrc r, int byte_off, float4 v) { std::memcpy(r.base + byte_off, &v, 16); }
static inline float4 dtqn_xch_load4(DtqnRsrc r, int byte_off) { float4 v; std::memcpy(&v, r.base + byte_off, 16); return v; }
static inline void dtqn_xch_store1(DtqnRsrc r, int byte_off, float v) { std::memcpy(r.base + byte_off, &v, 4); }
static inline float dtqn_xch_load1(DtqnRsrc r, int byte_off) { float v; std::memcpy(&v, r.base + byte_off, 4); return v; }
static inline float2 make_float2(float x, float y) { return float2{x, y}; }

typedef int hipError_t;
typedef void* hipStream_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorLaunchFailure = 719 };
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };

namespace hipemu {
struct Fiber;
struct Ctx {
    dim3 tid, bid, bdim, gdim;
};
extern thread_local Ctx* g_ctx;           // context of the running fiber
void block_barrier();                      // __syncthreads
// wave exchange: every lane of the wave publishes `n` 32-bit words, waits for the other lanes,
// then may read any lane's words through the returned pointer (slot-major: [lane][n]).
const uint32_t* wave_exchange(const uint32_t* mine, int n);
int lane_id();
void launch(dim3 grid, dim3 block, size_t smem, const std::function<void()>& body);
hipError_t last_error();
void atomic_add_f32(float* p, float v);
}  // namespace hipemu

#define threadIdx (hipemu::g_ctx->tid)
#define blockIdx (hipemu::g_ctx->bid)
#define blockDim (hipemu::g_ctx->bdim)
#define gridDim (hipemu::g_ctx->gdim)
static const int warpSize = 64;

// ---- synchronisation / collectives -----------------------------------------
static inline void __syncthreads() { hipemu::block_barrier(); }

template <typename T>
static inline T hipemu_shfl_from(T v, int src_lane) {
    static_assert(sizeof(T) == 4, "32-bit shuffles only");
    uint32_t w;
    std::memcpy(&w, &v, 4);
    const uint32_t* all = hipemu::wave_exchange(&w, 1);
    uint32_t r = all[src_lane & 63];
    T out;
    std::memcpy(&out, &r, 4);
    return out;
}
template <typename T> static inline T __shfl_xor(T v, int mask, int width = 64) { (void)width; return hipemu_shfl_from(v, hipemu::lane_id() ^ mask); }
template <typename T> static inline T __shfl(T v, int src, int width = 64) {
    int l = hipemu::lane_id();
    return hipemu_shfl_from(v, (l & ~(width - 1)) | (src & (width - 1)));
}
// gfx950 v_permlane32_swap / v_permlane16_swap applied to two copies of the same register (dtqn_device.hpp):
// [0] holds the value of the partner-aligned lower half / even row, [1] the upper half / odd row
struct hipemu_pair { float v[2]; float operator[](int k) const { return v[k]; } };
static inline hipemu_pair hipemu_lane_swap32(float x) {
    const int l = hipemu::lane_id();
    hipemu_pair r;
    r.v[0] = hipemu_shfl_from(x, l & 31);
    r.v[1] = hipemu_shfl_from(x, (l & 31) | 32);
    return r;
}
static inline hipemu_pair hipemu_lane_swap16(float x) {
    const int l = hipemu::lane_id();
    hipemu_pair r;
    r.v[0] = hipemu_shfl_from(x, l & ~16);
    r.v[1] = hipemu_shfl_from(x, l | 16);
    return r;
}
static inline float hipemu_row_ror(float x, int n) { const int l = hipemu::lane_id(); return hipemu_shfl_from(x, (l & ~15) | ((l - n) & 15)); }
#define DTQN_ROW_ROR(x, n) hipemu_row_ror(x, n)
#define DTQN_LANE_SWAP32(x) hipemu_lane_swap32(x)
#define DTQN_LANE_SWAP16(x) hipemu_lane_swap16(x)
template <typename T> static inline T __shfl_down(T v, unsigned d, int width = 64) {
    int l = hipemu::lane_id();
    int s = l + (int)d;
    if ((s & ~(width - 1)) != (l & ~(width - 1))) s = l;
    return hipemu_shfl_from(v, s);
}
static inline unsigned long long __ballot(int pred) {
    uint32_t w = pred ? 1u : 0u;
    const uint32_t* all = hipemu::wave_exchange(&w, 1);
    unsigned long long m = 0;
    for (int i = 0; i < 64; ++i) m |= (unsigned long long)(all[i] & 1u) << i;
    return m;
}

// ---- MFMA (f32 in / f32 accumulate), gfx950 lane maps ------------------------
typedef float hipemu_f32x4 __attribute__((ext_vector_type(4)));
typedef float hipemu_f32x16 __attribute__((ext_vector_type(16)));

// v_mfma_f32_16x16x4_f32: lane l supplies A[i=l&15][k=l>>4] and B[k=l>>4][j=l&15];
// D register r of lane l is D[row=(l>>4)*4+r][col=l&15].  k-ordered fmaf chain.
static inline hipemu_f32x4 hipemu_mfma_16x16x4(float a, float b, hipemu_f32x4 c, int, int, int) {
    uint32_t w[2];
    std::memcpy(&w[0], &a, 4);
    std::memcpy(&w[1], &b, 4);
    const uint32_t* all = hipemu::wave_exchange(w, 2);
    const int l = hipemu::lane_id(), col = l & 15, rq = l >> 4;
    hipemu_f32x4 d = c;
    for (int r = 0; r < 4; ++r) {
        const int row = rq * 4 + r;
        float acc = c[r];
        for (int k = 0; k < 4; ++k) {
            float av, bv;
            std::memcpy(&av, &all[(k * 16 + row) * 2 + 0], 4);
            std::memcpy(&bv, &all[(k * 16 + col) * 2 + 1], 4);
            acc = std::fmaf(av, bv, acc);
        }
        d[r] = acc;
    }
    return d;
}
// v_mfma_f32_32x32x2_f32: A[i=l&31][k=l>>5], B[k=l>>5][j=l&31];
// D reg r of lane l: row=(r&3)+8*(r>>2)+4*(l>>5), col=l&31.
static inline hipemu_f32x16 hipemu_mfma_32x32x2(float a, float b, hipemu_f32x16 c, int, int, int) {
    uint32_t w[2];
    std::memcpy(&w[0], &a, 4);
    std::memcpy(&w[1], &b, 4);
    const uint32_t* all = hipemu::wave_exchange(w, 2);
    const int l = hipemu::lane_id(), col = l & 31, hi = l >> 5;
    hipemu_f32x16 d = c;
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
        float acc = c[r];
        for (int k = 0; k < 2; ++k) {
            float av, bv;
            std::memcpy(&av, &all[(k * 32 + row) * 2 + 0], 4);
            std::memcpy(&bv, &all[(k * 32 + col) * 2 + 1], 4);
            acc = std::fmaf(av, bv, acc);
        }
        d[r] = acc;
    }
    return d;
}
#define __builtin_amdgcn_mfma_f32_16x16x4f32 hipemu_mfma_16x16x4
#define __builtin_amdgcn_mfma_f32_32x32x2f32 hipemu_mfma_32x32x2
#define __builtin_amdgcn_s_setprio(x) ((void)0)
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
static inline int hipemu_readfirstlane(int v) { return hipemu_shfl_from(v, 0); }
#define __builtin_amdgcn_readfirstlane hipemu_readfirstlane

// ---- math -------------------------------------------------------------------
static inline float hipemu_expf(float x) { return std::exp(x); }
static inline float hipemu_logf(float x) { return std::log(x); }
static inline float hipemu_fdividef(float a, float b) { return a / b; }
static inline float hipemu_rsqrtf(float x) { return 1.0f / std::sqrt(x); }
static inline float hipemu_frcp(float x) { return 1.0f / x; }
#define __expf hipemu_expf
#define __logf hipemu_logf
#define __fdividef hipemu_fdividef
#define rsqrtf hipemu_rsqrtf
#define __frcp_rn hipemu_frcp
using std::fmaxf;
using std::fminf;
static inline int min(int a, int b) { return a < b ? a : b; }
static inline int max(int a, int b) { return a > b ? a : b; }
using std::isfinite;

#include <chrono>
/* 100 MHz wall clock like the device's (bounded spins of the gradient exchange must be able to run out here too) */
static inline long long wall_clock64() { return (long long)(std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count() / 10); }
static inline float __int_as_float(int v) { float f; std::memcpy(&f, &v, 4); return f; }
static inline unsigned __float_as_uint(float f) { unsigned v; std::memcpy(&v, &f, 4); return v; }
static inline float __uint_as_float(unsigned v) { float f; std::memcpy(&f, &v, 4); return f; }
static inline long long clock64() { return 0; }

// ---- atomics ----------------------------------------------------------------
static inline float atomicAdd(float* p, float v) { hipemu::atomic_add_f32(p, v); return 0.f; }
static inline int atomicAdd(int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline unsigned atomicAdd(unsigned* p, unsigned v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline unsigned atomicOr(unsigned* p, unsigned v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }
static inline int atomicMax(int* p, int v) {
    int old = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (old < v && !__atomic_compare_exchange_n(p, &old, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return old;
}
static inline void __threadfence() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
static inline void __threadfence_system() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }

// ---- runtime API subset -------------------------------------------------------
#define hipLaunchKernelGGL(kernel, grid, block, smem, stream, ...) \
    hipemu::launch((grid), (block), (smem), [=]() { kernel(__VA_ARGS__); })
static inline hipError_t hipGetLastError() { return hipemu::last_error(); }
static inline hipError_t hipPeekAtLastError() { return hipemu::last_error(); }
static inline const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : "hipemu error"; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { std::memset(p, v, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { std::memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
// streams / events of the fork-join launches (dtqn_wgrad.hip): the emulation runs every launch to completion in issue order
typedef void* hipEvent_t;
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2 };
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = nullptr; return hipSuccess; }
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = nullptr; return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
template <typename F> static inline hipError_t hipFuncSetAttribute(F, hipFuncAttribute, int) { return hipSuccess; }
static inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount = 63 };
static inline hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t, int) { *v = 256; return hipSuccess; }
