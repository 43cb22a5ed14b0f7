// gfx950 instruction-level primitives the kernels use: lane swaps, DPP rotates, agent-scope (sc1) loads / stores through a
// buffer descriptor, scheduling fences, v_exp_f32.  Included as <dtqn_gfx950.hpp>: the product build finds this file; the
// test-only host emulation (tests/emu) puts its own directory first on the include path and supplies host equivalents
// there, so no test hook lives in the kernel sources.
#pragma once
#include <hip/hip_runtime.h>

// Keep-alive for prefetched registers: forces the compiler to place its s_waitcnt for the loads that
// produced `x` HERE (the test-only host build defines it away).
// gfx950 lane-swap instructions on two copies of one register: [0] / [1] = the value held by the lower / upper half
// (v_permlane32_swap) or the even / odd 16-lane row of the pair (v_permlane16_swap).  VALU-rate cross-lane exchange:
// no LDS round trip, unlike the ds_bpermute behind __shfl_xor.
struct LanePair {
    float v[2];
    __device__ __forceinline__ float operator[](int k) const { return v[k]; }
};
__device__ __forceinline__ LanePair dtqn_lane_swap32(float x) {
    const unsigned u = __float_as_uint(x);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return LanePair{{__uint_as_float(r[0]), __uint_as_float(r[1])}};
}
__device__ __forceinline__ LanePair dtqn_lane_swap16(float x) {
    const unsigned u = __float_as_uint(x);
    const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    return LanePair{{__uint_as_float(r[0]), __uint_as_float(r[1])}};
}
#define DTQN_LANE_SWAP32(x) dtqn_lane_swap32(x)
#define DTQN_LANE_SWAP16(x) dtqn_lane_swap16(x)
// Agent-scope (whole GPU, across XCDs) relaxed atomics: the two workgroups that share a sequence in row-split mode
// hand tiles to each other through global memory with these (sc1 loads / stores: coherent without L2 write-backs).
#define DTQN_AGENT_LOAD(p) __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define DTQN_AGENT_STORE(p, v) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
// System-scope (other GPUs over xGMI, other processes on this GPU) relaxed atomics: the gradient exchange of the data-parallel update
#define DTQN_SYSTEM_LOAD(p) __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)
#define DTQN_SYSTEM_STORE(p, v) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)
#define DTQN_AGENT_ADD(p, v) __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define DTQN_SPIN_PAUSE() __builtin_amdgcn_s_sleep(2)
#define DTQN_SPIN_PAUSE_LONG() __builtin_amdgcn_s_sleep(12)     /* pollers that wait for microseconds (one lane per workgroup) */
// all of this wave's outstanding global stores acknowledged at their scope (the workgroup barrier alone does not wait
// for global stores)
#define DTQN_WAIT_VMEM() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
// instruction-scheduling fence: keeps a block of prefetch loads ahead of the arithmetic that follows it (the machine
// scheduler otherwise sinks loads next to their uses to save registers, which serialises their latencies)
#define DTQN_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
// 16-byte write-through (sc1) stores / L1-bypassing (sc1) loads through a buffer descriptor: a dword sc1 store is one
// fabric write of its own and costs ~6x a 16-byte one per byte (MI355X_MICROARCH.md, inter-workgroup visibility)
typedef unsigned dtqn_u32x4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t DtqnRsrc;
#define DTQN_XCH_RSRC(ptr, bytes) __builtin_amdgcn_make_buffer_rsrc((void*)(ptr), 0, (int)(bytes), 0x00020000)
__device__ __forceinline__ void dtqn_xch_store4(DtqnRsrc r, int byte_off, float4 v) {
    dtqn_u32x4 u = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
    __builtin_amdgcn_raw_buffer_store_b128(u, r, byte_off, 0, 16);
}
__device__ __forceinline__ float4 dtqn_xch_load4(DtqnRsrc r, int byte_off) {
    const dtqn_u32x4 u = __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 16);
    return make_float4(__uint_as_float(u[0]), __uint_as_float(u[1]), __uint_as_float(u[2]), __uint_as_float(u[3]));
}
// dword forms (few, small records only: a dword sc1 store is one fabric write)
__device__ __forceinline__ void dtqn_xch_store1(DtqnRsrc r, int byte_off, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, byte_off, 0, 16);
}
__device__ __forceinline__ float dtqn_xch_load1(DtqnRsrc r, int byte_off) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, byte_off, 0, 16));
}
// DPP rotate of a 16-lane row by n lanes (VALU-rate; 0x120 + n = row_ror:n)
#define DTQN_ROW_ROR(x, n) __uint_as_float(__builtin_amdgcn_mov_dpp(__float_as_uint(x), 0x120 + (n), 0xf, 0xf, true))
// 2^x on the transcendental unit (v_exp_f32: -inf -> 0, no range reduction code)
#define DTQN_EXP2(x) __builtin_amdgcn_exp2f(x)
#define DTQN_ASM_KEEP(x) asm volatile("" : "+v"(x))

