// Gradient assembly, global-norm clip, Adam, hard target sync (gfx950; HBM/L2-bound elementwise work).
//
// Replaces torch.nn.utils.clip_grad_norm_(params, 1.0, error_if_nonfinite=True), optim.Adam.step and
// DqnAgent.target_update (dtqn/agents/dtqn.py:257-269, dtqn/agents/dqn.py:64,208-210), plus the seven
// `.item()` statistics of dtqn.py:245-253,263 (reduced on the device, read back asynchronously).
#include "dtqn_device.hpp"
#include "dtqn_adam_device.hpp"

namespace dtqn {

constexpr int kOptThreads = 256;
constexpr int kOptVec = 4;                 // floats per thread (one float4)
constexpr int kOptBlockElems = kOptThreads * kOptVec;

struct ReduceArgs {
    DtqnNet net;
    const float* gsplit;
    const float* small;
    const float* grd;
    float* grad;
    float* norm_partial;
    int32_t* step_counter;
    int batch, n_split;
    int n_parts;              // length of norm_partial (>= gridDim.x): the tail is zeroed
};

__device__ __forceinline__ float block_sum(float v, float* red, int tid) {
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    float s = 0.f;
    for (int w = 0; w < kOptThreads / 64; ++w) s += red[w];
    __syncthreads();
    return s;
}

// grad[p] = sum over the batch splits of gsplit[s][p].
__global__ __launch_bounds__(kOptThreads) void dtqn_reduce_kernel(ReduceArgs a) {
    __shared__ float red[kOptThreads / 64];
    const DtqnNet& net = a.net;
    const int tid = (int)threadIdx.x;
    const int p0 = ((int)blockIdx.x * kOptThreads + tid) * kOptVec;
    if (blockIdx.x == 0 && tid == 0) a.step_counter[0] = a.step_counter[1];   // publish the step count of the previous update
    if (blockIdx.x == 0)
        for (int i = (int)gridDim.x + tid; i < a.n_parts; i += kOptThreads) a.norm_partial[i] = 0.f;
    float v[kOptVec] = {0.f, 0.f, 0.f, 0.f};
    if (p0 < net.n_trainable) {
        // every gradient element (weight / bias GEMM blocks and the per-sequence partials folded in by the
        // extra blocks of dtqn_wgrad_kernel) is a sum over the batch splits; never-written padding is zero
        for (int s = 0; s < a.n_split; ++s) {
            const float4 x = ld4(a.gsplit + (size_t)s * net.n_trainable + p0);
            v[0] += x.x; v[1] += x.y; v[2] += x.z; v[3] += x.w;
        }
        st4(a.grad + p0, make_float4(v[0], v[1], v[2], v[3]));
    }
    const float ss = block_sum((v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]), red, tid);
    if (tid == 0) a.norm_partial[blockIdx.x] = ss;
}

struct NormArgs {
    const float* grad;
    float* norm_partial;
    int n, n_parts;
};
__global__ __launch_bounds__(kOptThreads) void dtqn_gradnorm_kernel(NormArgs a) {
    __shared__ float red[kOptThreads / 64];
    const int tid = (int)threadIdx.x;
    const int p0 = ((int)blockIdx.x * kOptThreads + tid) * kOptVec;
    if (blockIdx.x == 0)
        for (int i = (int)gridDim.x + tid; i < a.n_parts; i += kOptThreads) a.norm_partial[i] = 0.f;
    float ss = 0.f;
    if (p0 < a.n) {
        const float4 g = ld4(a.grad + p0);
        ss = (g.x * g.x + g.y * g.y) + (g.z * g.z + g.w * g.w);
    }
    ss = block_sum(ss, red, tid);
    if (tid == 0) a.norm_partial[blockIdx.x] = ss;
}

__global__ __launch_bounds__(kOptThreads) void dtqn_clip_adam_kernel(AdamArgs a) {
    __shared__ float red[kOptThreads / 64];
    __shared__ float mm4[4 * kOptThreads / 64];
    __shared__ double pw[2];
    const int tid = (int)threadIdx.x;
    // this thread's gradient / moments / parameters go in flight first: they do not depend on the norm, and the norm
    // reduction below (two barriers) would otherwise sit in front of their round trip
    const int p0 = ((int)blockIdx.x * kOptThreads + tid) * kOptVec;
    const bool mine = p0 < a.n;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 g4 = mine ? ld4(a.grad + p0) : z4;
    float4 m4 = mine ? ld4(a.m + p0) : z4, v4 = mine ? ld4(a.v + p0) : z4, p4 = mine ? ld4(a.theta + p0) : z4;
    // every block re-derives the global norm from the per-block partials (a few hundred floats)
    float part = 0.f;
    for (int i = tid; i < a.n_norm_parts; i += kOptThreads) part += a.norm_partial[i];
    const float total = block_sum(part, red, tid);
    const AdamCoef c = adam_coef(a, total, a.step_counter[0] + 1, pw, tid);   // 1-based index of this optimizer step
    if (c.finite && mine) {
        const float g[4] = {g4.x, g4.y, g4.z, g4.w};
        float mm[4] = {m4.x, m4.y, m4.z, m4.w}, vv[4] = {v4.x, v4.y, v4.z, v4.w}, pp[4] = {p4.x, p4.y, p4.z, p4.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) adam_elem(a, c, g[q], mm[q], vv[q], pp[q]);
        st4(a.m + p0, make_float4(mm[0], mm[1], mm[2], mm[3]));
        st4(a.v + p0, make_float4(vv[0], vv[1], vv[2], vv[3]));
        const float4 pn = make_float4(pp[0], pp[1], pp[2], pp[3]);
        st4(a.theta + p0, pn);
        if (c.sync_target) st4(a.theta_tgt + p0, pn);         // hard target update every tuf steps (dqn.py:208-210)
    }
    if (blockIdx.x == 0) adam_statistics<kOptThreads>(a, c, red, mm4, tid);
}

struct CopyArgs {
    const float* src;
    float* dst;
    int n;
};
__global__ __launch_bounds__(kOptThreads) void dtqn_copy_kernel(CopyArgs a) {
    const int p0 = ((int)blockIdx.x * kOptThreads + (int)threadIdx.x) * kOptVec;
    if (p0 < a.n) st4(a.dst + p0, ld4(a.src + p0));
}

}  // namespace dtqn

using namespace dtqn;

static int opt_blocks(int n) { return (n + kOptBlockElems - 1) / kOptBlockElems; }

extern "C" int dtqn_td_reduce(const DtqnNet* net, const DtqnTd* td, void* stream) {
    if (!net || !td) return DTQN_ERR_ARG;
    if (td->n_norm_blocks != opt_blocks(net->n_trainable)) return DTQN_ERR_ARG;
    if (dtqn_td_wgrad_is_direct(net, td->batch)) return DTQN_OK;      // the direct weight-gradient kernel wrote grad itself
    ReduceArgs a;
    a.net = *net;
    a.gsplit = td->gsplit; a.small = td->small; a.grd = td->grd; a.grad = td->grad;
    a.norm_partial = td->norm_partial; a.step_counter = td->step_counter;
    a.batch = td->batch; a.n_split = td->n_split; a.n_parts = dtqn_td_norm_partials(net);
    (void)hipGetLastError();   // drop stale errors left by other users of the runtime
    hipLaunchKernelGGL(dtqn_reduce_kernel, dim3(td->n_norm_blocks), dim3(kOptThreads), 0, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? DTQN_OK : DTQN_ERR_LAUNCH;
}

extern "C" int dtqn_td_gradnorm(const DtqnNet* net, const DtqnTd* td, void* stream) {
    if (!net || !td) return DTQN_ERR_ARG;
    if (td->n_norm_blocks != opt_blocks(net->n_trainable)) return DTQN_ERR_ARG;
    NormArgs a;
    a.grad = td->grad; a.norm_partial = td->norm_partial; a.n = net->n_trainable; a.n_parts = dtqn_td_norm_partials(net);
    (void)hipGetLastError();   // drop stale errors left by other users of the runtime
    hipLaunchKernelGGL(dtqn_gradnorm_kernel, dim3(td->n_norm_blocks), dim3(kOptThreads), 0, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? DTQN_OK : DTQN_ERR_LAUNCH;
}

extern "C" int dtqn_td_clip_adam(const DtqnNet* net, const DtqnTd* td, void* stream) {
    if (!net || !td) return DTQN_ERR_ARG;
    if (td->n_norm_blocks != opt_blocks(net->n_trainable)) return DTQN_ERR_ARG;
    const AdamArgs a = adam_args(net, td, dtqn_td_norm_partials(net));
    (void)hipGetLastError();   // drop stale errors left by other users of the runtime
    hipLaunchKernelGGL(dtqn_clip_adam_kernel, dim3(td->n_norm_blocks), dim3(kOptThreads), 0, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? DTQN_OK : DTQN_ERR_LAUNCH;
}

extern "C" int dtqn_target_sync(const DtqnNet* net, const float* theta_pol, float* theta_tgt, void* stream) {
    if (!net || !theta_pol || !theta_tgt) return DTQN_ERR_ARG;
    CopyArgs a;
    a.src = theta_pol; a.dst = theta_tgt; a.n = net->n_theta;
    (void)hipGetLastError();   // drop stale errors left by other users of the runtime
    hipLaunchKernelGGL(dtqn_copy_kernel, dim3(opt_blocks(net->n_theta)), dim3(kOptThreads), 0, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? DTQN_OK : DTQN_ERR_LAUNCH;
}
