// Gradient assembly, global-norm clip, Adam, hard target sync (gfx950; HBM/L2-bound elementwise work).
//
// Replaces torch.nn.utils.clip_grad_norm_(params, 1.0, error_if_nonfinite=True), optim.Adam.step and
// DqnAgent.target_update (dtqn/agents/dtqn.py:257-269, dtqn/agents/dqn.py:64,208-210), plus the seven
// `.item()` statistics of dtqn.py:245-253,263 (reduced on the device, read back asynchronously).
#include <cstdlib>

#include "dtqn_device.hpp"

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

// ---- device-side gradient exchange (include/dtqn_hip.h, dtqn_td_xreduce) ---------------------------------------------------
struct PublishArgs {
    int32_t* flag;
    int32_t gen;
};
__global__ void dtqn_xch_publish_kernel(PublishArgs a) {
    if (threadIdx.x == 0 && blockIdx.x == 0) DTQN_SYSTEM_STORE(a.flag, a.gen);
}
struct XreduceArgs {
    const float* const* peer_grad;
    int32_t* const* peer_flag;
    float* gsum;
    float* norm_partial;
    int32_t* status;
    int n, n_parts, world;
    int32_t gen;
    long long timeout_ticks;       // bounded wait, in ticks of the 100 MHz wall clock
    int32_t* own_flag;             // this rank's flag word: raised to `gen` by the launch itself (NULL: dtqn_xch_publish did it)
};
// Block b owns parameters [1024 b, 1024 b + 1024): wait for the `world` flag words, then one pass over the `world` buffers in rank
// order.  Peer buffers are read with system-scope loads (they were written by another GPU / process: nothing of them may come
// out of this GPU's caches), 16 floats per thread and peer at cfg 1.  HBM / xGMI-bound: 4 * world * n bytes in, 4 * n out.
__global__ __launch_bounds__(kOptThreads) void dtqn_xreduce_kernel(XreduceArgs a) {
    __shared__ float red[kOptThreads / 64];
    const int tid = (int)threadIdx.x;
    if (blockIdx.x == 0)
        for (int i = (int)gridDim.x + tid; i < a.n_parts; i += kOptThreads) a.norm_partial[i] = 0.f;
    // publish: the kernel boundary in front of THIS launch has made the rank's gradient visible; the first thing the launch does
    // is say so (every rank does, before it waits for anybody: no cycle) -- the one-thread publish launch of round 3 is gone
    if (a.own_flag != nullptr && blockIdx.x == 0 && tid == 0) DTQN_SYSTEM_STORE(a.own_flag, a.gen);
    if (tid < a.world) {
        const int32_t* f = a.peer_flag[tid];
        const long long t0 = wall_clock64();
        // generations only grow (and wrap after 2^31 updates): "reached" = not behind
        while ((int32_t)(DTQN_SYSTEM_LOAD(f) - a.gen) < 0) {
            // a peer died or never entered this update (default 5 s), or another block already gave up
            if (wall_clock64() - t0 > a.timeout_ticks || DTQN_SYSTEM_LOAD(a.status) != 0) {
                DTQN_SYSTEM_STORE(a.status, (int32_t)1);
                break;
            }
            DTQN_SPIN_PAUSE();
        }
    }
    __syncthreads();
    const int p0 = ((int)blockIdx.x * kOptThreads + tid) * kOptVec;
    float v[kOptVec] = {0.f, 0.f, 0.f, 0.f};
    if (p0 < a.n) {
        for (int r = 0; r < a.world; ++r) {
            const int32_t* g = reinterpret_cast<const int32_t*>(a.peer_grad[r] + p0);
#pragma unroll
            for (int c = 0; c < kOptVec; ++c) v[c] += __int_as_float(DTQN_SYSTEM_LOAD(g + c));
        }
        st4(a.gsum + p0, make_float4(v[0], v[1], v[2], v[3]));
    }
    const float ss = block_sum((v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]), red, tid);
    if (tid == 0) a.norm_partial[blockIdx.x] = ss;
}

struct AdamArgs {
    float* theta;
    float* theta_tgt;
    const float* grad;
    float* m;
    float* v;
    const float* norm_partial;
    const float* stats_partial;
    float* stats;
    float* stats_ring;
    int32_t* step_counter;
    const int32_t* xstatus;      // status word of the device-side gradient exchange (dtqn_td_xreduce), or NULL
    int n, n_norm_parts, batch, history, tuf, ring_slots;
    int n_stat_parts;            // batch * row_split per-workgroup statistics partials
    float lr, beta1, beta2, eps, clip, grad_scale;
};

__global__ __launch_bounds__(kOptThreads) void dtqn_clip_adam_kernel(AdamArgs a) {
    __shared__ float red[kOptThreads / 64];
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
    const float norm = sqrtf(total) * a.grad_scale;
    // The update is SKIPPED (theta, moments, step count untouched) when the norm is non-finite -- clip_grad_norm_(error_if_nonfinite=True)
    // raises before optimizer.step() in the reference (dtqn/agents/dtqn.py:257-265) --, when the device-side exchange gave up waiting
    // for a peer (its sum is then stale: applying it would silently split the replicas), and from then on for every later call
    // (step_counter[3], sticky): the host raises when it drains the statistics, up to a few calls late, and must find the state
    // the reference's exception leaves behind.
    const int why = !isfinite(norm) ? 1 : (a.xstatus != nullptr && *a.xstatus != 0) ? 2 : (a.step_counter[3] != 0) ? 3 : 0;
    const bool finite = why == 0;
    const int k = a.step_counter[0] + 1;                      // 1-based index of this optimizer step
    // bias corrections 1 - beta^k in f64 like torch's Python floats.  beta^k by repeated squaring: at most 62 dependent
    // multiplications (error a few 1e-16 relative) instead of two calls of the f64 pow() in front of every block's barrier;
    // the two powers go to two different waves
    if (tid == 0 || tid == 64) {
        const double beta = tid == 0 ? (double)a.beta1 : (double)a.beta2;
        double r = 1.0, b = beta;
        for (unsigned e = (unsigned)k; e != 0u; e >>= 1) {
            if (e & 1u) r *= b;
            b *= b;
        }
        pw[tid >> 6] = 1.0 - r;
    }
    __syncthreads();
    const float bc1 = (float)pw[0], bc2_sqrt = (float)sqrt(pw[1]);
    const float coef = fminf(1.0f, a.clip / (norm + 1e-6f)) * a.grad_scale;
    const float step_size = a.lr / bc1;
    const bool sync_target = finite && a.tuf > 0 && (k % a.tuf) == 0;
    if (finite && mine) {
        const float g[4] = {g4.x * coef, g4.y * coef, g4.z * coef, g4.w * coef};
        float mm[4] = {m4.x, m4.y, m4.z, m4.w}, vv[4] = {v4.x, v4.y, v4.z, v4.w}, pp[4] = {p4.x, p4.y, p4.z, p4.w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            mm[c] = mm[c] + (g[c] - mm[c]) * (1.0f - a.beta1);                  // exp_avg.lerp_(grad, 1 - beta1)
            vv[c] = vv[c] * a.beta2 + (1.0f - a.beta2) * g[c] * g[c];           // exp_avg_sq.mul_(b2).addcmul_(g, g, 1 - b2)
            const float denom = sqrtf(vv[c]) / bc2_sqrt + a.eps;
            pp[c] = pp[c] - step_size * (mm[c] / denom);
        }
        st4(a.m + p0, make_float4(mm[0], mm[1], mm[2], mm[3]));
        st4(a.v + p0, make_float4(vv[0], vv[1], vv[2], vv[3]));
        const float4 pn = make_float4(pp[0], pp[1], pp[2], pp[3]);
        st4(a.theta + p0, pn);
        if (sync_target) st4(a.theta_tgt + p0, pn);           // hard target update every tuf steps (dqn.py:208-210)
    }
    if (blockIdx.x == gridDim.x - 1) {
        // The LAST block owns no parameters (the launch carries one block more than the parameter tiles): the statistics of
        // dtqn.py:245-253,263, reduced over the per-sequence partials, the step counters and the host-visible ring slot -- a chain
        // of reductions and a PCIe write that used to sit behind block 0's Adam work and set the kernel's length
        float se = 0.f, sq = 0.f, sy = 0.f, mxq = -INFINITY, mnq = INFINITY, mxy = -INFINITY, mny = INFINITY;
        const int call_prev = a.step_counter[2];          // in flight with the partials: not a round trip of its own behind the reductions
        for (int b = tid; b < a.n_stat_parts; b += kOptThreads) {
            const float* sp = a.stats_partial + (size_t)b * 8;
            se += sp[0]; sq += sp[1]; mxq = fmaxf(mxq, sp[2]); mnq = fminf(mnq, sp[3]);
            sy += sp[4]; mxy = fmaxf(mxy, sp[5]); mny = fminf(mny, sp[6]);
        }
        se = block_sum(se, red, tid); sq = block_sum(sq, red, tid); sy = block_sum(sy, red, tid);
        for (int mk = 32; mk >= 1; mk >>= 1) {
            mxq = fmaxf(mxq, __shfl_xor(mxq, mk)); mnq = fminf(mnq, __shfl_xor(mnq, mk));
            mxy = fmaxf(mxy, __shfl_xor(mxy, mk)); mny = fminf(mny, __shfl_xor(mny, mk));
        }
        __shared__ float mm4[4][kOptThreads / 64];
        if ((tid & 63) == 0) { mm4[0][tid >> 6] = mxq; mm4[1][tid >> 6] = mnq; mm4[2][tid >> 6] = mxy; mm4[3][tid >> 6] = mny; }
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < kOptThreads / 64; ++w) {
                mxq = fmaxf(mxq, mm4[0][w]); mnq = fminf(mnq, mm4[1][w]); mxy = fmaxf(mxy, mm4[2][w]); mny = fminf(mny, mm4[3][w]);
            }
            const float cnt = (float)a.batch * (float)a.history;
            a.stats[0] = se / cnt;          // TD error (MSE loss)
            a.stats[1] = norm;              // pre-clip gradient norm
            a.stats[2] = mxq; a.stats[3] = sq / cnt; a.stats[4] = mnq;
            a.stats[5] = mxy; a.stats[6] = sy / cnt; a.stats[7] = mny;
            a.stats[8] = fminf(1.0f, a.clip / (norm + 1e-6f));
            a.stats[9] = (float)k;
            a.stats[10] = sync_target ? 1.f : 0.f;
            a.stats[11] = (float)why;         // 0 applied | 1 non-finite norm | 2 exchange timed out | 3 skipped behind an earlier 1 / 2
            if (finite) a.step_counter[1] = k;
            else a.step_counter[3] = 1;
            const int call = call_prev + 1;               // every call counts, also a skipped (non-finite) one
            a.step_counter[2] = call;
            if (a.stats_ring != nullptr) {
                // Host-visible copy: twelve 8-byte granules {value, tag}, each written by ONE system-scope store, tag = the call index
                // modulo 2^23 (exact in f32).  The host takes a slot when all twelve tags match: no fence between payload and tag.
                // (Rounds 1-4 wrote the payload, __threadfence_system(), then one tag: that fence is a write-back of this XCD's whole L2
                // -- dirty with the parameters / moments the Adam blocks have just stored -- in front of the kernel's last store.)
                unsigned long long* slot = reinterpret_cast<unsigned long long*>(a.stats_ring) + (size_t)((call - 1) % a.ring_slots) * 12;
                const unsigned long long tag = (unsigned long long)__float_as_uint((float)(call & 0x7fffff)) << 32;
                const float vals[12] = {se / cnt, norm, mxq, sq / cnt, mnq, mxy, sy / cnt, mny, fminf(1.0f, a.clip / (norm + 1e-6f)), (float)k,
                                        sync_target ? 1.f : 0.f, (float)why};
                for (int i = 0; i < 12; ++i) DTQN_SYSTEM_STORE(slot + i, tag | (unsigned long long)__float_as_uint(vals[i]));
            }
        }
    }
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

extern "C" int dtqn_xch_publish(int32_t* own_flag_dev, int32_t gen, void* stream) {
    if (!own_flag_dev) return DTQN_ERR_ARG;
    PublishArgs a;
    a.flag = own_flag_dev; a.gen = gen;
    (void)hipGetLastError();
    hipLaunchKernelGGL(dtqn_xch_publish_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? DTQN_OK : DTQN_ERR_LAUNCH;
}

extern "C" int dtqn_td_xreduce(const DtqnNet* net, const DtqnTd* td, const void* peer_grad_ptrs_dev, const void* peer_flag_ptrs_dev, int world,
                               int32_t gen, float* gsum_dev, int32_t* status_dev, int32_t* own_flag_dev, void* stream) {
    if (!net || !td || !peer_grad_ptrs_dev || !peer_flag_ptrs_dev || !gsum_dev || !status_dev) return DTQN_ERR_ARG;
    if (world < 1 || world > kOptThreads) return DTQN_ERR_ARG;
    if (td->n_norm_blocks != opt_blocks(net->n_trainable)) return DTQN_ERR_ARG;
    XreduceArgs a;
    a.peer_grad = static_cast<const float* const*>(peer_grad_ptrs_dev);
    a.peer_flag = static_cast<int32_t* const*>(peer_flag_ptrs_dev);
    a.gsum = gsum_dev; a.norm_partial = td->norm_partial; a.status = status_dev;
    a.n = net->n_trainable; a.n_parts = dtqn_td_norm_partials(net); a.world = world; a.gen = gen; a.own_flag = own_flag_dev;
    // DTQN_XCH_TIMEOUT_MS: how long a block waits for a peer's flag before it sets *status_dev (default 5000; tests use less)
    // DtqnTd.xch_timeout_ms (per engine) first, then the environment, then 5 s
    const char* tmo = getenv("DTQN_XCH_TIMEOUT_MS");
    const long long ms = td->xch_timeout_ms > 0 ? (long long)td->xch_timeout_ms : tmo != nullptr && atoll(tmo) > 0 ? atoll(tmo) : 5000ll;
    a.timeout_ticks = ms * 100000ll;
    (void)hipGetLastError();
    hipLaunchKernelGGL(dtqn_xreduce_kernel, dim3(td->n_norm_blocks), dim3(kOptThreads), 0, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? DTQN_OK : DTQN_ERR_LAUNCH;
}

extern "C" int dtqn_td_clip_adam(const DtqnNet* net, const DtqnTd* td, void* stream) {
    if (!net || !td) return DTQN_ERR_ARG;
    if (td->n_norm_blocks != opt_blocks(net->n_trainable)) return DTQN_ERR_ARG;
    AdamArgs a;
    a.theta = td->theta_pol; a.theta_tgt = td->theta_tgt; a.grad = td->grad; a.m = td->adam_m; a.v = td->adam_v;
    a.norm_partial = td->norm_partial; a.stats_partial = td->stats_partial; a.stats = td->stats;
    a.step_counter = td->step_counter;
    a.xstatus = td->xstatus;
    a.stats_ring = td->stats_ring; a.ring_slots = td->stats_ring_slots > 0 ? td->stats_ring_slots : 1;
    a.n = net->n_trainable; a.n_norm_parts = dtqn_td_norm_partials(net); a.batch = td->batch; a.history = td->history;
    a.n_stat_parts = td->batch * (td->row_split > 1 ? td->row_split : 1);
    a.tuf = td->target_update_frequency;
    a.lr = td->lr; a.beta1 = td->beta1; a.beta2 = td->beta2; a.eps = td->eps; a.clip = td->grad_norm_clip;
    a.grad_scale = td->grad_scale;
    (void)hipGetLastError();   // drop stale errors left by other users of the runtime
    hipLaunchKernelGGL(dtqn_clip_adam_kernel, dim3(td->n_norm_blocks + 1), dim3(kOptThreads), 0, (hipStream_t)stream, a);
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
