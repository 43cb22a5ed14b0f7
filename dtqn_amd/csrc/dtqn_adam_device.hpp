// Clip + Adam + statistics device code shared by dtqn_clip_adam_kernel (dtqn_optim.hip) and the fused tail of the
// one-launch weight-gradient kernel (dtqn_wgrad.hip).  Replaces torch.nn.utils.clip_grad_norm_(params, 1.0,
// error_if_nonfinite=True), optim.Adam.step, DqnAgent.target_update and the `.item()` statistics
// (dtqn/agents/dtqn.py:245-269, dtqn/agents/dqn.py:64,208-210).
#pragma once
#include "dtqn_device.hpp"

namespace dtqn {

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
    int n, n_norm_parts, batch, history, tuf, ring_slots;
    int n_stat_parts;            // batch * row_split per-workgroup statistics partials
    float lr, beta1, beta2, eps, clip, grad_scale;
};

// What every element update needs, derived from the global sum of squares and the step index k (1-based).
struct AdamCoef {
    float norm, coef, step_size, bc2_sqrt;
    int k;
    bool finite, sync_target;
};
// pw: two doubles of LDS.  Contains one __syncthreads().
__device__ __forceinline__ AdamCoef adam_coef(const AdamArgs& a, float total, int k, double* pw, int tid) {
    AdamCoef c;
    c.norm = sqrtf(total) * a.grad_scale;
    c.finite = isfinite(c.norm);
    c.k = k;
    if (tid == 0) {
        pw[0] = 1.0 - pow((double)a.beta1, (double)k);
        pw[1] = 1.0 - pow((double)a.beta2, (double)k);
    }
    __syncthreads();
    const float bc1 = (float)pw[0];
    c.bc2_sqrt = (float)sqrt(pw[1]);
    c.coef = fminf(1.0f, a.clip / (c.norm + 1e-6f)) * a.grad_scale;
    c.step_size = a.lr / bc1;
    c.sync_target = c.finite && a.tuf > 0 && (k % a.tuf) == 0;
    return c;
}
__device__ __forceinline__ void adam_elem(const AdamArgs& a, const AdamCoef& c, float g_raw, float& m, float& v, float& p) {
    const float g = g_raw * c.coef;
    m = m + (g - m) * (1.0f - a.beta1);                     // exp_avg.lerp_(grad, 1 - beta1)
    v = v * a.beta2 + (1.0f - a.beta2) * g * g;             // exp_avg_sq.mul_(b2).addcmul_(g, g, 1 - b2)
    const float denom = sqrtf(v) / c.bc2_sqrt + a.eps;
    p = p - c.step_size * (m / denom);
}

// Statistics of dtqn.py:245-253,263 reduced over the per-workgroup partials, step counters, host-visible ring slot.
// Called by ALL threads of ONE workgroup of NT threads; red: NT / 64 floats, mm4: 4 * NT / 64 floats of LDS.
template <int NT>
__device__ __forceinline__ void adam_statistics(const AdamArgs& a, const AdamCoef& c, float* red, float* mm4, int tid) {
    constexpr int NWV = NT / 64;
    auto bsum = [&](float v) {
        for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
        if ((tid & 63) == 0) red[tid >> 6] = v;
        __syncthreads();
        float s = 0.f;
        for (int w = 0; w < NWV; ++w) s += red[w];
        __syncthreads();
        return s;
    };
    float se = 0.f, sq = 0.f, sy = 0.f, mxq = -INFINITY, mnq = INFINITY, mxy = -INFINITY, mny = INFINITY;
    for (int b = tid; b < a.n_stat_parts; b += NT) {
        const float* sp = a.stats_partial + (size_t)b * 8;
        se += sp[0]; sq += sp[1]; mxq = fmaxf(mxq, sp[2]); mnq = fminf(mnq, sp[3]);
        sy += sp[4]; mxy = fmaxf(mxy, sp[5]); mny = fminf(mny, sp[6]);
    }
    se = bsum(se); sq = bsum(sq); sy = bsum(sy);
    for (int mk = 32; mk >= 1; mk >>= 1) {
        mxq = fmaxf(mxq, __shfl_xor(mxq, mk)); mnq = fminf(mnq, __shfl_xor(mnq, mk));
        mxy = fmaxf(mxy, __shfl_xor(mxy, mk)); mny = fminf(mny, __shfl_xor(mny, mk));
    }
    if ((tid & 63) == 0) {
        mm4[0 * NWV + (tid >> 6)] = mxq; mm4[1 * NWV + (tid >> 6)] = mnq;
        mm4[2 * NWV + (tid >> 6)] = mxy; mm4[3 * NWV + (tid >> 6)] = mny;
    }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < NWV; ++w) {
            mxq = fmaxf(mxq, mm4[0 * NWV + w]); mnq = fminf(mnq, mm4[1 * NWV + w]);
            mxy = fmaxf(mxy, mm4[2 * NWV + w]); mny = fminf(mny, mm4[3 * NWV + w]);
        }
        const float cnt = (float)a.batch * (float)a.history;
        a.stats[0] = se / cnt;          // TD error (MSE loss)
        a.stats[1] = c.norm;            // pre-clip gradient norm
        a.stats[2] = mxq; a.stats[3] = sq / cnt; a.stats[4] = mnq;
        a.stats[5] = mxy; a.stats[6] = sy / cnt; a.stats[7] = mny;
        a.stats[8] = fminf(1.0f, a.clip / (c.norm + 1e-6f));
        a.stats[9] = (float)c.k;
        a.stats[10] = c.sync_target ? 1.f : 0.f;
        a.stats[11] = c.finite ? 0.f : 1.f;
        if (c.finite) a.step_counter[1] = c.k;
        const int call = a.step_counter[2] + 1;       // every call counts, also a skipped (non-finite) one
        a.step_counter[2] = call;
        if (a.stats_ring != nullptr) {
            // host-visible copy: payload first, fence, then the tag the host polls
            float* slot = a.stats_ring + (size_t)((call - 1) % a.ring_slots) * 12;
            for (int i = 0; i < 12; ++i)
                if (i != 9) slot[i] = a.stats[i];
            __threadfence_system();
            slot[9] = (float)call;
        }
    }
}

// filled from the DtqnTd by both launchers
static inline AdamArgs adam_args(const DtqnNet* net, const DtqnTd* td, int n_norm_parts) {
    AdamArgs a;
    a.theta = td->theta_pol; a.theta_tgt = td->theta_tgt; a.grad = td->grad; a.m = td->adam_m; a.v = td->adam_v;
    a.norm_partial = td->norm_partial; a.stats_partial = td->stats_partial; a.stats = td->stats;
    a.step_counter = td->step_counter;
    a.stats_ring = td->stats_ring; a.ring_slots = td->stats_ring_slots > 0 ? td->stats_ring_slots : 1;
    a.n = net->n_trainable; a.n_norm_parts = n_norm_parts; a.batch = td->batch; a.history = td->history;
    a.n_stat_parts = td->batch * (td->row_split > 1 ? td->row_split : 1);
    a.tuf = td->target_update_frequency;
    a.lr = td->lr; a.beta1 = td->beta1; a.beta2 = td->beta2; a.eps = td->eps; a.clip = td->grad_norm_clip;
    a.grad_scale = td->grad_scale;
    return a;
}

}  // namespace dtqn
