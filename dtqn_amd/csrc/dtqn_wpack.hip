// dtqn_td_wpack: the fragment-major weight copies of dtqn_wpack.hpp, rewritten from theta_pol (F and B) and theta_tgt (F) in one launch.
#include "dtqn_device.hpp"
#include "dtqn_wpack.hpp"

namespace dtqn {

struct WPackArgs {
    WPackPlan plan;
    const float* theta_pol;
    const float* theta_tgt;
    float* pk_pol;
    float* pk_tgt;
    int f4_total;                      // float4 of all F copies
    int e_off, e_floats;               // embedding product table (dtqn_wpack.hpp), one per parameter set
    int O, V, E, KE, DO, off_tab, off_w;
};
// one float4 of output per thread: segment 0 = policy F, 1 = policy B, 2 = target F
__global__ __launch_bounds__(256) void dtqn_wpack_kernel(WPackArgs a) {
    const int per = a.f4_total, id = (int)(blockIdx.x * 256 + threadIdx.x);
    if (id >= 3 * per) {
        // embedding product table, one element per thread: P[j][v][d] = sum_c T[v][c] W_e[d][j e + c]
        int x = id - 3 * per;
        if (x >= 2 * a.e_floats) return;
        const bool tgt = x >= a.e_floats;
        x -= tgt ? a.e_floats : 0;
        const float* __restrict__ th = tgt ? a.theta_tgt : a.theta_pol;
        const int d = x % a.DO, jv = x / a.DO, v = jv % a.V, j = jv / a.V;
        const float* tr = th + a.off_tab + (size_t)v * a.E;
        const float* wr = th + a.off_w + (size_t)d * a.KE + (size_t)j * a.E;
        float p = 0.f;
        for (int c = 0; c < a.E; ++c) p = fmaf(tr[c], wr[c], p);
        (tgt ? a.pk_tgt : a.pk_pol)[a.e_off + x] = p;
        return;
    }
    const int seg = id / per;
    int o4 = id - seg * per;
    int j = 0;
    for (int k = 1; k < a.plan.n; ++k)
        if (a.plan.m[k].f_off / 4 <= o4) j = k;
    const WPackMat& m = a.plan.m[j];
    o4 -= m.f_off / 4;
    const float* __restrict__ W = (seg == 2 ? a.theta_tgt : a.theta_pol) + m.w_off;
    const int lane = o4 & 63, i = lane & 15, kq = lane >> 4, q = (o4 >> 6) & 7;
    float4 v;
    if (seg != 1) {
        const int kch = m.K / 128, kc = (o4 >> 9) % kch, ntile = (o4 >> 9) / kch;
        v = ld4(W + (size_t)(16 * ntile + i) * m.K + 128 * kc + 16 * q + 4 * kq);
        st4((seg == 0 ? a.pk_pol : a.pk_tgt) + m.f_off + (size_t)o4 * 4, v);
    } else {
        const int nch = m.N / 128, nc = (o4 >> 9) % nch, ktile = (o4 >> 9) / nch;
        const float* wp = W + (size_t)(128 * nc + 32 * kq + 4 * q) * m.K + 16 * ktile + i;
        v = make_float4(wp[0], wp[m.K], wp[2 * (size_t)m.K], wp[3 * (size_t)m.K]);
        st4(a.pk_pol + m.b_off + (size_t)o4 * 4, v);
    }
}

}  // namespace dtqn

using namespace dtqn;

extern "C" int dtqn_td_wpack_floats(const DtqnNet* net) {
    if (!net) return 0;
    return wpack_plan(*net).total;
}

extern "C" int dtqn_td_wpack(const DtqnNet* net, const DtqnTd* td, void* stream) {
    if (!net || !td) return DTQN_ERR_ARG;
    if (!td->wpack_pol || !td->wpack_tgt) return DTQN_OK;              // the caller runs on the parameter layout
    WPackArgs a;
    a.plan = wpack_plan(*net);
    if (a.plan.n == 0) return DTQN_OK;
    a.theta_pol = td->theta_pol; a.theta_tgt = td->theta_tgt; a.pk_pol = td->wpack_pol; a.pk_tgt = td->wpack_tgt;
    a.f4_total = a.plan.f_total / 4;
    a.e_off = a.plan.e_off; a.e_floats = a.plan.e_floats;
    a.O = net->obs_dim; a.V = net->vocab; a.E = net->embed_per_obs; a.KE = net->ke; a.DO = net->d_model - net->action_dim;
    a.off_tab = net->off_obs_tab; a.off_w = net->off_obs_w;
    const int blocks = (3 * a.f4_total + 2 * a.e_floats + 255) / 256;
    (void)hipGetLastError();
    hipLaunchKernelGGL(dtqn_wpack_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? DTQN_OK : DTQN_ERR_LAUNCH;
}
