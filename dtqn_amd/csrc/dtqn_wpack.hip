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
};
// one float4 of output per thread: segment 0 = policy F, 1 = policy B, 2 = target F
__global__ __launch_bounds__(256) void dtqn_wpack_kernel(WPackArgs a) {
    const int per = a.f4_total, id = (int)(blockIdx.x * 256 + threadIdx.x);
    if (id >= 3 * per) return;
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
    const int blocks = (3 * a.f4_total + 255) / 256;
    (void)hipGetLastError();
    hipLaunchKernelGGL(dtqn_wpack_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? DTQN_OK : DTQN_ERR_LAUNCH;
}
