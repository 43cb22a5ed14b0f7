// Weight gradients of the DTQN policy network: dW[N][K] = sum over all sampled tokens dY[tok][N]^T X[tok][K].
//
// Replaces the parameter-gradient half of loss.backward() (dtqn/agents/dtqn.py:256).  The contraction
// runs over the TOKEN axis (B*LP rows of the per-sequence act / grd records), so it is a real
// GEMM with K_contract = B*LP: it goes on the f32 matrix core.  A workgroup owns one 64x64 block of
// one dW and one split of the batch; a 64-wide block is FOUR INTERLEAVED 16-row MFMA tiles
// (tile c holds rows 4*i + c), so one lane's A operands for the four tiles are 4 consecutive floats
// of one dY row and its B operands 4 consecutive floats of one X row: every global access is a
// coalesced 16 B/lane load, 16 MFMAs per pair of loads.  The four waves take different sequences
// and are summed through LDS; splits are summed by dtqn_td_reduce (deterministic, no atomics).
#include "dtqn_device.hpp"

namespace dtqn {

struct WgradArgs {
    DtqnNet net;
    const DtqnWJob* jobs;
    const float* act;
    const float* grd;
    float* gsplit;          // [n_split][n_trainable]
    int batch, n_split, n_jobs;
};

__global__ __launch_bounds__(DTQN_THREADS) void dtqn_wgrad_kernel(WgradArgs a) {
    const Thr t = make_thr();
    const DtqnNet& net = a.net;
    const int LP = net.lp;
    // locate the job of this block
    const int tile = (int)blockIdx.x, split = (int)blockIdx.y;
    int j = 0;
    while (j + 1 < a.n_jobs && a.jobs[j + 1].tile0 <= tile) ++j;
    const DtqnWJob job = a.jobs[j];
    const int local = tile - job.tile0;
    const int bn = local / job.tiles_k, bk = local - bn * job.tiles_k;
    const int nbase = bn * 64, kbase = bk * 64;
    const float* xbase = (job.x_in_act ? a.act : a.grd) + job.x_off;
    const size_t xstride = job.x_in_act ? (size_t)net.act_stride : (size_t)net.grd_stride;
    const float* ybase = a.grd + job.dy_off;
    const size_t ystride = (size_t)net.grd_stride;
    // this lane's 4 consecutive dY columns / X columns
    const int ycol = nbase + 4 * t.i, xcol = kbase + 4 * t.i;
    const bool yok = ycol < job.ldy, xok = xcol < job.ldx;     // ld* are multiples of 4: whole float4 in range

    f32x4 acc[4][4];
#pragma unroll
    for (int cn = 0; cn < 4; ++cn)
#pragma unroll
        for (int ck = 0; ck < 4; ++ck) acc[cn][ck] = zero4();
    float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);

    const int b_lo = (int)((long long)a.batch * split / a.n_split);
    const int b_hi = (int)((long long)a.batch * (split + 1) / a.n_split);
    for (int b = b_lo + t.wave; b < b_hi; b += DTQN_WAVES) {
        const float* yp = ybase + (size_t)b * ystride + (size_t)t.kq * job.ldy + ycol;
        const float* xp = xbase + (size_t)b * xstride + (size_t)t.kq * job.ldx + xcol;
#pragma unroll 4
        for (int s = 0; s < LP / 4; ++s) {
            const float4 av = yok ? ld4(yp + (size_t)4 * s * job.ldy) : make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 bv = xok ? ld4(xp + (size_t)4 * s * job.ldx) : make_float4(0.f, 0.f, 0.f, 0.f);
            bsum.x += av.x; bsum.y += av.y; bsum.z += av.z; bsum.w += av.w;
            const float aa[4] = {av.x, av.y, av.z, av.w};
            const float bb[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
            for (int cn = 0; cn < 4; ++cn)
#pragma unroll
                for (int ck = 0; ck < 4; ++ck) acc[cn][ck] = mfma16(aa[cn], bb[ck], acc[cn][ck]);
        }
    }
    // bias: sum over the 4 token phases (kq) of this lane's columns
    bsum.x += __shfl_xor(bsum.x, 16); bsum.y += __shfl_xor(bsum.y, 16); bsum.z += __shfl_xor(bsum.z, 16); bsum.w += __shfl_xor(bsum.w, 16);
    bsum.x += __shfl_xor(bsum.x, 32); bsum.y += __shfl_xor(bsum.y, 32); bsum.z += __shfl_xor(bsum.z, 32); bsum.w += __shfl_xor(bsum.w, 32);

    // cross-wave sum through LDS: waves 1..3 publish, wave 0 accumulates and stores
    float* red = reinterpret_cast<float*>(dtqn_smem);          // [3][64 lanes][68]
    if (t.wave > 0) {
        float* rp = red + ((size_t)(t.wave - 1) * 64 + t.lane) * 68;
#pragma unroll
        for (int cn = 0; cn < 4; ++cn)
#pragma unroll
            for (int ck = 0; ck < 4; ++ck)
                st4(rp + (cn * 4 + ck) * 4, make_float4(acc[cn][ck][0], acc[cn][ck][1], acc[cn][ck][2], acc[cn][ck][3]));
        st4(rp + 64, bsum);
    }
    __syncthreads();
    if (t.wave == 0) {
        for (int w = 0; w < DTQN_WAVES - 1; ++w) {
            const float* rp = red + ((size_t)w * 64 + t.lane) * 68;
#pragma unroll
            for (int cn = 0; cn < 4; ++cn)
#pragma unroll
                for (int ck = 0; ck < 4; ++ck) {
                    const float4 v = ld4(rp + (cn * 4 + ck) * 4);
                    acc[cn][ck][0] += v.x; acc[cn][ck][1] += v.y; acc[cn][ck][2] += v.z; acc[cn][ck][3] += v.w;
                }
            const float4 bvv = ld4(rp + 64);
            bsum.x += bvv.x; bsum.y += bvv.y; bsum.z += bvv.z; bsum.w += bvv.w;
        }
        float* out = a.gsplit + (size_t)split * net.n_trainable;
        // acc[cn][ck][r]: n = nbase + 4*(kq*4 + r) + cn,  k = kbase + 4*i + ck
#pragma unroll
        for (int cn = 0; cn < 4; ++cn)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = nbase + 4 * (t.kq * 4 + r) + cn;
                if (n < job.N) {
#pragma unroll
                    for (int ck = 0; ck < 4; ++ck) {
                        const int k = kbase + 4 * t.i + ck;
                        if (k < job.K) out[job.w_off + (size_t)n * job.K + k] = acc[cn][ck][r];
                    }
                }
            }
        if (job.b_off >= 0 && bk == 0 && t.kq == 0) {
            const float bb[4] = {bsum.x, bsum.y, bsum.z, bsum.w};
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (ycol + c < job.N) out[job.b_off + ycol + c] = bb[c];
        }
    }
}

}  // namespace dtqn

using namespace dtqn;

extern "C" int dtqn_td_wgrad(const DtqnNet* net, const DtqnTd* td, void* stream) {
    if (!net || !td || td->batch < 1 || td->n_split < 1 || !td->wjobs) return DTQN_ERR_ARG;
    WgradArgs a;
    a.net = *net;
    a.jobs = td->wjobs;
    a.act = td->act; a.grd = td->grd; a.gsplit = td->gsplit;
    a.batch = td->batch; a.n_split = td->n_split; a.n_jobs = net->n_wjobs;
    const size_t lds = (size_t)3 * 64 * 68 * sizeof(float);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&dtqn_wgrad_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipGetLastError();   // drop stale errors left by other users of the runtime
    hipLaunchKernelGGL(dtqn_wgrad_kernel, dim3(net->n_wtiles, td->n_split), dim3(DTQN_THREADS), lds, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? DTQN_OK : DTQN_ERR_LAUNCH;
}
