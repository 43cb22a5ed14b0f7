// Arguments of the backward kernels (dtqn_backward.hip, dtqn_backward_wl.hip).
#pragma once
#include "dtqn_device.hpp"

namespace dtqn {

struct BwdArgs {
    DtqnNet net;
    const float* theta;          // policy parameters
    const float* act;            // [B][act_stride] saved by the training forward
    float* grd;                  // [B][grd_stride]
    float* small;                // [B][sp_stride]
    const float* q3;             // [3][B][LP][AP]
    float* stats_partial;        // [B][8]
    const float* obs;            // replay arrays (actions / rewards / dones of the sampled window)
    const uint8_t* actions;
    const float* rewards;
    const uint8_t* dones;
    long long obs_ep_stride, act_ep_stride, rew_ep_stride;
    const int32_t* ep_idx;
    const int32_t* start;
    int batch, history;
    float gamma;
    long long* prof;             // debug stage clock
    float* xch;                  // row-split hand-over buffer / flags (RS > 1 only)
    int32_t* xflags;
};

// Weights-through-LDS backward (dtqn_backward_wl.hip): residual gate, post-LN layers, D <= 64.  Returns DTQN_ERR_CONFIG when
// no instantiation covers (D, MT, HD, NW, RS); wl_ok / lds_bytes tell the dispatcher whether to take it at all.
bool bwd_wl_ok(const DtqnNet* net, int rs);
int launch_bwd_wl(const BwdArgs& a, int D, int MT, int HD, int NW, int RS, hipStream_t stream);

}  // namespace dtqn
