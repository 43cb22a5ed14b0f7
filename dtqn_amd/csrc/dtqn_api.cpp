// Entry points that only sequence other entry points (no device code of their own).
#include <cstdlib>

#include "dtqn_hip.h"

extern "C" const char* dtqn_build_info(void) {
#ifdef DTQN_BUILD_INFO
    return DTQN_BUILD_INFO;
#else
    return "dtqn_hip (unstamped build)";
#endif
}

static void* g_profile_buffer = nullptr;
extern "C" int dtqn_debug_set_profile_buffer(void* dev_buffer) {
    g_profile_buffer = dev_buffer;
    return DTQN_OK;
}
extern "C" void* dtqn_debug_profile_buffer(void) { return g_profile_buffer; }

// Latency mode (two workgroups per sequence): only where it pays and is covered -- the whole-sequence kernels with a
// 64-row context tile, residual gate, post-LN, and few enough sequences that every workgroup is resident at once.
extern "C" int dtqn_td_row_split(const DtqnNet* net, int batch) {
    if (!net || batch < 1) return 1;
    const char* e = getenv("DTQN_ROW_SPLIT");                 // tests / tuning: 0 = never, 1 = whenever covered
    if (e != nullptr && e[0] == '0') return 1;
    const bool covered = !net->tiled && net->lp == 64 && net->gate == DTQN_GATE_RES && !net->identity &&
                         (net->d_model == 64 || net->d_model == 128);
    if (!covered) return 1;
    if (e != nullptr && e[0] == '1') return 2;
    return 3 * batch * 2 <= 256 ? 2 : 1;                      // 256 CUs: all 3*B*2 forward workgroups resident
}
extern "C" int dtqn_td_xch_floats(const DtqnNet* net, int batch) {
    if (!net || batch < 1) return 0;
    return 3 * batch * net->num_layers * (net->lp / 2) * 2 * net->d_model;   // K | V (or dK | dV) of the lower half rows
}
extern "C" int dtqn_td_xch_flags(const DtqnNet* net, int batch) {
    if (!net || batch < 1) return 0;
    return 3 * batch * net->num_layers * 4;                   // backward: one per 64-column head group (<= 4)
}

// DtqnAgent.train() after sampling (dtqn/agents/dtqn.py:215-269) on one GPU: five launches.
extern "C" int dtqn_td_update(const DtqnNet* net, const DtqnReplay* rp, const DtqnTd* td, void* stream) {
    int rc;
    if ((rc = dtqn_td_forward(net, rp, td, stream)) != DTQN_OK) return rc;
    if ((rc = dtqn_td_backward(net, rp, td, stream)) != DTQN_OK) return rc;
    if ((rc = dtqn_td_wgrad(net, td, stream)) != DTQN_OK) return rc;
    if ((rc = dtqn_td_reduce(net, td, stream)) != DTQN_OK) return rc;
    return dtqn_td_clip_adam(net, td, stream);
}
