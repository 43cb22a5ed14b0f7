// Entry points that only sequence other entry points (no device code of their own).
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

// DtqnAgent.train() after sampling (dtqn/agents/dtqn.py:215-269) on one GPU: five launches.
extern "C" int dtqn_td_update(const DtqnNet* net, const DtqnReplay* rp, const DtqnTd* td, void* stream) {
    int rc;
    if ((rc = dtqn_td_forward(net, rp, td, stream)) != DTQN_OK) return rc;
    if ((rc = dtqn_td_backward(net, rp, td, stream)) != DTQN_OK) return rc;
    if ((rc = dtqn_td_wgrad(net, td, stream)) != DTQN_OK) return rc;
    if ((rc = dtqn_td_reduce(net, td, stream)) != DTQN_OK) return rc;
    return dtqn_td_clip_adam(net, td, stream);
}
