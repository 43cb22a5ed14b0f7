#include "dtqn_hip.h"
extern "C" {
int dtqn_lds_bytes_backward(const DtqnNet*) { return 0; }
int dtqn_replay_apply(const DtqnReplay*, const DtqnReplayRecord*, const float*, int, void*) { return 3; }
int dtqn_replay_sample(const DtqnReplay*, int, int, int, int, uint32_t, const int32_t*, int32_t*, int32_t*, void*) { return 3; }
int dtqn_td_backward(const DtqnNet*, const DtqnReplay*, const DtqnTd*, void*) { return 3; }
int dtqn_td_wgrad(const DtqnNet*, const DtqnTd*, void*) { return 3; }
int dtqn_td_reduce(const DtqnNet*, const DtqnTd*, void*) { return 3; }
int dtqn_td_gradnorm(const DtqnNet*, const DtqnTd*, void*) { return 3; }
int dtqn_td_clip_adam(const DtqnNet*, const DtqnTd*, void*) { return 3; }
int dtqn_target_sync(const DtqnNet*, const float*, float*, void*) { return 3; }
}
