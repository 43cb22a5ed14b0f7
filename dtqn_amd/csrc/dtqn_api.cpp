// Entry points that only sequence other entry points (no device code of their own).
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "dtqn_hip.h"
#include "dtqn_limits.h"

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
// compute units the residency rules below are written against: the device's own count (256 on an MI355X; fewer on a partition)
static int policy_cus() {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) return 256;
    return n;
}
extern "C" int dtqn_td_row_split(const DtqnNet* net, int batch) {
    if (!net || batch < 1) return 1;
    const int cus = policy_cus();
    const char* e = getenv("DTQN_ROW_SPLIT");                 // tests / tuning: 0 = never, 1 = whenever covered
    if (e != nullptr && e[0] == '0') return 1;
    const bool covered = !net->tiled && net->lp == 64 && !net->identity &&
                         (net->d_model == 64 || (net->d_model == 128 && net->gate == DTQN_GATE_RES));   // GRU: D <= 64
    if (!covered) return 1;
    if (e != nullptr && e[0] == '1') return 2;
    if (e != nullptr && e[0] == '4') return 4;
    // 256 CUs: all 3*B*2 forward (and B*4 backward) workgroups resident at once.  The value is the number of row slices
    // of the BACKWARD kernel (4 x 16 rows); the forward kernel never uses more than two (2 x 32 rows).
    if (3 * batch * 2 <= cus) return 4;
    // Beyond latency mode (round 5) the forward runs one workgroup per sequence, and the backward chain -- half / a quarter as long per
    // slice -- is still sliced while all its workgroups fit the chip at once.  Measured at config-1 shapes (TD-updates/s, whole |
    // sliced backward; profiles/r05_row_split_policy.txt): batch 48 5 559 | 8 136 (four slices), 64 5 450 | 7 883 (four), 96 4 405 |
    // 5 378 (two; four: 4 878), 128 4 245 | 5 142 (two; four: 4 683); at 192 / 256 (a round and a half / two rounds of two-slice
    // workgroups) 0.98 / 0.99 x, so BASELINE config 2 stays whole-sequence.  d_model 64 (both gates); d_model 128 was not measured.
    if (net->d_model == 64) {
        if (batch * 4 <= cus) return 4;
        if (batch * 2 <= cus) return 2;
    }
    return 1;
}
// 1: the TD update of `batch` sequences runs in latency mode -- sliced forward passes as well (two 32-row slices in the one-call update,
// four 16-row slices in the pipelined one, dtqn_td_fwd_slices4_ok) and, pipelined, the next update's target pass inside the backward
// launch.  0 with dtqn_td_row_split > 1: only the backward is sliced.
extern "C" int dtqn_td_latency_mode(const DtqnNet* net, int batch) {
    if (dtqn_td_row_split(net, batch) < 2) return 0;
    const char* e = getenv("DTQN_ROW_SPLIT");                 // forced slices (tests / tuning) are forced in both kernels:
    if (e != nullptr && (e[0] == '1' || e[0] == '4')) return 1;   // the two values dtqn_td_row_split treats as forced, nothing else
    return 3 * batch * 2 <= policy_cus() ? 1 : 0;
}
// Both kernel families cover D = 128 / residual gate / post-LN / 64-row contexts.  Measured at BASELINE config 3 shapes (updates/s,
// whole-sequence | row-block): B = 64 1681 | 1880, B = 128 1244 | 1391, B = 256 895 | 863, B = 512 459 | 499 -- the row-block
// kernels win wherever latency mode is out of reach, except around B = 256 (a round and a half of workgroups in several of their
// launches).  D = 64 stays whole-sequence (cfg 2: 2898 | 1697).
extern "C" int dtqn_td_prefers_tiled(const DtqnNet* net, int batch) {
    if (!net || batch < 1 || net->tiled) return 0;
    // head width 32 / width-padded networks on the whole-sequence side (dtqn_limits.h, dtqn_ws_lite) train there in latency mode only
    if (dtqn_ws_lite(net->tiled, net->d_model, net->head_dim, net->d_real))
        return dtqn_td_row_split(net, batch) == 4 && dtqn_td_latency_mode(net, batch) ? 0 : 1;
    const bool covered = net->d_model == 128 && net->gate == DTQN_GATE_RES && !net->identity && net->lp == 64 && net->dropout == 0.f &&
                         net->bag_size == 0;
    if (!covered) return 0;
    const char* e = getenv("DTQN_TRAIN_TILED");
    if (e != nullptr) return atoi(e) != 0 ? 1 : 0;
    return dtqn_td_row_split(net, batch) == 1 ? 1 : 0;
}
extern "C" int dtqn_td_xch_floats(const DtqnNet* net, int batch) {
    if (!net || batch < 1) return 0;
    // K | V (or dK | dV) hand-over records per (sequence of the three passes, layer): two slices send lp / 2 rows, four slices
    // 3 lp / 4 (dtqn_forward_body.hpp kv_xch_floats)
    return 3 * batch * net->num_layers * (3 * net->lp / 4) * 2 * net->d_model;
}
extern "C" int dtqn_td_xch_flags(const DtqnNet* net, int batch) {
    if (!net || batch < 1) return 0;
    // backward: one per 64-column head group (<= 4); + the event counters of the fused weight gradients (dtqn_wgrad_direct.hpp: kFuseWords)
    // forward with four slices: six (sender, receiver) pairs per (sequence, layer)
    return 3 * batch * net->num_layers * 6 + 16;
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

// The pipelined form of the same update (latency mode, dtqn_td_fwd_slices4_ok): the two policy passes as four row slices, the
// target pass inline unless the previous update's backward launch already carried it (have_target), the backward launch carrying the
// NEXT update's target pass when td_next is given, weight gradients, reduce, clip + Adam.  draw_step: the optimizer step this
// update carries (key of its window draw); the next update's is draw_step + 1.
extern "C" int dtqn_td_gradients_pipelined(const DtqnNet* net, const DtqnReplay* rp, const DtqnTd* td, const DtqnTd* td_next, int have_target,
                                           int draw_step, void* stream) {
    int rc;
    if (draw_step < 0) return DTQN_ERR_ARG;
    if ((rc = dtqn_td_forward_part(net, rp, td, 0, 2, 4, draw_step, stream)) != DTQN_OK) return rc;
    if (!have_target && (rc = dtqn_td_forward_part(net, rp, td, 2, 1, 4, draw_step, stream)) != DTQN_OK) return rc;
    rc = td_next != nullptr ? dtqn_td_backward_ahead(net, rp, td, td_next, draw_step + 1, stream) : dtqn_td_backward(net, rp, td, stream);
    if (rc != DTQN_OK) return rc;
    if ((rc = dtqn_td_wgrad(net, td, stream)) != DTQN_OK) return rc;
    return dtqn_td_reduce(net, td, stream);
}
extern "C" int dtqn_td_update_pipelined(const DtqnNet* net, const DtqnReplay* rp, const DtqnTd* td, const DtqnTd* td_next, int have_target,
                                        int draw_step, void* stream) {
    const int rc = dtqn_td_gradients_pipelined(net, rp, td, td_next, have_target, draw_step, stream);
    return rc != DTQN_OK ? rc : dtqn_td_clip_adam(net, td, stream);
}

// Rollout staging (north_star: "pinned hipMemcpyAsync into the device buffer").  Every HIP API call costs the host
// loop microseconds, so a step of the actor loop is one library call each, with as few runtime calls inside as possible:
//  * the producer's scatter kernel reads its records and observation rows straight out of the PINNED host staging
//    (device-mapped memory, a few dozen bytes per step): no copy is enqueued at all;
//  * the actor's context goes over with one hipMemcpyAsync (it is re-read by every lane of the embedding stage, so it
//    has to be in device memory), and the Q-values of the last row come back by the forward kernel writing them into
//    pinned host memory itself.
namespace dtqn {
int forward_infer(const DtqnNet* net, const float* theta, const float* obs, const uint8_t* actions, int batch, int n,
                  float* q_out, float* q_last_host, void* stream, float* xch, int32_t* xflags, const int32_t* last_rows, int in_rows,
                  uint32_t drop_seed, uint32_t drop_step, int train_mode);
int tiled_forward_actor(const DtqnNet* net, const float* theta, const float* obs, const uint8_t* actions, int batch, int n, int in_rows,
                        float* q_out, float* workspace, int train_mode, uint32_t drop_seed, uint32_t drop_step, hipStream_t stream, const int32_t* lens = nullptr);
}
extern "C" int dtqn_forward_tiled_strided(const DtqnNet* net, const float* theta, const float* obs, const uint8_t* actions,
                                          int batch, int n, int in_rows, float* q_out, float* workspace, void* stream);

extern "C" int dtqn_replay_push(const DtqnReplay* rp, const DtqnReplayRecord* recs_host, const float* obs_host, int n,
                                void* stream) {
    if (!rp || n < 0) return DTQN_ERR_ARG;
    if (n == 0) return DTQN_OK;
    if (!recs_host || !obs_host) return DTQN_ERR_ARG;
    return dtqn_replay_apply(rp, recs_host, obs_host, n, stream);
}

extern "C" int dtqn_actor_forward(const DtqnNet* net, const float* theta, const void* ctx_host, void* ctx_dev, int n,
                                  float* q_dev, float* q_last_host, float* workspace, int train_mode, uint32_t dropout_seed,
                                  uint32_t dropout_step, void* stream) {
    if (!net || !theta || !ctx_host || !ctx_dev || !q_dev || !q_last_host) return DTQN_ERR_ARG;
    if (n < 1 || n > net->ctx_len) return DTQN_ERR_ARG;                 // dtqn.py:170-173
    hipStream_t s = (hipStream_t)stream;
    const size_t obs_bytes = sizeof(float) * (size_t)net->ctx_len * net->obs_dim;
    if (hipMemcpyAsync(ctx_dev, ctx_host, obs_bytes + (size_t)net->ctx_len, hipMemcpyHostToDevice, s) != hipSuccess) return DTQN_ERR_LAUNCH;
    const float* obs = static_cast<const float*>(ctx_dev);
    const uint8_t* actions = static_cast<const uint8_t*>(ctx_dev) + obs_bytes;
    if (!net->tiled) {
        // batch 1 is the extreme of the small-batch regime: two workgroups for the one sequence once its upper half has
        // live rows (workspace = [hand-over tiles | flags], zeroed once by the caller)
        const bool split = workspace != nullptr && n > net->lp / 2 && dtqn_td_row_split(net, 1) >= 2;
        float* xch = split ? workspace : nullptr;
        int32_t* xflags = split ? reinterpret_cast<int32_t*>(workspace + dtqn_td_xch_floats(net, 1)) : nullptr;
        return dtqn::forward_infer(net, theta, obs, actions, 1, n, q_dev, q_last_host, stream, xch, xflags, nullptr, 0, dropout_seed, dropout_step, train_mode);
    }
    const int rc = dtqn::tiled_forward_actor(net, theta, obs, actions, 1, n, n, q_dev, workspace, train_mode, dropout_seed, dropout_step, s);
    if (rc != DTQN_OK) return rc;
    if (hipMemcpyAsync(q_last_host, q_dev + (size_t)(n - 1) * net->num_actions, sizeof(float) * net->num_actions,
                       hipMemcpyDeviceToHost, s) != hipSuccess)
        return DTQN_ERR_LAUNCH;
    return DTQN_OK;
}

// The same for N actors at once (vectorised rollout: N host environments per learner, one launch per vector step).
// ctx_host is PINNED: [N][ctx_len * obs_dim f32] observations | [N][ctx_len u8] actions | [N] int32 live rows n_i (1..ctx_len),
// copied to ctx_dev (same layout) in ONE hipMemcpyAsync.  Every sequence runs n_max = max n_i rows -- attention is causal, so
// the rows behind a shorter prefix cannot reach its last live row -- and Q of row n_i - 1 of sequence i lands in the pinned
// q_last_host[i][num_actions], written by the kernel (valid once `stream` has drained).  q_dev: [N][n_max][num_actions].
// workspace as dtqn_actor_forward with batch N: dtqn_forward_workspace_floats(net, N) floats, zeroed once.
extern "C" int dtqn_actor_forward_batch(const DtqnNet* net, const float* theta, const void* ctx_host, void* ctx_dev, int n_envs,
                                        int n_max, float* q_dev, float* q_last_host, float* workspace, int train_mode,
                                        uint32_t dropout_seed, uint32_t dropout_step, void* stream) {
    if (!net || !theta || !ctx_host || !ctx_dev || !q_dev || !q_last_host || n_envs < 1) return DTQN_ERR_ARG;
    if (n_max < 1 || n_max > net->ctx_len) return DTQN_ERR_ARG;                 // dtqn.py:170-173
    hipStream_t s = (hipStream_t)stream;
    const int L = net->ctx_len;
    const size_t obs_bytes = sizeof(float) * (size_t)n_envs * L * net->obs_dim;
    const size_t act_bytes = (((size_t)n_envs * L) + 3) & ~(size_t)3;       // keeps the int32 block 4-byte aligned
    const size_t total = obs_bytes + act_bytes + sizeof(int32_t) * (size_t)n_envs;
    // the live-row counts index q_dev (tiled path) and name the rows the kernel reports (whole-sequence path): the pinned host
    // copy is already written, so a bad count is refused here instead of becoming an out-of-bounds device access
    const int32_t* lens_h = reinterpret_cast<const int32_t*>(static_cast<const uint8_t*>(ctx_host) + obs_bytes + act_bytes);
    for (int i = 0; i < n_envs; ++i)
        if (lens_h[i] < 1 || lens_h[i] > n_max) return DTQN_ERR_ARG;
    if (hipMemcpyAsync(ctx_dev, ctx_host, total, hipMemcpyHostToDevice, s) != hipSuccess) return DTQN_ERR_LAUNCH;
    const float* obs = static_cast<const float*>(ctx_dev);
    const uint8_t* actions = static_cast<const uint8_t*>(ctx_dev) + obs_bytes;
    const int32_t* lens = reinterpret_cast<const int32_t*>(static_cast<const uint8_t*>(ctx_dev) + obs_bytes + act_bytes);
    if (net->tiled) {
        // row-block tiled nets: the forward leaves Q in q_dev; the last rows come back with one small copy per actor
        const int rc = dtqn::tiled_forward_actor(net, theta, obs, actions, n_envs, n_max, L, q_dev, workspace, train_mode, dropout_seed,
                                                 dropout_step, (hipStream_t)stream, lens);
        if (rc != DTQN_OK) return rc;
        for (int i = 0; i < n_envs; ++i)
            if (hipMemcpyAsync(q_last_host + (size_t)i * net->num_actions, q_dev + ((size_t)i * n_max + (lens_h[i] - 1)) * net->num_actions,
                               sizeof(float) * net->num_actions, hipMemcpyDeviceToHost, s) != hipSuccess)
                return DTQN_ERR_LAUNCH;
        return DTQN_OK;
    }
    // few actors: two workgroups per sequence once the longest prefix reaches the upper half (workspace = hand-over tiles | flags)
    const bool split = workspace != nullptr && n_max > net->lp / 2 && dtqn_td_row_split(net, n_envs) >= 2;
    float* xch = split ? workspace : nullptr;
    int32_t* xflags = split ? reinterpret_cast<int32_t*>(workspace + dtqn_td_xch_floats(net, n_envs)) : nullptr;
    return dtqn::forward_infer(net, theta, obs, actions, n_envs, n_max, q_dev, q_last_host, stream, xch, xflags, lens, L, dropout_seed, dropout_step, train_mode);
}
