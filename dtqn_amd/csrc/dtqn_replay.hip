// Device-resident replay buffer maintenance (gfx950).  Pure HBM byte work: coalesced fills and
// row writes, no arithmetic worth a matrix core.
//
// Replaces ReplayBuffer.store / store_obs / cleanse_episode (dtqn/buffers/replay_buffer.py:71-135)
// and the index draw of ReplayBuffer.sample (:141-158).  The window gather itself (:160-167) is
// fused into the forward / backward kernels, which read their (episode, start) rows in place.
#include "dtqn_device.hpp"

namespace dtqn {

struct ApplyArgs {
    DtqnReplay rp;
    const DtqnReplayRecord* recs;
    const float* obs_rows;
    int n;
};

// One workgroup applies the records IN ORDER (a store_obs cleanses the slot its later stores write), but not one record at a
// time: the records sit in pinned HOST memory (dtqn_replay_push), so every dependent read is a PCIe round trip.  The records of
// a chunk are pulled into LDS with one coalesced pass; a store_obs is a barrier-separated fill of its slot by all threads; the
// run of stores up to the next store_obs touches disjoint rows and is applied in ONE pass over (record, element) pairs, the
// observation rows read straight from the staging.  256 records of an env-step stream: 211 -> ~15 us.
__global__ __launch_bounds__(DTQN_THREADS) void dtqn_replay_apply_kernel(ApplyArgs a) {
    constexpr int CH = 256;
    constexpr int OBS_LDS = 4096;                         // floats: the chunk's observation rows ride along when they fit (vector observations)
    __shared__ DtqnReplayRecord recs[CH];
    __shared__ float obs_l[OBS_LDS];
    __shared__ int next_obs;
    static_assert(sizeof(DtqnReplayRecord) == 32, "eight dwords per record");
    const int tid = (int)threadIdx.x;
    const int T = a.rp.max_steps, O = a.rp.obs_dim;
    // Records carry obs_index = their position in the staging (ReplayBuffer._push), so a chunk's rows are the contiguous block
    // [c0, c0 + m) of obs_rows: pulled in the SAME pass as the records -- one PCIe round trip for the chunk instead of one for the
    // records, one for a store_obs row and one for the rows of every run of stores.
    const bool rows_in_lds = a.rp.obs_u8 == nullptr && CH * O <= OBS_LDS;
    for (int c0 = 0; c0 < a.n; c0 += CH) {
        const int m = a.n - c0 < CH ? a.n - c0 : CH;
        {
            const int32_t* src = reinterpret_cast<const int32_t*>(a.recs + c0);
            int32_t* dst = reinterpret_cast<int32_t*>(recs);
            for (int k = tid; k < m * 8; k += DTQN_THREADS) dst[k] = src[k];
            if (rows_in_lds)
                for (int k = tid; k < m * O; k += DTQN_THREADS) obs_l[k] = a.obs_rows[(size_t)c0 * O + k];
        }
        __syncthreads();
        // (a row outside the chunk's block -- a producer that numbers its rows differently -- is read from the staging as before)
        auto row = [&](int obs_index, int k) -> float {
            const int rel = obs_index - c0;
            return rows_in_lds && rel >= 0 && rel < m ? obs_l[rel * O + k] : a.obs_rows[(size_t)obs_index * O + k];
        };
        int i = 0;
        while (i < m) {                                   // uniform control flow: every thread walks the same LDS records
            const DtqnReplayRecord r = recs[i];
            if (r.kind == 0) {
                // cleanse_episode (:100-135): obs <- mask, actions <- 0, rewards <- 0, dones <- True, length <- 0; row 0 <- the observation
                const int ep = r.ep;
                float* obs = a.rp.obs + (size_t)ep * (T + 1) * O;
                uint8_t* act = a.rp.actions + (size_t)ep * (T + 1);
                float* rew = a.rp.rewards + (size_t)ep * T;
                uint8_t* don = a.rp.dones + (size_t)ep * T;
                if (a.rp.obs_u8 != nullptr) {      // image observations: uint8 rows (replay_buffer.py:36-45), staged as bytes
                    uint8_t* o8 = a.rp.obs_u8 + (size_t)ep * (T + 1) * O;
                    const uint8_t* s8 = reinterpret_cast<const uint8_t*>(a.obs_rows) + (size_t)r.obs_index * O;
                    for (int k = tid; k < (T + 1) * O; k += DTQN_THREADS) o8[k] = k < O ? s8[k] : (uint8_t)a.rp.obs_mask;
                } else
                for (int k = tid; k < (T + 1) * O; k += DTQN_THREADS) obs[k] = k < O ? row(r.obs_index, k) : a.rp.obs_mask;
                for (int k = tid; k < T + 1; k += DTQN_THREADS) act[k] = 0;
                for (int k = tid; k < T; k += DTQN_THREADS) { rew[k] = 0.f; don[k] = 1; }
                if (tid == 0) a.rp.ep_len[ep] = 0;
                __syncthreads();
                ++i;
                continue;
            }
            // store (:71-86) x run: obs at row t+1, action / reward / done at row t, episode length (the last store of a slot wins)
            // the run ends at the next store_obs: found by all threads at once (a serial walk over the LDS records -- one dependent
            // read and branch per record -- was 10 us of a 256-record commit)
            if (tid == 0) next_obs = -m;
            __syncthreads();
            for (int k = i + 1 + tid; k < m; k += DTQN_THREADS)
                if (recs[k].kind == 0) atomicMax(&next_obs, -k);
            __syncthreads();
            const int j = -next_obs;
            const int cnt = j - i;
            if (a.rp.obs_u8 != nullptr) {
                const uint8_t* s8 = reinterpret_cast<const uint8_t*>(a.obs_rows);
                for (int ri = i; ri < j; ++ri) {
                    const DtqnReplayRecord s = recs[ri];
                    uint8_t* dst = a.rp.obs_u8 + ((size_t)s.ep * (T + 1) + (s.t + 1)) * O;
                    for (int k = tid; k < O; k += DTQN_THREADS) dst[k] = s8[(size_t)s.obs_index * O + k];
                }
            } else
            for (int q = tid; q < cnt * O; q += DTQN_THREADS) {
                const int ri = i + q / O, k = q - (q / O) * O;
                const DtqnReplayRecord s = recs[ri];
                a.rp.obs[((size_t)s.ep * (T + 1) + (s.t + 1)) * O + k] = row(s.obs_index, k);
            }
            for (int ri = i + tid; ri < j; ri += DTQN_THREADS) {
                const DtqnReplayRecord s = recs[ri];
                a.rp.actions[(size_t)s.ep * (T + 1) + s.t] = (uint8_t)s.action;
                a.rp.rewards[(size_t)s.ep * T + s.t] = s.reward;
                a.rp.dones[(size_t)s.ep * T + s.t] = s.done ? 1 : 0;
                // episode length: the LAST store of that slot inside the run wins (the serial order of replay_buffer.py:71-86),
                // also when a producer interleaves the stores of two slots
                if (ri + 1 >= j || recs[ri + 1].ep != s.ep) {         // end of a stretch of one slot's stores (the common producer: one stretch)
                    bool last = true;
                    for (int rj = ri + 2; rj < j; ++rj)
                        if (recs[rj].ep == s.ep) { last = false; break; }
                    if (last) a.rp.ep_len[s.ep] = s.ep_len;
                }
            }
            __syncthreads();
            i = j;
        }
        __syncthreads();                                  // the next chunk overwrites the LDS records
    }
}

struct SampleArgs {
    const int32_t* ep_len;
    int n_valid, exclude, ctx_len, batch;
    uint32_t seed;
    int step;                          // >= 0: the draw's step; else step_counter[1]
    const int32_t* step_counter;
    int32_t* ep_idx;
    int32_t* start;
};

__global__ __launch_bounds__(DTQN_THREADS) void dtqn_replay_sample_kernel(SampleArgs a) {
    const int b = (int)blockIdx.x * DTQN_THREADS + (int)threadIdx.x;
    if (b >= a.batch) return;
    const uint32_t step = a.step >= 0 ? (uint32_t)a.step : a.step_counter != nullptr ? (uint32_t)a.step_counter[1] : 0u;
    int e, s;
    replay_draw(a.ep_len, a.n_valid, a.exclude, a.ctx_len, a.seed, step, b, e, s);
    a.ep_idx[b] = (int32_t)e;
    a.start[b] = (int32_t)s;
}

// ---- bags of the sampled windows (ReplayBuffer.sample_with_bag, replay_buffer.py:211-254) ---------------------------------
struct BagArgs {
    DtqnReplay rp;
    const int32_t *ep_idx, *start, *rows, *step_counter;
    int bag;
    uint32_t seed;
    float* bag_obs;
    uint8_t* bag_actions;
};
// One workgroup per window: pick the rows (host-drawn, or Floyd's sampling of `bag` distinct rows below the window), then copy
// them out of the episode -- observation row r and the action that PRECEDED it (the replay's action row r; row 0 is the dummy).
__global__ __launch_bounds__(DTQN_THREADS) void dtqn_replay_bag_kernel(BagArgs a) {
    __shared__ int32_t rows[2][DTQN_MAX_BAG];
    const int b = (int)blockIdx.x, tid = (int)threadIdx.x, bag = a.bag, O = a.rp.obs_dim, T = a.rp.max_steps;
    const int ep = a.ep_idx[b], start = a.start[b];
    if (a.rows != nullptr) {
        for (int k = tid; k < 2 * bag; k += DTQN_THREADS) rows[k / bag][k % bag] = a.rows[(size_t)b * 2 * bag + k];
    } else if (tid == 0) {
        if (start < bag) {
            for (int j = 0; j < bag; ++j) rows[0][j] = rows[1][j] = j < start ? j : -1;
        } else {
            const uint32_t step = a.step_counter != nullptr ? (uint32_t)a.step_counter[1] : 0u;
            for (int j = 0; j < bag; ++j) {            // Floyd: the j-th pick is uniform over [0, start-bag+j], or that bound if taken
                const int hi = start - bag + j;
                int r = (int)(((unsigned long long)hash_u32(a.seed, step, (uint32_t)b, 2u + (uint32_t)j) * (uint32_t)(hi + 1)) >> 32);
                for (int i = 0; i < j; ++i) if (rows[0][i] == r) { r = hi; break; }
                rows[0][j] = rows[1][j] = r;
            }
        }
    }
    __syncthreads();
    const float* obs = a.rp.obs + (size_t)ep * (T + 1) * O;
    const uint8_t* act = a.rp.actions + (size_t)ep * (T + 1);
    for (int k = tid; k < bag * O; k += DTQN_THREADS) {
        const int r = rows[0][k / O];
        a.bag_obs[((size_t)b * bag) * O + k] = r >= 0 ? obs[(size_t)r * O + k % O] : a.rp.obs_mask;
    }
    for (int k = tid; k < bag; k += DTQN_THREADS) a.bag_actions[(size_t)b * bag + k] = rows[1][k] >= 0 ? act[rows[1][k]] : (uint8_t)0;
}

}  // namespace dtqn

using namespace dtqn;

extern "C" int dtqn_replay_gather_bag(const DtqnReplay* rp, const int32_t* ep_idx_dev, const int32_t* start_dev, const int32_t* rows_dev,
                                      int batch, int bag_size, uint32_t seed, const int32_t* step_counter_dev, float* bag_obs_dev,
                                      uint8_t* bag_actions_dev, void* stream) {
    if (!rp || !ep_idx_dev || !start_dev || !bag_obs_dev || !bag_actions_dev || batch < 1) return DTQN_ERR_ARG;
    if (bag_size < 1 || bag_size > DTQN_MAX_BAG) return DTQN_ERR_ARG;
    BagArgs a;
    a.rp = *rp; a.ep_idx = ep_idx_dev; a.start = start_dev; a.rows = rows_dev; a.step_counter = step_counter_dev;
    a.bag = bag_size; a.seed = seed; a.bag_obs = bag_obs_dev; a.bag_actions = bag_actions_dev;
    (void)hipGetLastError();   // drop stale errors left by other users of the runtime
    hipLaunchKernelGGL(dtqn_replay_bag_kernel, dim3(batch), dim3(DTQN_THREADS), 0, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? DTQN_OK : DTQN_ERR_LAUNCH;
}

extern "C" int dtqn_replay_apply(const DtqnReplay* rp, const DtqnReplayRecord* recs_dev, const float* obs_rows_dev,
                                 int n, void* stream) {
    if (!rp || n < 0) return DTQN_ERR_ARG;
    if (n == 0) return DTQN_OK;
    if (!recs_dev || !obs_rows_dev) return DTQN_ERR_ARG;
    ApplyArgs a;
    a.rp = *rp; a.recs = recs_dev; a.obs_rows = obs_rows_dev; a.n = n;
    (void)hipGetLastError();   // drop stale errors left by other users of the runtime
    hipLaunchKernelGGL(dtqn_replay_apply_kernel, dim3(1), dim3(DTQN_THREADS), 0, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? DTQN_OK : DTQN_ERR_LAUNCH;
}

extern "C" int dtqn_replay_sample(const DtqnReplay* rp, int n_valid, int exclude, int ctx_len, int batch, uint32_t seed,
                                  const int32_t* step_counter_dev, int32_t* ep_idx_dev, int32_t* start_dev, void* stream) {
    return dtqn_replay_sample_at(rp, n_valid, exclude, ctx_len, batch, seed, -1, step_counter_dev, ep_idx_dev, start_dev, stream);
}
extern "C" int dtqn_replay_sample_at(const DtqnReplay* rp, int n_valid, int exclude, int ctx_len, int batch, uint32_t seed, int step,
                                     const int32_t* step_counter_dev, int32_t* ep_idx_dev, int32_t* start_dev, void* stream) {
    if (!rp || batch < 1 || !ep_idx_dev || !start_dev) return DTQN_ERR_ARG;
    const bool skip = exclude >= 0 && exclude < n_valid;
    if (n_valid - (skip ? 1 : 0) < 1) return DTQN_ERR_ARG;
    SampleArgs a;
    a.ep_len = rp->ep_len; a.n_valid = n_valid; a.exclude = exclude; a.ctx_len = ctx_len; a.batch = batch;
    a.seed = seed; a.step = step; a.step_counter = step_counter_dev; a.ep_idx = ep_idx_dev; a.start = start_dev;
    (void)hipGetLastError();   // drop stale errors left by other users of the runtime
    hipLaunchKernelGGL(dtqn_replay_sample_kernel, dim3((batch + DTQN_THREADS - 1) / DTQN_THREADS), dim3(DTQN_THREADS), 0,
                       (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? DTQN_OK : DTQN_ERR_LAUNCH;
}
