// DTQN forward: dispatch and C entry points.  The kernel bodies live in dtqn_forward_body.hpp; their instantiations are
// compiled in dtqn_forward_inst{a,b,c,d}.hip.
#include <cstdlib>

#include "dtqn_forward_body.hpp"

namespace dtqn {
DTQN_FWD_GROUP_A(DTQN_FWD_DECL)
DTQN_FWD_GROUP_B(DTQN_FWD_DECL)
DTQN_FWD_GROUP_C(DTQN_FWD_DECL)
DTQN_FWD_GROUP_D(DTQN_FWD2_DECL)
DTQN_FWD_GROUP_E(DTQN_FWDL_DECL)

static void set_dropout(FwdArgs& a, const DtqnNet* net, int passes, uint32_t seed, uint32_t step) {
    const bool on = net->dropout > 0.f && passes != 0;
    a.drop_thresh = on ? (uint32_t)((double)net->dropout * 4294967296.0) : 0u;
    a.drop_scale = on ? 1.0f / (1.0f - net->dropout) : 1.0f;
    a.drop_seed = seed; a.drop_step = step; a.drop_passes = on ? passes : 0;
}

// mt_rows: row tiles the launch really needs (0 = the network's padded context).  Inference on a short prefix of the
// context (the actor early in an episode) runs the instantiation with fewer row tiles when there is one.
static int dispatch_fwd(const FwdArgs& a, int nseq, int row_split, hipStream_t stream, int mt_rows = 0) {
    const int D = a.net.d_model, HD = a.net.head_dim;
    const int MT = mt_rows > 0 ? mt_rows : a.net.lp / 16;
    const int NW = mt_rows > 0 ? 8 : waves_for(a.net);
    if (dtqn_ws_lite(a.net.tiled, D, HD, a.net.d_real)) {      // four slices or one workgroup per sequence, nothing else (dtqn_limits.h)
        if (a.net.identity || a.net.gate != DTQN_GATE_RES || mt_rows > 0 || (row_split == 4 && (!a.xch || !a.xflags))) return DTQN_ERR_CONFIG;
        const bool pad = a.net.d_real > 0;
        if (HD == 8 && pad) return launch_fwd_lite<8, true>(a, nseq, row_split, stream);
        if (HD == 16 && pad) return launch_fwd_lite<16, true>(a, nseq, row_split, stream);
        if (HD == 32) return pad ? launch_fwd_lite<32, true>(a, nseq, row_split, stream) : launch_fwd_lite<32, false>(a, nseq, row_split, stream);
        return DTQN_ERR_CONFIG;
    }
    if (row_split == 4) {      // four workgroups per sequence: 16-row slices (TD update, weights-through-LDS body)
        if (a.net.lp != 64 || a.net.identity || a.net.gate != DTQN_GATE_RES || !a.xch || !a.xflags) return DTQN_ERR_CONFIG;
        if (D == 64 && HD == 8) return launch_fwd2<64, 1, 8, 8, false, 4>(a, nseq, stream);
        if (D == 64 && HD == 16) return launch_fwd2<64, 1, 16, 8, false, 4>(a, nseq, stream);
        return DTQN_ERR_CONFIG;
    }
    if (row_split == 2) {      // two workgroups per sequence (dtqn_td_row_split): 32-row slices of a 64-row tile, 8 waves
        if (a.net.lp != 64 || a.net.identity || !a.xch || !a.xflags) return DTQN_ERR_CONFIG;
        if (a.net.gate == DTQN_GATE_GRU) {
            if (D == 64 && HD == 8) return launch_fwd2<64, 2, 8, 8, true, 2>(a, nseq, stream);
            if (D == 64 && HD == 16) return launch_fwd2<64, 2, 16, 8, true, 2>(a, nseq, stream);
            return DTQN_ERR_CONFIG;
        }
        if (D == 64 && HD == 8) return launch_fwd2<64, 2, 8, 8, false, 2>(a, nseq, stream);
        if (D == 64 && HD == 16) return launch_fwd2<64, 2, 16, 8, false, 2>(a, nseq, stream);
        if (D == 128 && HD == 16) return launch_fwd2<128, 2, 16, 8, false, 2>(a, nseq, stream);
        return DTQN_ERR_CONFIG;
    }
#define DTQN_FWD_CASE(d, mt, hd, nw) \
    if (D == d && MT == mt && HD == hd && NW == nw) return launch_fwd<d, mt, hd, nw>(a, nseq, stream);
    DTQN_WS_TRAIN_INSTANCES(DTQN_FWD_CASE)
    DTQN_WS_FWD_ONLY_INSTANCES(DTQN_FWD_CASE)
#undef DTQN_FWD_CASE
    return DTQN_ERR_CONFIG;
}

}  // namespace dtqn

using namespace dtqn;

extern "C" int dtqn_lds_bytes_forward(const DtqnNet* net, int /*training*/) {
    if (!net) return 0;
    const size_t b = fwd_lds_bytes(net);
    return b <= 160 * 1024 ? (int)b : 0;
}

namespace dtqn {
int forward_infer(const DtqnNet* net, const float* theta, const float* obs, const uint8_t* actions, int batch, int n,
                  float* q_out, float* q_last_host, void* stream, float* xch, int32_t* xflags, const int32_t* last_rows, int in_rows,
                  uint32_t drop_seed, uint32_t drop_step, int train_mode);
}
extern "C" int dtqn_forward(const DtqnNet* net, const float* theta, const float* obs, const uint8_t* actions,
                            int batch, int n, float* q_out, void* stream) {
    return forward_infer(net, theta, obs, actions, batch, n, q_out, nullptr, stream, nullptr, nullptr, nullptr, 0, 0u, 0u, 0);
}
// xch / xflags != nullptr: latency mode, two workgroups per sequence (the caller decided it pays: dtqn_actor_forward)
int dtqn::forward_infer(const DtqnNet* net, const float* theta, const float* obs, const uint8_t* actions, int batch, int n,
                        float* q_out, float* q_last_host, void* stream, float* xch, int32_t* xflags, const int32_t* last_rows,
                        int in_rows, uint32_t drop_seed, uint32_t drop_step, int train_mode) {
    if (!net || !theta || !obs || !q_out || batch < 1) return DTQN_ERR_ARG;
    if (n < 1 || n > net->ctx_len) return DTQN_ERR_ARG;                 // dtqn.py:170-173
    if (net->tiled) return DTQN_ERR_CONFIG;                             // use dtqn_forward_tiled
    if (net->action_dim > 0 && !actions) return DTQN_ERR_ARG;
    FwdArgs a;
    a.net = *net;
    a.theta_a = theta; a.theta_b = theta;
    a.obs = obs; a.actions = actions;
    if (in_rows <= 0) in_rows = n;              // rows per sequence in the input arrays (the batched actor packs whole contexts)
    if (in_rows < n) return DTQN_ERR_ARG;
    a.obs_ep_stride = (long long)in_rows * net->obs_dim;
    a.act_ep_stride = in_rows;
    a.ep_idx = nullptr; a.start = nullptr;
    a.n = n; a.batch = batch; a.nseq = batch; a.block0 = 0; a.pass0 = 0; a.draw_step = -1;
    a.q_out = q_out;
    a.q_which_stride = 0;
    a.q_seq_stride = (long long)n * net->num_actions;
    a.q_row_stride = net->num_actions;
    a.q_last_host = q_last_host;
    a.last_rows = last_rows;
    a.act = nullptr;
    a.xch = xch; a.xflags = xflags;
    a.ep_len = nullptr; a.step_counter = nullptr; a.ep_out = nullptr; a.start_out = nullptr;
    a.s_n_valid = 0; a.s_exclude = -1; a.s_seed = 0;
    a.prof = nullptr;
    // dropout in a train-mode actor forward (the reference's policy network stays in train mode during rollouts)
    set_dropout(a, net, train_mode ? 1 : 0, drop_seed, drop_step);
    if (xch != nullptr && xflags != nullptr) {
        // latency mode of the actor: four 16-row workgroups per sequence where that body exists and all of them are resident at
        // once (one forward of 50 rows: 38 -> 29 us per launch of the stage chain), else two 32-row ones
        const char* es = getenv("DTQN_ACTOR_SLICES");      // A/B knob: 2 = the two-slice actor of round 3
        const bool four = batch * 4 <= 256 && dtqn_td_fwd_slices4_ok(net) != 0 && a.drop_thresh == 0u && !(es != nullptr && atoi(es) == 2);
        const bool lite = dtqn_ws_lite(net->tiled, net->d_model, net->head_dim, net->d_real) != 0;       // (no two-slice kernels)
        return dispatch_fwd(a, batch, four ? 4 : lite ? 1 : 2, (hipStream_t)stream);
    }
    // short prefix of a 64-row context: 16- or 32-row instantiation (same kernel, fewer row tiles), else the full tile
    if (net->lp == 64 && n <= 32 && net->gate == DTQN_GATE_RES && !dtqn_ws_lite(net->tiled, net->d_model, net->head_dim, net->d_real)) {
        const int rc = dispatch_fwd(a, batch, 1, (hipStream_t)stream, n <= 16 ? 1 : 2);
        if (rc != DTQN_ERR_CONFIG) return rc;
    }
    return dispatch_fwd(a, batch, 1, (hipStream_t)stream);
}

// FwdArgs of a TD-update forward (all passes or a part of them); shared with the backward launch that carries the NEXT update's
// target pass (dtqn_backward.hip, dtqn_td_backward_ahead)
void dtqn::td_forward_args(const DtqnNet* net, const DtqnReplay* rp, const DtqnTd* td, int pass0, int draw_step, FwdArgs* out) {
    FwdArgs& a = *out;
    const bool draw = td->sample_in_kernel != 0;
    a.net = *net;
    a.theta_a = td->theta_pol; a.theta_b = td->theta_tgt;
    a.obs = rp->obs; a.actions = rp->actions;
    a.obs_ep_stride = (long long)(rp->max_steps + 1) * rp->obs_dim;
    a.act_ep_stride = rp->max_steps + 1;
    a.ep_idx = td->ep_idx; a.start = td->start;
    a.ep_len = draw ? rp->ep_len : nullptr; a.step_counter = td->step_counter;
    a.ep_out = td->ep_idx; a.start_out = td->start;
    a.s_n_valid = td->sample_n_valid; a.s_exclude = td->sample_exclude; a.s_seed = td->sample_seed;
    a.n = net->ctx_len; a.batch = td->batch; a.nseq = 3 * td->batch; a.block0 = 0; a.pass0 = pass0; a.draw_step = draw_step;
    a.q_out = td->q3;
    a.q_which_stride = (long long)td->batch * net->lp * net->ap;
    a.q_seq_stride = (long long)net->lp * net->ap;
    a.q_row_stride = net->ap;
    a.q_last_host = nullptr;
    a.last_rows = nullptr;
    a.act = td->act;
    a.xch = td->xch; a.xflags = td->xflags;
    a.prof = static_cast<long long*>(dtqn_debug_profile_buffer());
    set_dropout(a, net, 0x3, td->dropout_seed, 0u);       // policy(o) and policy(o') run in train mode, the target net in eval mode (dtqn.py:215-230)
}

// pass0 / npasses: the passes this launch covers (0 policy(o), 1 policy(o'), 2 target(o')); slices: workgroups per sequence (0 = the
// policy of dtqn_td_forward); draw_step >= 0: key of the in-kernel window draw (else step_counter[1])
static int td_forward_part(const DtqnNet* net, const DtqnReplay* rp, const DtqnTd* td, int pass0, int npasses, int slices, int draw_step,
                           void* stream) {
    if (!net || !rp || !td || td->batch < 1) return DTQN_ERR_ARG;
    if (rp->obs_dim != net->obs_dim || rp->max_steps < net->ctx_len) return DTQN_ERR_ARG;
    if (pass0 < 0 || npasses < 1 || pass0 + npasses > 3) return DTQN_ERR_ARG;
    const bool whole = pass0 == 0 && npasses == 3;
    const bool draw = td->sample_in_kernel != 0;
    if (draw) {
        const bool skip = td->sample_exclude >= 0 && td->sample_exclude < td->sample_n_valid;
        if (td->sample_n_valid - (skip ? 1 : 0) < 1 || !td->step_counter) return DTQN_ERR_ARG;
    }
    // image nets: the windows must exist before dtqn_img_encode, which fills td->xemb in front of this call
    if (net->img_c > 0 && (draw || !td->xemb)) return DTQN_ERR_ARG;
    if (net->tiled) {       // the multi-kernel path reads the windows from td->ep_idx / td->start: draw them first
        if (!whole && !draw) return DTQN_ERR_ARG;
        if (draw) {
            const int rc = dtqn_replay_sample_at(rp, td->sample_n_valid, td->sample_exclude, net->ctx_len, td->batch, td->sample_seed, draw_step,
                                                 td->step_counter, td->ep_idx, td->start, stream);
            if (rc != DTQN_OK) return rc;
            if (net->bag_size > 0) {      // ... and the bags of those windows
                const int rb = dtqn_replay_gather_bag(rp, td->ep_idx, td->start, nullptr, td->batch, net->bag_size, td->sample_seed,
                                                      td->step_counter, td->bag_obs, td->bag_actions, stream);
                if (rb != DTQN_OK) return rb;
            }
        }
        return tiled_td_forward_part(net, rp, td, pass0, npasses, (hipStream_t)stream);
    }
    if (!whole && !draw) return DTQN_ERR_ARG;      // a partial launch re-derives its windows from the counter-based draw
    FwdArgs a;
    td_forward_args(net, rp, td, pass0, draw_step, &a);
    if (dtqn_ws_lite(net->tiled, net->d_model, net->head_dim, net->d_real)) slices = 4;      // the only training flavour of those shapes
    if (slices <= 0) {
        slices = td->row_split >= 2 && dtqn_td_latency_mode(net, td->batch) ? 2 : 1;      // the whole-update launch: two slices in latency mode ...
        const char* e = getenv("DTQN_FWD_SLICES");         // ... DTQN_FWD_SLICES=4: four (A/B knob; 3 B 4 workgroups do not fit the chip at once)
        if (e != nullptr && atoi(e) == 4 && td->row_split >= 2 && dtqn_td_fwd_slices4_ok(net)) slices = 4;
        if (e != nullptr && atoi(e) == 1) slices = 1;      // ... =1: whole 64-row workgroups under a sliced backward (A/B knob; the records do not depend on it)
    }
    if (slices == 4 && !dtqn_td_fwd_slices4_ok(net)) return DTQN_ERR_CONFIG;
    a.nseq = npasses * td->batch;
    return dispatch_fwd(a, npasses * td->batch, slices, (hipStream_t)stream);
}

extern "C" int dtqn_td_fwd_slices4_ok(const DtqnNet* net) {
    if (!net || net->tiled || net->lp != 64 || net->identity || net->gate != DTQN_GATE_RES || net->d_model != 64 || net->dropout > 0.f) return 0;
    if (net->head_dim != 8 && net->head_dim != 16 && net->head_dim != 32) return 0;
    return fwd_wl_ok(net) && fwd_lds_bytes(net, true) <= 160 * 1024 ? 1 : 0;
}

extern "C" int dtqn_td_forward(const DtqnNet* net, const DtqnReplay* rp, const DtqnTd* td, void* stream) {
    return td_forward_part(net, rp, td, 0, 3, 0, -1, stream);
}

extern "C" int dtqn_td_forward_part(const DtqnNet* net, const DtqnReplay* rp, const DtqnTd* td, int pass0, int npasses, int slices,
                                    int draw_step, void* stream) {
    if (slices != 1 && slices != 2 && slices != 4) return DTQN_ERR_ARG;
    return td_forward_part(net, rp, td, pass0, npasses, slices, draw_step, stream);
}
