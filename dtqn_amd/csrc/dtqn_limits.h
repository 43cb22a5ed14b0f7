// Coverage limits of the per-sequence fused kernels (one workgroup holds a whole context tile
// in LDS).  Anything beyond returns DTQN_ERR_CONFIG from dtqn_net_init.
#pragma once
#define DTQN_MAX_LP 64          /* padded context rows held in LDS (ctx_len <= 64) */
#define DTQN_MAX_D 128          /* d_model instantiations: 64, 128 (and 16/32 for tests) */
#define DTQN_MAX_HEAD_DIM 128   /* 4 .. 128 in the row-block attention kernels (128: contexts up to 64 rows, the LDS tile of one head); the whole-sequence kernels: 8, 16 (and 32: dtqn_ws_lite) */
#define DTQN_MAX_ACTIONS 64
#define DTQN_MAX_BAG 256        /* bag entries (bag_size <= padded context <= 256, the row-block tiled path's limit) */
#define DTQN_THREADS 256        /* 4 wave64 per workgroup */
#define DTQN_WAVES 4

// The whole-sequence kernels exist as explicit instantiations <d_model, 16-row tiles, head_dim, waves> (dtqn_forward.hip:
// dispatch_fwd, dtqn_backward.hip: td_backward): X(d, mt, hd, nw).  TRAIN = forward and backward exist; the first entry of a
// (d, mt, hd) is the default wave count, the others are reached with DTQN_WAVES (A/B switch).  dtqn_net_init places a network on
// the smallest row-tile count of its (d, hd) that holds the context, and on the row-block tiled path when there is none -- a
// shape it accepts has a kernel behind every launch.
#define DTQN_WS_TRAIN_INSTANCES(X) \
    X(64, 1, 8, 8) X(64, 2, 8, 8) X(64, 4, 8, 8) X(64, 4, 8, 4) X(64, 4, 8, 16) X(64, 4, 16, 8) \
    X(128, 4, 16, 8) X(128, 4, 16, 4) X(16, 1, 8, 4) X(16, 1, 8, 8) X(32, 2, 8, 4) X(32, 1, 16, 4)
// forward only (the actor on a short prefix of the context runs the instantiation with fewer row tiles)
#define DTQN_WS_FWD_ONLY_INSTANCES(X) X(64, 2, 16, 8) X(64, 1, 16, 8) X(128, 2, 16, 8) X(128, 1, 16, 8)

// Smallest instantiated row-tile count >= mt_needed of (d, hd) and its default wave count; 0 if there is none.
static inline int dtqn_ws_pick(int d, int hd, int mt_needed, int* nw_out) {
    int best = 0, nw = 0;
#define DTQN_WS_PICK_(D_, MT_, HD_, NW_) \
    if (d == D_ && hd == HD_ && MT_ >= mt_needed && (best == 0 || MT_ < best)) { best = MT_; nw = NW_; }
    DTQN_WS_TRAIN_INSTANCES(DTQN_WS_PICK_)
#undef DTQN_WS_PICK_
    if (nw_out) *nw_out = nw;
    return best;
}

// "Lite" whole-sequence shapes: d_model 64 (after width padding) with head width 32, or any width-padded network at head width 8 / 16 / 32
// -- residual gate, post-LN, no dropout, context <= 64 rows (dtqn_net_init checks those).  They exist as the weights-through-LDS
// forward (four 16-row slices or one 64-row tile per sequence) and the four-slice backward chain only: acting, inference and the
// latency-mode TD update (dtqn_td_row_split == 4) run there; a TD update at a larger batch runs on the row-block twin
// (dtqn_td_prefers_tiled).  Before round 5 these shapes ran on the row-block kernels throughout (2.8 x slower at batch 32).
static inline int dtqn_ws_lite_shape(int d, int hd, int padded) { return d == 64 && (hd == 32 || (padded && (hd == 8 || hd == 16))); }
static inline int dtqn_ws_lite(int tiled, int d, int hd, int d_real) { return !tiled && dtqn_ws_lite_shape(d, hd, d_real > 0); }
