/* dtqn_hip.h -- C ABI of libdtqn_hip.so, the MI355X (gfx950) engine behind dtqn_amd.
 *
 * The reference (kevslinger/DTQN) has no FFI / plugin layer: its hot path is Python calling
 * torch ops.  This header is therefore the boundary a reference maintainer would bind to
 * (ctypes stub in INTEGRATION.md): plain pointers to DEVICE memory, sizes, a hipStream_t passed
 * as void*, int status codes.  No torch types.  The library allocates nothing: every buffer is
 * owned by the caller (sizes come from dtqn_net_init).  All kernels are asynchronous on the
 * given stream.  Every entry point cites the reference interface it replaces (paths relative to
 * the reference root).
 *
 * Struct members are restricted to int32_t / uint32_t / float / pointers so that host bindings
 * can be generated from this file (dtqn_amd/_binding.py parses it).
 */
#ifndef DTQN_HIP_H
#define DTQN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DTQN_ABI_VERSION 18
#define DTQN_MAX_LAYERS 8

/* status codes */
#define DTQN_OK 0
#define DTQN_ERR_CONFIG 1      /* unsupported dims / variant (the reference would accept it) */
#define DTQN_ERR_ARG 2         /* bad argument: seq > ctx_len, NULL pointer, ... (AssertionError in dtqn.py:170-179) */
#define DTQN_ERR_LAUNCH 3      /* hip launch error */

/* gate / positional-encoding enums (dtqn/networks/dtqn.py:107-114, position_encodings.py:8-11) */
#define DTQN_GATE_RES 0
#define DTQN_GATE_GRU 1
#define DTQN_POS_LEARNED 0
#define DTQN_POS_SIN 1
#define DTQN_POS_NONE 2

/* ------------------------------------------------------------------------------------------
 * DtqnNet: network hyper-parameters (caller fills the first block) and every derived layout
 * (dtqn_net_init fills the rest).  Replaces the module tree built by DTQN.__init__
 * (dtqn/networks/dtqn.py:41-156): all parameters of one network live in ONE flat fp32 buffer
 * `theta` of n_theta floats; the first n_trainable floats are the trainable region the optimizer
 * walks (attn_mask is never materialised: causality is implicit; frozen sin / none position
 * tables sit after the trainable region).  Offsets are in floats; every tensor starts on a
 * 4-float boundary and padding stays zero.
 * ------------------------------------------------------------------------------------------ */
typedef struct DtqnNet {
    /* ---- inputs (dtqn.py:41-59 ctor arguments) ---- */
    int32_t obs_dim;          /* O: length of the observation vector */
    int32_t num_actions;      /* A */
    int32_t embed_per_obs;    /* e: per-dimension embedding width (discrete obs only) */
    int32_t action_dim;       /* a: action-embedding width, 0 = none */
    int32_t d_model;          /* D: inner_embed_size */
    int32_t num_heads;        /* H */
    int32_t num_layers;       /* NL */
    int32_t ctx_len;          /* L: history_len */
    int32_t gate;             /* DTQN_GATE_* */
    int32_t identity;         /* identity-map reordering (transformer.py:81-101) */
    int32_t pos;              /* DTQN_POS_* */
    int32_t discrete;         /* discrete observations -> Embedding(V,e)+Linear (representations.py:25-52) */
    int32_t vocab;            /* V */
    int32_t force_tiled;      /* input: 1 = lay the records out for the row-block tiled kernels even where the whole-sequence kernels
                               * cover the shape (dtqn_net_tiled_twin) */
    int32_t bag_size;         /* persistent-memory bag (utils/bag.py, dtqn.py:134-147,201-214): 0 = none.  Bag networks run on the row-block
                               * tiled path (bag_size <= padded context) */
    int32_t img_c, img_h, img_w; /* image observations (representations.py:77-130; obs_dim is the tuple (C, H, W) in the reference): > 0 selects
                               * the convolutional observation embedding -- five 3x3 convolutions (C->64 s2, 64->64, 64->64 s2, 64->128,
                               * 128->128 s2; padding 1; ReLU after each) + Flatten + Linear(128 h5 w5, D - a).  obs_dim must be C*H*W (uint8
                               * pixels, fed to the network as their float values like the reference).  Row-block tiled path; (D - a) % 16 == 0 */
    float dropout;            /* p of nn.Dropout / MultiheadAttention(dropout=p) (dtqn.py:51,105; transformer.py:34,41); train-mode
                               * forwards only; 0 = off.  Both kernel families (counter-based keep masks, recomputed in the backward) */
    int32_t d_real, heads_real, hd_real; /* width padding (0 on a fresh struct = none).  A shape outside the kernels' instantiations -- a d_model
                               * other than 64 / 128 / 256, a head width (d_model / num_heads) that is not 4, 8, 16, 32 or 64 -- is padded by
                               * dtqn_net_init (no bag, no images; head width <= 64, padded d_model <= 256; with dropout the keep masks are keyed by (row, real column)): every head
                               * to the next of those widths, then whole extra heads up to the next of 64 / 128 / 256 columns.  d_real /
                               * heads_real / hd_real keep the caller's d_model / num_heads / head width; d_model / num_heads / head_dim
                               * become the padded ones.  Every tensor of theta has the padded shape: the real entries in front, except
                               * along a head-structured axis (in_proj rows: [q | k | v][head][width]; out_proj columns: [head][width]),
                               * where each head's real entries are in front of that head's block.  The attention scale stays
                               * 1 / sqrt(hd_real).  Padded entries are zero and stay zero: zero weights and LayerNorm affines make the padded
                               * columns 0 in every activation (an all-zero head attends uniformly over zero values), the LayerNorm
                               * statistics run over the d_real real columns, its backward writes 0 into the padded ones, so every padded
                               * gradient entry is exactly 0 and Adam leaves the entry alone.  Row-block tiled path -- except a padded d_model of 64
                               * at head width 8 / 16 / 32 with the residual gate, post-LN, no dropout and a context of at most 64 rows, which
                               * (like the unpadded head width 32 at d_model 64) runs on the four-slice whole-sequence kernels: acting,
                               * inference and the latency-mode TD update there, larger batches on the row-block twin
                               * (dtqn_td_prefers_tiled; DTQN_WS_LITE_OFF=1: row-block throughout, as before round 5).  A struct that is
                               * initialised again keeps its padding (the fields are read as inputs when d_real > 0) */
    /* ---- derived: geometry ---- */
    int32_t abi_version;
    int32_t lp;               /* rows of the per-sequence records: L padded to the row count of the kernels the network runs on -- the smallest
                               * instantiated multiple of 16 of its (d_model, head_dim) on the whole-sequence kernels (dtqn_limits.h; 64 for
                               * ctx_len 50 at d_model 64), a multiple of 64 on the row-block tiled path.  Rows >= ctx_len are masked */
    int32_t ke;               /* embedding-linear fan-in: O (continuous) or O*e (discrete) */
    int32_t kep;              /* ke padded to 4 */
    int32_t ap;               /* A padded to 4 */
    int32_t head_dim;
    int32_t ffn_chunk;        /* columns of the 4D hidden processed per LDS pass */
    int32_t tiled;            /* 0: one workgroup holds a whole sequence in LDS (L <= 64, D <= 128);
                               * 1: row-block tiled kernels over the per-sequence records (L <= 256, D <= 256, and GRU gates /
                               *    identity layers at D >= 128, whose whole-sequence tile set exceeds LDS); forward and training */
    /* ---- derived: theta layout (floats) ---- */
    int32_t off_act_emb;      /* [A][a]            action_embedding.embedding.0.weight */
    int32_t off_obs_tab;      /* [V][e]            obs_embedding.observation_embedding.0.weight */
    int32_t off_obs_w;        /* [D-a][ke]         obs_embedding.observation_embedding(.2).weight  (image nets: .11.weight, ke = 128 h5 w5) */
    int32_t off_obs_b;        /* [D-a] */
    int32_t off_pos;          /* [L][D]            position_embedding.position_encoding */
    int32_t off_layer0;       /* first transformer layer block */
    int32_t layer_stride;
    int32_t lo_ln1_w, lo_ln1_b, lo_ln2_w, lo_ln2_b;           /* offsets inside a layer block */
    int32_t lo_in_w, lo_in_b, lo_out_w, lo_out_b;             /* attention.in_proj_* / out_proj.* */
    int32_t lo_f1_w, lo_f1_b, lo_f2_w, lo_f2_b;               /* ffn.0.* / ffn.2.* */
    int32_t off_gate_attn;    /* shared GRU gate (gates.py:5-31): w_r u_r w_z b_z u_z w_g u_g, each [D][D] ([D] for b_z) */
    int32_t off_gate_mlp;
    int32_t go_w_r, go_u_r, go_w_z, go_b_z, go_u_z, go_w_g, go_u_g;   /* offsets inside a gate block */
    int32_t off_head1_w, off_head1_b, off_head2_w, off_head2_b;       /* ffn.0.* / ffn.2.* (Q head; ffn.0.weight is [D][2D] with a bag) */
    int32_t off_bag_in_w, off_bag_in_b, off_bag_out_w, off_bag_out_b; /* bag_attention.in_proj_* / out_proj.* (bag_size > 0) */
    /* image nets: obs_embedding.observation_embedding.{0,2,4,6,8}.{weight [Cout][Cin][3][3], bias [Cout]} */
    int32_t off_cw0, off_cw1, off_cw2, off_cw3, off_cw4;
    int32_t off_cb0, off_cb1, off_cb2, off_cb3, off_cb4;
    int32_t img_h1, img_w1, img_h3, img_w3, img_h5, img_w5;   /* spatial sizes after the stride-2 convolutions 1, 3, 5 */
    int32_t img_feat;         /* 128 * img_h5 * img_w5 = ke of an image net */
    int32_t img_k1;           /* contraction length of the first convolution as a GEMM: 9 C padded to 16 */
    int32_t n_trainable;      /* floats the optimizer updates (multiple of 4) */
    int32_t n_theta;          /* total floats of theta */
    /* ---- derived: per-sequence saved-activation record written by the training forward ---- */
    int32_t act_stride;       /* floats per sequence */
    int32_t ao_ein, ao_x0, ao_layer0, act_layer_stride, ao_xf, ao_hh;
    int32_t al_u1, al_qkv, al_lse, al_o, al_m1, al_s1, al_st1, al_u2, al_h, al_mh, al_m2, al_s2, al_st2;
    /* al_m1 / al_mh / al_m2: ReLU activation patterns as wave ballots, one 64-bit word per
     * (16-row tile, 16-column tile, r): bit (kq*16 + i) <-> row tile*16 + kq*4 + r, column ctile*16 + i */
    int32_t al_gate1, al_gate2;   /* GRU gate records (attention / mlp gate): z, r, h~, r*x, x, y, each [LP][D] */
    /* bag branch (bag_size > 0), per sequence: embedding input [LP][kep] and embeddings [LP][D] of the bag entries (rows >= bag_size
     * zero), their k | v [LP][2D], the queries [LP][D], the attention weights [H][LP][bag_ld], the attention output [LP][D], and
     * xcat = [working memory | persistent memory] [LP][2D], the head's input */
    int32_t ao_bag_ein, ao_bag_e, ao_bag_kv, ao_bag_q, ao_bag_p, ao_bag_o, ao_xcat, bag_ld;
    /* ---- derived: per-sequence gradient record written by the backward-data kernel ---- */
    int32_t grd_stride;
    int32_t go_dx0, go_layer0, grd_layer_stride, go_dhh, go_dq;
    int32_t gl_dqkv, gl_da, gl_dhp, gl_df;
    int32_t gl_gate1, gl_gate2;   /* GRU: d z_pre, d r_pre, d h_pre, each [LP][D] */
    int32_t go_do;                /* tiled path only: dL/d(attention output) scratch [LP][D] */
    int32_t go_dcat, go_bag_do, go_bag_dq, go_bag_dkv, go_bag_de;   /* bag branch: d xcat [LP][2D], d attn-out, d q [LP][D], d k|v [LP][2D], d embeddings [LP][D] */
    /* ---- derived: per-sequence small partials (LayerNorm affine, embedding tables) ---- */
    int32_t sp_stride;
    int32_t so_ln, so_tab, so_act;      /* [NL][4][D], [V][e], [A][a] */
    int32_t sp_parts;                   /* small-partial records per sequence and backward workgroup slice: 1, or padded context / 64 on the
                                         * row-block tiled path (one per 64-row block) */
    /* ---- derived: weight-gradient job table ---- */
    int32_t n_wjobs;
    int32_t n_wtiles;         /* total 64x64 output blocks over all jobs */
} DtqnNet;

/* One weight-gradient GEMM:  dW[N][K] (+)= sum over tokens dY[t][N]^T X[t][K],  db[N] = sum_t dY[t][N].
 * x/dy offsets address the per-sequence act / grd records. */
typedef struct DtqnWJob {
    int32_t x_in_act;        /* 1: X lives in the act record, 0: in the grd record */
    int32_t x_off, ldx, K;
    int32_t dy_off, ldy, N;
    int32_t w_off;           /* offset of dW in the flat gradient */
    int32_t b_off;           /* offset of db, or -1 */
    int32_t tile0;           /* first global 64x64 block index of this job */
    int32_t tiles_n, tiles_k;
    int32_t n_layers;        /* > 1: the weights are shared by every layer (GRU gates); operands of layer l are  */
    int32_t x_lstride;       /*      at x_off + l * x_lstride / dy_off + l * dy_lstride and the products are      */
    int32_t dy_lstride;      /*      summed over l                                                               */
} DtqnWJob;

/* Fills every derived field of `net` from its inputs and decides which kernel family the network runs on (`tiled`, `lp`).  Returns
 * DTQN_ERR_CONFIG when the variant is outside what the gfx950 kernels cover; a network it accepts has a kernel
 * instantiation behind every entry point below. */
int dtqn_net_init(DtqnNet* net);
/* Writes net->n_wjobs jobs (host memory). */
int dtqn_net_wjobs(const DtqnNet* net, DtqnWJob* jobs);
/* Fills a HOST buffer of n_theta floats with the frozen tables (sinusoidal position encoding,
 * position_encodings.py:23-35); trainable entries are left untouched. */
int dtqn_net_fill_frozen(const DtqnNet* net, float* theta_host);
/* LDS bytes the forward / backward kernels request for this net (0 = does not fit). */
int dtqn_lds_bytes_forward(const DtqnNet* net, int training);
int dtqn_lds_bytes_backward(const DtqnNet* net);

/* ------------------------------------------------------------------------------------------
 * Device-resident episode-major replay (replaces dtqn/buffers/replay_buffer.py:19-69 storage).
 *   obs      [E][T+1][O] f32   (discrete observations are stored as exact small floats, as the
 *                               reference does, and converted to indices in-kernel)
 *   actions  [E][T+1]    u8
 *   rewards  [E][T]      f32
 *   dones    [E][T]      u8
 *   ep_len   [E]         i32   (the reference's uint8 wraps at 256)
 * ------------------------------------------------------------------------------------------ */
typedef struct DtqnReplay {
    float* obs;
    uint8_t* actions;
    float* rewards;
    uint8_t* dones;
    int32_t* ep_len;
    uint8_t* obs_u8;          /* image observations: [E][T+1][O] uint8 (replay_buffer.py:36-45 stores images as uint8); `obs` is NULL then */
    int32_t num_episodes;     /* E = buffer_size // max_episode_steps (replay_buffer.py:27) */
    int32_t max_steps;        /* T */
    int32_t obs_dim;          /* O */
    float obs_mask;           /* padding value of unwritten observations */
} DtqnReplay;

/* One producer record: what DtqnAgent.context_reset / observe push per env step
 * (replay_buffer.py:71-92).  kind 0 = store_obs (cleanse slot `ep`, write obs at row 0),
 * kind 1 = store (obs at row t+1, action/reward/done at row t, episode length = ep_len). */
typedef struct DtqnReplayRecord {
    int32_t kind;
    int32_t ep;
    int32_t t;
    int32_t action;
    float reward;
    int32_t done;
    int32_t obs_index;        /* row of the packed observation array that travels with the records */
    int32_t ep_len;           /* episode_lengths[ep] after this store (the `episode_length` argument) */
} DtqnReplayRecord;

/* Applies n records (device memory; staged by the host through pinned memory + hipMemcpyAsync)
 * to the replay arrays, in order.  obs_rows_dev holds one observation row [O] per RECORD, record i's row at index
 * recs[i].obs_index, and must be at least n rows long with obs_index == i for float observations of up to 16 values (the
 * kernel stages rows [c0, c0 + m) of a chunk of records in LDS in one pass; rows outside a chunk are fetched one by one). */
int dtqn_replay_apply(const DtqnReplay* rp, const DtqnReplayRecord* recs_dev, const float* obs_rows_dev,
                      int n, void* stream);
/* Draws `batch` (episode, start) pairs on the device with the reference's distribution
 * (replay_buffer.py:141-158): episode uniform over finished slots [0, n_valid) minus `exclude`,
 * start uniform on {0..max(0, len-L)}.  Counter-based RNG keyed by (seed, *step_counter). */
/* The bag half of ReplayBuffer.sample_with_bag (replay_buffer.py:211-254): for window b (episode ep_idx[b], first row start[b])
 * fill bag_obs[b] [bag_size][O] / bag_actions[b] [bag_size] with rows of the episode BEFORE the window; unused entries hold
 * (obs_mask, action 0).
 *   rows_dev != NULL: int32 [batch][2][bag_size] rows chosen by the host -- observation rows, then action rows (the reference
 *                     draws the two independently with random.sample, :232-252); -1 = unused entry.
 *   rows_dev == NULL: drawn here, keyed by (seed, step_counter[1], b): start < bag_size -> rows 0..start-1 (:221-229), else
 *                     bag_size distinct rows uniform over [0, start) (Floyd's sampling), the SAME rows for observations and actions. */
int dtqn_replay_gather_bag(const DtqnReplay* rp, const int32_t* ep_idx_dev, const int32_t* start_dev, const int32_t* rows_dev,
                           int batch, int bag_size, uint32_t seed, const int32_t* step_counter_dev, float* bag_obs_dev,
                           uint8_t* bag_actions_dev, void* stream);
int dtqn_replay_sample(const DtqnReplay* rp, int n_valid, int exclude, int ctx_len, int batch, uint32_t seed,
                       const int32_t* step_counter_dev, int32_t* ep_idx_dev, int32_t* start_dev, void* stream);
/* ... keyed by an explicit optimizer step (>= 0) instead of step_counter_dev[1]: the draw of an update other than the one in
 * flight (dtqn_td_forward_part with draw_step) */
int dtqn_replay_sample_at(const DtqnReplay* rp, int n_valid, int exclude, int ctx_len, int batch, uint32_t seed, int step,
                          const int32_t* step_counter_dev, int32_t* ep_idx_dev, int32_t* start_dev, void* stream);

/* ------------------------------------------------------------------------------------------
 * Inference / actor forward.  Replaces DTQN.forward (dtqn/networks/dtqn.py:158-218) for the
 * no-grad call sites (dtqn/agents/dtqn.py:81-107 get_action).
 *   obs      [B][n][O] f32 contiguous, actions [B][n] u8 (may be NULL when action_dim == 0)
 *   q_out    [B][n][A] f32
 * ------------------------------------------------------------------------------------------ */
int dtqn_forward(const DtqnNet* net, const float* theta, const float* obs, const uint8_t* actions,
                 int batch, int n, float* q_out, void* stream);

/* The same for nets with `tiled == 1` (contexts / widths that do not fit one workgroup's LDS, BASELINE
 * configs 4 and 5): row blocks of 64 tokens, GEMM stages on the matrix core over global-memory tensors,
 * attention per (sequence, head).  `workspace` holds dtqn_forward_workspace_floats(net, batch) floats
 * (= batch activation records). */
int dtqn_forward_workspace_floats(const DtqnNet* net, int batch);
int dtqn_forward_tiled(const DtqnNet* net, const float* theta, const float* obs, const uint8_t* actions,
                       int batch, int n, float* q_out, float* workspace, void* stream);

/* ------------------------------------------------------------------------------------------
 * One TD update = DtqnAgent.train() (dtqn/agents/dtqn.py:162-269) after sampling.
 * ------------------------------------------------------------------------------------------ */
typedef struct DtqnTd {
    /* network state */
    float* theta_pol;         /* [n_theta]  policy parameters (updated in place) */
    float* theta_tgt;         /* [n_theta]  target parameters */
    float* grad;              /* [n_trainable] flat gradient (mean over the local batch) */
    float* adam_m;            /* [n_trainable] */
    float* adam_v;            /* [n_trainable] */
    /* sampled windows */
    int32_t* ep_idx;          /* [B] sampled (episode, start) pairs: INPUT when sample_in_kernel == 0 (drawn by the host or by
                               * dtqn_replay_sample), OUTPUT of dtqn_td_forward when sample_in_kernel == 1 */
    int32_t* start;           /* [B] */
    /* workspaces */
    float* act;               /* [B][act_stride] */
    float* grd;               /* [B][grd_stride] */
    float* small;             /* [B * row_split * net.sp_parts][sp_stride] */
    float* q3;                /* [3][B][LP][AP]: Q_pol(o), Q_pol(o'), Q_tgt(o') */
    float* gsplit;            /* [n_split][n_trainable] split-K partials of the weight gradients */
    float* norm_partial;      /* [dtqn_td_norm_partials()] per-workgroup sums of squares of grad */
    float* stats_partial;     /* [B * row_split][8] */
    float* stats;             /* [12]: loss, grad_norm, q max/mean/min, target max/mean/min, clip coef, step, target-synced, non-finite flag */
    float* stats_ring;        /* optional [stats_ring_slots][12][2] in PINNED HOST memory (device-visible): every call of
                               * dtqn_td_clip_adam also writes its statistics to slot ((call_index - 1) % slots) as twelve 8-byte granules
                               * {value, tag}, tag = the 1-based call index modulo 2^23 (exact in f32), each granule ONE system-scope
                               * store: a slot is complete when all twelve tags carry the call's index.  The host polls it instead of
                               * enqueueing a device->host copy and an event per update; the kernel needs no fence */
    int32_t* step_counter;    /* [4]: [0] optimizer steps (published), [1] optimizer steps (next), [2] clip_adam calls, [3] poisoned: set by the
                               * first call that skipped its update (non-finite norm, exchange time-out); every later call skips too */
    const DtqnWJob* wjobs;    /* device copy of the job table */
    float* xch;               /* row-split exchange buffer, dtqn_td_xch_floats(net, B) floats (row_split > 1 only) */
    int32_t* xflags;          /* row-split hand-over flags, dtqn_td_xch_flags(net, B) ints, ZEROED once by the caller */
    float* bag_obs;           /* bag_size > 0: [B][bag_size][O] f32 bag observations of the sampled windows (ReplayBuffer.sample_with_bag,
                               * replay_buffer.py:171-264; filled by dtqn_replay_gather_bag -- by dtqn_td_forward itself when
                               * sample_in_kernel == 1); the same bag serves all three forwards of a sequence (dtqn.py:215-230) */
    uint8_t* bag_actions;     /* [B][bag_size] */
    const int32_t* xstatus;   /* data parallel, device-side exchange: the status word dtqn_td_xreduce sets when a peer's gradient never arrived
                               * (NULL otherwise).  dtqn_td_clip_adam skips the update while it is non-zero (stats[11] = 2) */
    const float* xemb;        /* image nets: [3 B][padded context][D - a] observation embeddings of the three forwards, produced by
                               * dtqn_img_encode_td in front of dtqn_td_forward (whose embedding stage then only adds action embeddings,
                               * positions and dropout) */
    float* wpack_pol;         /* optional, row-block networks: dtqn_td_wpack_floats(net) floats each -- fragment-major copies of the layer
                               * matrices (dtqn_td_wpack rewrites them from theta_pol / theta_tgt at the start of every row-block TD
                               * forward; the GEMM kernels then read their weight fragments 1 KB per load instruction).  NULL (either):
                               * the kernels read the parameter layout */
    float* wpack_tgt;
    /* hyper-parameters */
    int32_t batch;            /* B (local) */
    int32_t history;          /* loss over the last `history` positions (dtqn.py:240-241) */
    int32_t n_split;          /* token splits of the weight-gradient kernel */
    int32_t n_norm_blocks;
    int32_t stats_ring_slots;
    int32_t target_update_frequency;
    int32_t sample_in_kernel; /* 1: dtqn_td_forward draws the windows itself -- the same counter-based draw as dtqn_replay_sample
                               * (seed = sample_seed, step = step_counter[1]) evaluated by every workgroup for its own sequence:
                               * no separate sampling launch in front of the update */
    int32_t sample_n_valid;   /* finished episode slots [0, n_valid) */
    int32_t sample_exclude;   /* slot in progress (excluded), or -1 */
    uint32_t sample_seed;
    uint32_t dropout_seed;    /* keep masks of an update are a counter-based hash of (dropout_seed, step_counter[1], pass, sequence,
                               * site, layer, element): recomputed by the backward, nothing stored */
    int32_t row_split;        /* workgroups per sequence in the forward / backward kernels: the value
                               * dtqn_td_row_split(net, B) returned (1 = one workgroup per sequence) */
    int32_t xch_timeout_ms;   /* bounded wait of dtqn_td_xreduce for a peer's flag, in milliseconds; 0 = DTQN_XCH_TIMEOUT_MS from the
                               * environment, else 5000.  Per engine: a start-up check can use a short bound without touching the
                               * process environment */
    int32_t side_stream;      /* 1: the caller runs the NEXT update's target pass on a second stream beside this update's backward
                               * (row-block networks at small batches, learner.py enable_pipeline).  The backward then keeps 64-row
                               * workgroups where it would otherwise cut them to 32 to fill the chip (d_model 256: tl_chain_bwd_kernel,
                               * one launch per layer half) -- the other stream's kernels take the idle compute units */
    float gamma;
    float lr;
    float beta1;
    float beta2;
    float eps;
    float grad_norm_clip;
    float grad_scale;         /* multiplies the reduced gradient before clipping (1/world_size for DP) */
} DtqnTd;

/* ---- host <-> device staging of the rollout (pinned hipMemcpyAsync inside the library: one call per step) ----------- */

/* Producer side of observe() / context_reset(): applies n queued records and their observation rows to the replay.
 * recs_host / obs_host are PINNED (device-mapped) host staging; the scatter kernel reads them in place, so no copy is
 * enqueued.  The staging may be reused once work queued on `stream` after this call has completed. */
int dtqn_replay_push(const DtqnReplay* rp, const DtqnReplayRecord* recs_host, const float* obs_host, int n, void* stream);

/* get_action (dtqn/agents/dtqn.py:76-107): ctx_host is a PINNED buffer [ctx_len * obs_dim floats | ctx_len action bytes]
 * holding the n live rows of the rolling context; it is copied to ctx_dev (same layout), DTQN.forward runs on the n rows
 * (q_dev [ctx_len][num_actions]; `workspace` = dtqn_forward_workspace_floats(net, 1) floats, ZEROED once by the caller,
 * or NULL when that is 0: scratch of the tiled kernels, or the hand-over tiles of the two-workgroup latency mode) and
 * the Q-values of the LAST row land in the pinned q_last_host[num_actions] (written by the forward kernel itself; valid
 * once `stream` has drained).  All asynchronous on `stream`.  train_mode != 0 with net->dropout > 0: the forward applies
 * dropout (the reference keeps its policy network in train mode during rollouts, dqn.py:102-115), keyed by
 * (dropout_seed, dropout_step). */
int dtqn_actor_forward(const DtqnNet* net, const float* theta, const void* ctx_host, void* ctx_dev, int n, float* q_dev,
                       float* q_last_host, float* workspace, int train_mode, uint32_t dropout_seed, uint32_t dropout_step,
                       void* stream);

/* The same for N actors at once (vectorised rollout: N host environments per learner, ONE launch per vector step; the
 * reference steps one environment per forward, run.py:356-377).  ctx_host is PINNED and packs
 *   [N][ctx_len * obs_dim] f32 observations | [N][ctx_len] u8 actions (padded to a multiple of 4 bytes) | [N] int32 live rows n_i,
 * copied to ctx_dev (same size) with one hipMemcpyAsync.  Every sequence runs n_max = max n_i rows: attention is causal, so
 * rows behind a shorter prefix cannot reach its last live row.  Q of row n_i - 1 of actor i lands in the pinned
 * q_last_host[i][num_actions], written by the kernel itself (valid once `stream` has drained); q_dev is
 * [N][n_max][num_actions]; `workspace` = dtqn_forward_workspace_floats(net, N) floats, ZEROED once, or NULL when that is 0. */
int dtqn_actor_forward_batch(const DtqnNet* net, const float* theta, const void* ctx_host, void* ctx_dev, int n_envs, int n_max,
                             float* q_dev, float* q_last_host, float* workspace, int train_mode, uint32_t dropout_seed,
                             uint32_t dropout_step, void* stream);
/* DTQN.forward with the bag arguments (dtqn.py:158-218 incl. :201-214): bag_obs [B][bag_size][O] f32, bag_actions [B][bag_size] u8
 * (NULL when action_dim == 0).  Row-block tiled path; workspace as dtqn_forward_tiled.
 * train_mode != 0 with net->dropout > 0: a train-mode forward (the reference's policy network stays in train mode while the agent
 * acts and while it chooses what the bag keeps, dqn.py:102-115): keep masks keyed by (dropout_seed, dropout_step, sequence). */
int dtqn_forward_bag(const DtqnNet* net, const float* theta, const float* obs, const uint8_t* actions, const float* bag_obs,
                     const uint8_t* bag_actions, int batch, int n, float* q_out, float* workspace, int train_mode,
                     uint32_t dropout_seed, uint32_t dropout_step, void* stream);
/* dtqn_forward_tiled with `in_rows` (>= n) rows per sequence in the obs / actions arrays. */
int dtqn_forward_tiled_strided(const DtqnNet* net, const float* theta, const float* obs, const uint8_t* actions, int batch, int n,
                               int in_rows, float* q_out, float* workspace, void* stream);

/* Small-batch latency mode.  With B sampled sequences only 3B / B workgroups exist in the forward / backward
 * kernels, far fewer than the 256 CUs.  When this returns 2 the kernels run TWO workgroups per sequence, each
 * owning half of its rows (projections, LayerNorm, FFN, head and loss are row-local); causal attention is the one
 * place rows meet: the forward hands K | V of the lower rows to the upper slice, the backward hands the upper
 * queries' dK | dV contribution to the lower slice, through `xch` guarded by `xflags` (agent-scope atomics).
 * A return value of 4 means four 16-row slices in the backward kernel (pairwise dK | dV hand-overs from every slice to
 * the slices below it) and two in the forward.  Returns 1 when the shape / variant / batch does not profit (the chip
 * is already full) or is not covered.  "Fit the chip" is measured against the current device's compute-unit count
 * (hipDeviceAttributeMultiprocessorCount; 256 on an MI355X), so a partition with fewer units slices less. */
int dtqn_td_row_split(const DtqnNet* net, int batch);
/* 1: the update of `batch` sequences runs in latency mode -- sliced FORWARD passes too (dtqn_td_forward: two 32-row slices;
 * dtqn_td_update_pipelined: four 16-row slices and the next update's target pass inside the backward launch).  0 while
 * dtqn_td_row_split > 1: only the backward chain is sliced (batches past latency mode whose backward workgroups still fit the chip at
 * once: 43 ... 64 sequences in four slices, ... 128 in two, at d_model 64), the forward runs one workgroup per sequence and the
 * pipelined form is not used. */
int dtqn_td_latency_mode(const DtqnNet* net, int batch);
/* Training-path policy for shapes both kernel families cover: 1 when the TD update of `batch` sequences is faster on the
 * row-block tiled kernels than on the whole-sequence ones (D = 128, residual gate, post-LN, 64-row contexts, no dropout, batches
 * beyond latency mode: measured 460 -> 499 updates/s at BASELINE config 3).  The caller then trains with the twin of the net --
 * dtqn_net_tiled_twin: same parameters and theta layout, records laid out for the tiled kernels -- and keeps the original for the
 * actor's forwards.  DTQN_TRAIN_TILED=0|1 overrides (not for the shapes of the next sentence: they have no other kernels).  Also 1 for the shapes that exist on the whole-sequence side as four-slice
 * kernels only (head width 32 at d_model 64, width-padded networks of d_model 64) wherever dtqn_td_row_split is not 4. */
int dtqn_td_prefers_tiled(const DtqnNet* net, int batch);
int dtqn_net_tiled_twin(const DtqnNet* src, DtqnNet* dst);
int dtqn_td_xch_floats(const DtqnNet* net, int batch);
int dtqn_td_xch_flags(const DtqnNet* net, int batch);

/* The three forwards (dtqn.py:215,226,230) fused with the window gather (replay_buffer.py:160-167).
 * Grid = 3*B*row_split workgroups.  Writes q3 and the act record of the policy(o) pass. */
int dtqn_td_forward(const DtqnNet* net, const DtqnReplay* rp, const DtqnTd* td, void* stream);
/* A PART of the same launch (whole-sequence kernels, td->sample_in_kernel == 1): passes [pass0, pass0 + npasses) of
 * {0 policy(o), 1 policy(o'), 2 target(o')} with `slices` workgroups per sequence (1, 2, or 4 where dtqn_td_fwd_slices4_ok).
 * The target pass of update k + 1 depends on nothing update k produces (theta_tgt only moves at a hard sync), so it can run AHEAD,
 * beside update k's backward, which leaves half the chip idle (dtqn_td_backward_ahead carries it in the same launch) -- and passes
 * 0 - 1 of update k + 1 then run as 2 B 4 = 256 workgroups of 16 rows: every compute unit busy, half the rows per workgroup.
 * draw_step >= 0: key of the in-kernel window draw (the optimizer step the update carries); -1 = read step_counter[1].
 * Same dtqn.py:215-230. */
int dtqn_td_forward_part(const DtqnNet* net, const DtqnReplay* rp, const DtqnTd* td, int pass0, int npasses, int slices, int draw_step,
                         void* stream);
int dtqn_td_fwd_slices4_ok(const DtqnNet* net);
/* Double-DQN target, MSE, dL/dQ (dtqn.py:219-243) and the data-gradient chain of loss.backward()
 * (dtqn.py:256).  Writes the grd / small records and stats_partial. */
int dtqn_td_backward(const DtqnNet* net, const DtqnReplay* rp, const DtqnTd* td, void* stream);
/* dtqn_td_backward whose launch also carries the TARGET pass of the NEXT update (dtqn.py:230 of update k + 1) as 4 * batch more
 * workgroups: that pass depends on nothing update k computes, and the chain of dtqn_td_backward leaves half the compute units idle at
 * batch 32.  td_next: a DtqnTd whose q3 (where the pass leaves Q_tgt(o')), xch / xflags (its own hand-over buffers: the chain uses
 * td's at the same time) and sample_* fields describe update k + 1; draw_step_next: the optimizer step update k + 1 will carry.  The
 * caller runs update k + 1 as dtqn_td_forward_part(.., 0, 2, 4, draw_step_next, ..) + this function again, and uses the pass only if
 * nothing it read has changed since (replay writes, sampling range, hard target sync); otherwise dtqn_td_forward_part(.., 2, 1, 4, ..)
 * recomputes it.  Covered where dtqn_td_fwd_slices4_ok(net) and td->row_split == 4; DTQN_ERR_CONFIG elsewhere. */
int dtqn_td_backward_ahead(const DtqnNet* net, const DtqnReplay* rp, const DtqnTd* td, const DtqnTd* td_next, int draw_step_next, void* stream);
/* Weight gradients (the parameter half of loss.backward(), dtqn.py:256): token-contraction GEMMs over act x grd.
 * Large batches: split over the batch into gsplit, summed by dtqn_td_reduce.  Small batches
 * (dtqn_td_wgrad_is_direct): one launch writes grad and norm_partial itself and dtqn_td_reduce is a no-op. */
int dtqn_td_wgrad(const DtqnNet* net, const DtqnTd* td, void* stream);
int dtqn_td_wgrad_is_direct(const DtqnNet* net, int batch);
/* Recommended DtqnTd.n_split (token splits of the large-batch weight-gradient launches; gsplit holds n_split * n_trainable floats):
 * min(batch, 16), or -- row-block networks of d_model 128 / 256, whose large matrices are contracted in 128 x 128 tiles with the
 * operands staged through LDS (dtqn_wgrad_lds_kernel) -- as many as put one round of those tiles on the chip.  Any n_split >= 1 is
 * valid; the sums are deterministic for a given n_split. */
int dtqn_td_wgrad_splits(const DtqnNet* net, int batch);
/* Fragment-major weight copies of the row-block GEMM kernels (round 6; the matrices stay the reference's nn.Linear weights,
 * dtqn/networks/transformer.py:28-61 -- only the order in which a wave finds their elements changes).  dtqn_td_wpack_floats: floats
 * of DtqnTd.wpack_pol / wpack_tgt (0: the network is not covered -- d_model not a multiple of 128, bag networks, whole-sequence
 * networks, DTQN_WPACK=0).  dtqn_td_wpack: rewrite both from theta_pol / theta_tgt; dtqn_td_forward calls it itself on row-block
 * networks, so a caller only needs it in front of a dtqn_td_forward_part sequence it assembles by hand. */
int dtqn_td_wpack_floats(const DtqnNet* net);
int dtqn_td_wpack(const DtqnNet* net, const DtqnTd* td, void* stream);
/* Fused weight gradients (latency mode, row_split == 4, small batches; reference: the same loss.backward(), dtqn.py:256): the
 * dtqn_td_backward launch carries extra workgroups on the compute units its row slices leave idle; they contract a layer's
 * weight gradients as soon as every sequence has published that layer's gradient records (write-through stores + event
 * counters behind DtqnTd.xflags), while the data-gradient chain works on the layers below.  Returns 1 when dtqn_td_backward
 * does so for this (net, td): it then writes grad / norm_partial / step_counter[0] itself and dtqn_td_wgrad launches nothing. */
int dtqn_td_wgrad_is_fused(const DtqnNet* net, const DtqnTd* td);
/* Sums gsplit / small partials into grad (mean-loss scaling is already in dL/dQ), writes
 * norm_partial.  After this call `grad` is ready for a data-parallel all-reduce. */
int dtqn_td_reduce(const DtqnNet* net, const DtqnTd* td, void* stream);
/* Number of floats DtqnTd.norm_partial must hold (>= n_norm_blocks). */
int dtqn_td_norm_partials(const DtqnNet* net);
/* Recomputes norm_partial from `grad` (used after an all-reduce changed it). */
int dtqn_td_gradnorm(const DtqnNet* net, const DtqnTd* td, void* stream);
/* clip_grad_norm_(1.0) + Adam + step counters + hard target sync every tuf steps
 * (dtqn.py:257-269, dqn.py:64,208-210) + final stats.  Non-finite norm: sets stats[11] = 1 and skips
 * the update (the host raises RuntimeError like error_if_nonfinite=True); stats[11] = 2: the device-side exchange
 * timed out (DtqnTd.xstatus); 3: skipped because an earlier call was (sticky, step_counter[3]). */
int dtqn_td_clip_adam(const DtqnNet* net, const DtqnTd* td, void* stream);
/* ---- device-side gradient exchange of the data-parallel update (new; the reference is single-process) ----
 * One process per GPU.  Every rank owns an exchange buffer gx[2][n_trainable] (two generations) and a flag word, exported to its
 * peers over HIP IPC / peer access (the caller maps them; this library only sees device pointers).  Update k (k = 1, 2, ...):
 *   1. the rank points DtqnTd.grad at its gx[k & 1] and runs forward / backward / wgrad / reduce: its local mean gradient lands there;
 *   2. dtqn_xch_publish raises its flag word to k (system-scope store; the kernel boundary in front of it has made the gradient
 *      visible beyond this GPU's caches);
 *   3. dtqn_td_xreduce: every block waits (bounded spin, system-scope loads) until all `world` flag words have reached k, then sums
 *      its 1024 parameters over the `world` buffers IN RANK ORDER -- every rank computes the same bits, so replicas stay identical
 *      without an all-gather --, writes the sum to gsum and its sum of squares to DtqnTd.norm_partial;
 *   4. the rank points DtqnTd.grad at gsum and runs dtqn_td_clip_adam with grad_scale = 1 / world.
 * Replaces {RCCL all-reduce + dtqn_td_gradnorm} with one launch and no library call.  Two generations make step 1 of update k + 2
 * safe: a rank gets past step 3 of update k + 1 only after every peer published k + 1, i.e. finished reading generation k.
 *   peer_grad_ptrs_dev  device array of `world` pointers (const float*): generation (gen & 1) of every rank's buffer, own included
 *   peer_flag_ptrs_dev  device array of `world` pointers (int32_t*): every rank's flag word
 *   own_flag_dev        this rank's flag word, or NULL.  Not NULL: the launch raises it to `gen` itself before it waits (step 2 folded
 *                       into step 3: the kernel boundary in front of this launch is the same one dtqn_xch_publish relied on)
 *   status_dev          int32: set to 1 by a block whose wait ran out (5 s; DTQN_XCH_TIMEOUT_MS overrides): the caller raises instead of
 *                       hanging the GPU.  The sum is then stale: point DtqnTd.xstatus at this word and dtqn_td_clip_adam skips the update */
int dtqn_xch_publish(int32_t* own_flag_dev, int32_t gen, void* stream);
int dtqn_td_xreduce(const DtqnNet* net, const DtqnTd* td, const void* peer_grad_ptrs_dev, const void* peer_flag_ptrs_dev, int world,
                    int32_t gen, float* gsum_dev, int32_t* status_dev, int32_t* own_flag_dev, void* stream);
/* Convenience: forward, backward, wgrad, reduce, clip_adam back to back (single GPU). */
int dtqn_td_update(const DtqnNet* net, const DtqnReplay* rp, const DtqnTd* td, void* stream);
/* dtqn_td_update in its pipelined form (latency mode; dtqn_td_forward_part, dtqn_td_backward_ahead): policy passes as four row
 * slices, target pass inline unless have_target (the previous call's backward launch carried it into td->q3), the backward launch
 * carrying update k + 1's target pass when td_next != NULL, wgrad, reduce, clip_adam.  draw_step = optimizer steps taken so far. */
int dtqn_td_update_pipelined(const DtqnNet* net, const DtqnReplay* rp, const DtqnTd* td, const DtqnTd* td_next, int have_target,
                             int draw_step, void* stream);
/* ... the same up to the gradient (forward parts, backward, wgrad, reduce), for callers that put something between the gradient and the
 * optimizer launch: the data-parallel exchange, or the wait for an actor forward that still reads theta (two-stream rollout). */
int dtqn_td_gradients_pipelined(const DtqnNet* net, const DtqnReplay* rp, const DtqnTd* td, const DtqnTd* td_next, int have_target,
                                int draw_step, void* stream);
/* Hard target update theta_tgt <- theta_pol (dqn.py:208-210). */
int dtqn_target_sync(const DtqnNet* net, const float* theta_pol, float* theta_tgt, void* stream);

/* ------------------------------------------------------------------------------------------
 * Image observation embedding (representations.py:77-130, reached from dtqn/networks/dtqn.py:71-77 when obs_dim is a tuple).
 * Implicit-GEMM 3x3 convolutions on the f32 matrix core, activations NHWC in a caller-owned workspace; weights are read from
 * `theta` in the reference's layout through per-update transposed copies (`wprep`).  MFMA-bound: 2 * 9 * Cin * Cout FLOP per
 * output pixel.
 * ------------------------------------------------------------------------------------------ */
/* floats of the transposed-weight scratch (forward and backward operand layouts of every layer) */
int dtqn_img_prep_floats(const DtqnNet* net);
/* floats of an activation workspace for `tokens` images (the five NHWC feature maps) */
long long dtqn_img_act_floats(const DtqnNet* net, int tokens);
/* floats of the weight-gradient partial buffer */
long long dtqn_img_wpart_floats(const DtqnNet* net);
/* theta -> wprep (call once per parameter version, before encode / backward) */
int dtqn_img_prep(const DtqnNet* net, const float* theta, float* wprep, void* stream);
/* Encoder forward of `tokens` images.  images_u8: base of a uint8 image array (O = C*H*W bytes each); img_index [tokens]: image
 * number of every token (device).  act: dtqn_img_act_floats(net, tokens) floats (kept for dtqn_img_backward).  Every token t writes
 * its embedding [D - a] to out0 + dst0[t] * (D - a) and, when out1 != NULL and dst1[t] >= 0, also to out1 + dst1[t] * (D - a)
 * (dst arrays on the device; a window's rows 1..L-1 serve both policy(o) and policy(o')). */
int dtqn_img_encode(const DtqnNet* net, const float* theta, const float* wprep, const uint8_t* images_u8, const int32_t* img_index,
                    int tokens, float* act, float* out0, const int32_t* dst0, float* out1, const int32_t* dst1, void* stream);
/* Encoder backward for the `tokens` images of the LAST dtqn_img_encode on `act`: dxemb_base + dsrc[t] (float offsets, device
 * array; < 0: the token has no gradient) is dL/d(embedding) [D - a] of token t (the grd records' dx0 columns a..D).
 * gact: two gradient ping-pong buffers of max feature-map size (dtqn_img_gact_floats); wpart: dtqn_img_wpart_floats.  The
 * gradients of the convolutions and of the embedding linear are WRITTEN to grad_out (flat gradient layout, e.g. split 0 of
 * DtqnTd.gsplit) at their theta offsets. */
long long dtqn_img_gact_floats(const DtqnNet* net, int tokens);
int dtqn_img_backward(const DtqnNet* net, const float* theta, const float* wprep, const uint8_t* images_u8, const int32_t* img_index,
                      int tokens, const float* act, const float* dxemb_base, const int32_t* dsrc, float* gact, float* wpart,
                      float* grad_out, void* stream);
/* Token lists of a TD update: policy rows 0..L of every sampled window (B (L + 1) tokens; row r feeds policy(o) position r and
 * policy(o') position r - 1) and target rows 1..L (B L tokens).  Fills img_index / dst0 / dst1 / dsrc (device, B (L + 1) ints
 * each; the target list reuses the first B L entries of its own arrays). */
int dtqn_img_td_lists(const DtqnNet* net, const DtqnReplay* rp, const DtqnTd* td, int32_t* pol_index, int32_t* pol_dst0, int32_t* pol_dst1,
                      int32_t* pol_dsrc, int32_t* tgt_index, int32_t* tgt_dst0, void* stream);
/* DTQN.forward for image nets on precomputed embeddings: xemb [batch][n][D - a] (from dtqn_img_encode); otherwise as dtqn_forward_tiled */
int dtqn_forward_tiled_pre(const DtqnNet* net, const float* theta, const float* xemb, const uint8_t* actions, int batch, int n,
                           float* q_out, float* workspace, int train_mode, uint32_t dropout_seed, uint32_t dropout_step, void* stream);

/* Debug aid: when a device buffer of >= 2*64 int64 is registered, workgroup 0 of the forward (slots
 * 0..63) and backward (slots 64..127) TD kernels records the 100 MHz wall clock at its stage
 * boundaries into it.  NULL disables (default). */
int dtqn_debug_set_profile_buffer(void* dev_buffer);
/* Tests: workgroups of the last fused layer launch (tl_layer_kernel) if it walked packed rows (the live rows of policy(o') and target(o') 64 at
 * a time, dtqn_tiled.hip TlPack), else 0. */
int dtqn_debug_last_packed_blocks(void);

/* Library self-description. */
int dtqn_abi_version(void);
const char* dtqn_build_info(void);

#ifdef __cplusplus
}
#endif
#endif /* DTQN_HIP_H */
