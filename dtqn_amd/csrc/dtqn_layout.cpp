// Host-side layout of the flat parameter buffer, the per-sequence activation / gradient records
// and the weight-gradient job table (see include/dtqn_hip.h, DtqnNet).  Pure C++ (no device code).
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "dtqn_hip.h"
#include "dtqn_limits.h"

namespace {
inline int up4(int x) { return (x + 3) & ~3; }
inline int up16(int x) { return (x + 15) & ~15; }

struct Cursor {
    int pos = 0;
    int take(int n) {
        int at = pos;
        pos += up4(n);
        return at;
    }
};
}  // namespace

extern "C" int dtqn_abi_version(void) { return DTQN_ABI_VERSION; }

extern "C" int dtqn_net_tiled_twin(const DtqnNet* src, DtqnNet* dst) {
    if (!src || !dst) return DTQN_ERR_ARG;
    *dst = *src;                      // the input fields; dtqn_net_init recomputes everything derived
    dst->force_tiled = 1;
    return dtqn_net_init(dst);
}

extern "C" int dtqn_net_init(DtqnNet* net) {
    if (!net) return DTQN_ERR_ARG;
    const int O = net->obs_dim, A = net->num_actions, e = net->embed_per_obs, a = net->action_dim;
    const int NL = net->num_layers, L = net->ctx_len, V = net->vocab;
    if (net->d_real > 0) {            // a padded network initialised again (dtqn_net_tiled_twin, a copy): back to the caller's shape first
        net->d_model = net->d_real;
        net->num_heads = net->heads_real;
    }
    net->d_real = net->heads_real = net->hd_real = 0;
    if (O < 1 || A < 1 || net->d_model < 16 || net->num_heads < 1 || NL < 1 || NL > DTQN_MAX_LAYERS || L < 1) return DTQN_ERR_CONFIG;
    if (net->d_model % net->num_heads != 0 || a < 0 || a >= net->d_model || (a % 4) != 0) return DTQN_ERR_CONFIG;
    {
        // Width padding (include/dtqn_hip.h, d_real): a shape the kernels are not instantiated for runs as the next one that is -- every head
        // at the next instantiated head width, then whole extra heads (all-zero heads attend uniformly over zero values and contribute
        // nothing) up to the next instantiated d_model.
        const int d = net->d_model, h = net->num_heads, hd = d / h;
        const int hdp = hd <= 4 ? 4 : hd <= 8 ? 8 : hd <= 16 ? 16 : hd <= 32 ? 32 : hd <= 64 ? 64 : 128;
        bool native = (d == 16 || d == 32 || d == 64 || d == 128 || d == 256) && hd == hdp;
        // widths 16 / 32 exist on the whole-sequence kernels only (a few row counts): beyond those they run padded to 64 columns
        if (native && d < 64 && dtqn_ws_pick(d, hd, up16(L) / 16, nullptr) == 0) native = false;
        if (!native) {
            const int dmin = h * hdp, dp = dmin <= 64 ? 64 : dmin <= 128 ? 128 : 256;
            // (an action embedding is fine: a token is [action embedding (a) | observation embedding (D - a)], dtqn.py:192 -- the action
            //  columns come FIRST, so the real columns stay a prefix of the padded row and the padded rows of the observation Linear land
            //  behind them)
            if (hd > 128 || dmin > 256 || net->bag_size != 0 || net->img_c > 0) return DTQN_ERR_CONFIG;      // (dropout: the keep masks are keyed by (row, REAL column), TlDrop.dw)
            net->d_real = d;
            net->heads_real = h;
            net->hd_real = hd;
            net->d_model = dp;
            net->num_heads = dp / hdp;
        }
    }
    const int D = net->d_model, H = net->num_heads;
    if (D % 16 != 0 || D % H != 0) return DTQN_ERR_CONFIG;
    if (net->discrete && (V < 1 || e < 1)) return DTQN_ERR_CONFIG;
    if (net->gate != DTQN_GATE_RES && net->gate != DTQN_GATE_GRU) return DTQN_ERR_CONFIG;
    if (net->pos < DTQN_POS_LEARNED || net->pos > DTQN_POS_NONE) return DTQN_ERR_CONFIG;
    if (!(net->dropout >= 0.f && net->dropout < 1.f)) return DTQN_ERR_CONFIG;
    if (net->bag_size < 0) return DTQN_ERR_CONFIG;
    const bool img = net->img_c > 0;
    if (img) {
        // representations.py:77-130: obs_dim = (C, H, W); convolution strides 2, 1, 2, 1, 2, padding 1
        const int C = net->img_c, Hh = net->img_h, Ww = net->img_w;
        // (no action embedding next to the image embedding: the encoder's Linear writes all D columns; widths of the row-block path)
        if (Hh < 1 || Ww < 1 || C > 3 || O != C * Hh * Ww || net->discrete || a != 0 || net->bag_size > 0) return DTQN_ERR_CONFIG;
        if (!(D == 64 || D == 128 || D == 256)) return DTQN_ERR_CONFIG;
        auto half = [](int x) { return (x - 1) / 2 + 1; };
        net->img_h1 = half(Hh); net->img_w1 = half(Ww);
        net->img_h3 = half(net->img_h1); net->img_w3 = half(net->img_w1);
        net->img_h5 = half(net->img_h3); net->img_w5 = half(net->img_w3);
        net->img_feat = 128 * net->img_h5 * net->img_w5;
        net->img_k1 = up16(9 * C);
    } else {
        net->img_h = net->img_w = 0;
        net->img_h1 = net->img_w1 = net->img_h3 = net->img_w3 = net->img_h5 = net->img_w5 = net->img_feat = net->img_k1 = 0;
    }
    net->abi_version = DTQN_ABI_VERSION;
    net->lp = up16(L);
    net->tiled = 0;
    // head width 32 at d_model 64 and the width-padded shapes of d_model 64: the four-slice / whole-tile kernels of dtqn_limits.h
    // (dtqn_ws_lite) where the variant is covered -- residual gate, post-LN, no dropout, a context of at most 64 rows
    const bool lite = dtqn_ws_lite_shape(D, D / H, net->d_real > 0) && net->lp <= DTQN_MAX_LP && net->gate == DTQN_GATE_RES && !net->identity &&
                      net->dropout == 0.f && net->bag_size == 0 && !img && getenv("DTQN_FORCE_TILED") == nullptr && net->force_tiled == 0 &&
                      up4(img ? 0 : (net->discrete ? O * e : O)) <= 3 * D &&      // (the embedding operands of the whole-sequence kernels)
                      getenv("DTQN_WS_LITE_OFF") == nullptr;         // (A/B knob: the row-block path of rounds 1-4)
    if (net->lp > DTQN_MAX_LP || D > DTQN_MAX_D || getenv("DTQN_FORCE_TILED") != nullptr || net->force_tiled != 0 || net->bag_size > 0 || img ||
        (net->d_real > 0 && !lite)) {
        // does not fit one workgroup's LDS: row-block tiled path (64-row blocks)
        net->tiled = 1;
        net->lp = (L + 63) / 64 * 64;
    }
    net->ke = img ? net->img_feat : (net->discrete ? O * e : O);
    net->kep = up4(net->ke);
    net->ap = up4(A);
    net->head_dim = D / H;
    net->ffn_chunk = 2 * D;
    // kernels cover what fits the per-sequence LDS tile (DESIGN.md "coverage")
    if (net->head_dim > DTQN_MAX_HEAD_DIM || (net->head_dim % 4) != 0) return DTQN_ERR_CONFIG;
    if (!net->tiled) {
        // whole-sequence kernels are explicit instantiations (dtqn_limits.h): the smallest row-tile count of this (d_model, head_dim)
        // that holds the context (a context of 8 at d_model 64 runs the 16-row kernels, head_dim 16 at d_model 64 the 64-row
        // ones: rows past the context are masked like rows 50..63 of BASELINE config 1); none -> the row-block tiled path
        const int mt = lite ? 4 : dtqn_ws_pick(D, net->head_dim, net->lp / 16, nullptr);
        if (mt > 0) net->lp = 16 * mt;
        if (mt == 0 || ((D == 128 || D == 256) && (dtqn_lds_bytes_backward(net) == 0 || net->gate == DTQN_GATE_GRU))) {
            // ... or the whole-sequence tile set of this variant (identity-reordered layers or the GRU gate at D = 128) exceeds
            // 160 KB of LDS
            net->tiled = 1;
            net->lp = (L + 63) / 64 * 64;
        }
    }
    const int LP = net->lp;
    // the bag branch is composed from the row-block kernels: as many bag entries as the records have rows
    if (net->bag_size > 0 && (net->bag_size > LP || !(D == 64 || D == 128 || D == 256))) return DTQN_ERR_CONFIG;
    if (net->tiled) {
        // tiled kernels: D in {64, 128, 256}, context up to 512 (north_star's bound) as long as the attention tile q | k | v | dO of ONE head
        // fits LDS (the test below): head widths up to 16 at 257 .. 512 rows, up to 32 at 256; wider heads at long contexts would need the
        // K-blocked attention loop (out of scope)
        if (!(D == 64 || D == 128 || D == 256) || LP > 512) return DTQN_ERR_CONFIG;
        const int hd = net->head_dim;                    // tl_attn_kernel / tl_attn_bwd_kernel instantiations
        if (!(hd == 4 || hd == 8 || hd == 16 || hd == 32 || hd == 64 || hd == 128)) return DTQN_ERR_CONFIG;
        if (((size_t)LP * (4 * net->head_dim + 4) + 2 * (size_t)LP) * sizeof(float) > 160 * 1024) return DTQN_ERR_CONFIG;
    }
    if (A > DTQN_MAX_ACTIONS || (!net->tiled && net->kep > 3 * D)) return DTQN_ERR_CONFIG;

    // ---- theta: trainable region first ----
    Cursor c;
    net->off_act_emb = a > 0 ? c.take(A * a) : -1;
    net->off_obs_tab = net->discrete ? c.take(V * e) : -1;
    net->off_obs_w = c.take((D - a) * net->ke);
    net->off_obs_b = c.take(D - a);
    {
        const int C = net->img_c;
        net->off_cw0 = img ? c.take(64 * C * 9) : -1;    net->off_cb0 = img ? c.take(64) : -1;
        net->off_cw1 = img ? c.take(64 * 64 * 9) : -1;   net->off_cb1 = img ? c.take(64) : -1;
        net->off_cw2 = img ? c.take(64 * 64 * 9) : -1;   net->off_cb2 = img ? c.take(64) : -1;
        net->off_cw3 = img ? c.take(128 * 64 * 9) : -1;  net->off_cb3 = img ? c.take(128) : -1;
        net->off_cw4 = img ? c.take(128 * 128 * 9) : -1; net->off_cb4 = img ? c.take(128) : -1;
    }
    const bool pos_trainable = net->pos == DTQN_POS_LEARNED;
    if (pos_trainable) net->off_pos = c.take(L * D);
    {
        Cursor lc;
        // the 13 D floats of LayerNorm affines and biases first, contiguous: the kernels that stage parameters through LDS
        // copy them as one block per layer (dtqn_wl.hpp)
        net->lo_ln1_w = lc.take(D);
        net->lo_ln1_b = lc.take(D);
        net->lo_ln2_w = lc.take(D);
        net->lo_ln2_b = lc.take(D);
        net->lo_in_b = lc.take(3 * D);
        net->lo_out_b = lc.take(D);
        net->lo_f1_b = lc.take(4 * D);
        net->lo_f2_b = lc.take(D);
        net->lo_in_w = lc.take(3 * D * D);
        net->lo_out_w = lc.take(D * D);
        net->lo_f1_w = lc.take(4 * D * D);
        net->lo_f2_w = lc.take(4 * D * D);
        net->layer_stride = lc.pos;
    }
    net->off_layer0 = c.take(net->layer_stride * NL);
    {
        Cursor gc;
        net->go_w_r = gc.take(D * D);
        net->go_u_r = gc.take(D * D);
        net->go_w_z = gc.take(D * D);
        net->go_b_z = gc.take(D);
        net->go_u_z = gc.take(D * D);
        net->go_w_g = gc.take(D * D);
        net->go_u_g = gc.take(D * D);
        if (net->gate == DTQN_GATE_GRU) {
            net->off_gate_attn = c.take(gc.pos);
            net->off_gate_mlp = c.take(gc.pos);
        } else {
            net->off_gate_attn = net->off_gate_mlp = -1;
        }
    }
    const int bag = net->bag_size;
    net->off_bag_in_w = bag > 0 ? c.take(3 * D * D) : -1;
    net->off_bag_in_b = bag > 0 ? c.take(3 * D) : -1;
    net->off_bag_out_w = bag > 0 ? c.take(D * D) : -1;
    net->off_bag_out_b = bag > 0 ? c.take(D) : -1;
    net->off_head1_w = c.take(D * (bag > 0 ? 2 * D : D));      // Linear(2D, D) on [working | persistent memory] with a bag (dtqn.py:140-144)
    net->off_head1_b = c.take(D);
    net->off_head2_w = c.take(A * D);
    net->off_head2_b = c.take(A);
    net->n_trainable = c.pos;
    if (!pos_trainable) net->off_pos = c.take(L * D);
    net->n_theta = c.pos;

    // ---- activation record ----
    const bool gru = net->gate == DTQN_GATE_GRU;
    Cursor ac;
    net->ao_ein = ac.take(img ? 0 : LP * net->kep);        // image nets: the encoder's feature maps live in its own workspace
    net->ao_x0 = ac.take(LP * D);
    {
        Cursor lc;
        net->al_u1 = lc.take(LP * D);
        net->al_qkv = lc.take(LP * 3 * D);
        net->al_lse = lc.take(H * LP);
        net->al_o = lc.take(LP * D);
        net->al_m1 = lc.take((LP / 16) * (D / 16) * 4 * 2);
        net->al_s1 = lc.take(LP * D);
        net->al_st1 = lc.take(LP * 2);
        net->al_u2 = lc.take(LP * D);
        net->al_h = lc.take(LP * 4 * D);
        net->al_mh = lc.take((LP / 16) * (4 * D / 16) * 4 * 2);
        net->al_m2 = lc.take((LP / 16) * (D / 16) * 4 * 2);
        net->al_s2 = lc.take(LP * D);
        net->al_st2 = lc.take(LP * 2);
        net->al_gate1 = gru ? lc.take(6 * LP * D) : -1;
        net->al_gate2 = gru ? lc.take(6 * LP * D) : -1;
        net->act_layer_stride = lc.pos;
    }
    net->ao_layer0 = ac.take(net->act_layer_stride * NL);
    net->ao_xf = ac.take(LP * D);
    net->ao_hh = ac.take(LP * D);
    net->bag_ld = bag > 0 ? up4(bag) : 0;
    net->ao_bag_ein = bag > 0 ? ac.take(LP * net->kep) : -1;
    net->ao_bag_e = bag > 0 ? ac.take(LP * D) : -1;
    net->ao_bag_kv = bag > 0 ? ac.take(LP * 2 * D) : -1;
    net->ao_bag_q = bag > 0 ? ac.take(LP * D) : -1;
    net->ao_bag_p = bag > 0 ? ac.take(H * LP * net->bag_ld) : -1;
    net->ao_bag_o = bag > 0 ? ac.take(LP * D) : -1;
    net->ao_xcat = bag > 0 ? ac.take(LP * 2 * D) : -1;
    net->act_stride = ac.pos;

    // ---- gradient record ----
    Cursor gc;
    net->go_dx0 = gc.take(LP * D);
    {
        Cursor lc;
        net->gl_dqkv = lc.take(LP * 3 * D);
        net->gl_da = lc.take(LP * D);
        net->gl_dhp = lc.take(LP * 4 * D);
        net->gl_df = lc.take(LP * D);
        net->gl_gate1 = gru ? lc.take(3 * LP * D) : -1;
        net->gl_gate2 = gru ? lc.take(3 * LP * D) : -1;
        net->grd_layer_stride = lc.pos;
    }
    net->go_layer0 = gc.take(net->grd_layer_stride * NL);
    net->go_dhh = gc.take(LP * D);
    net->go_dq = gc.take(LP * net->ap);
    net->go_do = net->tiled ? gc.take(LP * D) : -1;
    net->go_dcat = bag > 0 ? gc.take(LP * 2 * D) : -1;
    net->go_bag_do = bag > 0 ? gc.take(LP * D) : -1;
    net->go_bag_dq = bag > 0 ? gc.take(LP * D) : -1;
    net->go_bag_dkv = bag > 0 ? gc.take(LP * 2 * D) : -1;
    net->go_bag_de = bag > 0 ? gc.take(LP * D) : -1;
    net->grd_stride = gc.pos;

    // ---- small partials ----
    Cursor sc;
    net->so_ln = sc.take(NL * 4 * D);
    net->so_tab = net->discrete ? sc.take(V * e) : -1;
    net->so_act = a > 0 ? sc.take(A * a) : -1;
    net->sp_stride = sc.pos;
    net->sp_parts = net->tiled ? LP / 64 : 1;

    // ---- weight-gradient jobs ----
    int njobs = 0, ntiles = 0;
    auto count = [&](int N, int K) {
        njobs++;
        ntiles += ((N + 63) / 64) * ((K + 63) / 64);
    };
    if (!img) count(D - a, net->ke);       // image nets: the embedding linear's gradient comes out of dtqn_img_backward
    for (int l = 0; l < NL; ++l) {
        count(3 * D, D);
        count(D, D);
        count(4 * D, D);
        count(D, 4 * D);
    }
    if (gru) {
        // shared gate weights see the tokens of every layer: one job per (gate, matrix), looping over layers
        for (int g = 0; g < 2; ++g) for (int m = 0; m < 6; ++m) count(D, D);
        if (D > 64 && !net->tiled) return DTQN_ERR_CONFIG;   // the whole-sequence GRU backward keeps five [LP][D] tiles in LDS
    }
    if (bag > 0) {
        count(D, D);          // bag_attention in-projection, query rows
        count(2 * D, D);      // ... key | value rows
        count(D, D);          // bag_attention out-projection
    }
    count(D, bag > 0 ? 2 * D : D);
    count(A, D);
    net->n_wjobs = njobs;
    net->n_wtiles = ntiles;
    return DTQN_OK;
}

extern "C" int dtqn_net_wjobs(const DtqnNet* net, DtqnWJob* jobs) {
    if (!net || !jobs) return DTQN_ERR_ARG;
    const int a = net->action_dim, D = net->d_model, NL = net->num_layers, A = net->num_actions;
    int j = 0, tile = 0;
    auto add = [&](int x_in_act, int x_off, int ldx, int K, int dy_off, int ldy, int N, int w_off, int b_off) {
        DtqnWJob& w = jobs[j++];
        w.x_in_act = x_in_act;
        w.x_off = x_off; w.ldx = ldx; w.K = K;
        w.dy_off = dy_off; w.ldy = ldy; w.N = N;
        w.w_off = w_off; w.b_off = b_off;
        w.tile0 = tile;
        w.tiles_n = (N + 63) / 64;
        w.tiles_k = (K + 63) / 64;
        w.n_layers = 1;
        w.x_lstride = w.dy_lstride = 0;
        tile += w.tiles_n * w.tiles_k;
    };
    // embedding linear: dY = dx0[:, a:], X = e_in
    if (net->img_c <= 0) add(1, net->ao_ein, net->kep, net->ke, net->go_dx0 + a, D, D - a, net->off_obs_w, net->off_obs_b);
    if (net->bag_size > 0) {
        // the embedding linear also embeds the bag entries (dtqn.py:203-210): a second token set of the same job, summed
        // like the layers of a shared GRU gate (records of the bag entries at a fixed distance from the context's)
        jobs[j - 1].n_layers = 2;
        jobs[j - 1].x_lstride = net->ao_bag_ein - net->ao_ein;
        jobs[j - 1].dy_lstride = net->go_bag_de - net->go_dx0;
    }
    for (int l = 0; l < NL; ++l) {
        const int ab = net->ao_layer0 + l * net->act_layer_stride;
        const int gb = net->go_layer0 + l * net->grd_layer_stride;
        const int tb = net->off_layer0 + l * net->layer_stride;
        add(1, ab + net->al_u1, D, D, gb + net->gl_dqkv, 3 * D, 3 * D, tb + net->lo_in_w, tb + net->lo_in_b);
        add(1, ab + net->al_o, D, D, gb + net->gl_da, D, D, tb + net->lo_out_w, tb + net->lo_out_b);
        add(1, ab + net->al_u2, D, D, gb + net->gl_dhp, 4 * D, 4 * D, tb + net->lo_f1_w, tb + net->lo_f1_b);
        add(1, ab + net->al_h, 4 * D, 4 * D, gb + net->gl_df, D, D, tb + net->lo_f2_w, tb + net->lo_f2_b);
    }
    if (net->gate == DTQN_GATE_GRU) {
        // gate record: z, r, h~, r*x, x, y ; gradient record: dz_pre, dr_pre, dh_pre  (each [LP][D])
        const int LPD = net->lp * D;
        for (int g = 0; g < 2; ++g) {
            const int gate_w = g == 0 ? net->off_gate_attn : net->off_gate_mlp;
            const int ga = net->ao_layer0 + (g == 0 ? net->al_gate1 : net->al_gate2);
            const int gg = net->go_layer0 + (g == 0 ? net->gl_gate1 : net->gl_gate2);
            const int x_off = ga + 4 * LPD, y_off = ga + 5 * LPD, rx_off = ga + 3 * LPD;
            const int first = j;
            add(1, y_off, D, D, gg + 1 * LPD, D, D, gate_w + net->go_w_r, -1);
            add(1, x_off, D, D, gg + 1 * LPD, D, D, gate_w + net->go_u_r, -1);
            add(1, y_off, D, D, gg + 0 * LPD, D, D, gate_w + net->go_w_z, gate_w + net->go_b_z);
            add(1, x_off, D, D, gg + 0 * LPD, D, D, gate_w + net->go_u_z, -1);
            add(1, y_off, D, D, gg + 2 * LPD, D, D, gate_w + net->go_w_g, -1);
            add(1, rx_off, D, D, gg + 2 * LPD, D, D, gate_w + net->go_u_g, -1);
            for (int q = first; q < j; ++q) {
                jobs[q].n_layers = NL;
                jobs[q].x_lstride = net->act_layer_stride;
                jobs[q].dy_lstride = net->grd_layer_stride;
            }
        }
    }
    // the head reads the final stream: ao_xf, except for identity-reordered layers on the row-block tiled path, whose last
    // layer leaves it in its own s2 field
    const int xf_off = net->tiled && net->identity ? net->ao_layer0 + (NL - 1) * net->act_layer_stride + net->al_s2 : net->ao_xf;
    if (net->bag_size > 0) {
        // bag_attention (dtqn.py:134-139,211-213): q = W_q xf, k | v = W_kv E_bag, persistent memory = W_o attn + b_o
        add(1, net->ao_xcat, 2 * D, D, net->go_bag_dq, D, D, net->off_bag_in_w, net->off_bag_in_b);
        add(1, net->ao_bag_e, D, D, net->go_bag_dkv, 2 * D, 2 * D, net->off_bag_in_w + D * D, net->off_bag_in_b + D);
        add(1, net->ao_bag_o, D, D, net->go_dcat + D, 2 * D, D, net->off_bag_out_w, net->off_bag_out_b);
        add(1, net->ao_xcat, 2 * D, 2 * D, net->go_dhh, D, D, net->off_head1_w, net->off_head1_b);
    } else
    add(1, xf_off, D, D, net->go_dhh, D, D, net->off_head1_w, net->off_head1_b);
    add(1, net->ao_hh, D, D, net->go_dq, net->ap, A, net->off_head2_w, net->off_head2_b);
    return (j == net->n_wjobs && tile == net->n_wtiles) ? DTQN_OK : DTQN_ERR_CONFIG;
}

extern "C" int dtqn_net_fill_frozen(const DtqnNet* net, float* theta) {
    if (!net || !theta) return DTQN_ERR_ARG;
    const int L = net->ctx_len, D = net->d_model;
    const int DR = net->d_real > 0 ? net->d_real : D;        // the table of the caller's width in front of each padded row, zeros behind
    if (net->pos == DTQN_POS_SIN) {
        // position_encodings.py:23-35: P[t,2i] = sin(t * exp(2i * -ln(1e4)/D)), P[t,2i+1] = cos(same);
        // the reference evaluates it in fp32 torch ops: exp in fp32, product in fp32, sin/cos in fp32.
        for (int t = 0; t < L; ++t) {
            for (int i = 0; i < DR; i += 2) {
                const float div = std::exp((float)i * (float)(-std::log(10000.0) / DR));
                const float ang = (float)t * div;
                theta[net->off_pos + t * D + i] = std::sin(ang);
                if (i + 1 < DR) theta[net->off_pos + t * D + i + 1] = std::cos(ang);
            }
            for (int i = DR; i < D; ++i) theta[net->off_pos + t * D + i] = 0.f;
        }
    } else if (net->pos == DTQN_POS_NONE) {
        std::memset(theta + net->off_pos, 0, sizeof(float) * (size_t)L * D);
    }
    return DTQN_OK;
}
