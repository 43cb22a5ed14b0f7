// Fragment-major copies of the weight matrices the row-block GEMM kernels read (round 6).
//
// tools/microbench/mfma_probe.hip: the row-block GEMM loop is bound by the way every wave pulls its weight fragments out of L2 -- in
// the parameter layout of the reference (W[N][K] row-major, utils/agent_utils.py / nn.Linear) lane (i, kq) of a fragment load reads 16
// bytes of row i, so one load instruction touches sixteen half-used 128-byte lines (forward X W^T), or -- backward, dY W -- is a dword
// load that touches four 64-byte row pieces.  The same loop with the fragment stored in the order the lanes consume it (every load
// instruction reads 1 KB contiguous) runs 0.65 -> 0.75 of the f32 matrix peak at 32 rows per workgroup, 0.77 -> 0.85 at 64.
// So the TD update keeps two re-ordered copies of the layer matrices (in_proj, out_proj, ffn.0, ffn.2) and of the first head matrix:
//   F (forward, X W^T, W[N][K]):   F[ntile][kc][q][lane] = float4 W[16 ntile + i][128 kc + 16 q + 4 kq .. + 3]        (q < 8)
//   B (backward, dY W, W[N][K], contraction over N):   B[ktile][nc][q4][lane] = { W[128 nc + 32 kq + 4 q4 + e][16 ktile + i] }, e < 4
// with lane = 16 kq + i: exactly the registers frag16_fetch / frag_dyw_fetch fill.  dtqn_td_wpack rewrites them from theta_pol /
// theta_tgt at the start of every row-block TD forward (a few MB: 3 - 10 us of a 1.1 - 2.1 ms update), so they are never stale --
// whoever wrote theta in between (Adam, a hard target sync, load_state_dict).  Networks the plan does not cover (d_model not a
// multiple of 128, bag networks) and callers that leave DtqnTd.wpack_* NULL run on the parameter layout as before.
#pragma once
#include <cstdlib>
#include "dtqn_hip.h"

namespace dtqn {

struct WPackMat {
    int w_off, N, K;                   // W[N][K] at theta + w_off
    int f_off, b_off;                  // float offsets of its F / B copy in the packed buffer
};
constexpr int kMaxPackMats = 4 * 8 + 1;
struct WPackPlan {
    int n;                             // matrices (0: the network is not covered)
    int f_total, total;                // floats of all F copies; of everything in the buffer (F + B copies + the embedding product table)
    int e_off, e_floats;               // embedding product table (discrete observations): offset in the buffer, floats (0: none)
    WPackMat m[kMaxPackMats];
};
// Embedding product table (round 6).  A discrete observation row is embedded as concat_j(T[tok_j]) W_e^T + b (representations.py:25-52): O table
// rows of e floats each against the [D - a][O e] matrix.  Both are parameters, the tokens come from a vocabulary of V: the products
//   P[j][v][d] = sum_{c < e} T[v][c] * W_e[d][j e + c]
// are O V (D - a) floats (46 KB at BASELINE config 3), rewritten with the fragment-major weight copies at the start of every TD forward, and the
// embedding of a row becomes b + sum_j P[j][tok_j] -- O gathered rows instead of a [rows][O e] x [O e][D] product whose operands are
// gathered element by element (tl_embed_kernel: 38 / 71 us per TD forward at configs 4 / 3, an HBM-bound kernel's job).  The sum runs
// over j in the order of the columns; inside a slot over c: the same terms as the reference's product in a different association.
static inline int wpack_etab_floats(const DtqnNet& net) {
    if (!net.discrete || net.img_c > 0 || net.action_dim % 4 != 0 || net.d_model % 4 != 0) return 0;
    const long long fl = (long long)net.obs_dim * net.vocab * (net.d_model - net.action_dim);
    return fl > 0 && fl <= (1 << 20) ? (int)fl : 0;     // (a 4 MB table would no longer stay cache-resident next to the records)
}
static inline WPackPlan wpack_plan(const DtqnNet& net) {
    WPackPlan p = {};
    const char* e = getenv("DTQN_WPACK");
    if (!net.tiled || net.d_model % 128 != 0 || net.bag_size > 0 || net.num_layers > 8 || (e != nullptr && atoi(e) == 0)) return p;
    const int D = net.d_model;
    auto add = [&](int w_off, int N, int K) {
        WPackMat& m = p.m[p.n++];
        m.w_off = w_off; m.N = N; m.K = K; m.f_off = p.f_total; m.b_off = 0;
        p.f_total += N * K;
    };
    for (int l = 0; l < net.num_layers; ++l) {
        const int tb = net.off_layer0 + l * net.layer_stride;
        add(tb + net.lo_in_w, 3 * D, D);
        add(tb + net.lo_out_w, D, D);
        add(tb + net.lo_f1_w, 4 * D, D);
        add(tb + net.lo_f2_w, D, 4 * D);
    }
    add(net.off_head1_w, D, D);
    for (int j = 0; j < p.n; ++j) p.m[j].b_off = p.f_total + p.m[j].f_off;
    p.total = 2 * p.f_total;
    const char* ee = getenv("DTQN_EMBED_TABLE");
    p.e_floats = (ee != nullptr && atoi(ee) == 0) ? 0 : wpack_etab_floats(net);
    p.e_off = p.total;
    p.total += p.e_floats;
    return p;
}
// F / B copy of the matrix at theta + w_off inside the packed buffer `pk`, or nullptr (no buffer, or not a packed matrix)
static inline const float* wpack_f(const WPackPlan& p, const float* pk, int w_off) {
    if (pk == nullptr) return nullptr;
    for (int j = 0; j < p.n; ++j)
        if (p.m[j].w_off == w_off) return pk + p.m[j].f_off;
    return nullptr;
}
static inline const float* wpack_etab(const WPackPlan& p, const float* pk) {
    return pk != nullptr && p.n > 0 && p.e_floats > 0 ? pk + p.e_off : nullptr;
}
static inline const float* wpack_b(const WPackPlan& p, const float* pk, int w_off) {
    if (pk == nullptr) return nullptr;
    for (int j = 0; j < p.n; ++j)
        if (p.m[j].w_off == w_off) return pk + p.m[j].b_off;
    return nullptr;
}

}  // namespace dtqn
