// Coverage limits of the per-sequence fused kernels (one workgroup holds a whole context tile
// in LDS).  Anything beyond returns DTQN_ERR_CONFIG from dtqn_net_init.
#pragma once
#define DTQN_MAX_LP 64          /* padded context rows held in LDS (ctx_len <= 64) */
#define DTQN_MAX_D 128          /* d_model instantiations: 64, 128 (and 16/32 for tests) */
#define DTQN_MAX_HEAD_DIM 32
#define DTQN_MAX_ACTIONS 64
#define DTQN_MAX_BAG 256        /* bag entries (bag_size <= padded context <= 256, the row-block tiled path's limit) */
#define DTQN_THREADS 256        /* 4 wave64 per workgroup */
#define DTQN_WAVES 4
