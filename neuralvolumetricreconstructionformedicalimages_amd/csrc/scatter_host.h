// scatter_host.h -- host side of the binned gradient scatter: the launch plan (tile shape, buckets, record blocks) shared by the
// training path (render_fused.hip) and the drop-in operator with a workspace (hash_encode.hip: naf_hash_encode_backward_ws).
#pragma once

#include <algorithm>

#include "naf_host.h"
#include "scatter_binned.h"
#include "scatter_v2.h"

namespace naf {

static inline bool per_level_launches(const naf_render_cfg *cfg) { return (cfg->flags & NAF_CFG_PER_LEVEL_LAUNCHES) != 0u; }

constexpr uint64_t kBinMinPoints = 1u << 13;             // measured: 128 rays x 192 samples 0.62 ms (atomics) vs 0.36 ms (binned) per step
constexpr size_t kBinBudgetBytes = (size_t)40 << 30;     // record buffer per pass (HBM is 288 GB): all 16 levels of a 65 536-ray
                                                         // step fit, so the reducer gets 1024 workgroups to balance over 256 CUs

// The canonical shape (two bf16 channels) takes the compact 8-byte records and the kernels of scatter_v2.h
static inline bool scatter_v2(const naf_render_cfg *cfg) {
    return cfg->mlp_precision == NAF_BF16 && cfg->C == 2u && (cfg->flags & NAF_CFG_SCATTER_PAIR12) == 0u;
}
// bytes of a pair record (scatter_binned.h): head + two corners x C values (fp32 in parity mode, bf16 packed in pairs otherwise)
static inline size_t record_bytes(const naf_render_cfg *cfg) {
    if (scatter_v2(cfg)) return sizeof(PairFx);
    return cfg->mlp_precision == NAF_F32 ? 4u * (1u + 2u * cfg->C) : 4u * (1u + 2u * ((cfg->C + 1u) / 2u));
}
// pass-1 tile shape, the host mirror of BinShape<Rec>
static inline uint32_t bin_threads(const naf_render_cfg *cfg) { return record_bytes(cfg) <= 12 ? 512u : 256u; }
static inline uint32_t bin_points_per_thread(const naf_render_cfg *cfg) { return record_bytes(cfg) <= 20 ? 2u : 1u; }
constexpr uint32_t kBigTileLog2Nb = 7u;                  // buckets per level from which pass 1 uses its 1024-thread shape

static inline bool make_bin_plan(const naf_render_cfg *cfg, uint64_t n_points, BinPlan *plan) {
    if (cfg->scatter_mode == NAF_SCATTER_ATOMIC || cfg->log2_hashmap_size == 0 || cfg->log2_hashmap_size > 28 || n_points == 0) return false;
    if (cfg->scatter_mode == NAF_SCATTER_AUTO && n_points < kBinMinPoints) return false;
    const uint64_t maxT = 1ull << cfg->log2_hashmap_size;
    uint32_t log2_nb = 6;
    while (((maxT >> log2_nb) * cfg->C * 8u) > (128u << 10)) ++log2_nb;         // reducer rows (64-bit) must fit LDS
    // NAF_CFG_MIN_BUCKETS: more, smaller buckets on request (a bucket keeps at least 64 rows)
    const uint32_t want_nb = 6u + ((cfg->flags & NAF_CFG_MIN_BUCKETS_MASK) >> NAF_CFG_MIN_BUCKETS_SHIFT);
    const uint32_t cap_nb = cfg->log2_hashmap_size >= 12u ? cfg->log2_hashmap_size - 6u : 6u;
    log2_nb = std::max(log2_nb, std::min(want_nb, cap_nb));
    const size_t rec = record_bytes(cfg);
    // Tables of 2^20 rows per level and more need 128 .. 512 buckets (the reducer's rows must fit the LDS), which cuts a
    // 1024-point tile into runs of 8 .. 32 records -- short, ragged reads in pass 2.  With 12-byte records a 2048-point
    // tile (ONE workgroup of 1024 threads per CU instead of two of 512: the same 16 waves) still fits the LDS.
    const bool big = rec <= 12 && log2_nb >= kBigTileLog2Nb;
    const uint32_t tile = bin_threads(cfg) * bin_points_per_thread(cfg) * (big ? 2u : 1u);
    plan->tile_points = tile;
    plan->n_tiles = (uint32_t)((n_points + tile - 1) / tile);
    plan->log2_nb = log2_nb;
    // A tile's block holds four pair records per point plus the second halves of unpaired pairs: 1.6 % on average, but ALL
    // pairs of a ray that keeps an x cell with index 63 mod 64 for its whole length (seen at T = 2^22, where 1.25x was not
    // always enough).  1.375x; a tile that still fills its block spills the excess to atomics (correct, counted).  More
    // would fit the LDS next to a second workgroup, but the larger allocation measured 2-3 % slower.  A multiple of 32
    // records: blocks start on 128-byte lines.
    plan->slots = std::min<uint32_t>(65504u, ((tile * 11u / 2u) + 31u) & ~31u);
    if ((cfg->flags & NAF_CFG_TEST_TINY_BLOCKS) != 0u) plan->slots = std::max(32u, tile & ~31u);      // a quarter of a tile's records fit
    // pass 2 reads a bucket's run of a tile with W lanes: W = the power of two >= 1.25 x the mean run length, at most a wave
    const uint32_t mean_run = std::max<uint32_t>(1u, (tile * 4u) >> log2_nb);
    plan->log2_w = 3u;
    while (plan->log2_w < 6u && (1u << plan->log2_w) < mean_run + mean_run / 4u) ++plan->log2_w;
    plan->max_local_rows = (uint32_t)((((maxT + (1ull << log2_nb) - 1) >> log2_nb) + 63u) & ~63ull);
    const size_t per_level = (size_t)plan->n_tiles * plan->slots * rec;
    plan->levels_per_pass = (uint32_t)std::min<size_t>(cfg->L, std::max<size_t>(1, kBinBudgetBytes / per_level));
    if (per_level_launches(cfg)) plan->levels_per_pass = 1;
    return true;
}

// Workgroups a reducer launch over `nl` levels splits each bucket's tiles between: 1 when buckets x levels give every CU a
// workgroup (one owner per row, sums formed in a fixed order: the table gradient is bit-reproducible -- this covers the four-level
// buckets of a data-parallel step, 64 x 4 = 256), more (with per-row fp32 atomics at the end, whose order is not fixed) only when a
// pass holds fewer than four levels' worth of buckets.
static inline uint32_t reducer_split(uint32_t NB, uint32_t nl) { return NB * nl >= 256u ? 1u : std::max(1u, std::min(16u, 1024u / (NB * nl))); }

}  // namespace naf
