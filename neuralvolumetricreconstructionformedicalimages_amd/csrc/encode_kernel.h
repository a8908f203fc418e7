// encode_kernel.h -- the hash-grid encoder kernel of the training / evaluation path (hashencoder.cu:77-198) and its launch rules.
// Shared by render_fused.hip (points generated from rays, grids, point lists) and hash_encode.hip (the drop-in operator
// naf_hash_encode_forward: caller-supplied coordinates in [0, 1], hashencoder.h:13), so that a reference maintainer who binds
// the operator gets the window gathers, the multi-point loop and the XCD groups of the fused path.
#pragma once

#include <algorithm>

#include "hash_kernels.h"
#include "naf_host.h"

namespace naf {

// Feature tensors are written in the MLP's operand precision: TT = table storage, P::feat_t = feature storage.
// The encoder kernel is templated on one storage type for table and output, so when they differ the
// features are produced by a converting instantiation below.
__host__ __device__ constexpr uint32_t encode_points_per_thread(uint32_t C) { return C <= 2 ? 4u : C == 4 ? 2u : 1u; }

template <typename TT, typename FT, uint32_t C, typename Src, uint32_t kWindow>        // kWindow: 0 = two gathers per pair, else points per thread
__global__ void __launch_bounds__(256)
encode_kernel(Src src, const typename TT::store_t *__restrict__ table, const int32_t *__restrict__ offsets,
              typename FT::store_t *__restrict__ feat, uint32_t B, uint32_t H, uint32_t level_base, uint32_t order, uint32_t n_levels, uint32_t total_levels,
              uint32_t tiles, uint32_t chunk, uint32_t blc) {
    // order 0 -- level-major (large batches): blocks are dispatched x-fastest, so the whole chip works on ONE level at a time and
    //   that level's slice of the table stays in the L2s (98.6 % hits).
    // order 1 -- NAF_CFG_LEVELS_INTERLEAVED (diagnostic): the level in x, every XCD walks levels k and k + 8 at once.
    const uint32_t table_rows = (uint32_t)offsets[total_levels];
    // order 4 + log2 G -- XCD GROUPS (small and medium batches): the eight XCDs form G groups of 8 / G; group g takes the levels
    //   g, g + G, ... one after the other and spreads each level's `tiles` workgroup-sized pieces over its XCDs, dispatched in
    //   order.  A level is then pulled through 8 / G L2s instead of all eight -- at the reference's batch size the table fills are
    //   most of what the encoder fetches (0.24 GB per launch by PMC: 8 x 28.5 MB) -- while even / odd levels (G = 2) or levels mod 4
    //   (G = 4) balance within 0.5 % / 4 % (profiles/round3_bench_per_level_16384rays.json); G = 8 (one level per XCD at a time)
    //   is 14 % off balance however the levels are paired.  Workgroup b sits on XCD b mod 8 (round-robin dispatch: a property
    //   used for speed only, any placement gives the same bits).
    uint32_t g_level = 0u, g_block = 0u;
    bool g_idle = false;
    if (order >= 4u) {
        const uint32_t log2g = order - 4u, G = 1u << log2g, per = 8u >> log2g, xcd = blockIdx.x & 7u;
        const uint32_t u = (blockIdx.x >> 3) * per + (xcd % per);           // piece index inside the group
        g_level = xcd / per + G * (u / tiles);
        g_block = u % tiles;
        g_idle = g_level >= n_levels;                                       // padding workgroups of a grid rounded up to 8
    }
    if (g_idle) return;
    {
    const uint32_t level = level_base + (order >= 4u ? g_level : order == 1u ? blockIdx.x : blockIdx.y);
    const uint32_t block_x = order >= 4u ? g_block : order == 1u ? blockIdx.y : blockIdx.x;
    const uint32_t grid_x = order >= 4u ? tiles : order == 1u ? gridDim.y : gridDim.x;
    const LevelMeta m = make_level_meta<3>(offsets, level, H);
    const typename TT::store_t *__restrict__ grid = table + (size_t)m.offset * C;
    // Where the C features of point b go: [level][B points] -- or, for a level-parallel rank (naf_levels_encode, chunk = points of one
    // rank), [rank = b / chunk][level - level_base][chunk points]: one contiguous block per destination of the all-to-all.
    const uint32_t chunk_magic = chunk != 0u ? (uint32_t)(0x100000000ull / chunk) + 1u : 0u;
    auto slot = [&](uint32_t b) -> size_t {
        if (blc != 0u) return (size_t)b * total_levels + level;
        if (chunk == 0u) return (size_t)level * B + b;
        uint32_t r = __umulhi(b, chunk_magic);               // b / chunk, at most one too large (b < 2^31)
        r -= r * chunk > b ? 1u : 0u;
        return ((size_t)r * n_levels + (level - level_base)) * chunk + (b - r * chunk);
    };
    // (Round 4 tried to stream a level's slice of the table into the L2s of its XCD group in front of the gathers -- every workgroup a
    // share, coalesced 16-byte loads -- on the theory that small batches wait for sector-by-sector fills at the fabric's request rate.
    // Same-box A/B, encode_kernel with / without: 256 rays 0.030 / 0.023 ms, 1 024: 0.066 / 0.060, 2 048: 0.114 / 0.108, 4 096: 0.199 /
    // 0.199 -- the extra loads cost what they were meant to save and more; profiles/round4_ab_encoder_l2_warmup.jsonl.  Not kept.)
    dispatch_mode<Src::kInRange>(m.mode, [&](auto mode_tag) {
    constexpr uint32_t MODE = decltype(mode_tag)::value;
    const uint32_t stride = grid_x * blockDim.x;
    if constexpr (kWindow != 0u) {
        // x-neighbour corners through one 16-byte window each (PairWindow, naf_device.h): 4.7 instead of 8 L1 accesses per point
        using PW = PairWindow<TT, C>;
        constexpr uint32_t kPts = kWindow;                       // 2: 8 windows + their far rows in flight per lane
        const uint32_t safe_last = table_rows - PW::kWin - m.offset;
        for (uint32_t b0 = block_x * blockDim.x + threadIdx.x; b0 < B; b0 += kPts * stride) {
            float w[kPts][8];
            PW pw[kPts][4];
#pragma unroll
            for (uint32_t k = 0; k < kPts; ++k) {
                const uint32_t b = min(b0 + k * stride, B - 1u);
                float x[3], frac[3];
                uint32_t pg[3];
                src.get(b, x);
                locate<3>(x, m.scale, frac, pg);
                uint32_t row[8];
                cell_corners<MODE, 3>(m, frac, pg, w[k], row);
#pragma unroll
                for (uint32_t j = 0; j < 4; ++j) pw[k][j].issue(grid, row[2 * j], row[2 * j + 1], safe_last);
            }
#pragma unroll
            for (uint32_t k = 0; k < kPts; ++k) {
                const uint32_t b = b0 + k * stride;
                float a[C];
#pragma unroll
                for (uint32_t ch = 0; ch < C; ++ch) a[ch] = 0.0f;
#pragma unroll
                for (uint32_t j = 0; j < 4; ++j) {
                    float va[C], vb[C];
                    pw[k][j].finish(va, vb);
#pragma unroll
                    for (uint32_t ch = 0; ch < C; ++ch) a[ch] = __fmaf_rn(w[k][2 * j], va[ch], a[ch]);
#pragma unroll
                    for (uint32_t ch = 0; ch < C; ++ch) a[ch] = __fmaf_rn(w[k][2 * j + 1], vb[ch], a[ch]);
                }
                if (b < B) store_vec<FT, C>(feat + slot(b) * C, a);
            }
        }
    } else {
    // several points per thread and iteration: 32 independent gathers in flight per lane (measured on the chest step,
    // C = 2: 1 / 2 / 4 / 8 points -> 3.28 / 2.99 / 2.81 / 2.90 ms)
    constexpr uint32_t kPts = encode_points_per_thread(C);
    for (uint32_t b0 = block_x * blockDim.x + threadIdx.x; b0 < B; b0 += kPts * stride) {
        float w[kPts][8], v[kPts][8][C];
#pragma unroll
        for (uint32_t k = 0; k < kPts; ++k) {
            const uint32_t b = min(b0 + k * stride, B - 1u);
            float x[3], frac[3];
            uint32_t pg[3];
            src.get(b, x);
            locate<3>(x, m.scale, frac, pg);
            uint32_t row[8];
            cell_corners<MODE, 3>(m, frac, pg, w[k], row);
#pragma unroll
            for (uint32_t c = 0; c < 8; ++c) load_vec<TT, C>(grid + (size_t)row[c] * C, v[k][c]);
        }
#pragma unroll
        for (uint32_t k = 0; k < kPts; ++k) {
            const uint32_t b = b0 + k * stride;
            float a[C];
#pragma unroll
            for (uint32_t ch = 0; ch < C; ++ch) a[ch] = 0.0f;
#pragma unroll
            for (uint32_t c = 0; c < 8; ++c)
#pragma unroll
                for (uint32_t ch = 0; ch < C; ++ch) a[ch] = __fmaf_rn(w[k][c], v[k][c][ch], a[ch]);
            if (b < B) store_vec<FT, C>(feat + slot(b) * C, a);
        }
    }
    }
    });
    }
}

template <typename TT, typename FT, uint32_t C, typename Src>
static int launch_encode(const Src &src, const void *table, const int32_t *offsets, void *feat, uint32_t B, uint32_t H, uint32_t L, uint32_t flags,
                         hipStream_t s, uint32_t lv_begin = 0u, uint32_t lv_end = ~0u, uint32_t chunk = 0u, bool blc = false) {
    // flags: the NAF_CFG_ENCODE_* / _PER_LEVEL_LAUNCHES / _LEVELS_INTERLEAVED bits of naf_render_cfg.flags (0: the measured defaults).
    // blc: features go to [B, L, C] (the drop-in operator's second layout) instead of [L, B, C].
    // [lv_begin, lv_end): the levels to encode (all by default; a level-parallel rank encodes the levels it owns, naf_levels_encode).
    // `feat` is indexed by the ABSOLUTE level: level l of point b sits at (l * B + b) * C -- unless `chunk` != 0 (a level range only):
    // then the output is [b / chunk][l - lv_begin][b % chunk][C], see encode_kernel.
    lv_end = std::min(lv_end, L);
    const uint32_t nl = lv_end - lv_begin;
    constexpr bool kCanWindow = PairWindow<TT, C>::kUsable;
    // x-neighbour corners through ONE 16-byte window per pair (fewer L1 accesses: what small batches are bound by) or through two
    // gathers with four points per lane in flight (more misses outstanding: what large batches are bound by).  Measured
    // (encode_kernel, window / two gathers, ms): 1 024 rays 0.067 / 0.088, 2 048: 0.115 / 0.117, 4 096: 0.216 / 0.211, 16 384:
    // 0.777 / 0.753, 65 536: 3.00 / 2.89; T = 2^22 fp16 (foot) 6.20 / 5.85; fp32 tables (a window covers only the mask-1 pairs, but
    // the L1 is the tighter resource there) 0.908 / 1.008 at 16 384 rays.  NAF_CFG_ENCODE_TWO_GATHERS / _WINDOWS force one form.
    bool window = kCanWindow && (B < 600000u || sizeof(typename TT::store_t) == 4u);
    if ((flags & NAF_CFG_ENCODE_WINDOWS) != 0u) window = kCanWindow;
    if ((flags & NAF_CFG_ENCODE_TWO_GATHERS) != 0u) window = false;
    auto kern = encode_kernel<TT, FT, C, Src, 0u>;
    if constexpr (kCanWindow) { if (window) kern = encode_kernel<TT, FT, C, Src, 2u>; }
    const uint32_t kPts = window ? 2u : encode_points_per_thread(C);
    if ((flags & NAF_CFG_PER_LEVEL_LAUNCHES) != 0u && chunk == 0u) {
        static const char *const names[32] = NAF_LEVEL_NAMES("encode_kernel_L");
        for (uint32_t l = lv_begin; l < lv_end; ++l) {
            ProfScope prof_(level_name(names, l), s);
            hipLaunchKernelGGL(kern, dim3(hash_grid_x((B + kPts - 1u) / kPts), 1), dim3(256), 0, s, src,
                               (const typename TT::store_t *)table, offsets, (typename FT::store_t *)feat, B, H, l, 0u, 1u, L, 0u, 0u, blc ? 1u : 0u);
        }
        return check_launch("encode_kernel");
    }
    const uint32_t gx = hash_grid_x((B + kPts - 1u) / kPts);
    const bool interleaved = (flags & NAF_CFG_LEVELS_INTERLEAVED) != 0u && gx <= 65535u;
    // XCD groups (order 4 + log2 G; see encode_kernel): a level is pulled through 8 / G L2s instead of eight.  Measured on the chest
    // step (encode_kernel, ms; level-major / G = 2 / 4 / 8): 256 rays 0.036 / 0.029 / 0.023 / 0.023, 512: 0.047 / 0.038 / 0.034 / 0.037,
    // 1 024: 0.068 / 0.061 / 0.061 / 0.067, 2 048: 0.111 / 0.108 / 0.110 / 0.121, 4 096: 0.202 / 0.202 / 0.207 / 0.230, 16 384:
    // 0.735 / 0.753 / 0.791 / 0.884 -- four groups below 160 000 points, two below 500 000, level-major above.
    // NAF_CFG_ENCODE_GROUPS_2 / _4 / both (= 8) force G, NAF_CFG_ENCODE_LEVEL_MAJOR forces none.
    uint32_t log2g = B < 160000u ? 2u : B < 500000u ? 1u : 0u;
    if ((flags & (NAF_CFG_ENCODE_GROUPS_2 | NAF_CFG_ENCODE_GROUPS_4)) != 0u)
        log2g = ((flags & NAF_CFG_ENCODE_GROUPS_2) != 0u ? 1u : 0u) + ((flags & NAF_CFG_ENCODE_GROUPS_4) != 0u ? 2u : 0u);
    if ((flags & NAF_CFG_ENCODE_LEVEL_MAJOR) != 0u) log2g = 0u;
    while (log2g != 0u && nl % (1u << log2g) != 0u) --log2g;                 // a level range: as many groups as divide it
    const bool grouped = !interleaved && log2g != 0u && (nl >= 8u || nl != L);
    const uint32_t order = interleaved ? 1u : grouped ? 4u + log2g : 0u;
    const dim3 grid = interleaved ? dim3(nl, gx) : grouped ? dim3((nl * gx + 7u) / 8u * 8u) : dim3(gx, nl);
    { ProfScope prof_("encode_kernel", s); hipLaunchKernelGGL(kern, grid, dim3(256), 0, s, src,
                       (const typename TT::store_t *)table, offsets, (typename FT::store_t *)feat, B, H, lv_begin, order, nl, L, gx, chunk, blc ? 1u : 0u); }
    return check_launch("encode_kernel");
}

}  // namespace naf
