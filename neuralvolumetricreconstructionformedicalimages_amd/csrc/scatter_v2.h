// scatter_v2.h -- the binned hash-table gradient scatter for the canonical NAF shape (two bf16 channels), on an instruction diet
// (round 4).  Same two passes as scatter_binned.h (exact per-tile multisplit -> per-bucket reduction in 64-bit fixed point,
// hashencoder.cu:201-272 is what both replace) and the same workspace layout (BinPlan: blocks [level][tile][slots], run words
// [level][bucket][tile]); what changed is what a record is and how little arithmetic each one costs.  Counter evidence that asked for
// it (profiles/round3_sq_counters.md, round3_wave_state.md): pass 1 spent 283 vector instructions per point and level and two LDS
// atomics per record, pass 2 ran its vector pipe 54 % busy converting through fp64.
//
// RECORD (8 bytes instead of 12).  The x-neighbour corners of a cell take w_x = 1 - f_x and f_x of the same product g * w_y * w_z, so
// a record carries that product ONCE and the fraction next to the row:
//     head = local row of the first corner (13 bits) | e << 13 (4 bits) | f_x << 17 (15-bit fixed point)
//     pay  = 2 x bf16:  g[c] * (w_y * w_z)
// and the reducer forms pay * (1 - f_x) for the row `local` and pay * f_x for the row `local ^ (2^e - 1)`.  (bf16 x 15 bits = 23 bits:
// both products and the difference are EXACT in fp32, so the only roundings are the bf16 payload and the fixed-point conversion.)
// e = 0 marks a SINGLE record (f_x = 0: the whole payload goes to `local`); it is used where the two corners have different owners
// and on the coarse levels whose equal-cell runs are merged inside the wave (the merged sums of the two corners are not one
// product times one fraction any more).  512 instead of 768 bytes per point and pass through HBM, ds_write_b64 instead of
// ds_write_b96 into the staging block.
//
// BUCKET = the TOP bits of the row: bucket = row >> sh, local = row & (2^sh - 1), sh = ceil(log2 T_l) - log2 NB per level.  One shift
// and one mask instead of five instructions; x-neighbour rows differ in their LOW bits only (r ^ r' = 2^e - 1: the prime of dimension
// 0 is 1 on hashed levels, r' = r + 1 on dense ones), so a pair leaves its bucket with probability 2^-sh (2^-13 at T = 2^19) instead of
// 2^-6 -- the unpaired path is all but dead on the hashed levels -- and a bucket owns CONTIGUOUS table rows, which makes the Adam
// tail of pass 2 a plain stream.  (The round-2 choice of the bits above the low six spread a ray's consecutive cells over the
// buckets of dense levels; the levels where that matters are the merged ones, whose runs collapse to one record anyway.)
//
// ONE LDS atomic per record in pass 1: the returning ds_add_rtn_u32 that counts a bucket also hands the record its rank inside the
// bucket; after the scan the slot is start[bucket] + rank (a plain LDS read).  Pass 2 converts to fixed point on the integer pipe:
// with kFixHead2 = 26 a record (up to 16 merged contributions) stays below 2^30, so v * 2^shift -> v_cvt_i32_f32 -> sign extension
// replaces the fp64 route (26 significant bits below the step's largest gradient: two more than the fp32 mantissa the atomic path sums with).
#pragma once

#include "draw_device.h"
#include "hash_kernels.h"
#include "scatter_binned.h"

namespace naf {

struct __attribute__((aligned(8))) PairFx {
    uint32_t head, pay;
};
constexpr uint32_t kFxLocalBits = 13u, kFxBits = 15u;
constexpr uint32_t kFxEShift = kFxLocalBits, kFxShift = kFxLocalBits + 4u;
constexpr int kFixHead2 = 26;

__device__ __forceinline__ int fixed_shift2(uint32_t gmax_bits) {
    const int e = (int)((gmax_bits >> 23) & 0xffu);          // biased exponent of gmax; 0 -> all gradients are zero
    if (e == 0 || e == 255) return 0;
    const int s = kFixHead2 - (e - 127) - 1;
    return s > 120 ? 120 : s;                                 // 2^shift must be an fp32 (gradients below 2^-94: their low bits are not missed)
}

__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    bf16x2 pk;
    pk[0] = (__bf16)lo;
    pk[1] = (__bf16)hi;
    return __builtin_bit_cast(uint32_t, pk);
}

// A record that found its tile's block full (see scatter_binned.h: never at T = 2^19, rare single tiles at T = 2^22, every tile under
// NAF_CFG_TEST_TINY_BLOCKS): straight to the gradient table with float atomics.  Out of line: the hot path keeps its registers.
__device__ __attribute__((noinline)) void spill_record(float *__restrict__ gg, uint32_t row_a, uint32_t head, uint32_t pay) {
    const uint32_t e = (head >> kFxEShift) & 15u;
    const uint32_t row_b = row_a ^ ((1u << e) - 1u);
    const float fx = (float)(head >> kFxShift) * (1.0f / 32768.0f);
    const float p0 = __uint_as_float(pay << 16), p1 = __uint_as_float(pay & 0xffff0000u);
    const float b0 = p0 * fx, b1 = p1 * fx;
    atomicAdd(gg + (size_t)row_a * 2u, p0 - b0);
    atomicAdd(gg + (size_t)row_a * 2u + 1u, p1 - b1);
    if (e != 0u) {
        atomicAdd(gg + (size_t)row_b * 2u, b0);
        atomicAdd(gg + (size_t)row_b * 2u + 1u, b1);
    }
}

// entries of pass 1's side list (second-corner singles of a tile and level; what does not fit goes to the table with atomics): a
// merged run of two or more points emits four of them, i.e. at most two per point = half of the tile's pair count, 3/8 of the
// block's slots (which are 11/8 of the pair count)
__host__ __device__ constexpr uint32_t side_list_capacity(uint32_t slots) { return slots / 8u * 3u; }

// wave-wide inclusive prefix sum on DPP (row shifts inside the 16-lane rows, then the two row broadcasts of gfx9)
__device__ __forceinline__ uint32_t wave_inclusive_sum(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);      // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);      // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);      // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);      // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);      // row_bcast:15 -> rows 1, 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);      // row_bcast:31 -> rows 2, 3
    return v;
}

// wave-wide maximum the same way; the result is valid in lane 63
__device__ __forceinline__ uint32_t wave_max_to_lane63(uint32_t v) {
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false));
    return v;
}

// ---- pass 1 ---------------------------------------------------------------------------------------------------
// LDS: hist[2][NB] | start[NB] | misc (4 dwords) | staging[slots + 1] records | side list [3/8 slots] x 12 bytes.
// One workgroup = one tile of NT x 2 points x LV levels.  Per level: (A) every thread builds the records of its points in registers;
// ds_add_rtn_u32 on the bucket's counter returns the record's rank; barrier; (B) the counters become exclusive offsets -- with 64
// buckets EVERY wave does that for itself (one LDS read + a DPP prefix sum, a bucket per lane; ds_bpermute then hands a record its
// bucket's offset), so there is no serial section and no second barrier; with more buckets wave 0 scans into start[] behind one more
// barrier; (C) every record goes to staging[offset + rank]; barrier; (D) the dense, bucket-sorted block leaves for HBM as whole
// 128-byte lines.  Counters and the side list's length are double-buffered by level parity: the copy the NEXT level will use is
// cleared between this level's two barriers.
// What the round-4 counters said about the first form of this kernel (profiles/round4_scatter_counters.md): its waves were parked
// 57 % of the time and issued as many scalar as vector instructions -- exec-mask juggling around per-record conditions.  So the
// common level (not merged, every lane emits: lanes past the end of the batch carry a zero gradient and their records add nothing)
// has a body of its own without a single per-lane branch; a record that finds its block full is noticed by ONE vote per level.
// kFromBlocks (level-parallel steps, naf_levels_scatter): the feature gradients still lie as the all-to-all delivered them -- one block per
// source rank, [rank][owned level][that rank's points][2] -- and pass 1 reads them in place instead of behind a pass that re-orders them
// (levels_gather_kernel: 0.014 of a rank's 0.233 ms per step at 8 ranks).  The step's maximum |gradient|, which the fixed-point reducer
// scales by and which a single-GPU step gets from its MLP backward, is taken on the way: every gradient of the owned levels passes
// through exactly one lane here.
struct GradBlocks {
    uint32_t rank_points;          // points of one rank (B / n_ranks), >= 2
    uint32_t block_points;         // distance between two ranks' blocks, in points (4 bytes each)
    uint32_t level0;               // first owned level: (level l, point b of rank r) sits at r * block_points + (l - level0) * rank_points + b
    uint32_t magic, shift;         // b / rank_points = mulhi(b, magic) >> shift for b < 2^31 (make_grad_blocks)
    uint32_t *gmax_bits;           // bit pattern of max |gradient| (zeroed by the host before the launch)
};
// magic = ceil(2^(31 + l) / d), l = ceil(log2 d): exact quotients for every numerator below 2^31 (Granlund & Montgomery with one bit
// to spare, so the multiplier fits 32 bits); d >= 2.
static inline GradBlocks make_grad_blocks(uint32_t rank_points, uint32_t block_points, uint32_t level0, uint32_t *gmax_bits) {
    uint32_t l = 0;
    while ((1ull << l) < rank_points) ++l;
    const uint64_t two = 1ull << (31u + l);
    return GradBlocks{rank_points, block_points, level0, (uint32_t)((two + rank_points - 1u) / rank_points), l - 1u, gmax_bits};
}

template <uint32_t NT, uint32_t LV, uint32_t kLog2NB, bool kFromBlocks = false>      // kLog2NB: 6 = the 64-bucket plan, 0 = as the plan says (T >= 2^20)
__global__ void __launch_bounds__(NT, 4)                       // 512 threads: two workgroups per CU; 1024: one -- 16 waves either way
scatter_bin2_kernel(SrcRays src, const uint16_t *__restrict__ grad, const int32_t *__restrict__ offsets, float *__restrict__ grad_table,
                    PairFx *__restrict__ blocks, uint32_t *__restrict__ runs, uint32_t *__restrict__ overflow, uint32_t B, uint32_t H,
                    uint32_t level_base, uint32_t n_levels, BinPlan plan, SlabReduce slab_job, DrawJob draw_job, GradBlocks gb = GradBlocks{}) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if (blockIdx.x >= plan.n_tiles) {
        // spare workgroups behind the tiles: the MLP backward's slab reduction (scatter_binned.h), then -- naf_render_train_adam_draw --
        // the pixel draw of the NEXT step (draw_device.h): nothing in this step depends on either
        if (blockIdx.y == 0u) {
            const uint32_t spare = blockIdx.x - plan.n_tiles, n_slab = slab_job.slabs != nullptr ? kSlabReduceBlocks : 0u;
            if (spare < n_slab) slab_reduce_block<NT / kReduceParams>(reinterpret_cast<float (*)[kReduceParams]>(smem), slab_job, spare, threadIdx.x);
            else {
                const uint32_t t = (spare - n_slab) * NT + threadIdx.x;
                if (t < draw_job.count) draw_one(draw_job, t);
            }
        }
        return;
    }
    constexpr uint32_t PTS = 2u;
    constexpr bool kWaveScan = kLog2NB == 6u;
    const uint32_t log2_nb = kLog2NB != 0u ? kLog2NB : plan.log2_nb, NB = 1u << log2_nb, nb_mask = NB - 1u, SLOTS = plan.slots;
    uint32_t *hist = reinterpret_cast<uint32_t *>(smem);           // [2][NB]
    uint32_t *start = hist + 2u * NB;                              // [NB]   (only without kWaveScan)
    uint32_t *misc = start + NB;                                   // [0] records of the level, [1], [2] length of the side list (by level parity)
    uint2 *staging = reinterpret_cast<uint2 *>(misc + 4);          // [SLOTS + 1]: slot SLOTS swallows the records of a full block
    // Second-corner singles do not travel in registers (eight more record slots per thread for something that happens to 2^-13 of
    // the pairs on the hashed levels would cost the kernel its occupancy): phase A appends them to a side list in LDS
    // { bucket << 16 | rank, head, pay } and phase C places them together with everything else.
    const uint32_t side_cap = side_list_capacity(SLOTS);
    uint32_t *side = reinterpret_cast<uint32_t *>(staging + SLOTS + 1u);      // [side_cap][3]
    const uint32_t tile = blockIdx.x, lane = threadIdx.x & 63u, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (uint32_t i = threadIdx.x; i < 2u * NB; i += NT) hist[i] = 0u;
    if (threadIdx.x < 4u) misc[threadIdx.x] = 0u;
    // kFromBlocks: misc[3] = the largest |gradient| the workgroup knows of, at first what earlier workgroups of the step have published
    // (an agent-scope load: a plain one is served by this XCD's L2, which keeps the zero it saw first whatever the memory-side atomics
    // of other workgroups have done since)
    uint32_t published = 0u;                                       // (lane 3 of wave 0 keeps its copy: see `seed` below)
    if constexpr (kFromBlocks) {
        if (threadIdx.x == 3u) { published = __hip_atomic_load(gb.gmax_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); misc[3] = published; }
    }

    float x[PTS][3];
    uint32_t bp[PTS];
    bool valid[PTS];
#pragma unroll
    for (uint32_t q = 0; q < PTS; ++q) {
        const uint32_t b_raw = (tile * PTS + q) * NT + threadIdx.x;
        valid[q] = b_raw < B;
        bp[q] = valid[q] ? b_raw : B - 1u;
        src.get(bp[q], x[q]);
    }
    const float spacing = src.sample_spacing();
    // where (level, point) sits: [level][B][2], or inside its rank's block (bp[] then holds the point's offset for level0)
    if constexpr (kFromBlocks) {
#pragma unroll
        for (uint32_t q = 0; q < PTS; ++q) {
            const uint32_t r = __umulhi(bp[q], gb.magic) >> gb.shift;          // bp / rank_points
            bp[q] += r * (gb.block_points - gb.rank_points);
        }
    }
    auto grad_at = [&](uint32_t level, uint32_t q) __attribute__((always_inline)) -> uint32_t {
        if constexpr (kFromBlocks) return *reinterpret_cast<const uint32_t *>(grad + ((size_t)(level - gb.level0) * gb.rank_points + bp[q]) * 2u);
        else return *reinterpret_cast<const uint32_t *>(grad + ((size_t)level * B + bp[q]) * 2u);
    };
    uint32_t graw[PTS];                                            // this level's feature gradients (2 x bf16), requested one level ahead
    uint32_t wmax = 0u;                                            // kFromBlocks: largest |gradient| this wave has seen (fp32 bit pattern, wave-uniform)
    if (blockIdx.y * LV < n_levels) {
#pragma unroll
        for (uint32_t q = 0; q < PTS; ++q) graw[q] = grad_at(level_base + blockIdx.y * LV, q);
    }
    __syncthreads();
    // wmax may start from misc[3] even if a faster wave has raised the word already (whatever is in the word is accounted for);
    // `seed`, against which thread 0 decides whether the workgroup has anything to publish, may not -- it comes from the register of
    // the lane that loaded the published value (wave 0; the other waves never use theirs)
    if constexpr (kFromBlocks) wmax = (uint32_t)__builtin_amdgcn_readfirstlane((int)misc[3]);
    const uint32_t seed = (uint32_t)__builtin_amdgcn_readlane((int)published, 3);
    uint32_t n_overflow = 0, n_overflow_level = 0;

    for (uint32_t it = 0; it < LV; ++it) {
        const uint32_t ly = blockIdx.y * LV + it;
        if (ly >= n_levels) break;                                   // uniform
        const uint32_t level = level_base + ly, cur = it & 1u;
        uint32_t *hist_c = hist + cur * NB, *side_n = misc + 1u + cur;
        const LevelMeta m = make_level_meta<3>(offsets, level, H);
        float *__restrict__ gg = grad_table + (size_t)m.offset * 2u;
        const RowMap map = make_row_map(m.mode, m.size, log2_nb);
        // merge runs of equal cells inside the wave where cells are at least two sample spacings wide (an invariant of the level, see
        // scatter_binned.h: fewer than 2^16 cells per axis, so that the cell key packs)
        const bool merging = m.scale * spacing < 0.5f && m.scale < 65535.0f;
        const bool fast = !merging && bucket_shift(m.size, 0u) <= map.hs;             // not merged, one chunk per bucket: row >> hs == 0

#ifndef NAF_DIAG_BLOCKS_NO_MAX
        if constexpr (kFromBlocks) {
            // largest |gradient| of the level in this wave, as an fp32 bit pattern (NaN > Inf > finite: a non-finite gradient poisons the
            // step as it does on one GPU; a lane past the end of the batch holds the last point's gradient, a value of the batch).  Per
            // level and through LDS rather than in a register across the level bodies: the 128-register variants have none to spare.
            // The wave keeps the largest value it knows of in a scalar register (at first the workgroup's word) and does nothing while no
            // lane exceeds it -- gradients of one step are of one magnitude, so a wave mostly pays three vector instructions per point
            // and a vote.
            uint32_t gm = 0u;
#pragma unroll
            for (uint32_t q = 0; q < PTS; ++q) gm = max(gm, max((graw[q] << 16) & 0x7fffffffu, graw[q] & 0x7fff0000u));
            if (__ballot(gm > wmax) != 0ull) {
                wmax = (uint32_t)__builtin_amdgcn_readlane((int)wave_max_to_lane63(gm), 63);
                if (lane == 0u) atomicMax(&misc[3], wmax);
            }
        }
#endif
        uint32_t head[PTS][4], pay[PTS][4], bkt[PTS][4], rank[PTS][4];
        bool on[PTS];
        // a second corner that travels as a record of its own: counted and ranked like every record, kept in the side list
        auto emit_b = [&](uint32_t rb, uint32_t p) __attribute__((always_inline)) {
            const uint32_t i = atomicAdd(side_n, 1u);
            if (i < side_cap) {
                const uint32_t b = map.bucket(rb, nb_mask), r = atomicAdd(&hist_c[b], 1u);
                side[3u * i] = (b << 16) | r; side[3u * i + 1u] = map.local(rb); side[3u * i + 2u] = p;
            } else {                                                 // a tile of nothing but singles with a full list: see spill_record
                spill_record(gg, rb, 0u, p);
                ++n_overflow;
                ++n_overflow_level;
            }
        };
        // ---- A: records (registers) + rank inside the bucket ------------------------------------------------------------------
#pragma unroll
        for (uint32_t q = 0; q < PTS; ++q) {
            float g[2];
            g[0] = valid[q] ? __uint_as_float(graw[q] << 16) : 0.0f;
            g[1] = valid[q] ? __uint_as_float(graw[q] & 0xffff0000u) : 0.0f;
            // SrcRays guarantees coordinates in [0, 1]: pos >= 0.5, so the conversion IS the floor and v_fract the fraction (locate()
            // spells out the reference's behaviour for coordinates nobody checked; a NaN lands in cell 0 with a NaN fraction there too)
            float frac[3];
            uint32_t pg[3];
#pragma unroll
            for (uint32_t d = 0; d < 3; ++d) {
                const float pos = __fmaf_rn(x[q][d], m.scale, 0.5f);
                pg[d] = (uint32_t)pos;
                frac[d] = __builtin_amdgcn_fractf(pos);
            }
            uint32_t ra[4], xm[4];                                   // row of the first corner of pair k, xor distance to the second
            bool formed[4];                                          // xm has the form 2^e - 1 (always, except behind a true modulo)
            dispatch_mode<true>(m.mode, [&](auto mode_tag) {
                constexpr uint32_t MODE = decltype(mode_tag)::value;
                if constexpr (is_hash_mode(MODE)) {
                    const uint32_t ty0 = pg[1] * kPrime1, ty1 = ty0 + kPrime1, tz0 = pg[2] * kPrime2, tz1 = tz0 + kPrime2;
                    const uint32_t h[4] = {ty0 ^ tz0, ty1 ^ tz0, ty0 ^ tz1, ty1 ^ tz1};
#pragma unroll
                    for (uint32_t k = 0; k < 4; ++k) {
                        if constexpr (MODE == kHashMask) {
                            ra[k] = (pg[0] ^ h[k]) & (m.size - 1u);
                            xm[k] = (pg[0] ^ (pg[0] + 1u)) & (m.size - 1u);
                            formed[k] = true;
                        } else {
                            ra[k] = (pg[0] ^ h[k]) % m.size;
                            xm[k] = ra[k] ^ (((pg[0] + 1u) ^ h[k]) % m.size);
                            formed[k] = (xm[k] & (xm[k] + 1u)) == 0u;
                        }
                    }
                } else {
                    const uint32_t ty0 = pg[1] * m.stride1, tz0 = pg[2] * m.stride2;
                    const uint32_t base = pg[0] + ty0 + tz0;
                    const uint32_t raw[4] = {base, base + m.stride1, base + m.stride2, base + m.stride1 + m.stride2};
#pragma unroll
                    for (uint32_t k = 0; k < 4; ++k) {
                        uint32_t rb;
                        if constexpr (MODE == kDenseNoMod) { ra[k] = raw[k]; rb = raw[k] + 1u; formed[k] = true; }
                        else if constexpr (MODE == kDenseMask) { ra[k] = raw[k] & (m.size - 1u); rb = (raw[k] + 1u) & (m.size - 1u); formed[k] = true; }
                        else { ra[k] = raw[k] % m.size; rb = (raw[k] + 1u) % m.size; }
                        xm[k] = ra[k] ^ rb;
                        if constexpr (MODE == kDenseMod) formed[k] = (xm[k] & (xm[k] + 1u)) == 0u;
                    }
                }
            });
            const float wy[2] = {1.0f - frac[1], frac[1]}, wz[2] = {1.0f - frac[2], frac[2]};
            const float fx = frac[0], gx = 1.0f - frac[0];
            const uint32_t fxq = (uint32_t)(fx * 32768.0f);          // < 32768: fx < 1
            on[q] = valid[q];
            // Merged levels: all eight corner contributions are summed over each run of equal cells with a segmented inclusive scan on
            // DPP row shifts (runs cut at 16-lane rows; scatter_binned.h) and only the last lane of a run emits.  A run of ONE point
            // (len == 0 at its last lane) still is one product times one fraction: it travels as pair records like everywhere else;
            // a truly merged run emits eight singles.  Either way a point costs at most four record slots and two side-list
            // entries, whatever the geometry.
            float va[4][2], vb[4][2];
            bool merged_lane = false;                                // this lane closes a run of two or more points
            if (merging) {
#pragma unroll
                for (uint32_t k = 0; k < 4; ++k) {
                    const float wyz = wy[k & 1u] * wz[k >> 1];
#pragma unroll
                    for (uint32_t c = 0; c < 2; ++c) { const float p = wyz * g[c]; va[k][c] = p * gx; vb[k][c] = p * fx; }
                }
                const uint32_t c_lo = pg[0] | (pg[1] << 16), c_hi = pg[2] | (valid[q] ? 0u : 0x80000000u);
                const uint32_t p_lo = dpp_row_shr<1>(c_lo), p_hi = dpp_row_shr<1>(c_hi);
                const uint64_t heads = __ballot((lane & 15u) == 0u || c_lo != p_lo || c_hi != p_hi);
                const uint32_t first = 63u - (uint32_t)__clzll(heads & (~0ull >> (63u - lane)));
                const uint32_t len = lane - first;
                auto fold = [&](auto shift_tag) {
                    constexpr uint32_t d = decltype(shift_tag)::value;
                    const float take = len >= d ? 1.0f : 0.0f;
#pragma unroll
                    for (uint32_t k = 0; k < 4; ++k)
#pragma unroll
                        for (uint32_t c = 0; c < 2; ++c) {
                            va[k][c] = __builtin_fmaf(dpp_row_shr<d>(va[k][c]), take, va[k][c]);
                            vb[k][c] = __builtin_fmaf(dpp_row_shr<d>(vb[k][c]), take, vb[k][c]);
                        }
                };
                fold(std::integral_constant<uint32_t, 1>{});
                fold(std::integral_constant<uint32_t, 2>{});
                fold(std::integral_constant<uint32_t, 4>{});
                fold(std::integral_constant<uint32_t, 8>{});
                on[q] = valid[q] && (lane == 63u || ((heads >> (lane + 1u)) & 1ull));
                merged_lane = on[q] && len != 0u;
            }
            {
                // the common case first, for every lane: a pair record per k ...
                bool single[4], any = false;
#pragma unroll
                for (uint32_t k = 0; k < 4; ++k) {
                    const float wyz = wy[k & 1u] * wz[k >> 1];
                    single[k] = on[q] && !merged_lane && !(formed[k] && xm[k] <= map.smask);
                    any = any || single[k];
                    bkt[q][k] = fast ? ra[k] >> map.s : map.bucket(ra[k], nb_mask);
                    head[q][k] = (fast ? ra[k] & map.smask : map.local(ra[k])) | ((uint32_t)__builtin_popcount(xm[k]) << kFxEShift) | (fxq << kFxShift);
                    pay[q][k] = pack_bf16x2(wyz * g[0], wyz * g[1]);
                }
                // ... then the lanes whose corners have different owners (2^-sh of the pairs: a wave-level branch that is almost never
                // taken on the hashed levels) turn theirs into two singles
                if (__ballot(any) != 0ull) {
#pragma unroll
                    for (uint32_t k = 0; k < 4; ++k) {
                        if (single[k]) {
                            const float wyz = wy[k & 1u] * wz[k >> 1];
                            const float p0 = wyz * g[0], p1 = wyz * g[1];
                            head[q][k] &= (1u << kFxLocalBits) - 1u; pay[q][k] = pack_bf16x2(p0 * gx, p1 * gx);
                            emit_b(ra[k] ^ xm[k], pack_bf16x2(p0 * fx, p1 * fx));
                        }
                    }
                }
            }
            if (merging && __ballot(merged_lane) != 0ull) {
#pragma unroll
                for (uint32_t k = 0; k < 4; ++k) {
                    if (merged_lane) {
                        head[q][k] &= (1u << kFxLocalBits) - 1u; pay[q][k] = pack_bf16x2(va[k][0], va[k][1]);
                        emit_b(ra[k] ^ xm[k], pack_bf16x2(vb[k][0], vb[k][1]));
                    }
                }
            }
            if (fast) {                                              // every lane emits (see the header): no exec juggling around the atomics
#pragma unroll
                for (uint32_t k = 0; k < 4; ++k) rank[q][k] = atomicAdd(&hist_c[bkt[q][k]], 1u);           // ds_add_rtn_u32
            } else {
#pragma unroll
                for (uint32_t k = 0; k < 4; ++k) {
                    rank[q][k] = 0u;
                    if (on[q]) rank[q][k] = atomicAdd(&hist_c[bkt[q][k]], 1u);
                }
            }
        }
        // the next level's gradients: requested here, consumed right after the barrier and before this level's stores are issued (vmcnt
        // retires in order: a wait placed behind the stores would sit out their round trip -- scatter_binned.h)
        if (it + 1u < LV && ly + 1u < n_levels) {
#pragma unroll
            for (uint32_t q = 0; q < PTS; ++q) graw[q] = grad_at(level + 1u, q);
        }
        lds_barrier();
#pragma unroll
        for (uint32_t q = 0; q < PTS; ++q) asm volatile("" : "+v"(graw[q]) : : "memory");      // pin the wait here
#ifndef NAF_DIAG_BLOCKS_NO_TAIL
        if constexpr (kFromBlocks) {
            // The workgroup publishes its maximum ONCE, behind the first barrier of its last level (every wave has looked at every level by
            // then), and only if it beats what the workgroup found published at its start.  Same-address atomics retire ~12 ns apart and a
            // wave's slot is not released before its atomic is acknowledged; here the rest of the level hides part of that.  Measured on
            // one box, us per launch on top of a build that takes no maximum at all (50.5 / 41.5 us at 8 / 4 ranks x 1 024 rays,
            // profiles/round4_ab_levels_in_place_maximum_variants.jsonl): here +2 / +5.5; at the very end of the workgroup +6 / +9, behind a
            // fresh look at the global word there +3.5 / +7; an atomic per raise of the workgroup's word +8.5 / +27.  The per-level part above
            // costs 0.7.  No memory wait follows in this wave: nothing is prefetched any more.
            if ((it + 1u == LV || ly + 1u == n_levels) && threadIdx.x == 0u) {
                const uint32_t m = misc[3];
                if (m > seed) atomicMax(gb.gmax_bits, m);
            }
        }
#endif

        // ---- B: counters -> exclusive offsets, run words, total ---------------------------------------------------------------------
        // (the counters and the side-list length of the OTHER parity are what the next level uses: nobody reads them any more -- their
        // last readers passed the previous level's second barrier -- and nobody adds to them before this level's second barrier)
        if (wave == 0u) {
            for (uint32_t i = lane; i < NB; i += 64u) hist[(cur ^ 1u) * NB + i] = 0u;
            if (lane == 0u) misc[1u + (cur ^ 1u)] = 0u;
        }
        uint32_t excl = 0u, total = 0u;
        if constexpr (kWaveScan) {
            const uint32_t n = hist_c[lane];
            const uint32_t incl = wave_inclusive_sum(n);
            excl = incl - n;
            total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            if (wave == 0u) {
                const uint32_t start_c = min(excl, SLOTS), n_c = min(n, SLOTS - start_c);      // what fits the block
                runs[run_index(plan, ly, lane, tile)] = start_c | (n_c << 16);
            }
        } else {
            if (wave == 0u) {
                const uint32_t per = NB >> 6;
                uint32_t mine = 0u;
                for (uint32_t j = 0; j < per; ++j) mine += hist_c[lane * per + j];
                const uint32_t incl = wave_inclusive_sum(mine);
                uint32_t run_start = incl - mine;
                for (uint32_t j = 0; j < per; ++j) {
                    const uint32_t b = lane * per + j, n = hist_c[b];
                    start[b] = run_start;
                    const uint32_t start_c = min(run_start, SLOTS), n_c = min(n, SLOTS - start_c);
                    runs[run_index(plan, ly, b, tile)] = start_c | (n_c << 16);
                    run_start += n;
                }
                if (lane == 63u) misc[0] = incl;
            }
            lds_barrier();
            total = misc[0];
        }

        // ---- C: placement -------------------------------------------------------------------------------------------------
        auto offset_of = [&](uint32_t b) __attribute__((always_inline)) -> uint32_t {
            if constexpr (kWaveScan) return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(b << 2), (int)excl);
            else return start[b];
        };
        bool over = false;
#pragma unroll
        for (uint32_t q = 0; q < PTS; ++q) {
            if (fast) {
#pragma unroll
                for (uint32_t k = 0; k < 4; ++k) {
                    const uint32_t pos = offset_of(bkt[q][k]) + rank[q][k];
                    over = over || pos >= SLOTS;
                    staging[min(pos, SLOTS)] = make_uint2(head[q][k], pay[q][k]);
                }
            } else {
#pragma unroll
                for (uint32_t k = 0; k < 4; ++k) {
                    const uint32_t pos = offset_of(bkt[q][k]) + rank[q][k];          // every lane takes part in the bpermute
                    if (on[q]) {
                        over = over || pos >= SLOTS;
                        staging[min(pos, SLOTS)] = make_uint2(head[q][k], pay[q][k]);
                    }
                }
            }
        }
        {
            const uint32_t n_side = min(*side_n, side_cap);
            for (uint32_t i0 = 0; i0 < n_side; i0 += NT) {                            // uniform trip count: bpermute needs the whole wave
                const uint32_t i = i0 + threadIdx.x;
                const bool live = i < n_side;
                const uint32_t w0 = live ? side[3u * i] : 0u;
                const uint32_t pos = offset_of(w0 >> 16) + (w0 & 0xffffu);
                if (live) {
                    const uint32_t h = side[3u * i + 1u], p = side[3u * i + 2u];
                    if (pos < SLOTS) staging[pos] = make_uint2(h, p);
                    else {
                        spill_record(gg, map.row(w0 >> 16, h), h, p);
                        ++n_overflow;
                        ++n_overflow_level;
                    }
                }
            }
        }
        if (__ballot(over) != 0ull) {                                     // a full block (see spill_record): one vote per level
#pragma unroll
            for (uint32_t q = 0; q < PTS; ++q)
#pragma unroll
                for (uint32_t k = 0; k < 4; ++k) {
                    const uint32_t pos = offset_of(bkt[q][k]) + rank[q][k];
                    if ((fast || on[q]) && pos >= SLOTS) {
                        spill_record(gg, map.row(bkt[q][k], head[q][k] & ((1u << kFxLocalBits) - 1u)), head[q][k], pay[q][k]);
                        ++n_overflow;
                        ++n_overflow_level;
                    }
                }
        }
        lds_barrier();

        // ---- D: the dense block leaves as whole 128-byte lines (16 bytes per lane; stale slots past the end are harmless) -----
        {
            const uint32_t n_rec = min(total, SLOTS);
            const uint32_t n_chunk = min(((n_rec * 8u + 127u) >> 7) << 3, (SLOTS * 8u) >> 4);
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            u32x4 *__restrict__ dst = reinterpret_cast<u32x4 *>(blocks + block_index(plan, ly, tile));
            const u32x4 *from = reinterpret_cast<const u32x4 *>(staging);
            // non-temporal: written once, read once by pass 2 -- kept out of the Infinity Cache they leave the optimiser state there
            for (uint32_t c = threadIdx.x; c < n_chunk; c += NT) __builtin_nontemporal_store(from[c], dst + c);
        }
        if (__ballot(n_overflow_level != 0u)) {                          // diagnostics: overflow per level (overflow[1 + level])
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) n_overflow_level += __shfl_xor(n_overflow_level, off, 64);
            if (lane == 0u) atomicAdd(overflow + 1u + level, n_overflow_level);
            n_overflow_level = 0u;
        }
        // no barrier here: the next level adds to the counters of the other parity and writes the staging block only behind ITS first
        // barrier -- every wave has finished this copy before it arrives there
    }
    if (__ballot(n_overflow != 0u)) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) n_overflow += __shfl_xor(n_overflow, off, 64);
        if (lane == 0u) atomicAdd(overflow, n_overflow);
    }
}

// ---- pass 2 ---------------------------------------------------------------------------------------------------
// One workgroup = one (bucket, level): the table rows [bucket << sh, (bucket + 1) << sh) of the level -- contiguous.  Record phase as in
// scatter_binned.h (every wave streams one contiguous range of tiles, eight runs in flight per lane, tails of two runs share an
// instruction); accumulators acc[channel][local row], 64-bit fixed point (ds_add_u64), filled through the integer pipe (see the header).
// kAdam: the workgroup finishes its rows with their Adam update (naf_render_train_adam) -- a plain stream over contiguous memory,
// 16 bytes per lane and array (two rows x two channels), operands requested in front of the record phase; kFast picks the form of
// adam_math.h (tables with a 16-bit shadow).
__device__ __forceinline__ int cvt_rpi(float v) {                // floor(v + 0.5): one instruction, no bias towards zero
    int i;                                                        // (plain truncation measured the same: profiles/round4_ab_*)
    asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(i) : "v"(v));
    return i;
}

// a record as one 8-byte load; NAF_V2_NT_LOADS (A/B builds, tools/build_variant.sh): non-temporal, so that records read once do not
// displace the optimiser state -- measured flat at every batch size (profiles/round4_ab_reducer_nt_loads_and_levels_per_bin_workgroup.jsonl)
__device__ __forceinline__ PairFx load_record(const PairFx *p) {
#ifdef NAF_V2_NT_LOADS
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    const u32x2 v = __builtin_nontemporal_load(reinterpret_cast<const u32x2 *>(p));
    return PairFx{v.x, v.y};
#else
    return *p;
#endif
}

#ifdef NAF_REDUCE_STAMPS        // diagnostic builds (tools/reduce_stamps.py): shader-clock stamps of pass 2's phases, eight words per workgroup
__device__ uint32_t g_reduce_stamps[2048 * 8];
#define NAF_RSTAMP(i) rs_[i] = clock64()
#else
#define NAF_RSTAMP(i)
#endif

template <bool kAdam, bool kFast>
__global__ void __launch_bounds__(1024)
scatter_reduce2_kernel(const PairFx *__restrict__ blocks, const uint32_t *__restrict__ runs, const int32_t *__restrict__ offsets,
                       float *__restrict__ grad_table, const uint32_t *__restrict__ gmax_bits, uint32_t level_base, uint32_t ly_begin,
                       uint32_t H, BinPlan plan, AdamTail adam) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
#ifdef NAF_REDUCE_STAMPS
    long long rs_[5] = {0, 0, 0, 0, 0};
    const long long rwall_ = wall_clock64();
#endif
    NAF_RSTAMP(0);
    const uint32_t bucket = blockIdx.x, ly = ly_begin + blockIdx.y, level = level_base + ly;      // ly: level slot of the bin pass
    const LevelMeta lm = make_level_meta<3>(offsets, level, H);
    const uint32_t off = lm.offset, T = lm.size;
    const RowMap map = make_row_map(lm.mode, T, plan.log2_nb);
    if ((bucket << map.s) >= T) return;                          // a level with fewer chunks than buckets: no rows, no records
    // local rows this bucket may own: whole chunks of 2^s rows (the last chunk of the level may be cut short: row < T is checked per row)
    const uint32_t rows_local = ((T + ((1u << map.hs) - 1u)) >> map.hs) << map.s;
    const uint32_t gbits = *gmax_bits;
    const int shift = fixed_shift2(gbits);
    const bool poison = fixed_nonfinite(gbits);
    const float scale = __uint_as_float((uint32_t)(shift + 127) << 23);               // 2^shift, -102 <= shift <= 120
    const float scale_fx = __uint_as_float((uint32_t)(shift + 127 - (int)kFxBits) << 23);
    const uint32_t pitch = plan.max_local_rows, T_ = blockDim.x;
    unsigned long long *acc0 = reinterpret_cast<unsigned long long *>(smem), *acc1 = acc0 + pitch;

    struct __attribute__((packed, aligned(4))) Quad { float x, y, z, w; };              // level offsets may be odd: 8-byte aligned only
#ifndef NAF_RED_PREQ
#define NAF_RED_PREQ 2u          // Two of a thread's four quads at T = 2^19: tools/reduce_stamps.py shows the workgroup's loads returning at
#endif                           // the rate the memory system delivers them, in order -- with all four quads (196 KB) requested in front, the
                                 // records queue behind them and the record phase starts at 18 000 of the workgroup's 43 000 cycles.  Same-box
                                 // A/B, 1 / 2 / 3 / 4 quads: 0.092 / 0.090 / 0.090-0.091 / 0.092-0.093 ms at 1 024 rays, flat at 65 536
                                 // (profiles/round4_ab_reducer_adam_prefetch_depth.jsonl; A/B builds override the macro).
    constexpr uint32_t kPreQ = kAdam ? NAF_RED_PREQ : 0u;       // quads (two rows x two channels) prefetched per thread
    Quad preq_p[kPreQ ? kPreQ : 1u], preq_m[kPreQ ? kPreQ : 1u], preq_v[kPreQ ? kPreQ : 1u];
    // quad q = local rows 2q, 2q + 1 = two CONSECUTIVE table rows when chunks hold at least two rows (s >= 1) and both exist
    const uint32_t n_quads = (rows_local + 1u) >> 1;
    auto quad_row = [&](uint32_t q) { return map.row(bucket, 2u * q); };
    auto quad_full = [&](uint32_t row0) { return map.s != 0u && row0 + 1u < T; };
    if constexpr (kPreQ != 0u) {
#pragma unroll
        for (uint32_t k = 0; k < kPreQ; ++k) {
            const uint32_t q = threadIdx.x + k * T_;
            const uint32_t row0 = quad_row(q);
            preq_p[k] = preq_m[k] = preq_v[k] = Quad{0.0f, 0.0f, 0.0f, 0.0f};
            if (q < n_quads && quad_full(row0)) {
                const size_t e = ((size_t)off + row0) * 2u;
                preq_p[k] = *reinterpret_cast<const Quad *>(adam.param + e);
                preq_m[k] = *reinterpret_cast<const Quad *>(adam.m + e);
                preq_v[k] = *reinterpret_cast<const Quad *>(adam.v + e);
            }
        }
    }
    const uint32_t lane = threadIdx.x & 63u, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), n_waves = T_ >> 6;
    const size_t run0 = run_index(plan, ly, bucket, 0);
    const uint32_t split_tiles = (plan.n_tiles + gridDim.z - 1u) / gridDim.z;
    const uint32_t split_begin = blockIdx.z * split_tiles, split_end = min(plan.n_tiles, split_begin + split_tiles);
    const uint32_t share = (split_tiles + n_waves - 1u) / n_waves;
    const uint32_t per_wave = share >= 64u ? (share + 63u) & ~63u : (share + 7u) & ~7u;
    const uint32_t t_begin = split_begin + wave * per_wave, t_end = min(split_end, t_begin + per_wave);
    const uint32_t first_runs = t_begin + lane < t_end ? runs[run0 + t_begin + lane] : 0u;
    {
        typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
        u64x2 *a2 = reinterpret_cast<u64x2 *>(smem);
        for (uint32_t i = threadIdx.x; i < pitch; i += T_) a2[i] = u64x2{0ull, 0ull};      // pitch * 2 cells of 8 bytes
    }
    __syncthreads();
    NAF_RSTAMP(1);

    auto add = [&](const PairFx &r) {
        const uint32_t la = r.head & ((1u << kFxLocalBits) - 1u);
        const uint32_t e = (r.head >> kFxEShift) & 15u;
        const uint32_t lb = la ^ ((1u << e) - 1u);
        const float t = (float)(r.head >> kFxShift) * scale_fx;                          // f_x * 2^shift
        const float p0 = __uint_as_float(r.pay << 16), p1 = __uint_as_float(r.pay & 0xffff0000u);
        const float b0 = p0 * t, b1 = p1 * t;                                            // exact: 8 x 15 bits
        const float a0 = p0 * scale - b0, a1 = p1 * scale - b1;                          // exact: p * 2^shift * (1 - f_x)
        atomicAdd(&acc0[la], (unsigned long long)(long long)cvt_rpi(a0));                // ds_add_u64
        atomicAdd(&acc1[la], (unsigned long long)(long long)cvt_rpi(a1));
        if (e != 0u) {                                                                   // singles (the merged levels' records) add nothing there
            atomicAdd(&acc0[lb], (unsigned long long)(long long)cvt_rpi(b0));
            atomicAdd(&acc1[lb], (unsigned long long)(long long)cvt_rpi(b1));
        }
    };
    constexpr uint32_t kGroup = 8u, kTail = 32u;
    if (plan.log2_w < 6u) {
        // many buckets (T >= 2^20): short runs -- a wave takes G = 64 / W runs per instruction, W lanes each (scatter_binned.h)
        const uint32_t W = 1u << plan.log2_w, G = 64u >> plan.log2_w, g = lane >> plan.log2_w, j = lane & (W - 1u);
        constexpr uint32_t kSteps = 4;
        for (uint32_t t0 = t_begin; t0 < t_end; t0 += 64u) {
            const uint32_t mine = t0 == t_begin ? first_runs : (t0 + lane < t_end ? runs[run0 + t0 + lane] : 0u);
            const uint32_t n_here = min(64u, t_end - t0);
            for (uint32_t s0 = 0; s0 < n_here; s0 += kSteps * G) {
                uint32_t n4[kSteps];
                const PairFx *b4[kSteps];
                PairFx r4[kSteps];
#pragma unroll
                for (uint32_t u = 0; u < kSteps; ++u) {
                    const uint32_t tl = s0 + u * G + g;
                    const uint32_t word = (uint32_t)__shfl((int)mine, (int)min(tl, 63u), 64);
                    n4[u] = tl < n_here ? word >> 16 : 0u;
                    b4[u] = blocks + block_index(plan, ly, t0 + min(tl, n_here - 1u)) + (word & 0xffffu);
                    r4[u] = load_record(b4[u] + (j < n4[u] ? j : 0u));
                }
#pragma unroll
                for (uint32_t u = 0; u < kSteps; ++u)
                    if (j < n4[u]) add(r4[u]);
#pragma unroll 1
                for (uint32_t u = 0; u < kSteps; ++u)
                    for (uint32_t i = W + j; __ballot(i < n4[u]) != 0ull; i += W)
                        if (i < n4[u]) add(load_record(b4[u] + i));
            }
        }
    } else
    for (uint32_t t0 = t_begin; t0 < t_end; t0 += 64u) {
        const uint32_t mine = t0 == t_begin ? first_runs : (t0 + lane < t_end ? runs[run0 + t0 + lane] : 0u);
        const uint32_t n_here = min(64u, t_end - t0);
        for (uint32_t j = 0; j < n_here; j += kGroup) {
            uint32_t n[kGroup], n_max = 0u;
            const PairFx *base[kGroup];
            PairFx ra[kGroup];
#pragma unroll
            for (uint32_t u = 0; u < kGroup; ++u) {
                const uint32_t tj = min(j + u, n_here - 1u);
                const uint32_t word = (uint32_t)__builtin_amdgcn_readlane((int)mine, (int)tj);              // scalar (tj is wave-uniform)
                n[u] = j + u < n_here ? word >> 16 : 0u;
                n_max = max(n_max, n[u]);
                base[u] = blocks + block_index(plan, ly, t0 + tj) + (word & 0xffffu);
                ra[u] = load_record(base[u] + (lane < n[u] ? lane : 0u));
            }
            {
                constexpr uint32_t kPer = 64u / kTail;                      // runs per tail instruction
                PairFx rt[kGroup / kPer];
                uint32_t nt[kGroup / kPer];
#pragma unroll
                for (uint32_t q = 0; q < kGroup / kPer; ++q) {
                    const uint32_t sub = lane / kTail, slot = 64u + (lane % kTail);
                    nt[q] = n[q * kPer];
                    const PairFx *bq = base[q * kPer];
#pragma unroll
                    for (uint32_t k = 1; k < kPer; ++k) {
                        nt[q] = sub == k ? n[q * kPer + k] : nt[q];
                        bq = sub == k ? base[q * kPer + k] : bq;
                    }
                    rt[q] = load_record(bq + (slot < nt[q] ? slot : 0u));
                }
#pragma unroll
                for (uint32_t u = 0; u < kGroup; ++u)
                    if (lane < n[u]) add(ra[u]);
#pragma unroll
                for (uint32_t q = 0; q < kGroup / kPer; ++q)
                    if (64u + (lane % kTail) < nt[q]) add(rt[q]);
            }
            if (n_max > 64u + kTail) {                                      // long runs (clustered tiles, merged levels' singles): the rest, run by run
#pragma unroll 1
                for (uint32_t u = 0; u < kGroup; ++u)
                    for (uint32_t i = 64u + kTail + lane; i < n[u]; i += 64u) add(load_record(base[u] + i));
            }
        }
    }
    NAF_RSTAMP(2);
    __syncthreads();
    NAF_RSTAMP(3);

    float *__restrict__ gg = grad_table + (size_t)off * 2u;
    const float nan = __builtin_nanf("");
    // fixed point -> fp32 of the four sums of a quad (rows 2q, 2q + 1 x channels 0, 1).  Sums that fit 32 bits -- nearly all -- convert
    // with v_cvt_f32_i32 + v_ldexp_f32: one rounding of the same exact value as the fp64 route; chosen per wave (scatter_binned.h).
    auto sums = [&](uint32_t q, float (&g)[4]) {
        const unsigned long long *a0 = &acc0[2u * q], *a1 = &acc1[2u * q];
        const long long x[4] = {(long long)a0[0], (long long)a1[0], (long long)a0[1], (long long)a1[1]};
        bool small = true;
#pragma unroll
        for (uint32_t j = 0; j < 4; ++j) small = small && (x[j] == (long long)(int)x[j]);
        if (__ballot(!small) == 0ull) {
#pragma unroll
            for (uint32_t j = 0; j < 4; ++j) g[j] = ldexpf((float)(int)x[j], -shift);
        } else {
#pragma unroll
            for (uint32_t j = 0; j < 4; ++j) g[j] = (float)ldexp((double)x[j], -shift);
        }
        if (poison) g[0] = g[1] = g[2] = g[3] = nan;
    };
    if constexpr (kAdam) {
        // A level whose tiles overflowed their blocks has contributions in the gradient table already (atomics of pass 1): they are
        // added and cleared here.
        const bool spilled = adam.overflow[1u + level] != 0u;
        float *__restrict__ pp = adam.param + (size_t)off * 2u, *__restrict__ pm = adam.m + (size_t)off * 2u, *__restrict__ pv = adam.v + (size_t)off * 2u;
        auto store_lp = [&](size_t el, uint32_t count, const float (&p)[4]) {       // 16-bit shadow of `count` elements from level element `el`
            if (adam.lp == nullptr) return;
            uint16_t *lp = reinterpret_cast<uint16_t *>(adam.lp) + (size_t)off * 2u + el;
            uint32_t lo, hi;
            if (adam.lp_dtype == kAdamLpF16) {
                const _Float16 h0 = (_Float16)p[0], h1 = (_Float16)p[1], h2 = (_Float16)p[2], h3 = (_Float16)p[3];
                lo = (uint32_t)__builtin_bit_cast(uint16_t, h0) | ((uint32_t)__builtin_bit_cast(uint16_t, h1) << 16);
                hi = (uint32_t)__builtin_bit_cast(uint16_t, h2) | ((uint32_t)__builtin_bit_cast(uint16_t, h3) << 16);
            } else {
                lo = pack_bf16x2(p[0], p[1]);
                hi = pack_bf16x2(p[2], p[3]);
            }
            struct __attribute__((packed, aligned(4))) Half4 { uint32_t lo, hi; };
            if (count == 4u) *reinterpret_cast<Half4 *>(lp) = Half4{lo, hi};
            else *reinterpret_cast<uint32_t *>(lp) = lo;
        };
        auto one_row = [&](uint32_t row, float g0, float g1) {                       // two elements of a row that has no partner in its quad
            const size_t el = (size_t)row * 2u;
            float p[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            const float g[2] = {g0, g1};
#pragma unroll
            for (uint32_t c = 0; c < 2; ++c) {
                float gc = g[c];
                if (spilled) {
                    const float extra = gg[el + c];
                    if (extra != 0.0f) { gc = extra + gc; gg[el + c] = 0.0f; }
                }
                float m = pm[el + c], v = pv[el + c];
                p[c] = pp[el + c];
                adam_one<kFast>(p[c], m, v, gc, adam.a);
                pp[el + c] = p[c]; pm[el + c] = m; pv[el + c] = v;
            }
            store_lp(el, 2u, p);
        };
        auto finish = [&](uint32_t q, bool prefetched, const Quad &p4, const Quad &m4, const Quad &v4) {
            if (q >= n_quads) return;
            float g[4];
            sums(q, g);
            const uint32_t row0 = quad_row(q);
            if (quad_full(row0)) {
                const size_t el = (size_t)row0 * 2u;
                Quad P = p4, M = m4, V = v4;
                if (!prefetched) {
                    P = *reinterpret_cast<const Quad *>(pp + el);
                    M = *reinterpret_cast<const Quad *>(pm + el);
                    V = *reinterpret_cast<const Quad *>(pv + el);
                }
                if (spilled) {
                    const Quad extra = *reinterpret_cast<const Quad *>(gg + el);
                    const float x[4] = {extra.x, extra.y, extra.z, extra.w};
                    bool any = false;
#pragma unroll
                    for (uint32_t j = 0; j < 4; ++j)
                        if (x[j] != 0.0f) { g[j] = x[j] + g[j]; any = true; }          // the order of the separate route: table += sum
                    if (any) *reinterpret_cast<Quad *>(gg + el) = Quad{0.0f, 0.0f, 0.0f, 0.0f};
                }
                float p[4] = {P.x, P.y, P.z, P.w}, m[4] = {M.x, M.y, M.z, M.w}, v[4] = {V.x, V.y, V.z, V.w};
#pragma unroll
                for (uint32_t j = 0; j < 4; ++j) adam_one<kFast>(p[j], m[j], v[j], g[j], adam.a);
                *reinterpret_cast<Quad *>(pp + el) = Quad{p[0], p[1], p[2], p[3]};
                *reinterpret_cast<Quad *>(pm + el) = Quad{m[0], m[1], m[2], m[3]};
                *reinterpret_cast<Quad *>(pv + el) = Quad{v[0], v[1], v[2], v[3]};
                store_lp(el, 4u, p);
            } else {                                          // the level's last row when its count is odd, or one-row chunks (tiny tables)
                if (row0 < T) one_row(row0, g[0], g[1]);
                const uint32_t row1 = map.row(bucket, 2u * q + 1u);
                if (2u * q + 1u < rows_local && row1 < T) one_row(row1, g[2], g[3]);
            }
        };
#pragma unroll
        for (uint32_t k = 0; k < kPreQ; ++k) finish(threadIdx.x + k * T_, true, preq_p[k], preq_m[k], preq_v[k]);
        const Quad none{0.0f, 0.0f, 0.0f, 0.0f};
        for (uint32_t q = threadIdx.x + kPreQ * T_; q < n_quads; q += T_) finish(q, false, none, none, none);
    } else {
        // gradient table += sums: in place and coalesced when this workgroup is the sole owner of its rows, with one fp32 atomic per
        // element when the bucket's tiles were split between gridDim.z workgroups
        for (uint32_t q = threadIdx.x; q < n_quads; q += T_) {
            float g[4];
            sums(q, g);
            const uint32_t row0 = quad_row(q);
            if (gridDim.z == 1u && quad_full(row0)) {
                Quad *dst = reinterpret_cast<Quad *>(gg + (size_t)row0 * 2u);
                const Quad old = *dst;
                *dst = Quad{old.x + g[0], old.y + g[1], old.z + g[2], old.w + g[3]};
            } else {
#pragma unroll
                for (uint32_t r = 0; r < 2; ++r) {
                    const uint32_t row = map.row(bucket, 2u * q + r);
                    if (2u * q + r < rows_local && row < T) {
#pragma unroll
                        for (uint32_t c = 0; c < 2; ++c) {
                            if (gridDim.z == 1u) gg[(size_t)row * 2u + c] += g[2u * r + c];
                            else atomicAdd(gg + (size_t)row * 2u + c, g[2u * r + c]);
                        }
                    }
                }
            }
        }
    }
#ifdef NAF_REDUCE_STAMPS
    NAF_RSTAMP(4);
    if (threadIdx.x == 0u) {
        uint32_t *dbg = g_reduce_stamps + ((blockIdx.y * gridDim.x + blockIdx.x) & 2047u) * 8u;
        for (int i = 0; i < 5; ++i) dbg[i] = (uint32_t)(rs_[i] - rs_[0]);
        dbg[5] = (uint32_t)rwall_; dbg[6] = (uint32_t)wall_clock64(); dbg[7] = level;
    }
#endif
}
#undef NAF_RSTAMP

}  // namespace naf
