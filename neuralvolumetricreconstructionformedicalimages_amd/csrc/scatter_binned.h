// scatter_binned.h -- hash-table gradient scatter without per-contribution global atomics (gfx950).
//
// Why: on MI355X a global float atomic is executed at the memory side, one 64-byte request per touched line; scattered
// 4-byte adds top out at ~2e10 requests/s chip-wide (MI355X_MICROARCH.md "Global float atomics"; measured here:
// 2.4 ms per level for 25 M adds, 12 ms on level 0 where 4913 rows take all of them).  The reference design
// (hashencoder.cu:257-269, one atomicAdd per corner and channel) is ~25x slower than the rest of the training step.
// A level of the table gets ~50-200 contributions per row per step, but they are hash-scattered, so no locality trick
// removes them -- they have to be ROUTED to an owner instead (a one-digit radix multisplit on the row index):
//
//   pass 1  bin      one workgroup = one tile of NT x PTS points x LV consecutive levels (all 16, or 4 when tiles are scarce).
//                    Per level an EXACT multisplit of the tile's records by bucket, in LDS: histogram (one LDS atomic per
//                    record), exclusive scan of the NB counters, placement (one returning LDS atomic per record) -- the
//                    tile's records end up bucket-sorted and DENSE in a staging block, which is copied to the tile's
//                    global block [level][tile][slots] with 16-byte stores of whole 128-byte lines; (start, length) of every
//                    bucket's run goes to runs[level][bucket][tile].  No global atomics, no prefix sums between
//                    workgroups, and no per-bucket capacity: however skewed a tile is (all samples of a ray in one cell,
//                    rays parallel to an axis, ...) its runs simply have different lengths.  Only a tile whose TOTAL
//                    exceeds the block (more than a quarter of its x-neighbour pairs unpaired; never seen) spills the
//                    excess to global atomics, so the result is correct for any input.
//   pass 2  reduce   one workgroup = one (bucket, level): streams the bucket's run of every tile (several loads in
//                    flight per lane), accumulates its rows in LDS as 64-bit fixed point (ds_add_u64), then adds the finished
//                    rows to the gradient table with plain, coalesced read-modify-writes -- it is the only owner of those rows.
//
// PAIR RECORDS.  The two x-neighbour corners of a cell sit in rows r and r' with r ^ r' = 2^e - 1: on hashed levels
// r = x ^ h(y,z) and r' = (x+1) ^ h(y,z) (hashencoder.cu:36-52: the prime of dimension 0 is 1), on dense levels r' = r + 1,
// and x ^ (x+1) is the mask of the trailing ones of x plus one bit.  With probability 1 - 2^-6 the mask is below 64, i.e. both
// rows lie in the same aligned block of 64 rows.  So the bucket is taken from the bits ABOVE the low six, both corners
// reach the same owner, and ONE record carries both:
//     { local row of the first corner | xor mask << 16,  C values of the first corner,  C values of the second }
// 12 bytes instead of 2 x 8 for two bf16 channels -- a quarter fewer bytes through HBM in both passes, half the LDS
// atomics and staging writes in pass 1, half the record loads and index arithmetic in pass 2.  The rare pair whose mask is
// >= 64 travels as two records with mask 0 and a zero second half.
//
// Bucket = bits [6, 6 + log2 NB) of the row, local row = remaining high bits : low six bits.  A bucket's local rows
// 64 k .. 64 k + 63 are 64 CONSECUTIVE table rows, so the reducer adds its sums to the table in place, coalesced.
// Consecutive cells of a ray that runs along x share their 64-row block and therefore their bucket: runs of such tiles come
// in clusters (measured with a fixed per-bucket capacity of mean + 4 sigma: 1-2 % of the records of levels 4..6 overflowed,
// 10 % on some projections) -- which is why the multisplit is exact instead of slotted.
//
// On the coarsest levels (cells wider than the sample spacing) consecutive samples of a ray fall into the same cell: each
// run of equal cells is merged inside the wave first (segmented scan over DPP row shifts) and only its last lane emits.
//
// What the memory system wants (tools/write_pattern_bench.hip, MI355X): line-aligned, dense stores reach > 5 TB/s, ragged
// 16-byte-granular ones ~3 TB/s.  Waves never wait for these stores: the barriers inside the level loop order LDS traffic
// only (lds_barrier), and the global loads of the loop are consumed before the stores.
#pragma once

#include "adam_math.h"
#include "mlp_slabs.h"
#include "naf_device.h"

namespace naf {

// ---- records ---------------------------------------------------------------------------------------------------
// w[0] = local row of the first corner (< 2^16) | xor mask to the second corner's local row << 16 (0: no second corner)
template <uint32_t C>
struct PairF32 {                      // 2 x C fp32 values
    static constexpr uint32_t kHalf = C;
    uint32_t w[1 + 2 * C];
    __device__ __forceinline__ void set(uint32_t head, const float (&a)[C], const float (&b)[C]) {
        w[0] = head;
#pragma unroll
        for (uint32_t c = 0; c < C; ++c) { w[1 + c] = __float_as_uint(a[c]); w[1 + C + c] = __float_as_uint(b[c]); }
    }
    __device__ __forceinline__ float value(uint32_t half, uint32_t c) const { return __uint_as_float(w[1 + half * C + c]); }
};

template <uint32_t C>
struct PairBF16 {                     // 2 x C bf16 values (packed two per dword)
    static constexpr uint32_t kHalf = (C + 1) / 2;
    uint32_t w[1 + 2 * kHalf];
    static __device__ __forceinline__ uint32_t pack2(float lo, float hi) {      // plain casts: one v_cvt_pk_bf16_f32 per pair
        typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
        bf16x2 pk;
        pk[0] = (__bf16)lo;
        pk[1] = (__bf16)hi;
        return __builtin_bit_cast(uint32_t, pk);
    }
    __device__ __forceinline__ void set(uint32_t head, const float (&a)[C], const float (&b)[C]) {
        w[0] = head;
#pragma unroll
        for (uint32_t c = 0; c < C; c += 2) {
            w[1 + c / 2] = pack2(a[c], c + 1 < C ? a[c + 1] : 0.0f);
            w[1 + kHalf + c / 2] = pack2(b[c], c + 1 < C ? b[c + 1] : 0.0f);
        }
    }
    __device__ __forceinline__ float value(uint32_t half, uint32_t c) const {
        const uint32_t p = w[1 + half * kHalf + c / 2];
        return __uint_as_float((c & 1u) ? (p & 0xffff0000u) : (p << 16));
    }
};

struct BinPlan {
    uint32_t tile_points;     // points per pass-1 workgroup (threads x points per thread)
    uint32_t n_tiles;
    uint32_t log2_nb;         // NB = buckets per level
    uint32_t slots;           // records per (level, tile) block = LDS staging slots of pass 1 (a multiple of 32, < 2^16)
    uint32_t levels_per_pass;
    uint32_t max_local_rows;  // ceil(max T_l / NB), rounded up to a multiple of 64
    uint32_t log2_w;          // pass 2: lanes per run (6: a whole wave per run; less when many buckets make the runs short)
};

// Pass-1 shape by record size (compile time): 12-byte records (two bf16 channels) get 512 threads x 2 points -- 1024-point
// tiles, a mean of 64 pair records per bucket, so a reducer wave is full; larger records get smaller tiles (LDS staging).
// With 128 buckets per level and more (tables of 2^20 rows and up) the host launches the 12-byte shape with 2 x kThreads:
// one 1024-thread workgroup per CU on a 2048-point tile, which doubles the run length pass 2 reads (make_bin_plan).
template <typename Rec> struct BinShape {
    static constexpr uint32_t kThreads = sizeof(Rec) <= 12 ? 512u : 256u;
    static constexpr uint32_t kPoints = sizeof(Rec) <= 20 ? 2u : 1u;
};

// Fixed-point scale of the reducer.  `gmax_bits` = bit pattern of max |feature gradient| of the step (written by the MLP
// backward kernel, BEFORE the gradients are rounded to their storage type: rounding can lift a value by at most one ulp of
// bf16, 2^-8 relative, which the bound below absorbs by using the NEXT power of two).  A contribution is w * g with
// 0 <= w <= 1, so |v| <= gmax < 2^(E+1) with E = exponent(gmax); pass 1 may merge the up to 16 same-cell contributions of a
// DPP row into one record, so a record is bounded by 16 * 2^(E+1).
// fixed = v * 2^(kFixHead - E - 1) keeps a single contribution below 2^kFixHead and a record below 2^(kFixHead + 4): with
// kFixHead = 31 that leaves 63 - 35 = 28 bits for the sum (2.7e8 maximal records per row) and 31 significant bits below
// the largest gradient -- seven more than the fp32 mantissa the atomic path accumulates with.
// A non-finite gradient anywhere in the step makes the exponent field 255: the reducer then writes NaN into every row sum
// (the atomic path would have poisoned the touched rows; the divergence stays visible instead of turning into garbage).
constexpr int kFixHead = 31;
__device__ __forceinline__ bool fixed_nonfinite(uint32_t gmax_bits) { return ((gmax_bits >> 23) & 0xffu) == 0xffu; }
__device__ __forceinline__ int fixed_shift(uint32_t gmax_bits) {
    const int e = (int)((gmax_bits >> 23) & 0xffu);          // biased exponent of gmax; 0 -> all gradients are zero
    return (e == 0 || e == 255) ? 0 : kFixHead - (e - 127) - 1;
}
// fp32 -> 64-bit fixed point round(v * 2^shift), branch-free: the product is exact in double, and adding 1.5 * 2^52
// leaves the (two's complement) integer in the low mantissa bits for |v * 2^shift| < 2^51 (here < 2^36 per record).
__device__ __forceinline__ long long to_fixed(float v, double scale) {
    const double d = (double)v * scale + 6755399441055744.0;
    return __double_as_longlong(d) - 0x4338000000000000ll;
}

// ---- row <-> (bucket, local row) -------------------------------------------------------------------------------------
// (round 4: the chunked map below replaced "bucket = bits [6, 6 + log2 NB)" in these kernels too.  With 64-row chunks on EVERY level
// 1.6 % of the x pairs were unpaired -- and ALL pairs of a ray that runs parallel to an axis with its x cell at 63 mod 64: on such
// projections single tiles of the fp32 parity mode (512-point tiles = 2.7 rays) filled their blocks and spilled records to float
// atomics, whose order is not fixed -- two runs of the same fp32 training differed from the step that drew projection 25 of 50 on,
// tools/determinism_probe.py.)
// rows a bucket of a level with T rows owns: [bucket << sh, (bucket + 1) << sh)
__device__ __forceinline__ uint32_t bucket_shift(uint32_t T, uint32_t log2_nb) {
    const uint32_t bits = T > 1u ? 32u - (uint32_t)__builtin_clz(T - 1u) : 0u;      // ceil(log2 T); scalar, once per level
    return bits > log2_nb ? bits - log2_nb : 0u;
}

// ---- row <-> (bucket, local row) of a level -----------------------------------------------------------------------------
// A level's rows are dealt to the NB buckets in chunks of 2^s consecutive rows: bucket = (row >> s) & (NB - 1), local row = the
// remaining high bits : the low s bits.  s = sh (the top-bit form of the header: one chunk per bucket) on hashed and wrapped-dense
// levels; the small dense levels (T_l = (R + 1)^3 < 2^log2T, never a power of two: top bits would leave up to half of the buckets
// without rows and send a ray's cells to two or three of the rest) keep 64-row chunks -- their records are merged singles anyway.
struct RowMap {
    uint32_t s, smask, hs;            // chunk shift, 2^s - 1, s + log2 NB
    __device__ __forceinline__ uint32_t bucket(uint32_t row, uint32_t nb_mask) const { return (row >> s) & nb_mask; }
    __device__ __forceinline__ uint32_t local(uint32_t row) const { return (row & smask) | ((row >> hs) << s); }
    __device__ __forceinline__ uint32_t row(uint32_t bucket, uint32_t local) const { return ((local >> s) << hs) | (bucket << s) | (local & smask); }
};
__device__ __forceinline__ RowMap make_row_map(uint32_t mode, uint32_t T, uint32_t log2_nb) {
    const uint32_t sh = bucket_shift(T, log2_nb);
    RowMap m;
    m.s = max(1u, mode == kDenseNoMod ? min(sh, 6u) : sh);      // at least two rows per chunk: the reducers finish rows in pairs (16-byte quads)
    m.smask = (1u << m.s) - 1u;
    m.hs = m.s + log2_nb;
    return m;
}

// runs: [level slot][bucket][tile] = start | length << 16 (one coalesced load per 64 tiles in pass 2);
// records: [level slot][tile][slots], the tile's records sorted by bucket
__device__ __forceinline__ size_t run_index(const BinPlan &plan, uint32_t ly, uint32_t bucket, uint32_t tile) {
    return (((size_t)ly << plan.log2_nb) + bucket) * plan.n_tiles + tile;
}
__device__ __forceinline__ size_t block_index(const BinPlan &plan, uint32_t ly, uint32_t tile) {
    return ((size_t)ly * plan.n_tiles + tile) * plan.slots;
}

// ---- pass 1 ---------------------------------------------------------------------------------------------------
// LDS: hist[NB] | cursor[NB] | total | staging[slots] records.  One workgroup = one tile of NT x PTS points x LV consecutive
// levels: the sample positions are evaluated once, and the stores of one level drain while the next level is being computed.
// Per level: (A) every thread builds the records of its points in registers and counts them per bucket (ds_add_u32),
// (B) wave 0 turns the histogram into exclusive offsets and publishes (start, length) of every run, (C) every thread takes
// its slots (ds_add_rtn_u32 on the cursors) and writes its records, (D) the dense, bucket-sorted block leaves for HBM.
template <typename FT, uint32_t C, typename Src, typename Rec, uint32_t NT, uint32_t PTS, uint32_t LV>
__global__ void __launch_bounds__(NT, NT >= 512u ? 4 : 2)       // 512 threads: two workgroups (16 waves) per CU -> <= 128 VGPRs; 1024: one
scatter_bin_kernel(Src src, const typename FT::store_t *__restrict__ grad, const int32_t *__restrict__ offsets,
                   float *__restrict__ grad_table, Rec *__restrict__ blocks, uint32_t *__restrict__ runs,
                   uint32_t *__restrict__ overflow, uint32_t B, uint32_t H, uint32_t level_base, uint32_t n_levels,
                   BinPlan plan, SlabReduce slab_job, uint32_t grad_stride_l, uint32_t grad_stride_b) {
    // the C gradients of (level l, point b) sit at grad + (l * grad_stride_l + b * grad_stride_b) * C: (B, 1) for [L, B, C], (1, L) for [B, L, C]
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // Spare workgroups behind the tiles (first launch of a training step): the reduction of the MLP backward's weight-gradient
    // slabs -- nothing in this kernel needs its results, the reducer that follows does (the gradient maximum), so it runs beside the
    // tiles instead of as a dependent launch of its own in front of them.
    if (blockIdx.x >= plan.n_tiles) {
        if (blockIdx.y == 0u && slab_job.slabs != nullptr)
            slab_reduce_block<NT / kReduceParams>(reinterpret_cast<float (*)[kReduceParams]>(smem), slab_job, blockIdx.x - plan.n_tiles, threadIdx.x);
        return;
    }
    constexpr uint32_t K = sizeof(Rec) / 4u;                    // dwords per record
    static_assert(PTS == 1u || PTS == 2u, "one or two points per thread");
    const uint32_t NB = 1u << plan.log2_nb, SLOTS = plan.slots;
    uint32_t *hist = reinterpret_cast<uint32_t *>(smem);
    uint32_t *cursor = hist + NB;
    uint32_t *total_p = cursor + NB;                             // 4 dwords (keeps the staging block 16-byte aligned)
    uint32_t *staging = total_p + 4;                             // [SLOTS][K]
    const uint32_t tile = blockIdx.x, lane = threadIdx.x & 63u, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (uint32_t i = threadIdx.x; i < NB; i += NT) hist[i] = 0u;

    // the thread's points.  Threads past the end of the batch take the last point with a zero gradient: their records
    // add nothing, and no validity test is needed further down.
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
    // this level's feature gradients, requested one level ahead (see the note at the barrier)
    RawVec<FT, C> graw[PTS];
    float g[PTS][C];
    if (blockIdx.y * LV < n_levels) {
#pragma unroll
        for (uint32_t q = 0; q < PTS; ++q)
            raw_load<FT, C>(grad + ((size_t)(level_base + blockIdx.y * LV) * grad_stride_l + (size_t)bp[q] * grad_stride_b) * C, graw[q]);
    }
#pragma unroll
    for (uint32_t q = 0; q < PTS; ++q) raw_unpack<FT, C>(graw[q], g[q]);
    __syncthreads();
    uint32_t n_overflow = 0, n_overflow_level = 0;               // records that did not fit (whole kernel / current level)

    constexpr uint32_t kNoRow = 0xffffffffu;
    for (uint32_t it = 0; it < LV; ++it) {
        const uint32_t ly = blockIdx.y * LV + it;
        if (ly >= n_levels) break;                                   // uniform
        const uint32_t level = level_base + ly;
        const LevelMeta m = make_level_meta<3>(offsets, level, H);
        float *__restrict__ gg = grad_table + (size_t)m.offset * C;
        // wave-uniform, batch-wide.  The merge key below packs two cell coordinates into 16 bits each: an invariant of the level
        // (fewer than 2^16 cells per axis), not of ray 0's spacing -- a degenerate first ray (spacing 0) must not switch merging on
        // for the fine levels, whose keys would alias.
        const bool merging = m.scale * spacing < 0.75f && m.scale < 65535.0f;
        const RowMap map = make_row_map(m.mode, m.size, plan.log2_nb);
        const uint32_t nb_mask = NB - 1u;

        // ---- A: records of the thread's points (registers), histogram -------------------------------------------------
        Rec rec[PTS][4];                                             // the four x-neighbour pairs of a cell
        uint32_t bkt[PTS][4], lone[PTS][4];                          // lone: row of a second corner that travels alone, or kNoRow
        bool on[PTS];
#pragma unroll
        for (uint32_t q = 0; q < PTS; ++q) {
            uint32_t row[8];
            float val[8][C];
            float frac[3];
            uint32_t pg[3];
            locate<3>(x[q], m.scale, frac, pg);
            if (!valid[q]) {
#pragma unroll
                for (uint32_t ch = 0; ch < C; ++ch) g[q][ch] = 0.0f;
            }
            dispatch_mode<Src::kInRange>(m.mode, [&](auto mode_tag) {
                constexpr uint32_t MODE = decltype(mode_tag)::value;
                float w[8];
                cell_corners<MODE, 3>(m, frac, pg, w, row);
#pragma unroll
                for (uint32_t c = 0; c < 8; ++c)
#pragma unroll
                    for (uint32_t ch = 0; ch < C; ++ch) val[c][ch] = w[c] * g[q][ch];
            });
            // padding threads of the last tile (clones of the last point, zero gradient) emit nothing: if that point sits on a
            // face of the volume its x pairs are unpaired on every level from 2 on, and 8 records per clone would fill the
            // block and push the tile's real records out to atomics
            on[q] = valid[q];
            bool keyed = valid[q];                                       // takes part in the cell runs of a merged level
            if constexpr (!Src::kInRange) {
                // Coordinates nobody range-checked (the stand-alone operator): outside [0, 1] the reference's weights are whatever its
                // arithmetic gives (|w| > 1, NaN), which the fixed-point reducer -- scaled for |w g| <= max |g| -- cannot take.  Such a
                // point goes to the table the reference's way, with atomics (hashencoder.cu:266-269), and emits no record.
                bool inside = true;
#pragma unroll
                for (uint32_t d = 0; d < 3; ++d) inside = inside && x[q][d] >= 0.0f && x[q][d] <= 1.0f;
                if (valid[q] && !inside) {
#pragma unroll
                    for (uint32_t c = 0; c < 8; ++c)
#pragma unroll
                        for (uint32_t ch = 0; ch < C; ++ch) atomicAdd(gg + (size_t)row[c] * C + ch, val[c][ch]);
                }
                on[q] = keyed = valid[q] && inside;
#pragma unroll
                for (uint32_t c = 0; c < 8; ++c)
#pragma unroll
                    for (uint32_t ch = 0; ch < C; ++ch) val[c][ch] = keyed ? val[c][ch] : 0.0f;
            }
            if (merging) {
                // Merge each run of equal cells with a segmented inclusive scan; only the last lane of a run emits.  Runs
                // are cut at 16-lane rows: the scan then moves its operands with DPP row shifts (a modifier of the VALU
                // instruction) instead of 6 x 16 trips through the LDS crossbar (ds_bpermute).
                // (padding lanes get a cell key of their own -- bit 31 of c_hi -- so that they never share a run with real points)
                const uint32_t c_lo = pg[0] | (pg[1] << 16), c_hi = pg[2] | (keyed ? 0u : 0x80000000u);    // merging levels have < 2^16 cells per axis
                const uint32_t p_lo = dpp_row_shr<1>(c_lo), p_hi = dpp_row_shr<1>(c_hi);
                const uint64_t heads = __ballot((lane & 15u) == 0u || c_lo != p_lo || c_hi != p_hi);
                const uint32_t start = 63u - (uint32_t)__clzll(heads & (~0ull >> (63u - lane)));
                const uint32_t len = lane - start;                               // my position inside the run (same row)
                auto fold = [&](auto shift_tag) {
                    constexpr uint32_t d = decltype(shift_tag)::value;
                    const float take = len >= d ? 1.0f : 0.0f;          // val += shifted * take: the product with 0 / 1 is exact
#pragma unroll
                    for (uint32_t c = 0; c < 8; ++c)
#pragma unroll
                        for (uint32_t ch = 0; ch < C; ++ch) val[c][ch] = __builtin_fmaf(dpp_row_shr<d>(val[c][ch]), take, val[c][ch]);
                };
                fold(std::integral_constant<uint32_t, 1>{});
                fold(std::integral_constant<uint32_t, 2>{});
                fold(std::integral_constant<uint32_t, 4>{});
                fold(std::integral_constant<uint32_t, 8>{});
                on[q] = keyed && (lane == 63u || ((heads >> (lane + 1u)) & 1ull));
            }
            // x-neighbour corners (2k, 2k+1) travel together when their rows share a 64-row block; otherwise (1.6 % of the
            // pairs) the record keeps both value halves with mask 0 and phase C splits it into two single records
#pragma unroll
            for (uint32_t k = 0; k < 4; ++k) {
                const uint32_t ra = row[2 * k], rb = row[2 * k + 1], mask = ra ^ rb;
                const bool paired = mask <= map.smask;                      // both rows in one chunk of the bucket
                rec[q][k].set(map.local(ra) | (paired ? mask << 16 : 0u), val[2 * k], val[2 * k + 1]);
                bkt[q][k] = map.bucket(ra, nb_mask);
                lone[q][k] = paired ? kNoRow : rb;
                if (on[q]) {
                    atomicAdd(&hist[bkt[q][k]], 1u);
                    if (!paired) atomicAdd(&hist[map.bucket(rb, nb_mask)], 1u);
                }
            }
        }
        // The next level's gradients are requested here and CONSUMED right after the barrier, before this level's stores are
        // issued: vmcnt retires in order and the compiler can only wait for "everything", so any wait placed after the stores
        // would sit out their whole round trip.  This way the stores drain behind the next level's arithmetic.
        if (it + 1u < LV && ly + 1u < n_levels) {
#pragma unroll
            for (uint32_t q = 0; q < PTS; ++q) raw_load<FT, C>(grad + ((size_t)(level + 1u) * grad_stride_l + (size_t)bp[q] * grad_stride_b) * C, graw[q]);
        }
        lds_barrier();
#pragma unroll
        for (uint32_t q = 0; q < PTS; ++q) {
            raw_unpack<FT, C>(graw[q], g[q]);
#pragma unroll
            for (uint32_t ch = 0; ch < C; ++ch) asm volatile("" : "+v"(g[q][ch]) : : "memory");      // pin the wait here
        }

        // ---- B: histogram -> exclusive offsets (wave 0: NB / 64 consecutive buckets per lane), run words, total --------
        if (wave == 0u) {
            const uint32_t per = NB >> 6;                                // 1 for 64 buckets, 8 for 512
            uint32_t mine = 0u;
            for (uint32_t j = 0; j < per; ++j) mine += hist[lane * per + j];
            uint32_t incl = mine;
#pragma unroll
            for (uint32_t d = 1; d < 64u; d <<= 1) {
                const uint32_t up = (uint32_t)__shfl_up((int)incl, d, 64);
                if (lane >= d) incl += up;
            }
            uint32_t run_start = incl - mine;
            for (uint32_t j = 0; j < per; ++j) {
                const uint32_t b = lane * per + j, n = hist[b];
                cursor[b] = run_start;
                hist[b] = 0u;                                            // ready for the next level
                const uint32_t start_c = min(run_start, SLOTS), n_c = min(n, SLOTS - start_c);      // what fits the block
                runs[run_index(plan, ly, b, tile)] = start_c | (n_c << 16);
                run_start += n;
            }
            if (lane == 63u) total_p[0] = incl;
        }
        lds_barrier();

        // ---- C: placement -------------------------------------------------------------------------------------------------
        auto place = [&](uint32_t b, const uint32_t (&words)[K], uint32_t row_a, uint32_t row_b, const Rec &r) __attribute__((always_inline)) {
            const uint32_t pos = atomicAdd(&cursor[b], 1u);
            if (pos < SLOTS) {
                uint32_t *dst = staging + (size_t)pos * K;
#pragma unroll
                for (uint32_t i = 0; i < K; ++i) dst[i] = words[i];
            } else {                                                     // the tile's block is full (see the header): straight to the table
                // (values are read out BEFORE the row test: with fp32 records the compiler otherwise selects the record WORD by the
                // test -- a dynamically indexed private array, which sent all of `rec` to scratch memory: 176 bytes per lane in the
                // fp32-mode kernel for the sake of a branch that is almost never taken)
                float va[C], vb[C];
#pragma unroll
                for (uint32_t ch = 0; ch < C; ++ch) { va[ch] = r.value(0, ch); vb[ch] = r.value(1, ch); }
#pragma unroll
                for (uint32_t ch = 0; ch < C; ++ch) {
                    atomicAdd(gg + (size_t)row_a * C + ch, va[ch]);
                    if (row_b != row_a) atomicAdd(gg + (size_t)row_b * C + ch, vb[ch]);
                }
                ++n_overflow;
                ++n_overflow_level;
            }
        };
#pragma unroll
        for (uint32_t q = 0; q < PTS; ++q) {
            if (!on[q]) continue;
#pragma unroll
            for (uint32_t k = 0; k < 4; ++k) {
                const Rec &r = rec[q][k];
                const uint32_t la = r.w[0] & 0xffffu;
                const uint32_t row_a = map.row(bkt[q][k], la);
                if (lone[q][k] == kNoRow) {
                    place(bkt[q][k], r.w, row_a, map.row(bkt[q][k], la ^ (r.w[0] >> 16)), r);
                } else {                                                  // two single records: first corner, second corner
                    Rec a, b2;
                    float va[C], vb[C], zero[C];
#pragma unroll
                    for (uint32_t ch = 0; ch < C; ++ch) { va[ch] = r.value(0, ch); vb[ch] = r.value(1, ch); zero[ch] = 0.0f; }
                    a.set(la, va, zero);
                    b2.set(map.local(lone[q][k]), vb, zero);
                    place(bkt[q][k], a.w, row_a, row_a, a);
                    place(map.bucket(lone[q][k], nb_mask), b2.w, lone[q][k], lone[q][k], b2);
                }
            }
        }
        lds_barrier();

        // ---- D: the dense block leaves as whole 128-byte lines (16 bytes per lane; stale slots past the end are harmless) -----
        {
            const uint32_t n_rec = min(total_p[0], SLOTS);
            const uint32_t n_chunk = min(((n_rec * (uint32_t)sizeof(Rec) + 127u) >> 7) << 3, (SLOTS * (uint32_t)sizeof(Rec)) >> 4);
            uint4 *__restrict__ dst = reinterpret_cast<uint4 *>(blocks + block_index(plan, ly, tile));
            const uint4 *from = reinterpret_cast<const uint4 *>(staging);
            // Non-temporal stores: the records are written once and read once by pass 2, and at the reference's batch they (151 MB)
            // and the optimiser state pass 2 streams (171 MB of fp32 master + moments, every step) do not both fit the 256 MB
            // Infinity Cache -- kept out of it, they leave the state there.  Same-box A/B, 1 024 rays: step 0.311 -> 0.301 ms (pass 1
            // 0.056 -> 0.053, pass 2 0.114 -> 0.109); flat from 16 384 rays on.  (The same hint on the feature-gradient and feature
            // stores helps their writers at large batches and costs their readers more at this one, and non-temporal record LOADS in pass 2
            // cost it 2-4 %: not used.)
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            for (uint32_t c = threadIdx.x; c < n_chunk; c += NT) __builtin_nontemporal_store(reinterpret_cast<const u32x4 *>(from)[c], reinterpret_cast<u32x4 *>(dst) + c);
        }
        if (__ballot(n_overflow_level != 0u)) {                          // diagnostics: overflow per level (overflow[1 + level])
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) n_overflow_level += __shfl_xor(n_overflow_level, off, 64);
            if (lane == 0u) atomicAdd(overflow + 1u + level, n_overflow_level);
            n_overflow_level = 0u;
        }
        // no barrier here: the next level touches only `hist` (cleared in B) before its own first barrier, and writes the
        // staging block after its second one -- every wave has finished this copy before it arrives at the first.
    }
    if (__ballot(n_overflow != 0u)) {                                // statistics only: one atomic per wave that overflowed
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) n_overflow += __shfl_xor(n_overflow, off, 64);
        if (lane == 0u) atomicAdd(overflow, n_overflow);
    }
}

// ---- pass 2 ---------------------------------------------------------------------------------------------------
// Accumulators are 64-bit FIXED POINT updated with ds_add_u64: integer LDS atomics run at 4.7 lane-ops/clk/CU on
// gfx950 against 2.46 for ds_add_f64 and 0.33 for ds_add_f32 (tools/lds_atomic_bench.hip).  Integer addition is
// associative, so the reduction is bit-reproducible from run to run.  The scale follows the largest feature gradient of
// the step (fixed_shift): 31 significant bits below it, 28 bits of headroom above for the sum of merged records.
// kAdam (naf_render_train_adam, single-GPU steps): the workgroup is the sole owner of its rows and has their finished sums in
// LDS, so instead of writing the gradient out for a separate Adam pass to read back and clear it applies the update itself --
// the 57 MB gradient table is then neither written, re-read nor zeroed (only rows that pass 1 reached with atomics are).
template <uint32_t C, typename Rec, bool kAdam = false>
__global__ void __launch_bounds__(1024)
scatter_reduce_kernel(const Rec *__restrict__ blocks, const uint32_t *__restrict__ runs, const int32_t *__restrict__ offsets,
                      float *__restrict__ grad_table, const uint32_t *__restrict__ gmax_bits, uint32_t level_base,
                      uint32_t ly_begin, uint32_t H, BinPlan plan, AdamTail adam) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t gbits = *gmax_bits;
    const int shift = fixed_shift(gbits);
    const bool poison = fixed_nonfinite(gbits);
    const double scale = ldexp(1.0, shift);
    unsigned long long *acc = reinterpret_cast<unsigned long long *>(smem);
    const uint32_t T_ = blockDim.x;
    const uint32_t bucket = blockIdx.x, ly = ly_begin + blockIdx.y, level = level_base + ly;      // ly: level slot of the bin pass
    const LevelMeta lm = make_level_meta<3>(offsets, level, H);
    const uint32_t off = lm.offset, T = lm.size;
    const RowMap map = make_row_map(lm.mode, T, plan.log2_nb);
    if ((bucket << map.s) >= T) return;                      // a level with fewer chunks than buckets: no rows, no records
    // local rows this bucket may own: whole chunks of 2^s rows (row < T is checked per row)
    const uint32_t rows_local = ((T + ((1u << map.hs) - 1u)) >> map.hs) << map.s;
    // accumulators are channel-major, acc[ch][local row]: the two 8-byte cells of a row would otherwise sit 8 bytes apart and
    // one ds_add_u64 instruction (one channel of 64 rows) could reach only every other bank pair
    const uint32_t pitch = plan.max_local_rows;
    // kAdam: the parameters and moments of this workgroup's rows are requested NOW, before the records are streamed.  The update at
    // the end is pure streaming (26 bytes per table element, the fixed cost of a step whatever its batch), the record phase in
    // between is latency-bound at small batches: issued here, the two overlap instead of following each other (1 024-ray step:
    // 0.131 -> see DESIGN.md 4.2).  rows_local * C <= 16 * 1024 by the LDS bound of the plan (make_bin_plan); whatever lies beyond is
    // loaded in the tail as before.
    // C == 2 (every shipped configuration): a thread owns QUADS of table elements -- two consecutive rows x two channels = 16
    // contiguous bytes of param / m / v (rows 64 k .. 64 k + 63 of a bucket are consecutive table rows, see row_of) -- so the tail
    // moves 16 bytes per lane and instruction like the stand-alone adam_kernel (6.2 TB/s) instead of 4 (the reducer with its
    // element-wise tail: 522 MB in 0.137 ms = 3.8 TB/s at 1 024 rays).  Level offsets may be odd, i.e. the quads are only 8-byte
    // aligned: gfx950 global accesses need dword alignment (Quad is declared with alignment 4, the compiler still emits dwordx4).
    struct __attribute__((packed, aligned(4))) Quad { float x, y, z, w; };
    constexpr uint32_t kPreQ = (kAdam && C == 2u) ? 4u : 0u;                 // quads prefetched per thread (16 elements)
    constexpr uint32_t kPre = (kAdam && C == 1u) ? 16u : 0u;                 // other channel counts: element-wise
    Quad preq_p[kPreQ ? kPreQ : 1u], preq_m[kPreQ ? kPreQ : 1u], preq_v[kPreQ ? kPreQ : 1u];
    float pre_p[kPre ? kPre : 1u], pre_m[kPre ? kPre : 1u], pre_v[kPre ? kPre : 1u];
    const uint32_t n_quads = rows_local * C / 4u;                            // rows_local is a multiple of 64
    // quad q holds local rows 2q, 2q + 1 (same 64-row block) = table rows row0, row0 + 1; a level with an odd row count (dense
    // levels) ends with a quad whose second row does not exist: that one row is finished element-wise by its thread in the tail
    auto quad_row = [&](uint32_t q) { return map.row(bucket, 2u * q); };
    if constexpr (kPreQ != 0u) {
        const float *__restrict__ pp = adam.param + (size_t)off * C;
        const float *__restrict__ pm = adam.m + (size_t)off * C;
        const float *__restrict__ pv = adam.v + (size_t)off * C;
#pragma unroll
        for (uint32_t k = 0; k < kPreQ; ++k) {
            const uint32_t q = threadIdx.x + k * T_;
            const uint32_t row0 = quad_row(q);
            preq_p[k] = preq_m[k] = preq_v[k] = Quad{0.0f, 0.0f, 0.0f, 0.0f};
            if (q < n_quads && row0 + 1u < T) {
                const size_t e = (size_t)row0 * C;
                preq_p[k] = *reinterpret_cast<const Quad *>(pp + e);
                preq_m[k] = *reinterpret_cast<const Quad *>(pm + e);
                preq_v[k] = *reinterpret_cast<const Quad *>(pv + e);
            }
        }
    }
    if constexpr (kPre != 0u) {
        const float *__restrict__ pp = adam.param + (size_t)off * C;
        const float *__restrict__ pm = adam.m + (size_t)off * C;
        const float *__restrict__ pv = adam.v + (size_t)off * C;
#pragma unroll
        for (uint32_t k = 0; k < kPre; ++k) {
            const uint32_t i = threadIdx.x + k * T_;
            const uint32_t local = i / C, ch = i - local * C;
            const uint32_t row = map.row(bucket, local);
            pre_p[k] = pre_m[k] = pre_v[k] = 0.0f;
            if (i < rows_local * C && row < T) {
                const size_t e = (size_t)row * C + ch;
                pre_p[k] = pp[e]; pre_m[k] = pm[e]; pre_v[k] = pv[e];
            }
        }
    }
    // the wave index is made scalar explicitly: tile ranges, run-word addresses and block bases then live in SGPRs
    const uint32_t lane = threadIdx.x & 63u, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), n_waves = T_ >> 6;
    const size_t run0 = run_index(plan, ly, bucket, 0);
    // every wave streams ONE contiguous range of tiles
    // gridDim.z > 1 (few levels per pass at very large batches): the tiles are split between gridDim.z workgroups
    const uint32_t split_tiles = (plan.n_tiles + gridDim.z - 1u) / gridDim.z;
    const uint32_t split_begin = blockIdx.z * split_tiles, split_end = min(plan.n_tiles, split_begin + split_tiles);
    // (in whole 64-tile blocks when there are that many; a 1 024-ray step has 192 tiles, and 64-tile blocks would leave 13 of
    // the 16 waves without work: 0.136 -> 0.1 ms)
    const uint32_t share = (split_tiles + n_waves - 1u) / n_waves;
    const uint32_t per_wave = share >= 64u ? (share + 63u) & ~63u : (share + 7u) & ~7u;
    const uint32_t t_begin = split_begin + wave * per_wave, t_end = min(split_end, t_begin + per_wave);
    // the run words of the wave's first 64 tiles are requested here, with the Adam operands and in front of the clear and its
    // barrier: the record phase then starts with ONE dependent round trip (run words -> records) less on its critical path
    const uint32_t first_runs = t_begin + lane < t_end ? runs[run0 + t_begin + lane] : 0u;
    for (uint32_t i = threadIdx.x; i < pitch * C; i += T_) acc[i] = 0ull;
    __syncthreads();

    auto add = [&](const Rec &r) {          // both halves unconditionally: a record without a second corner adds zeros to its own row
        const uint32_t la = r.w[0] & 0xffffu, lb = la ^ (r.w[0] >> 16);
#pragma unroll
        for (uint32_t ch = 0; ch < C; ++ch) {
            atomicAdd(&acc[ch * pitch + la], (unsigned long long)to_fixed(r.value(0, ch), scale));   // ds_add_u64
            atomicAdd(&acc[ch * pitch + lb], (unsigned long long)to_fixed(r.value(1, ch), scale));
        }
    };
    // (An integer route for this conversion -- v * 2^shift is exact in fp32, v_rndne + v_cvt_i32 give the same integer below 2^31 --
    // was measured with a per-wave fallback to fp64: the second code path costs 36 spilled registers in the Adam variants and the
    // kernel got slower at every batch size: 0.103 -> 0.145 ms at 1 024 rays, 2.04 -> 2.19 ms at 65 536.)
    // each wave owns blocks of 64 consecutive tiles: one coalesced load brings their run words.  Runs are read kGroup at a
    // time with straight-line code: kGroup loads for records 0..63 plus kGroup * kTail / 64 loads for the tails are in
    // flight per lane before the first LDS atomic (lanes past a run's end read its first record, a line that is fetched
    // anyway -> no extra traffic); whatever a run holds beyond 64 + kTail records follows in a plain loop.
    constexpr uint32_t kGroup = sizeof(Rec) <= 12 ? 8u : 4u, kTail = 32;
    if (plan.log2_w < 6u) {
        // Many buckets (T >= 2^20: 128 .. 512 per level) make the runs short -- 8 to 32 records: a wave takes G = 64 / W runs per
        // instruction, W lanes each, four instructions' worth of loads in flight; whatever a run holds beyond W follows in a loop.
        const uint32_t W = 1u << plan.log2_w, G = 64u >> plan.log2_w, g = lane >> plan.log2_w, j = lane & (W - 1u);
        constexpr uint32_t kSteps = 4;
        for (uint32_t t0 = t_begin; t0 < t_end; t0 += 64u) {
            const uint32_t mine = t0 == t_begin ? first_runs : (t0 + lane < t_end ? runs[run0 + t0 + lane] : 0u);
            const uint32_t n_here = min(64u, t_end - t0);
            for (uint32_t s0 = 0; s0 < n_here; s0 += kSteps * G) {
                uint32_t n4[kSteps];
                const Rec *b4[kSteps];
                Rec r4[kSteps];
#pragma unroll
                for (uint32_t u = 0; u < kSteps; ++u) {
                    const uint32_t tl = s0 + u * G + g;
                    const uint32_t word = (uint32_t)__shfl((int)mine, (int)min(tl, 63u), 64);
                    n4[u] = tl < n_here ? word >> 16 : 0u;
                    b4[u] = blocks + block_index(plan, ly, t0 + min(tl, n_here - 1u)) + (word & 0xffffu);
                    r4[u] = b4[u][j < n4[u] ? j : 0u];
                }
#pragma unroll
                for (uint32_t u = 0; u < kSteps; ++u)
                    if (j < n4[u]) add(r4[u]);
#pragma unroll 1
                for (uint32_t u = 0; u < kSteps; ++u)
                    for (uint32_t i = W + j; __ballot(i < n4[u]) != 0ull; i += W)
                        if (i < n4[u]) add(b4[u][i]);
            }
        }
    } else
    for (uint32_t t0 = t_begin; t0 < t_end; t0 += 64u) {
        const uint32_t mine = t0 == t_begin ? first_runs : (t0 + lane < t_end ? runs[run0 + t0 + lane] : 0u);
        const uint32_t n_here = min(64u, t_end - t0);
        for (uint32_t j = 0; j < n_here; j += kGroup) {
            uint32_t n[kGroup], n_max = 0u;
            const Rec *base[kGroup];
            Rec ra[kGroup];
#pragma unroll
            for (uint32_t u = 0; u < kGroup; ++u) {
                const uint32_t tj = min(j + u, n_here - 1u);
                const uint32_t word = (uint32_t)__builtin_amdgcn_readlane((int)mine, (int)tj);              // scalar (tj is wave-uniform)
                n[u] = j + u < n_here ? word >> 16 : 0u;
                n_max = max(n_max, n[u]);
                base[u] = blocks + block_index(plan, ly, t0 + tj) + (word & 0xffffu);
                ra[u] = base[u][lane < n[u] ? lane : 0u];
            }
            {
                // The tails (records 64 .. 64 + kTail) of 64 / kTail runs share one load and one set of LDS atomics, kTail lanes
                // per run, instead of a nearly empty wave each.
                constexpr uint32_t kPer = 64u / kTail;                      // runs per tail instruction
                Rec rt[kGroup / kPer];
                uint32_t nt[kGroup / kPer];
#pragma unroll
                for (uint32_t q = 0; q < kGroup / kPer; ++q) {
                    const uint32_t sub = lane / kTail, slot = 64u + (lane % kTail);
                    nt[q] = n[q * kPer];                                    // select among the kPer scalar run lengths / bases
                    const Rec *bq = base[q * kPer];
#pragma unroll
                    for (uint32_t k = 1; k < kPer; ++k) {
                        nt[q] = sub == k ? n[q * kPer + k] : nt[q];
                        bq = sub == k ? base[q * kPer + k] : bq;
                    }
                    rt[q] = bq[slot < nt[q] ? slot : 0u];
                }
#pragma unroll
                for (uint32_t u = 0; u < kGroup; ++u)
                    if (lane < n[u]) add(ra[u]);
#pragma unroll
                for (uint32_t q = 0; q < kGroup / kPer; ++q)
                    if (64u + (lane % kTail) < nt[q]) add(rt[q]);
            }
            if (n_max > 64u + kTail) {                                      // long runs (clustered tiles): the rest, run by run
#pragma unroll 1
                for (uint32_t u = 0; u < kGroup; ++u)
                    for (uint32_t i = 64u + kTail + lane; i < n[u]; i += 64u) add(base[u][i]);
            }
        }
    }
    __syncthreads();
    float *__restrict__ gg = grad_table + (size_t)off * C;
    const float nan = __builtin_nanf("");
    if constexpr (kAdam) {
        // Same traversal as below (64 consecutive rows x C per step, coalesced on every array).  A level whose tiles overflowed
        // their blocks has contributions in the gradient table already (atomics of pass 1): they are added and cleared here.
        const bool spilled = adam.overflow[1u + level] != 0u;
        float *__restrict__ pp = adam.param + (size_t)off * C;
        float *__restrict__ pm = adam.m + (size_t)off * C;
        float *__restrict__ pv = adam.v + (size_t)off * C;
        auto update = [&](uint32_t local, uint32_t ch, uint32_t row, float p, float m, float v) {
            const size_t e = (size_t)row * C + ch;
            float g = poison ? nan : (float)ldexp((double)(long long)acc[ch * pitch + local], -shift);
            if (spilled) {
                const float extra = gg[e];
                if (extra != 0.0f) { g = extra + g; gg[e] = 0.0f; }          // the order of the separate route: table += sum
            }
            if (adam.lp != nullptr) adam_one<true>(p, m, v, g, adam.a);          // the form adam_kernel uses for tables with a 16-bit shadow
            else adam_one<false>(p, m, v, g, adam.a);
            pp[e] = p; pm[e] = m; pv[e] = v;
            if (adam.lp != nullptr) {
                const size_t el = (size_t)off * C + e;
                if (adam.lp_dtype == kAdamLpF16) reinterpret_cast<_Float16 *>(adam.lp)[el] = (_Float16)p;
                else reinterpret_cast<uint16_t *>(adam.lp)[el] = f32_to_bf16(p);
            }
        };
        if constexpr (kPreQ != 0u) {
            // four elements at a time: e0 = (row0, ch 0), e1 = (row0, ch 1), e2 = (row0 + 1, ch 0), e3 = (row0 + 1, ch 1)
            auto update4 = [&](uint32_t q, uint32_t row0, Quad p4, Quad m4, Quad v4) {
                const uint32_t local0 = 2u * q;
                const size_t e = (size_t)row0 * C;
                const unsigned long long *a0 = &acc[0u * pitch + local0], *a1 = &acc[1u * pitch + local0];
                const long long x[4] = {(long long)a0[0], (long long)a1[0], (long long)a0[1], (long long)a1[1]};
                // fixed point -> fp32.  A sum that fits 32 bits (every row whose gradient is below about twice the step's largest
                // single contribution: nearly all of them) converts with v_cvt_f32_i32 + v_ldexp_f32 -- one rounding of the same
                // exact value, hence the same bits as the fp64 route below, which costs six half-rate instructions per element.
                // The choice is made per wave so that the common case runs without the fp64 code at all.
                bool small = true;
#pragma unroll
                for (uint32_t j = 0; j < 4; ++j) small = small && (x[j] == (long long)(int)x[j]);
                float g[4];
                if (__ballot(!small) == 0ull && shift < 120) {
#pragma unroll
                    for (uint32_t j = 0; j < 4; ++j) g[j] = ldexpf((float)(int)x[j], -shift);
                } else {
#pragma unroll
                    for (uint32_t j = 0; j < 4; ++j) g[j] = (float)ldexp((double)x[j], -shift);
                }
                if (poison) g[0] = g[1] = g[2] = g[3] = nan;
                if (spilled) {
                    const Quad extra = *reinterpret_cast<const Quad *>(gg + e);
                    const float x[4] = {extra.x, extra.y, extra.z, extra.w};
                    bool any = false;
#pragma unroll
                    for (uint32_t j = 0; j < 4; ++j)
                        if (x[j] != 0.0f) { g[j] = x[j] + g[j]; any = true; }          // the order of the separate route: table += sum
                    if (any) *reinterpret_cast<Quad *>(gg + e) = Quad{0.0f, 0.0f, 0.0f, 0.0f};
                }
                float p[4] = {p4.x, p4.y, p4.z, p4.w}, m[4] = {m4.x, m4.y, m4.z, m4.w}, v[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
                for (uint32_t j = 0; j < 4; ++j) {
                    if (adam.lp != nullptr) adam_one<true>(p[j], m[j], v[j], g[j], adam.a);
                    else adam_one<false>(p[j], m[j], v[j], g[j], adam.a);
                }
                *reinterpret_cast<Quad *>(pp + e) = Quad{p[0], p[1], p[2], p[3]};
                *reinterpret_cast<Quad *>(pm + e) = Quad{m[0], m[1], m[2], m[3]};
                *reinterpret_cast<Quad *>(pv + e) = Quad{v[0], v[1], v[2], v[3]};
                if (adam.lp != nullptr) {
                    struct __attribute__((packed, aligned(4))) Half4 { uint32_t lo, hi; };
                    const size_t el = (size_t)off * C + e;
                    Half4 h;
                    if (adam.lp_dtype == kAdamLpF16) {
                        const _Float16 h0 = (_Float16)p[0], h1 = (_Float16)p[1], h2 = (_Float16)p[2], h3 = (_Float16)p[3];
                        h.lo = (uint32_t)__builtin_bit_cast(uint16_t, h0) | ((uint32_t)__builtin_bit_cast(uint16_t, h1) << 16);
                        h.hi = (uint32_t)__builtin_bit_cast(uint16_t, h2) | ((uint32_t)__builtin_bit_cast(uint16_t, h3) << 16);
                    } else {
                        h.lo = (uint32_t)f32_to_bf16(p[0]) | ((uint32_t)f32_to_bf16(p[1]) << 16);
                        h.hi = (uint32_t)f32_to_bf16(p[2]) | ((uint32_t)f32_to_bf16(p[3]) << 16);
                    }
                    *reinterpret_cast<Half4 *>(reinterpret_cast<uint16_t *>(adam.lp) + el) = h;
                }
            };
            auto finish = [&](uint32_t q, bool prefetched, const Quad &p4, const Quad &m4, const Quad &v4) {
                if (q >= n_quads) return;
                const uint32_t row0 = quad_row(q);
                if (row0 + 1u < T) {
                    if (prefetched) update4(q, row0, p4, m4, v4);
                    else {
                        const size_t e = (size_t)row0 * C;
                        update4(q, row0, *reinterpret_cast<const Quad *>(pp + e), *reinterpret_cast<const Quad *>(pm + e),
                                *reinterpret_cast<const Quad *>(pv + e));
                    }
                } else if (row0 < T) {                       // the level's last row when its row count is odd
#pragma unroll
                    for (uint32_t ch = 0; ch < C; ++ch) {
                        const size_t e = (size_t)row0 * C + ch;
                        update(2u * q, ch, row0, pp[e], pm[e], pv[e]);
                    }
                }
            };
#pragma unroll
            for (uint32_t k = 0; k < kPreQ; ++k) finish(threadIdx.x + k * T_, true, preq_p[k], preq_m[k], preq_v[k]);
            const Quad none{0.0f, 0.0f, 0.0f, 0.0f};
            for (uint32_t q = threadIdx.x + kPreQ * T_; q < n_quads; q += T_) finish(q, false, none, none, none);
        } else {
        if constexpr (kPre != 0u) {
#pragma unroll
            for (uint32_t k = 0; k < kPre; ++k) {
                const uint32_t i = threadIdx.x + k * T_;
                const uint32_t local = i / C, ch = i - local * C;
                const uint32_t row = map.row(bucket, local);
                if (i < rows_local * C && row < T) update(local, ch, row, pre_p[k], pre_m[k], pre_v[k]);
            }
        }
#pragma unroll 4
        for (uint32_t i = threadIdx.x + kPre * T_; i < rows_local * C; i += T_) {
            const uint32_t local = i / C, ch = i - local * C;
            const uint32_t row = map.row(bucket, local);
            if (row < T) {
                const size_t e = (size_t)row * C + ch;
                update(local, ch, row, pp[e], pm[e], pv[e]);
            }
        }
        }
    } else if (gridDim.z == 1u) {
        // sole owner, and the bucket's local rows 64 k .. 64 k + 63 are 64 consecutive table rows: add the sums in place,
        // coalesced (64 x C floats per block)
        if constexpr (C == 2u) {                             // 16 bytes per lane, like the Adam tail above
            for (uint32_t q = threadIdx.x; q < n_quads; q += T_) {
                const uint32_t row0 = quad_row(q), local0 = 2u * q;
                const unsigned long long *a0 = &acc[0u * pitch + local0], *a1 = &acc[1u * pitch + local0];
                if (row0 + 1u < T) {
                    Quad *dst = reinterpret_cast<Quad *>(gg + (size_t)row0 * C);
                    const Quad old = *dst;
                    const long long x[4] = {(long long)a0[0], (long long)a1[0], (long long)a0[1], (long long)a1[1]};
                    bool small = true;                       // 32-bit sums: the same bits without fp64 (see the Adam tail)
#pragma unroll
                    for (uint32_t j = 0; j < 4; ++j) small = small && (x[j] == (long long)(int)x[j]);
                    float g[4];
                    if (__ballot(!small) == 0ull && shift < 120) {
#pragma unroll
                        for (uint32_t j = 0; j < 4; ++j) g[j] = ldexpf((float)(int)x[j], -shift);
                    } else {
#pragma unroll
                        for (uint32_t j = 0; j < 4; ++j) g[j] = (float)ldexp((double)x[j], -shift);
                    }
                    if (poison) g[0] = g[1] = g[2] = g[3] = nan;
                    *dst = Quad{old.x + g[0], old.y + g[1], old.z + g[2], old.w + g[3]};
                } else if (row0 < T) {
                    gg[(size_t)row0 * C] += poison ? nan : (float)ldexp((double)(long long)a0[0], -shift);
                    gg[(size_t)row0 * C + 1u] += poison ? nan : (float)ldexp((double)(long long)a1[0], -shift);
                }
            }
        } else
        for (uint32_t i = threadIdx.x; i < rows_local * C; i += T_) {
            const uint32_t local = i / C, ch = i - local * C;
            const uint32_t row = map.row(bucket, local);
            if (row < T) gg[(size_t)row * C + ch] += poison ? nan : (float)ldexp((double)(long long)acc[ch * pitch + local], -shift);
        }
    } else {
        for (uint32_t i = threadIdx.x; i < rows_local * C; i += T_) {
            const uint32_t local = i / C, ch = i - local * C;
            const uint32_t row = map.row(bucket, local);
            if (row < T)
                atomicAdd(gg + (size_t)row * C + ch,
                          poison ? nan : (float)ldexp((double)(long long)acc[ch * pitch + local], -shift));   // one add per row and split
        }
    }
}

}  // namespace naf
