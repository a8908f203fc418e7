// scatter_binned.h -- hash-table gradient scatter without per-contribution global atomics (gfx950).
//
// Why: on MI355X a global float atomic is executed at the memory side, one 64-byte request per touched line; scattered
// 4-byte adds top out at ~2e10 requests/s chip-wide (MI355X_MICROARCH.md "Global float atomics"; measured here:
// 2.4 ms per level for 25 M adds, 12 ms on level 0 where 4913 rows take all of them).  The reference design
// (hashencoder.cu:257-269, one atomicAdd per corner and channel) is ~25x slower than the rest of the training step.
// A level of the table gets ~50-200 contributions per row per step, but they are hash-scattered, so no locality trick
// removes them -- they have to be ROUTED to an owner instead (a one-digit radix multisplit on the row index):
//
//   pass 1  bin      one workgroup = one tile of 512 points x LV consecutive levels (all 16, or 4 when tiles are scarce).  Per level every thread takes
//                    an LDS slot for each of its 8 contributions in the bucket's staging run (bucket = row & (NB-1), one
//                    returning LDS atomic per record) and writes the record (row >> log2 NB, w*g[0..C)) there; the runs are
//                    then copied to fixed-size global REGIONS [level][tile][bucket][slot_cap] with 16-byte stores that
//                    cover whole 128-byte lines, and the run lengths go to counts[level][bucket][tile].  No global
//                    atomics, no prefix sums between workgroups; a run that outgrows its region (~3e-5 of them) falls
//                    back to global atomics for the excess, so any capacity is CORRECT.
//   pass 2  reduce   one workgroup = one (bucket, level): streams the bucket's regions of all tiles (several loads in
//                    flight per lane), accumulates rows  row >> log2(NB)  in LDS as 64-bit fixed point (ds_add_u64), then
//                    adds the finished rows to the gradient table with plain read-modify-writes -- it is the only owner
//                    of those rows.
//
// What the memory system wants (tools/write_pattern_bench.hip, MI355X): the 64 runs of a tile written next to each other
// ([tile][bucket] order) and line-aligned at both ends reach > 5 TB/s; the same bytes as ragged 16-byte-granular runs, or
// [bucket][tile] order at a 768-byte stride, only ~3 TB/s.  Waves never wait for these stores: the barriers inside the
// level loop order LDS traffic only (lds_barrier), and the one global load of the loop is consumed before the stores.
//
// Bucket = LOW bits of the row (rotated, see bucket_of), so dense coarse levels (whose rows are spatially ordered and
// heavily skewed toward the volume centre) spread as evenly as the hashed ones.
#pragma once

#include "naf_device.h"

namespace naf {

template <uint32_t C>
struct RecF32 {                       // row + C fp32 values
    uint32_t w[1 + C];
    __device__ __forceinline__ void set(uint32_t row, const float (&v)[C]) {
        w[0] = row;
#pragma unroll
        for (uint32_t c = 0; c < C; ++c) w[1 + c] = __float_as_uint(v[c]);
    }
    __device__ __forceinline__ float value(uint32_t c) const { return __uint_as_float(w[1 + c]); }
};

template <uint32_t C>
struct RecBF16 {                      // row + C bf16 values (packed two per dword)
    uint32_t w[1 + (C + 1) / 2];
    __device__ __forceinline__ void set(uint32_t row, const float (&v)[C]) {
        w[0] = row;
#pragma unroll
        for (uint32_t c = 0; c < C; c += 2) {              // plain casts: hipcc emits one v_cvt_pk_bf16_f32 per pair
            typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
            bf16x2 pk;
            pk[0] = (__bf16)v[c];
            pk[1] = c + 1 < C ? (__bf16)v[c + 1] : (__bf16)0.0f;
            w[1 + c / 2] = __builtin_bit_cast(uint32_t, pk);
        }
    }
    __device__ __forceinline__ float value(uint32_t c) const {
        const uint32_t p = w[1 + c / 2];
        return __uint_as_float((c & 1u) ? (p & 0xffff0000u) : (p << 16));
    }
};

struct BinPlan {
    uint32_t tile_points;     // points per pass-1 workgroup (= its thread count)
    uint32_t n_tiles;
    uint32_t log2_nb;         // NB = buckets per level
    uint32_t slot_cap;        // records per (level, bucket, tile) region = LDS slots per bucket in pass 1
    uint32_t levels_per_pass;
    uint32_t max_local_rows;  // ceil(max T_l / NB)
};

// Fixed-point scale of the reducer.  `gmax_bits` = bit pattern of max |feature gradient| of the step (written by the MLP
// backward kernel, BEFORE the gradients are rounded to their storage type: rounding can lift a value by at most one ulp of
// bf16, 2^-8 relative, which the bound below absorbs by using the NEXT power of two).  A contribution is w * g with
// 0 <= w <= 1, so |v| <= gmax < 2^(E+1) with E = exponent(gmax); pass 1 may merge the up to 64 same-cell contributions of a
// wave into one record, so a record is bounded by 64 * 2^(E+1).
// fixed = v * 2^(kFixHead - E - 1) keeps a single contribution below 2^kFixHead and a record below 2^(kFixHead + 6): with
// kFixHead = 31 that leaves 63 - 37 = 26 bits for the sum (6.7e7 maximal records per row) and 31 significant bits below
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
// leaves the (two's complement) integer in the low mantissa bits for |v * 2^shift| < 2^51 (here < 2^38 per record).
__device__ __forceinline__ long long to_fixed(float v, double scale) {
    const double d = (double)v * scale + 6755399441055744.0;
    return __double_as_longlong(d) - 0x4338000000000000ll;
}

// Row <-> (bucket, local row).  local = row >> log2 NB; the bucket is the low log2 NB bits of the row ROTATED by a function
// of the local row, (row + local + (local >> 10)) & (NB-1): for a fixed local row the NB candidates still map one-to-one to
// the buckets, so (bucket, local) identifies the row, but neighbouring corners no longer collide.  On dense and
// uint32-wrapped dense levels row = x + s1 y + s2 z with s1 = s2 = 1 (mod 64): with the plain low bits the eight corners
// of a cell fall into buckets b + {0,1,1,2,1,2,2,3}, and the triple hits overflow the staging runs (0.5 ms per step of
// fallback atomics); rotated they fall into b + {0,1,2,3,3,4,5,6}.
// Hashed levels scatter their rows anyway and skip the rotation (three VALU operations per record in pass 1): `twist` is 1
// on dense / wrapped-dense levels, 0 on hashed ones -- the same rule in all three passes (level_twist).
__device__ __forceinline__ uint32_t bucket_twist(uint32_t local) { return local + (local >> 10); }
__device__ __forceinline__ uint32_t level_twist(const LevelMeta &m) { return m.mode < kHashMask ? 1u : 0u; }
__device__ __forceinline__ uint32_t bucket_of(uint32_t row, uint32_t log2_nb, uint32_t twist) {
    return (row + (twist ? bucket_twist(row >> log2_nb) : 0u)) & ((1u << log2_nb) - 1u);
}
__device__ __forceinline__ uint32_t row_of(uint32_t bucket, uint32_t local, uint32_t log2_nb, uint32_t twist) {
    return (local << log2_nb) | ((bucket - (twist ? bucket_twist(local) : 0u)) & ((1u << log2_nb) - 1u));
}

// rows per bucket in the sums buffer of pass 2 (a multiple of 64: pass 3 works on 64-row blocks)
__host__ __device__ __forceinline__ size_t sums_rows(const BinPlan &plan) { return ((size_t)plan.max_local_rows + 63u) & ~(size_t)63u; }

// run lengths: [level][bucket][tile] (one coalesced load per 64 tiles in pass 2); records: [level][tile][bucket][slot_cap]
__device__ __forceinline__ size_t count_index(const BinPlan &plan, uint32_t ly, uint32_t bucket, uint32_t tile) {
    return (((size_t)ly << plan.log2_nb) + bucket) * plan.n_tiles + tile;
}
__device__ __forceinline__ size_t region_index(const BinPlan &plan, uint32_t ly, uint32_t bucket, uint32_t tile) {
    return ((((size_t)ly * plan.n_tiles) + tile) << plan.log2_nb) + bucket;
}

// ---- pass 1 ---------------------------------------------------------------------------------------------------
// LDS: cnt[2][NB] | staging[NB][slot_cap] records.  One workgroup = one tile of NT points (one per thread) x LV
// consecutive levels: the sample position is evaluated once, and the stores of one level drain while the next level
// is being computed.  With 8-byte records the launch uses NT = 512: LDS allows two or three workgroups per CU either
// way, and 16+ resident waves hide the input loads and the staging round trips far better than 8 (5.6 -> 4.7 ms/step).
template <typename FT, uint32_t C, typename Src, typename Rec, uint32_t NT, uint32_t LV>
__global__ void __launch_bounds__(NT, NT == 512u ? 6 : NT == 1024u ? 4 : 1)   // 512 threads: three workgroups (24 waves) per CU -> <= 80 VGPRs
scatter_bin_kernel(Src src, const typename FT::store_t *__restrict__ grad, const int32_t *__restrict__ offsets,
                   float *__restrict__ grad_table, Rec *__restrict__ regions, uint32_t *__restrict__ counts,
                   uint32_t *__restrict__ overflow, uint32_t B, uint32_t H, uint32_t level_base, uint32_t n_levels,
                   BinPlan plan) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr uint32_t NW = NT / 64u;
    const uint32_t NB = 1u << plan.log2_nb, CAP = plan.slot_cap;
    uint32_t *cnt2 = reinterpret_cast<uint32_t *>(smem);
    Rec *staging = reinterpret_cast<Rec *>(cnt2 + 2u * NB);
    const uint32_t tile = blockIdx.x, lane = threadIdx.x & 63u, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // scalar: copy-out bucket arithmetic on the SALU
    for (uint32_t i = threadIdx.x; i < 2u * NB; i += NT) cnt2[i] = 0u;

    // the thread's point.  Threads past the end of the batch take the last point with a zero gradient: their records
    // add nothing, and no validity test is needed further down.
    const uint32_t b_raw = tile * NT + threadIdx.x;
    const bool valid = b_raw < B;
    const uint32_t b = valid ? b_raw : B - 1u;
    float x[3];
    src.get(b, x);
    const float spacing = src.sample_spacing();
    // this level's feature gradient, requested one level ahead (see the note at the copy-out)
    RawVec<FT, C> graw;
    float g[C];
    if (blockIdx.y * LV < n_levels) raw_load<FT, C>(grad + ((size_t)(level_base + blockIdx.y * LV) * B + b) * C, graw);
    raw_unpack<FT, C>(graw, g);
    __syncthreads();
    uint32_t n_overflow = 0;

    for (uint32_t it = 0; it < LV; ++it) {
        const uint32_t ly = blockIdx.y * LV + it;
        if (ly >= n_levels) break;                                   // uniform
        const uint32_t level = level_base + ly;
        uint32_t *cnt = cnt2 + (it & 1u) * NB;
        const LevelMeta m = make_level_meta<3>(offsets, level, H);
        float *__restrict__ gg = grad_table + (size_t)m.offset * C;

        // A: rows + values of this thread's eight contributions
        uint32_t row[8];
        float val[8][C];
        float frac[3];
        uint32_t pg[3];
        locate<3>(x, m.scale, frac, pg);
        if (!valid) {
#pragma unroll
            for (uint32_t ch = 0; ch < C; ++ch) g[ch] = 0.0f;
        }
        dispatch_mode<Src::kInRange>(m.mode, [&](auto mode_tag) {
            constexpr uint32_t MODE = decltype(mode_tag)::value;
            float w[8];
            cell_corners<MODE, 3>(m, frac, pg, w, row);
#pragma unroll
            for (uint32_t c = 0; c < 8; ++c)
#pragma unroll
                for (uint32_t ch = 0; ch < C; ++ch) val[c][ch] = w[c] * g[ch];
        });
        // Cells wider than the sample spacing (wave-uniform decision): consecutive samples of a ray that fall into the
        // same cell hit the same 8 rows.  Merge each run of equal cells with a segmented inclusive scan; only the last
        // lane of a run emits records.  Runs are cut at 16-lane rows: the scan then moves its operands with DPP row shifts
        // (a modifier of the VALU instruction) instead of 6 x 16 trips through the LDS crossbar (ds_bpermute) -- on the
        // three coarse levels that merge, the scan used to cost more than the records it saves (per-level launches:
        // 0.078 ms against 0.066 ms for a hashed level).  A cut costs at most one extra record set per 16 samples.
        bool emit = true;
        if (m.scale * spacing < 0.75f) {
            const uint32_t c_lo = pg[0] | (pg[1] << 16), c_hi = pg[2];        // merging levels have < 2^16 cells per axis
            const uint32_t p_lo = dpp_row_shr<1>(c_lo), p_hi = dpp_row_shr<1>(c_hi);
            const uint64_t heads = __ballot((lane & 15u) == 0u || c_lo != p_lo || c_hi != p_hi);
            const uint32_t start = 63u - (uint32_t)__clzll(heads & (~0ull >> (63u - lane)));
            const uint32_t len = lane - start;                               // my position inside the run (same row)
            auto fold = [&](auto shift_tag) {
                constexpr uint32_t d = decltype(shift_tag)::value;
                const float take = len >= d ? 1.0f : 0.0f;          // val += shifted * take: one v_fmac with a DPP operand per value
#pragma unroll                                                       // (the product with 0 / 1 is exact, so this IS the masked add)
                for (uint32_t c = 0; c < 8; ++c)
#pragma unroll
                    for (uint32_t ch = 0; ch < C; ++ch) val[c][ch] = __builtin_fmaf(dpp_row_shr<d>(val[c][ch]), take, val[c][ch]);
            };
            fold(std::integral_constant<uint32_t, 1>{});
            fold(std::integral_constant<uint32_t, 2>{});
            fold(std::integral_constant<uint32_t, 4>{});
            fold(std::integral_constant<uint32_t, 8>{});
            emit = lane == 63u || ((heads >> (lane + 1u)) & 1ull);
        }
        // take a slot per record and write it; a full bucket (rare: slot_cap = 1.5 x mean + 8) adds straight to the table
        if (emit) {
            const uint32_t twist = level_twist(m);                      // wave-uniform
#pragma unroll
            for (uint32_t half = 0; half < 2; ++half) {                 // four corners at a time: fewer live registers
                uint32_t pos[4], bkt[4];
                if (twist) {
#pragma unroll
                    for (uint32_t c = 0; c < 4; ++c) bkt[c] = bucket_of(row[4 * half + c], plan.log2_nb, 1u);
                } else {
#pragma unroll
                    for (uint32_t c = 0; c < 4; ++c) bkt[c] = bucket_of(row[4 * half + c], plan.log2_nb, 0u);
                }
#pragma unroll
                for (uint32_t c = 0; c < 4; ++c) pos[c] = atomicAdd(&cnt[bkt[c]], 1u);
#pragma unroll
                for (uint32_t c = 0; c < 4; ++c) {
                    const uint32_t cc = 4 * half + c;
                    if (pos[c] < CAP) {
                        Rec r;
                        r.set(row[cc] >> plan.log2_nb, val[cc]);
                        staging[bkt[c] * CAP + pos[c]] = r;
                    } else {
#pragma unroll
                        for (uint32_t ch = 0; ch < C; ++ch) atomicAdd(gg + (size_t)row[cc] * C + ch, val[cc][ch]);
                        ++n_overflow;
                    }
                }
            }
        }
        // The next level's gradient is requested here and CONSUMED right after the barrier, before this level's stores are
        // issued: vmcnt retires in order and the compiler can only wait for "everything", so any wait placed after the stores
        // would sit out their whole round trip.  This way the stores drain behind the next level's arithmetic.
        if (it + 1u < LV && ly + 1u < n_levels) raw_load<FT, C>(grad + ((size_t)(level + 1u) * B + b) * C, graw);
        lds_barrier();
        raw_unpack<FT, C>(graw, g);
#pragma unroll
        for (uint32_t ch = 0; ch < C; ++ch) asm volatile("" : "+v"(g[ch]) : : "memory");      // pin the wait here

        // B: copy each bucket run to its region.  A wave owns buckets wave, wave+NW, ...; it takes them kCopy at a time
        //    with straight-line code (run lengths, then staging reads, then stores) so the LDS / global round trips of
        //    several buckets overlap.  8-byte records travel two per lane as 16-byte LDS reads / global stores
        //    (slot_cap even, <= 128).  The other counter set is cleared for the next level meanwhile.
        for (uint32_t i = threadIdx.x; i < NB; i += NT) cnt2[((it & 1u) ^ 1u) * NB + i] = 0u;
        if constexpr (sizeof(Rec) == 8) {
            constexpr uint32_t kCopy = 4;                              // NB (>= 64, a power of two) is a multiple of NW * kCopy
            const uint32_t pairs = CAP >> 1;
            const uint32_t my_pair = min(lane, pairs - 1u);     // (slot_cap is a multiple of 16 records = 128 B here)
            Rec *__restrict__ tile_regions = regions + region_index(plan, ly, 0u, tile) * CAP;      // the NB regions of this tile are adjacent
            for (uint32_t base = wave; base < NB; base += NW * kCopy) {
                uint32_t nrun[kCopy];
                uint4 v[kCopy];
#pragma unroll
                for (uint32_t k = 0; k < kCopy; ++k) nrun[k] = min(cnt[base + NW * k], CAP);
#pragma unroll
                for (uint32_t k = 0; k < kCopy; ++k)
                    v[k] = reinterpret_cast<const uint4 *>(staging + (base + NW * k) * CAP)[my_pair];
#pragma unroll
                for (uint32_t k = 0; k < kCopy; ++k)
                    if (2u * lane < min((nrun[k] + 15u) & ~15u, CAP))        // whole 128-byte lines; stale slots are harmless
                        reinterpret_cast<uint4 *>(tile_regions + (size_t)(base + NW * k) * CAP)[lane] = v[k];
                // the kCopy run lengths of the batch leave with ONE store: lane k writes the count of bucket base + NW k
                if (lane < kCopy) {
                    uint32_t mine = nrun[0];
#pragma unroll
                    for (uint32_t k = 1; k < kCopy; ++k) mine = lane == k ? nrun[k] : mine;
                    counts[count_index(plan, ly, base + NW * lane, tile)] = mine;
                }
            }
        } else {
            for (uint32_t bkt = wave; bkt < NB; bkt += NW) {
                const uint32_t n = min(cnt[bkt], CAP);
                const size_t reg = region_index(plan, ly, bkt, tile);
                Rec *__restrict__ dst = regions + reg * CAP;
                for (uint32_t slot = lane; slot < n; slot += 64u) dst[slot] = staging[bkt * CAP + slot];
                if (lane == 0u) counts[count_index(plan, ly, bkt, tile)] = n;
            }
        }
        lds_barrier();                                               // staging and this counter set are free again
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
// the step (fixed_shift): 31 significant bits below it, 26 bits of headroom above for the sum of merged records.
template <uint32_t C, typename Rec>
__global__ void __launch_bounds__(1024)
scatter_reduce_kernel(const Rec *__restrict__ regions, const uint32_t *__restrict__ counts, const int32_t *__restrict__ offsets,
                      float *__restrict__ grad_table, float *__restrict__ sums, const uint32_t *__restrict__ gmax_bits,
                      uint32_t H, uint32_t level_base, uint32_t ly_begin, BinPlan plan) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t gbits = *gmax_bits;
    const int shift = fixed_shift(gbits);
    const bool poison = fixed_nonfinite(gbits);
    const double scale = ldexp(1.0, shift);
    unsigned long long *acc = reinterpret_cast<unsigned long long *>(smem);
    const uint32_t T_ = blockDim.x, CAP = plan.slot_cap;
    const uint32_t bucket = blockIdx.x, ly = ly_begin + blockIdx.y, level = level_base + ly;      // ly: level slot of the bin pass
    const uint32_t off = (uint32_t)offsets[level], T = (uint32_t)offsets[level + 1] - off;
    // rows of this bucket: one per complete group of NB rows, plus one if the bucket's row of the last, partial group exists
    const uint32_t full_groups = T >> plan.log2_nb;
    const uint32_t twist = level_twist(make_level_meta<3>(offsets, level, H));
    const uint32_t rows_local = full_groups + (row_of(bucket, full_groups, plan.log2_nb, twist) < T ? 1u : 0u);
    // accumulators are channel-major, acc[ch][local row]: the two 8-byte cells of a row would otherwise sit 8 bytes apart and
    // one ds_add_u64 instruction (one channel of 64 rows) could reach only every other bank pair
    const uint32_t pitch = plan.max_local_rows;
    for (uint32_t i = threadIdx.x; i < pitch * C; i += T_) acc[i] = 0ull;
    __syncthreads();

    // the wave index is made scalar explicitly: tile ranges, run-length block addresses and region bases then live in SGPRs
    const uint32_t lane = threadIdx.x & 63u, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), n_waves = T_ >> 6;
    const size_t cnt0 = count_index(plan, ly, bucket, 0);
    auto add = [&](const Rec &r) {
#pragma unroll
        for (uint32_t ch = 0; ch < C; ++ch) atomicAdd(&acc[ch * pitch + r.w[0]], (unsigned long long)to_fixed(r.value(ch), scale));   // ds_add_u64
    };
    // each wave owns blocks of 64 consecutive tiles: one coalesced load brings their run lengths.  Regions are read
    // kGroup at a time with straight-line code: kGroup loads for records 0..63 plus kGroup * kTail / 64 loads for the tails
    // are in flight per lane before the first LDS atomic (lanes past a run length read slot 0, a line that is fetched
    // anyway -> no extra traffic).
    constexpr uint32_t kGroup = 8, kTail = 16;
    // every wave streams ONE contiguous range of tiles
    // gridDim.z > 1 (few levels per pass at very large batches): the tiles are split between gridDim.z workgroups
    const uint32_t split_tiles = (plan.n_tiles + gridDim.z - 1u) / gridDim.z;
    const uint32_t split_begin = blockIdx.z * split_tiles, split_end = min(plan.n_tiles, split_begin + split_tiles);
    const uint32_t per_wave = (((split_tiles + n_waves - 1u) / n_waves) + 63u) & ~63u;
    const uint32_t t_begin = split_begin + wave * per_wave, t_end = min(split_end, t_begin + per_wave);
    for (uint32_t t0 = t_begin; t0 < t_end; t0 += 64u) {
        const uint32_t mine = t0 + lane < t_end ? counts[cnt0 + t0 + lane] : 0u;
        const uint32_t n_here = min(64u, t_end - t0);
        for (uint32_t j = 0; j < n_here; j += kGroup) {
            uint32_t n[kGroup], n_max = 0u;
            Rec ra[kGroup];
#pragma unroll
            for (uint32_t u = 0; u < kGroup; ++u) {
                const uint32_t tj = min(j + u, n_here - 1u);
                n[u] = j + u < n_here ? (uint32_t)__builtin_amdgcn_readlane((int)mine, (int)tj) : 0u;      // scalar (tj is wave-uniform)
                n_max = max(n_max, n[u]);
                ra[u] = (regions + region_index(plan, ly, bucket, t0 + tj) * CAP)[lane < n[u] ? lane : 0u];
            }
            if (n_max <= 64u + kTail) {
                // the usual case: no run is longer than 64 + kTail records.  The tails (records 64..) of 64 / kTail regions
                // share one load and one pair of LDS atomics, kTail lanes per region, instead of a nearly empty wave each.
                constexpr uint32_t kPer = 64u / kTail;                      // regions per tail instruction
                Rec rt[kGroup / kPer];
                uint32_t nt[kGroup / kPer];
#pragma unroll
                for (uint32_t q = 0; q < kGroup / kPer; ++q) {
                    const uint32_t sub = lane / kTail, u = q * kPer + sub, slot = 64u + (lane % kTail);
                    const uint32_t tj = min(j + u, n_here - 1u);
                    nt[q] = n[q * kPer];                                    // select among the kPer scalar run lengths
#pragma unroll
                    for (uint32_t k = 1; k < kPer; ++k) nt[q] = sub == k ? n[q * kPer + k] : nt[q];
                    rt[q] = (regions + region_index(plan, ly, bucket, t0 + tj) * CAP)[slot < nt[q] ? slot : 0u];
                }
#pragma unroll
                for (uint32_t u = 0; u < kGroup; ++u)
                    if (lane < n[u]) add(ra[u]);
#pragma unroll
                for (uint32_t q = 0; q < kGroup / kPer; ++q)
                    if (64u + (lane % kTail) < nt[q]) add(rt[q]);
            } else {
                Rec rb[kGroup];                                             // slot_cap <= 128 (planner invariant): two loads
#pragma unroll                                                              // per lane cover a whole region
                for (uint32_t u = 0; u < kGroup; ++u) {
                    const uint32_t tj = min(j + u, n_here - 1u);
                    rb[u] = (regions + region_index(plan, ly, bucket, t0 + tj) * CAP)[lane + 64u < n[u] ? lane + 64u : 0u];
                }
#pragma unroll
                for (uint32_t u = 0; u < kGroup; ++u) {
                    if (lane < n[u]) add(ra[u]);
                    if (lane + 64u < n[u]) add(rb[u]);
                }
            }
        }
    }
    __syncthreads();
    if (gridDim.z == 1u) {
        // sole owner of these rows, but they are NB rows apart in the table: write the finished sums as one contiguous block
        // [level][bucket][local][C]; scatter_apply_kernel adds them to the table with coalesced accesses on both sides
        float *__restrict__ dst = sums + (((size_t)ly << plan.log2_nb) + bucket) * sums_rows(plan) * C;
        for (uint32_t i = threadIdx.x; i < rows_local * C; i += T_) {
            const uint32_t local = i / C, ch = i - local * C;
            dst[i] = poison ? __builtin_nanf("") : (float)ldexp((double)(long long)acc[ch * pitch + local], -shift);
        }
    } else {
        float *__restrict__ gg = grad_table + (size_t)off * C;
        for (uint32_t i = threadIdx.x; i < rows_local * C; i += T_) {
            const uint32_t local = i / C, ch = i - local * C;
            atomicAdd(gg + (size_t)row_of(bucket, local, plan.log2_nb, twist) * C + ch,
                      poison ? __builtin_nanf("") : (float)ldexp((double)(long long)acc[ch * pitch + local], -shift));   // one add per row and split
        }
    }
}

// ---- pass 3 ---------------------------------------------------------------------------------------------------
// grad_table[row] += sums[level][bucket][local] for row = row_of(bucket, local).  One workgroup = 64 local rows x 64
// buckets: the block is read bucket-major (64 local rows x C floats contiguous per bucket), turned in LDS and added to
// the table row-major (the 64 buckets of one local row are 64 consecutive rows, in rotated order).  Doing this
// read-modify-write straight from the reducer touches every 64-byte sector of the table for 8 useful bytes.
template <uint32_t C>
__global__ void __launch_bounds__(256)
scatter_apply_kernel(const float *__restrict__ sums, const int32_t *__restrict__ offsets, float *__restrict__ grad_table,
                     uint32_t H, uint32_t level_base, uint32_t ly_begin, BinPlan plan) {
    __shared__ float tile[64][64 * C + 1];                      // [bucket in group][local in block][C], odd pitch
    const uint32_t ly = ly_begin + blockIdx.y, level = level_base + ly, local0 = blockIdx.x * 64u, bucket0 = blockIdx.z * 64u;
    const uint32_t off = (uint32_t)offsets[level], T = (uint32_t)offsets[level + 1] - off;
    if (((size_t)local0 << plan.log2_nb) >= T) return;          // block past the end of this level (uniform)
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const size_t rows = sums_rows(plan);
    const uint32_t twist = level_twist(make_level_meta<3>(offsets, level, H));
    for (uint32_t b = wave; b < 64u; b += 4u) {
        const float *src_b = sums + ((((size_t)ly << plan.log2_nb) + bucket0 + b) * rows + local0) * C;
#pragma unroll
        for (uint32_t k = 0; k < C; ++k) tile[b][lane + 64u * k] = src_b[lane + 64u * k];      // element e = local * C + ch
    }
    __syncthreads();
    float *__restrict__ gg = grad_table + (size_t)off * C;
    for (uint32_t j = wave; j < 64u; j += 4u) {
        const uint32_t row = row_of(bucket0 + lane, local0 + j, plan.log2_nb, twist);
        if (row < T) {
#pragma unroll
            for (uint32_t ch = 0; ch < C; ++ch) gg[(size_t)row * C + ch] += tile[lane][j * C + ch];
        }
    }
}

}  // namespace naf
