// scatter_binned.h -- hash-table gradient scatter without per-contribution global atomics (gfx950).
//
// Why: on MI355X a global float atomic is executed at the memory side, one 64-byte request per touched line; scattered
// 4-byte adds top out at ~2e10 requests/s chip-wide (MI355X_MICROARCH.md "Global float atomics"; measured here:
// 2.4 ms per level for 25 M adds, 12 ms on level 0 where 4913 rows take all of them).  The reference design
// (hashencoder.cu:257-269, one atomicAdd per corner and channel) is ~25x slower than the rest of the training step.
// A level of the table gets ~50-200 contributions per row per step, but they are hash-scattered, so no locality trick
// removes them -- they have to be ROUTED to an owner instead (a one-digit radix multisplit on the row index):
//
//   pass 1  bin      one workgroup = one tile of points x one level.  It counts its contributions per bucket
//                    (bucket = row & (NB-1)) in LDS, reserves a contiguous range in each bucket's global stream with ONE
//                    returning atomic per non-empty bucket, counting-sorts the records (row, w*g[0..C)) in LDS and
//                    copies them out in bucket order, so stores are coalesced runs instead of 64 scattered 8-byte
//                    pieces per wave.  A full stream falls back to a global atomic for that contribution, so any
//                    capacity is CORRECT; streams are sized 1.25x the uniform expectation.
//   pass 2  reduce   one workgroup = one (bucket, level): streams its records (contiguous, coalesced, several loads
//                    in flight per lane), accumulates rows  row >> log2(NB)  in LDS (ds_add_f32), then adds the finished
//                    rows to the gradient table with plain read-modify-writes -- it is the only owner of those rows.
//
// Bucket = LOW bits of the row, so dense coarse levels (whose rows are spatially ordered and heavily skewed toward the
// volume centre) spread as evenly as the hashed ones.
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
        for (uint32_t c = 0; c < C; c += 2) {
            const uint32_t lo = f32_to_bf16(v[c]);
            const uint32_t hi = c + 1 < C ? f32_to_bf16(v[c + 1]) : 0u;
            w[1 + c / 2] = lo | (hi << 16);
        }
    }
    __device__ __forceinline__ float value(uint32_t c) const {
        const uint32_t p = w[1 + c / 2];
        return __uint_as_float((c & 1u) ? (p & 0xffff0000u) : (p << 16));
    }
};

struct BinPlan {
    uint32_t tile_points;     // points per pass-1 workgroup = 256 * PPT
    uint32_t n_tiles;
    uint32_t log2_nb;         // NB = buckets per level (multiple of 256)
    uint32_t stream_cap;      // records per (level, bucket, sub) stream
    uint32_t log2_sub;        // each bucket has 2^log2_sub sub-streams (tile % 2^log2_sub) so the cursors are not hot spots
    uint32_t levels_per_pass;
    uint32_t max_local_rows;  // ceil(max T_l / NB)
};

// exclusive prefix sum of one value per thread over a 256-thread workgroup; `scratch` holds >= 4 uint32
__device__ __forceinline__ uint32_t block_exclusive_scan_256(uint32_t v, uint32_t *scratch, uint32_t &total) {
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t t = __shfl_up(inc, off, 64);
        if ((int)lane >= off) inc += t;
    }
    if (lane == 63u) scratch[wave] = inc;
    __syncthreads();
    uint32_t base = 0;
    for (uint32_t w = 0; w < wave; ++w) base += scratch[w];
    total = scratch[0] + scratch[1] + scratch[2] + scratch[3];
    __syncthreads();
    return base + inc - v;
}

// ---- pass 1 ---------------------------------------------------------------------------------------------------
// LDS: cnt[NB] | off[NB] | gbase[NB] | scratch[4] | staging[256 * PPT * 8] records.
// PPT = points per thread: the rows and weights of a thread's PPT*8 contributions stay in registers between the
// counting and the placement phase, so the index arithmetic runs once.
template <typename FT, uint32_t C, typename Src, typename Rec, uint32_t PPT>
__global__ void __launch_bounds__(256)
scatter_bin_kernel(Src src, const typename FT::store_t *__restrict__ grad, const int32_t *__restrict__ offsets,
                   float *__restrict__ grad_table, Rec *__restrict__ streams, uint32_t *__restrict__ cursors,
                   uint32_t *__restrict__ overflow, uint32_t B, uint32_t H, uint32_t level_base, BinPlan plan) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t NB = 1u << plan.log2_nb, mask = NB - 1u;
    uint32_t *cnt = reinterpret_cast<uint32_t *>(smem);
    uint32_t *off = cnt + NB;
    uint32_t *gbase = off + NB;
    uint32_t *scratch = gbase + NB;
    Rec *staging = reinterpret_cast<Rec *>(scratch + 4);
    const uint32_t ly = blockIdx.y, level = level_base + ly, tile = blockIdx.x;
    for (uint32_t i = threadIdx.x; i < NB; i += 256u) cnt[i] = 0u;
    __syncthreads();

    const LevelMeta m = make_level_meta<3>(offsets, level, H);
    const uint32_t b0 = tile * (256u * PPT);
    // merge same-cell runs only where they exist: cells wider than the sample spacing (wave-uniform decision)
    const bool dedup = m.scale * src.sample_spacing() < 0.75f;

    // A: rows + weights of this thread's contributions; count per bucket
    uint32_t row[PPT][8];
    float val[PPT][8][C];
    uint64_t cell[PPT];
    const uint32_t lane = threadIdx.x & 63u;
    dispatch_mode(m.mode, [&](auto mode_tag) {
        constexpr uint32_t MODE = decltype(mode_tag)::value;
#pragma unroll
        for (uint32_t k = 0; k < PPT; ++k) {
            const uint32_t b = b0 + k * 256u + threadIdx.x;
            const bool valid = b < B;
            float x[3], frac[3], g[C];
            uint32_t pg[3];
            src.get(valid ? b : B - 1u, x);
            locate<3>(x, m.scale, frac, pg);
            load_vec<FT, C>(grad + ((size_t)level * B + (valid ? b : B - 1u)) * C, g);
            cell[k] = valid ? ((uint64_t)pg[0] | ((uint64_t)pg[1] << 21) | ((uint64_t)pg[2] << 42)) : ~0ull;
#pragma unroll
            for (uint32_t c = 0; c < 8; ++c) {
                uint32_t pl[3];
                const float w = corner<3>(c, frac, pg, pl);
                row[k][c] = valid ? grid_row<MODE, 3>(m, pl) : 0xffffffffu;
#pragma unroll
                for (uint32_t ch = 0; ch < C; ++ch) val[k][c][ch] = w * g[ch];
            }
        }
    });
    if (dedup) {
        // Consecutive samples of a ray that fall into the same cell hit the same 8 rows: merge each run of equal
        // cells inside the wave (segmented inclusive scan); only the last lane of a run emits records.
        for (uint32_t k = 0; k < PPT; ++k) {
            const uint64_t prev = __shfl_up(cell[k], 1, 64);
            const uint64_t heads = __ballot(lane == 0u || cell[k] != prev);
            const uint32_t start = 63u - (uint32_t)__clzll(heads & (~0ull >> (63u - lane)));
#pragma unroll
            for (uint32_t d = 1; d < 64; d <<= 1) {
                const bool take = lane >= start + d;
#pragma unroll
                for (uint32_t c = 0; c < 8; ++c)
#pragma unroll
                    for (uint32_t ch = 0; ch < C; ++ch) {
                        const float t = __shfl_up(val[k][c][ch], d, 64);
                        if (take) val[k][c][ch] += t;
                    }
            }
            const bool tail = lane == 63u || ((heads >> (lane + 1u)) & 1ull);
            if (!tail) {
#pragma unroll
                for (uint32_t c = 0; c < 8; ++c) row[k][c] = 0xffffffffu;
            }
        }
    }
#pragma unroll
    for (uint32_t k = 0; k < PPT; ++k)
#pragma unroll
        for (uint32_t c = 0; c < 8; ++c)
            if (row[k][c] != 0xffffffffu) atomicAdd(&cnt[row[k][c] & mask], 1u);
    __syncthreads();

    // B: exclusive scan of the bucket counts (each thread owns NB/256 consecutive buckets) + global reservation
    const uint32_t per = NB >> 8;
    uint32_t mine = 0;
    for (uint32_t k = 0; k < per; ++k) mine += cnt[threadIdx.x * per + k];
    uint32_t total;
    uint32_t run = block_exclusive_scan_256(mine, scratch, total);
    const uint32_t sub = tile & ((1u << plan.log2_sub) - 1u);
    // stream id = ((ly * NB + bucket) << log2_sub) + sub
    uint32_t *cur = cursors + (((size_t)ly << plan.log2_nb) << plan.log2_sub) + sub;
    for (uint32_t k = 0; k < per; ++k) {
        const uint32_t bkt = threadIdx.x * per + k, n = cnt[bkt];
        off[bkt] = run;
        run += n;
        gbase[bkt] = n ? atomicAdd(&cur[(size_t)bkt << plan.log2_sub], n) : 0u;       // one returning atomic per non-empty bucket
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < NB; i += 256u) cnt[i] = off[i];     // running write positions
    __syncthreads();

    // C: place the records in bucket order in LDS
    uint32_t pos[PPT][8];
#pragma unroll
    for (uint32_t k = 0; k < PPT; ++k)
#pragma unroll
        for (uint32_t c = 0; c < 8; ++c)
            pos[k][c] = row[k][c] != 0xffffffffu ? atomicAdd(&cnt[row[k][c] & mask], 1u) : 0xffffffffu;
#pragma unroll
    for (uint32_t k = 0; k < PPT; ++k)
#pragma unroll
        for (uint32_t c = 0; c < 8; ++c)
            if (pos[k][c] != 0xffffffffu) {
                Rec r;
                r.set(row[k][c], val[k][c]);
                staging[pos[k][c]] = r;
            }
    __syncthreads();

    // D: copy out; consecutive threads copy consecutive records of a bucket -> coalesced runs
    float *__restrict__ gg = grad_table + (size_t)m.offset * C;
    for (uint32_t j = threadIdx.x; j < total; j += 256u) {
        Rec r = staging[j];
        const uint32_t rw = r.w[0], bkt = rw & mask;
        const uint32_t dst = gbase[bkt] + (j - off[bkt]);
        if (dst < plan.stream_cap) {
            r.w[0] = rw >> plan.log2_nb;
            streams[((((size_t)ly << plan.log2_nb) + bkt) << plan.log2_sub | sub) * plan.stream_cap + dst] = r;
        } else {                                               // stream full: still correct, just slower
#pragma unroll
            for (uint32_t ch = 0; ch < C; ++ch) atomicAdd(gg + (size_t)rw * C + ch, r.value(ch));
            atomicAdd(overflow, 1u);
        }
    }
}

// ---- pass 2 ---------------------------------------------------------------------------------------------------
// Accumulation is in DOUBLE: on gfx950 ds_add_f64 runs at ~2.5 lane-ops/clk/CU while ds_add_f32 manages 0.33
// (tools/lds_atomic_bench.hip), and the sums come out more accurate than fp32 atomics as a bonus.
template <uint32_t C, typename Rec>
__global__ void __launch_bounds__(256)
scatter_reduce_kernel(const Rec *__restrict__ streams, const uint32_t *__restrict__ cursors, const int32_t *__restrict__ offsets,
                      float *__restrict__ grad_table, uint32_t level_base, BinPlan plan) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double *acc = reinterpret_cast<double *>(smem);
    const uint32_t NB = 1u << plan.log2_nb;
    const uint32_t bucket = blockIdx.x, ly = blockIdx.y, level = level_base + ly;
    const uint32_t off = (uint32_t)offsets[level], T = (uint32_t)offsets[level + 1] - off;
    const uint32_t rows_local = bucket < T ? (T - bucket + NB - 1u) >> plan.log2_nb : 0u;    // rows with row % NB == bucket
    for (uint32_t i = threadIdx.x; i < rows_local * C; i += 256u) acc[i] = 0.0;
    __syncthreads();

    constexpr uint32_t U = 8;                                   // records in flight per lane
    for (uint32_t sub = 0; sub < (1u << plan.log2_sub); ++sub) {
        const size_t stream = ((((size_t)ly << plan.log2_nb) + bucket) << plan.log2_sub) + sub;
        const uint32_t n = min(cursors[stream], plan.stream_cap);
        const Rec *__restrict__ recs = streams + stream * plan.stream_cap;
        for (uint32_t i0 = 0; i0 < n; i0 += 256u * U) {
            Rec r[U];
#pragma unroll
            for (uint32_t u = 0; u < U; ++u) {                          // unconditional (clamped) loads: all U in flight at once
                const uint32_t i = i0 + u * 256u + threadIdx.x;
                r[u] = recs[i < n ? i : n - 1u];
            }
#pragma unroll
            for (uint32_t u = 0; u < U; ++u) {
                const uint32_t i = i0 + u * 256u + threadIdx.x;
                if (i < n) {
#pragma unroll
                    for (uint32_t ch = 0; ch < C; ++ch) atomicAdd(&acc[r[u].w[0] * C + ch], (double)r[u].value(ch));     // ds_add_f64
                }
            }
        }
    }
    __syncthreads();
    float *__restrict__ gg = grad_table + (size_t)off * C;
    for (uint32_t i = threadIdx.x; i < rows_local * C; i += 256u) {
        const uint32_t local = i / C, ch = i - local * C;
        const size_t dst = ((size_t)local << plan.log2_nb) + bucket;
        gg[dst * C + ch] += (float)acc[i];                        // sole owner of these rows in this launch
    }
}

}  // namespace naf
