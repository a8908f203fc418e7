// render_fused.hip -- fused NAF ray-march pipeline for gfx950: rays -> samples -> hash features -> sigma-MLP (MFMA)
// -> attenuation line integral -> masked MSE gradient -> MLP backward (MFMA) -> hash-table scatter.
//
// Replaces the ATen / cuBLAS kernel soup behind reference src/render/render.py:82-212, src/network/network.py:34-58,
// src/loss/loss.py:37-39 and the autograd of train.py:69-127.  Points are never materialised: every kernel recomputes
// its sample position from the 32-byte ray record and the jitter (explicit t_rand or the counter-based generator).
//
// Kernels of one training step (B = n_rays * S points, feature tensors are [L, B, C]):
//   1 encode_kernel                  gathers  -> feat                 level-major, or XCD groups taking alternate levels below 500 k
//                                                                      points; 16-byte window gathers below 600 k points, two gathers above
//   2 mlp_forward_kernel             feat -> sigma -> acc[r]          one wave per ray, wave-reduced line integral
//   3 mlp_backward_kernel            feat, (acc, target, weight | d acc) -> dfeat, per-workgroup slabs: dW, loss share, max |dfeat|
//   4 scatter_bin / scatter_reduce (scatter_binned.h)   dfeat -> grad table (or, with the Adam tail, straight into the table's update),
//     no global atomics; spare workgroups of the first scatter_bin launch reduce the slabs of 3 (mlp_slabs.h): MLP gradient or the
//     MLP's Adam update, loss, gradient maximum.  hash_backward_kernel<SrcRays> (hash_kernels.h, fp32 atomics like the reference)
//     below 2^13 points per call, with mlp_grad_reduce_kernel as a launch of its own (also on data-parallel steps).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <type_traits>

#include "naf_host.h"
#include "hash_kernels.h"
#include "encode_kernel.h"
#include "field_mlp.h"
#include "field_mlp16.h"
#include "fused_forward.h"
#include "mlp_slabs.h"
#include "scatter_binned.h"
#include "scatter_v2.h"
#include "scatter_host.h"

namespace naf {

// ---- feature tile I/O ------------------------------------------------------------------------------------
// Lane (n,h) owns features f = 8q + 4h + i (q,i = 0..3) of point p; feature f is channel f % C of level f / C.
// The load is split in two so that a kernel can issue the (raw) loads of its NEXT tile before it starts the arithmetic of
// the current one and convert them an iteration later: the MLP kernels run at one or two waves per SIMD, nothing else
// hides the ~2 us HBM round trip.
template <uint32_t kBytes> struct RawWord;
template <> struct RawWord<2> { typedef uint16_t type; };
template <> struct RawWord<4> { typedef uint32_t type; };
template <> struct RawWord<8> { typedef uint2 type; };
template <> struct RawWord<16> { typedef uint4 type; };

template <typename FT, uint32_t C>
struct FeatRaw {
    static constexpr uint32_t V = C < 4 ? C : 4;     // contiguous channels per access
    using word_t = typename RawWord<V * sizeof(typename FT::store_t)>::type;
    word_t w[16 / V];
};
template <typename FT, uint32_t C>
__device__ __forceinline__ void load_feat_raw(const typename FT::store_t *__restrict__ feat, uint32_t B, uint32_t p,
                                              uint32_t h, FeatRaw<FT, C> &raw) {
    constexpr uint32_t V = FeatRaw<FT, C>::V;
#pragma unroll
    for (uint32_t q = 0; q < 4; ++q)
#pragma unroll
        for (uint32_t i = 0; i < 4; i += V) {
            const uint32_t f = 8u * q + 4u * h + i;
            raw.w[(4 * q + i) / V] =
                *reinterpret_cast<const typename FeatRaw<FT, C>::word_t *>(feat + ((size_t)(f / C) * B + p) * C + (f % C));
        }
}
template <typename FT, uint32_t C>
__device__ __forceinline__ void unpack_feat_raw(const FeatRaw<FT, C> &raw, float (&x)[16]) {
    constexpr uint32_t V = FeatRaw<FT, C>::V;
    using S = typename FT::store_t;
#pragma unroll
    for (uint32_t j = 0; j < 16 / V; ++j) {
        S e[V];
        __builtin_memcpy(e, &raw.w[j], sizeof(e));
#pragma unroll
        for (uint32_t k = 0; k < V; ++k) x[V * j + k] = Conv<FT>::load(&e[k]);
    }
}
template <typename FT, uint32_t C>
__device__ __forceinline__ void load_feat_slots(const typename FT::store_t *__restrict__ feat, uint32_t B, uint32_t p,
                                                uint32_t h, float (&x)[16]) {
    FeatRaw<FT, C> raw;
    load_feat_raw<FT, C>(feat, B, p, h, raw);
    unpack_feat_raw<FT, C>(raw, x);
}
template <typename FT, uint32_t C>
__device__ __forceinline__ void store_feat_slots(typename FT::store_t *__restrict__ feat, uint32_t B, uint32_t p,
                                                 uint32_t h, const float (&x)[16]) {
    constexpr uint32_t V = C < 4 ? C : 4;
#pragma unroll
    for (uint32_t q = 0; q < 4; ++q)
#pragma unroll
        for (uint32_t i = 0; i < 4; i += V) {
            const uint32_t f = 8u * q + 4u * h + i;
            float v[V];
#pragma unroll
            for (uint32_t k = 0; k < V; ++k) v[k] = x[4 * q + i + k];
            store_vec<FT, V>(feat + ((size_t)(f / C) * B + p) * C + (f % C), v);
        }
}

__device__ __forceinline__ float wave_sum32(float v) {   // sum over the 32 lanes that share a lane half
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// Per-wave depth buffer: the S depths of the current ray are evaluated once (lanes stride the samples) and kept in LDS,
// so the sample distances of all tiles of the ray are two LDS reads instead of two jitter/depth evaluations per point.
constexpr uint32_t kMaxSamplesLds = 1024;
__device__ __forceinline__ void fill_depths(const SrcRays &src, uint32_t r, float near, float far, float *zbuf, uint32_t lane) {
    for (uint32_t s = lane; s < src.S; s += 64u) zbuf[s] = src.depth(r, s, near, far);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ float buffered_dist(const float *zbuf, uint32_t s, uint32_t S, float dnorm) {
    if (s + 1u >= S) return 1e-10f * dnorm;
    return (zbuf[s + 1u] - zbuf[s]) * dnorm;
}

// dist of sample s on ray r (render.py:192-194): (z[s+1]-z[s]) * |d|, last sample 1e-10 * |d|
__device__ __forceinline__ float sample_dist(const SrcRays &src, uint32_t r, uint32_t s, float near, float far, float dnorm) {
    if (s + 1u >= src.S) return 1e-10f * dnorm;
    return (src.depth(r, s + 1u, near, far) - src.depth(r, s, near, far)) * dnorm;
}

// Per-sample outputs of the forward pass (what the fine pass consumes, render.py:113-126,203-211): the network output
// sigma[r,s] and the RUNNING optical depth tau[r,s] = sum_{s' <= s} sigma * dist.  The W samples of a tile sit in W
// consecutive lanes; the running sum is an inclusive wave prefix sum over them (log2 W shuffle steps) plus the carry
// of the ray's earlier tiles.  Every lane of the wave takes part in the shuffles; `writer` lanes store.  Returns the new carry.
template <uint32_t W>
__device__ __forceinline__ float emit_samples(float term, float sigma, float carry, bool writer, uint32_t pos,
                                              float *__restrict__ sigma_out, float *__restrict__ depth_out, size_t idx) {
    float scan = term;
#pragma unroll
    for (uint32_t d = 1; d < W; d <<= 1) {
        const float up = __shfl_up(scan, d, W);
        if (pos >= d) scan += up;
    }
    scan += carry;
    if (writer) {
        if (sigma_out != nullptr) sigma_out[idx] = sigma;
        if (depth_out != nullptr) depth_out[idx] = scan;
    }
    return __shfl(scan, W - 1u, W);          // the tile's last sample carries the total so far (lanes past S add 0)
}

// Where point p of a point-list launch writes its output: identity, or (grid queries, SrcGrid traversal) the position of
// grid point p in the caller's [n0, n1, n2] array.
struct OutMap {
    uint32_t n0, n1, n2;          // n0 == 0: identity
    uint32_t first;               // grid queries in chunks: point p of the launch is point first + p of the traversal
    __device__ __forceinline__ size_t at(uint32_t p) const {
        if (n0 == 0u) return p;
        p += first;
        const uint32_t i0 = p % n0, rest = p / n0, i2 = rest % n2, i1 = rest / n2;
        return ((size_t)i0 * n1 + i1) * n2 + i2;
    }
};

// ---- 2: MLP forward + line integral ----------------------------------------------------------------------
// kRays: one wave per ray, acc[r] = sum_s sigma*dist.   !kRays: plain point list, out[p] = sigma(p).
template <typename P, uint32_t C, bool kRays>
__global__ void __launch_bounds__(256, 2)        // 3 waves per SIMD cost the fp32 kernel 91 spilled registers per lane
mlp_forward_kernel(const typename P::feat_t::store_t *__restrict__ feat, const float *__restrict__ mlp, SrcRays src,
                   float *__restrict__ out, float *__restrict__ sigma_out, float *__restrict__ depth_out, uint32_t n_items,
                   uint32_t B, int act, OutMap omap) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    MlpShared<P>::build(smem, mlp, 4);
    const uint32_t lane = threadIdx.x & 63u, n = lane & 31u, h = lane >> 5;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = (gridDim.x * blockDim.x) >> 6;
    typename P::Frag x0f;
    float x0[16], h1[16], h2[16], h3[16];

    if constexpr (kRays) {
        const uint32_t S = src.S, tiles = (S + 31u) / 32u;
        const bool use_zbuf = S <= kMaxSamplesLds;
        float *zbuf = reinterpret_cast<float *>(smem + ((MlpShared<P>::kBytes + 15u) & ~15u)) + (threadIdx.x >> 6) * kMaxSamplesLds;
        FeatRaw<typename P::feat_t, C> ahead;                 // features of the next tile, in flight during this one
        if (wave < n_items) load_feat_raw<typename P::feat_t, C>(feat, B, wave * S + min(n, S - 1u), h, ahead);
        for (uint32_t r = wave; r < n_items; r += n_waves) {
            const float *ray = src.rays + (size_t)r * 8;
            const float near = ray[6], far = ray[7];
            const float dnorm = sqrtf(ray[3] * ray[3] + ray[4] * ray[4] + ray[5] * ray[5]);
            if (use_zbuf) fill_depths(src, r, near, far, zbuf, lane);
            float part = 0.0f, carry = 0.0f;
            for (uint32_t k = 0; k < tiles; ++k) {
                const uint32_t s = 32u * k + n;
                const bool valid = s < S;
                const FeatRaw<typename P::feat_t, C> now = ahead;
                {   // next tile of this ray, else first tile of this wave's next ray, else (nothing left) this tile again
                    const bool more = k + 1u < tiles;
                    const uint32_t rn = more ? r : (r + n_waves < n_items ? r + n_waves : r);
                    const uint32_t sn = more ? s + 32u : n;
                    load_feat_raw<typename P::feat_t, C>(feat, B, rn * S + min(sn, S - 1u), h, ahead);
                }
                unpack_feat_raw<typename P::feat_t, C>(now, x0);
                const float z4 = mlp_tile_forward<P>(smem, lane, x0, x0f, h1, h2, h3);
                const float sigma = last_act(act, z4);
                const float term = !valid ? 0.0f
                                 : sigma * (use_zbuf ? buffered_dist(zbuf, s, S, dnorm) : sample_dist(src, r, s, near, far, dnorm));
                if (sigma_out != nullptr || depth_out != nullptr)          // per-sample outputs (wave-uniform test)
                    carry = emit_samples<32>(term, sigma, carry, valid && h == 0, n, sigma_out, depth_out, (size_t)r * S + s);
                if (h == 0) part += term;
            }
            part = wave_sum32(part);
            if (lane == 0) out[r] = part;
        }
    } else {
        const uint32_t tiles = (n_items + 31u) / 32u;
        for (uint32_t k = wave; k < tiles; k += n_waves) {
            const uint32_t p0 = 32u * k + n;
            const bool valid = p0 < n_items;
            const uint32_t p = valid ? p0 : n_items - 1u;
            load_feat_slots<typename P::feat_t, C>(feat, B, p, h, x0);
            const float z4 = mlp_tile_forward<P>(smem, lane, x0, x0f, h1, h2, h3);
            if (valid && h == 0) out[omap.at(p)] = last_act(act, z4);
        }
    }
}

// ---- 2b: the same on 16-point tiles (bf16 mode, C = 2; field_mlp16.h) -------------------------------------------
template <bool kRays>
__global__ void __launch_bounds__(256, 4)
mlp16_forward_kernel(const uint16_t *__restrict__ feat, const float *__restrict__ mlp, SrcRays src, float *__restrict__ out,
                     float *__restrict__ sigma_out, float *__restrict__ depth_out, uint32_t n_items, uint32_t B, int act,
                     OutMap omap) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t lane = threadIdx.x & 63u, c = lane & 15u, g = lane >> 4;
    const uint32_t wave = __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) >> 6), n_waves = (gridDim.x * blockDim.x) >> 6;
    Act16 a;
    if constexpr (kRays) {
        const uint32_t S = src.S, tiles = (S + 15u) / 16u;
        const bool use_zbuf = S <= kMaxSamplesLds;
        float *zbuf = reinterpret_cast<float *>(smem + ((Mlp16Shared::kBytes + 15u) & ~15u)) + (threadIdx.x >> 6) * kMaxSamplesLds;
        Feat16Raw ahead;                                     // features of the next tile, in flight during this one
        if (wave < n_items) load_feat16(feat, B, wave * S + min(c, S - 1u), g, ahead);
        // (after the first feature request: the two round trips overlap; the depth buffers are filled later: they stage the weights)
        Mlp16Shared::build_staged<4>(smem, mlp, reinterpret_cast<float *>(smem + ((Mlp16Shared::kBytes + 15u) & ~15u)));
        Mlp16InRegs wt;
        wt.load(smem, lane);
        for (uint32_t r = wave; r < n_items; r += n_waves) {
            const float *ray = src.rays + (size_t)r * 8;
            const float near = ray[6], far = ray[7];
            const float dnorm = sqrtf(ray[3] * ray[3] + ray[4] * ray[4] + ray[5] * ray[5]);
            if (use_zbuf) fill_depths(src, r, near, far, zbuf, lane);
            float part = 0.0f, carry = 0.0f;
            if (sigma_out == nullptr && depth_out == nullptr) {            // (wave-uniform) line integrals only: training, projections
                // All four lane groups of a tile hold the same z4, so the activation and the sample distance -- 40 of a tile's ~125
                // vector instructions -- would be evaluated four times over.  Four tiles share one evaluation instead: lane group j
                // keeps the z4 of tile k0 + j, i.e. lane (c, g) ends up with sample 16 (k0 + g) + c.  The terms then return to lane
                // group 0 row by row (v_permlane*_swap) and are added in tile order -- the order of the loop below: same bits.
                for (uint32_t k0 = 0; k0 < tiles; k0 += 4u) {
                    float zsel = 0.0f;
#pragma unroll
                    for (uint32_t j = 0; j < 4u; ++j) {
                        const uint32_t k = k0 + j;
                        if (k >= tiles) break;                               // uniform
                        const uint32_t sk = 16u * k + c;
                        const Feat16Raw now = ahead;
                        {   // next tile of this ray, else first tile of this wave's next ray, else (nothing left) this tile again
                            const bool more = k + 1u < tiles;
                            const uint32_t rn = more ? r : (r + n_waves < n_items ? r + n_waves : r);
                            const uint32_t sn = more ? sk + 16u : c;
                            load_feat16(feat, B, rn * S + min(sn, S - 1u), g, ahead);
                        }
                        const float z4 = mlp16_tile_forward(wt, feat16_operand(now), a);
                        zsel = g == j ? z4 : zsel;
                    }
                    const uint32_t s = 16u * (k0 + g) + c;
                    const float sigma = last_act16(act, zsel);
                    const float term = s >= S ? 0.0f
                                     : sigma * (use_zbuf ? buffered_dist(zbuf, s, S, dnorm) : sample_dist(src, r, s, near, far, dnorm));
                    const uint32_t tb = __float_as_uint(term);
                    const auto r16 = __builtin_amdgcn_permlane16_swap(tb, tb, false, false);          // [1]: rows 1 1 3 3
                    const auto r32 = __builtin_amdgcn_permlane32_swap(tb, tb, false, false);          // [1]: rows 2 3 2 3
                    const auto r48 = __builtin_amdgcn_permlane16_swap(r32[1], r32[1], false, false);  // [1]: rows 3 3 3 3
                    if (g == 0u) {
                        part += term;
                        if (k0 + 1u < tiles) part += __uint_as_float(r16[1]);
                        if (k0 + 2u < tiles) part += __uint_as_float(r32[1]);
                        if (k0 + 3u < tiles) part += __uint_as_float(r48[1]);
                    }
                }
            } else
            for (uint32_t k = 0; k < tiles; ++k) {
                const uint32_t s = 16u * k + c;
                const bool valid = s < S;
                const Feat16Raw now = ahead;
                {   // next tile of this ray, else first tile of this wave's next ray, else (nothing left) this tile again
                    const bool more = k + 1u < tiles;
                    const uint32_t rn = more ? r : (r + n_waves < n_items ? r + n_waves : r);
                    const uint32_t sn = more ? s + 16u : c;
                    load_feat16(feat, B, rn * S + min(sn, S - 1u), g, ahead);
                }
                const float z4 = mlp16_tile_forward(wt, feat16_operand(now), a);
                const float sigma = last_act16(act, z4);
                const float term = !valid ? 0.0f
                                 : sigma * (use_zbuf ? buffered_dist(zbuf, s, S, dnorm) : sample_dist(src, r, s, near, far, dnorm));
                if (sigma_out != nullptr || depth_out != nullptr)          // per-sample outputs (wave-uniform test)
                    carry = emit_samples<16>(term, sigma, carry, valid && g == 0u, c, sigma_out, depth_out, (size_t)r * S + s);
                if (g == 0u) part += term;
            }
#pragma unroll
            for (int off = 8; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);     // the 16 points of lane group 0
            if (lane == 0) out[r] = part;
        }
    } else {
        Mlp16Shared::build_staged<4>(smem, mlp, reinterpret_cast<float *>(smem + ((Mlp16Shared::kBytes + 15u) & ~15u)));
        Mlp16InRegs wt;
        wt.load(smem, lane);
        const uint32_t tiles = (n_items + 15u) / 16u;
        for (uint32_t k = wave; k < tiles; k += n_waves) {
            const uint32_t p0 = 16u * k + c;
            const bool valid = p0 < n_items;
            const uint32_t p = valid ? p0 : n_items - 1u;
            Feat16Raw raw;
            load_feat16(feat, B, p, g, raw);
            const float z4 = mlp16_tile_forward(wt, feat16_operand(raw), a);
            if (valid && g == 0u) out[omap.at(p)] = last_act16(act, z4);
        }
    }
}

// ---- 2b': the same line integrals with FOUR waves per ray (steps with fewer rays than two waves per SIMD: the reference's 1 024) ---
// One wave per ray walks a ray's tiles one after the other, and with one wave per SIMD nothing hides the dependent MFMA / epilogue
// chain of a tile: the forward of a 1 024-ray step is a single ray's latency (12 tiles, ~18 000 cycles) whatever the chip could do
// beside it.  Here a workgroup takes a ray at a time and its four waves a quarter of the tiles each; the terms sigma * dist meet in
// LDS and wave 0 adds them in tile order, then across the 16 lanes like mlp16_forward_kernel does: the same sums in the same order,
// bit-identical line integrals.  Each wave fills the depths of its own tiles (+ 1) in its depth buffer; the term buffers (double-
// buffered: one barrier per ray) live in the upper halves of the first two depth buffers, which is why S <= kMaxSamplesLds / 2.
__global__ void __launch_bounds__(256, 4)
mlp16_forward_split_kernel(const uint16_t *__restrict__ feat, const float *__restrict__ mlp, SrcRays src, float *__restrict__ out,
                           uint32_t n_rays, uint32_t B, int act) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t lane = threadIdx.x & 63u, c = lane & 15u, g = lane >> 4, wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t S = src.S, tiles = (S + 15u) / 16u;
    const uint32_t k_begin = wib * tiles / 4u, k_end = (wib + 1u) * tiles / 4u;
    float *zall = reinterpret_cast<float *>(smem + ((Mlp16Shared::kBytes + 15u) & ~15u));
    float *zbuf = zall + wib * kMaxSamplesLds;
    float *terms[2] = {zall + kMaxSamplesLds / 2u, zall + kMaxSamplesLds + kMaxSamplesLds / 2u};
    Act16 a;
    Feat16Raw ahead;
    if (blockIdx.x < n_rays) load_feat16(feat, B, blockIdx.x * S + min(16u * k_begin + c, S - 1u), g, ahead);
    Mlp16Shared::build_staged<4>(smem, mlp, zall);
    Mlp16InRegs wt;
    wt.load(smem, lane);
    uint32_t turn = 0u;
    for (uint32_t r = blockIdx.x; r < n_rays; r += gridDim.x, turn ^= 1u) {
        const float *ray = src.rays + (size_t)r * 8;
        const float near = ray[6], far = ray[7];
        const float dnorm = sqrtf(ray[3] * ray[3] + ray[4] * ray[4] + ray[5] * ray[5]);
        for (uint32_t s = 16u * k_begin + lane; s < min(S, 16u * k_end + 1u); s += 64u) zbuf[s] = src.depth(r, s, near, far);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        float *T = terms[turn];
        for (uint32_t k0 = k_begin; k0 < k_end; k0 += 4u) {
            float zsel = 0.0f;
#pragma unroll
            for (uint32_t j = 0; j < 4u; ++j) {
                const uint32_t k = k0 + j;
                if (k >= k_end) break;                                   // uniform
                const Feat16Raw now = ahead;
                {   // next tile of this wave's range, else the first one of the workgroup's next ray, else (nothing left) this tile again
                    const bool more = k + 1u < k_end;
                    const uint32_t rn = more ? r : (r + gridDim.x < n_rays ? r + gridDim.x : r);
                    const uint32_t sn = 16u * (more ? k + 1u : k_begin) + c;
                    load_feat16(feat, B, rn * S + min(sn, S - 1u), g, ahead);
                }
                const float z4 = mlp16_tile_forward(wt, feat16_operand(now), a);
                zsel = g == j ? z4 : zsel;
            }
            const uint32_t s = 16u * (k0 + g) + c;
            const float sigma = last_act16(act, zsel);
            const float term = s >= S ? 0.0f : sigma * buffered_dist(zbuf, s, S, dnorm);
            if (k0 + g < k_end) T[s] = term;
        }
        __syncthreads();
        if (wib == 0u) {
            float part = 0.0f;
            if (g == 0u)
                for (uint32_t k = 0; k < tiles; ++k) part += T[16u * k + c];
#pragma unroll
            for (int off = 8; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);     // the 16 points of lane group 0
            if (lane == 0) out[r] = part;
        }
        // (no second barrier: the next ray's terms go to the other buffer, and this buffer is written again only behind the next
        // ray's barrier, which wave 0 reaches after it has read it)
    }
}

// ---- 2c: gathers + MLP + line integral in one kernel, features in registers (fused_forward.h; forward-only calls) -----------
// Src = SrcRays: one wave per ray (out[r] = sum_s sigma * dist, optional per-sample outputs).  Any other source: a plain point
// list / generated grid, out[omap.at(p)] = sigma(p).  `feat` != nullptr additionally stores the features (diagnostic:
// NAF_CFG_FUSED_STORE_FEATURES times what the feature traffic itself costs this kernel).
template <typename TT, typename Src>
__global__ void __launch_bounds__(256, 3)
fused_forward_kernel(Src src, const typename TT::store_t *__restrict__ table, const int32_t *__restrict__ offsets,
                     const float *__restrict__ mlp, float *__restrict__ out, float *__restrict__ sigma_out,
                     float *__restrict__ depth_out, uint16_t *__restrict__ feat, uint32_t n_items, uint32_t B, uint32_t H, int act,
                     OutMap omap) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr uint32_t kShAligned = (Mlp16Shared::kBytes + 15u) & ~15u;
    LevelRec *recs = reinterpret_cast<LevelRec *>(smem + kShAligned);
    build_level_recs(recs, offsets, kFusedLevels, H);
    Mlp16Shared::build<4>(smem, mlp);                          // ends with a workgroup barrier: the level records are visible too
    const uint32_t lane = threadIdx.x & 63u, c = lane & 15u, g = lane >> 4;
    const uint32_t wave = __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) >> 6), n_waves = (gridDim.x * blockDim.x) >> 6;
    const uint32_t table_rows = (uint32_t)offsets[kFusedLevels];
    Act16 a;
    if constexpr (std::is_same<Src, SrcRays>::value) {
        const uint32_t S = src.S, tiles = (S + 15u) / 16u;
        const bool use_zbuf = S <= kMaxSamplesLds;
        float *zbuf = reinterpret_cast<float *>(smem + kShAligned + kFusedLevels * (uint32_t)sizeof(LevelRec)) + (threadIdx.x >> 6) * kMaxSamplesLds;
        for (uint32_t r = wave; r < n_items; r += n_waves) {
            const float4 *ray = reinterpret_cast<const float4 *>(src.rays + (size_t)r * 8);
            const float4 ra = ray[0], rc = ray[1];
            const float near = rc.z, far = rc.w;
            const float dnorm = sqrtf(ra.w * ra.w + rc.x * rc.x + rc.y * rc.y);
            if (use_zbuf) fill_depths(src, r, near, far, zbuf, lane);
            float part = 0.0f, carry = 0.0f;
            for (uint32_t k = 0; k < tiles; ++k) {
                const uint32_t s = 16u * k + c;
                const bool valid = s < S;
                const uint32_t sc = valid ? s : S - 1u;
                float x[3];
                src.position(ra, rc, use_zbuf ? zbuf[sc] : src.depth(r, sc, near, far), x);
                const Feat16Raw f = gather_point_features<TT>(recs, g, x, table, table_rows);
                if (feat != nullptr && valid) store_point_features(feat, B, r * S + s, g, f);
                const float z4 = mlp16_tile_forward(smem, lane, feat16_operand(f), a);
                const float sigma = last_act16(act, z4);
                const float term = !valid ? 0.0f
                                 : sigma * (use_zbuf ? buffered_dist(zbuf, s, S, dnorm) : sample_dist(src, r, s, near, far, dnorm));
                if (sigma_out != nullptr || depth_out != nullptr)          // per-sample outputs (wave-uniform test)
                    carry = emit_samples<16>(term, sigma, carry, valid && g == 0u, c, sigma_out, depth_out, (size_t)r * S + s);
                if (g == 0u) part += term;
            }
#pragma unroll
            for (int off = 8; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);     // the 16 points of lane group 0
            if (lane == 0) out[r] = part;
        }
    } else {
        const uint32_t tiles = (n_items + 15u) / 16u;
        for (uint32_t k = wave; k < tiles; k += n_waves) {
            const uint32_t p0 = 16u * k + c;
            const bool valid = p0 < n_items;
            const uint32_t p = valid ? p0 : n_items - 1u;
            float x[3];
            src.get(p, x);
            const Feat16Raw f = gather_point_features<TT>(recs, g, x, table, table_rows);
            if (feat != nullptr && valid) store_point_features(feat, B, p, g, f);
            const float z4 = mlp16_tile_forward(smem, lane, feat16_operand(f), a);
            if (valid && g == 0u) out[omap.at(p)] = last_act16(act, z4);
        }
    }
}

// ---- 3: MLP backward --------------------------------------------------------------------------------------
// slab layout == parameter block layout (kW0 .. kB3), one slab of kSlabStride floats per workgroup.
// Training steps hand the backward kernels what the loss needs instead of a precomputed d loss / d acc: the masked squared error
// of train.py:127 / loss.py:37 in weighted form, loss = sum_r w_r (acc_r - y_r)^2, d loss / d acc_r = 2 w_r (acc_r - y_r), is two
// flops per ray -- a kernel launch of its own (loss_grad_kernel) cost a fortieth of the reference-size step.
struct LossInputs { const float *acc, *target, *weight; };
constexpr uint32_t kClearWords = 33;               // overflow counters of the binned scatter: total + one per level (Workspace::overflow)

template <typename P>
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <typename P, uint32_t C>
__global__ void __launch_bounds__(256)
mlp_backward_kernel(const typename P::feat_t::store_t *__restrict__ feat, const float *__restrict__ mlp, SrcRays src,
                    const float *__restrict__ grad_acc, LossInputs loss, typename P::feat_t::store_t *__restrict__ dfeat,
                    float *__restrict__ slabs, uint32_t n_rays, uint32_t B, int act, uint32_t *__restrict__ clear_words) {
    using Sh = MlpShared<P>;
    if (clear_words != nullptr && blockIdx.x == 0u && threadIdx.x < kClearWords) clear_words[threadIdx.x] = 0u;      // see StepExtras
    using TR = typename P::tr_t;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    Sh::build(smem, mlp, 8);
    const uint32_t lane = threadIdx.x & 63u, n = lane & 31u, h = lane >> 5, wib = threadIdx.x >> 6;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = (gridDim.x * blockDim.x) >> 6;
    constexpr uint32_t kImg = 32u * P::kTrPitch;                                  // elements per transpose image
    constexpr uint32_t kShAligned = (Sh::kBytes + 15u) & ~15u;
    TR *imgA = reinterpret_cast<TR *>(smem + kShAligned) + (size_t)wib * 3u * kImg;   // gradient tile  G
    TR *imgB = imgA + kImg;                                                        // input tile     X0
    TR *imgC = imgB + kImg;                                                        // hidden tile    H2 / H1
    constexpr uint32_t kImgBytes = ((4u * 3u * kImg * (uint32_t)sizeof(TR)) + 15u) & ~15u;
    float *zbuf = reinterpret_cast<float *>(smem + kShAligned + kImgBytes) + wib * kMaxSamplesLds;
    const bool use_zbuf = src.S <= kMaxSamplesLds;

    // weight-gradient accumulators: lane (c,h), register t  <->  dW[out = slot_row(t,h)][in = c]
    f32x16 dW0 = {0}, dW1 = {0}, dW2a = {0}, dW2b = {0};
    float db0 = 0.0f, db1 = 0.0f, db2 = 0.0f, db3 = 0.0f;
    uint32_t dmax = 0u;                                      // bit pattern of max |feature gradient| (scale of the binned scatter);
                                                             // compared as integers so that Inf / NaN win and stay visible
    float loss_part = 0.0f;                                  // this wave's rays: sum w (acc - y)^2
    float dw3[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) dw3[t] = 0.0f;
    float w3s[16];
    Sh::slot_vector(smem, 3, h, w3s);

    const uint32_t S = src.S, tiles = (S + 31u) / 32u;
    FeatRaw<typename P::feat_t, C> ahead;                     // features of the next tile, in flight during this one
    if (wave < n_rays) load_feat_raw<typename P::feat_t, C>(feat, B, wave * S + min(n, S - 1u), h, ahead);
    for (uint32_t r = wave; r < n_rays; r += n_waves) {
        const float *ray = src.rays + (size_t)r * 8;
        const float near = ray[6], far = ray[7];
        const float dnorm = sqrtf(ray[3] * ray[3] + ray[4] * ray[4] + ray[5] * ray[5]);
        float dacc;
        if (loss.target != nullptr) {                        // training step: the loss lives here (wave-uniform arithmetic)
            const float err = loss.acc[r] - loss.target[r], w = loss.weight[r];
            dacc = 2.0f * w * err;
            loss_part += w * err * err;
        } else dacc = grad_acc[r];
        if (use_zbuf) fill_depths(src, r, near, far, zbuf, lane);

        for (uint32_t k = 0; k < tiles; ++k) {
            const uint32_t s = 32u * k + n;
            const bool valid = s < S;
            const uint32_t p = r * S + (valid ? s : S - 1u);
            float x0[16], h1[16], h2[16], h3[16];
            typename P::Frag x0f;
            const FeatRaw<typename P::feat_t, C> now = ahead;
            {   // next tile of this ray, else first tile of this wave's next ray, else (nothing left) this tile again
                const bool more = k + 1u < tiles;
                const uint32_t rn = more ? r : (r + n_waves < n_rays ? r + n_waves : r);
                const uint32_t sn = more ? s + 32u : n;
                load_feat_raw<typename P::feat_t, C>(feat, B, rn * S + min(sn, S - 1u), h, ahead);
            }
            unpack_feat_raw<typename P::feat_t, C>(now, x0);
            const float z4 = mlp_tile_forward<P>(smem, lane, x0, x0f, h1, h2, h3);
            const float sigma = last_act(act, z4);
            const float gsig = !valid ? 0.0f
                             : dacc * (use_zbuf ? buffered_dist(zbuf, s, S, dnorm) : sample_dist(src, r, s, near, far, dnorm));
            const float g4 = gsig * last_act_grad(act, z4, sigma);

            // output layer: dw3 += g4 * h3, db3 += g4 ; G3 = (w3 g4) * lrelu'(z3)
            float g[16];
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                dw3[t] = __fmaf_rn(g4, h3[t], dw3[t]);
                g[t] = w3s[t] * g4 * (h3[t] > 0.0f ? 1.0f : kLeaky);
            }
            if (h == 0) db3 += g4;

            // transpose G3, X0, H2 through LDS (P::tr_put / P::tr_load): the weight gradients contract over points
            const typename P::Frag g3f = P::pack(g);
            P::tr_put(imgA, n, h, g, g3f);
            P::tr_put(imgB, n, h, x0, x0f);
            P::tr_put(imgC, n, h, h2, P::pack(h2));
            wave_lds_fence<P>();
            typename P::Frag gA = P::tr_load(imgA, n, h);             // A[m = out][k = point]
            const typename P::Frag xB = P::tr_load(imgB, n, h);        // B[k = point][n = in]
            typename P::Frag hB = P::tr_load(imgC, n, h);
            dW2a = P::mma(gA, xB, dW2a);
            dW2b = P::mma(gA, hB, dW2b);
            db2 += P::frag_sum(gA);
            wave_lds_fence<P>();

            // back through layer 2 (skip layer): d[input] and d[h2]
            f32x16 zero = {0};
            f32x16 dx0 = P::mma(P::load_wfrag(smem, kFW2aT, lane), g3f, zero);
            f32x16 dh = P::mma(P::load_wfrag(smem, kFW2bT, lane), g3f, zero);
#pragma unroll
            for (int t = 0; t < 16; ++t) g[t] = dh[t] * (h2[t] > 0.0f ? 1.0f : kLeaky);      // G2

            const typename P::Frag g2f = P::pack(g);
            P::tr_put(imgA, n, h, g, g2f);
            P::tr_put(imgC, n, h, h1, P::pack(h1));
            wave_lds_fence<P>();
            gA = P::tr_load(imgA, n, h);
            hB = P::tr_load(imgC, n, h);
            dW1 = P::mma(gA, hB, dW1);
            db1 += P::frag_sum(gA);
            wave_lds_fence<P>();

            dh = P::mma(P::load_wfrag(smem, kFW1T, lane), g2f, zero);
#pragma unroll
            for (int t = 0; t < 16; ++t) g[t] = dh[t] * (h1[t] > 0.0f ? 1.0f : kLeaky);      // G1
            const typename P::Frag g1f = P::pack(g);
            P::tr_put(imgA, n, h, g, g1f);
            wave_lds_fence<P>();
            gA = P::tr_load(imgA, n, h);
            dW0 = P::mma(gA, xB, dW0);
            db0 += P::frag_sum(gA);
            wave_lds_fence<P>();

            dx0 = P::mma(P::load_wfrag(smem, kFW0T, lane), g1f, dx0);
            if (valid) {
                float o[16];
#pragma unroll
                for (int t = 0; t < 16; ++t) { o[t] = dx0[t]; dmax = max(dmax, __float_as_uint(o[t]) & 0x7fffffffu); }
                store_feat_slots<typename P::feat_t, C>(dfeat, B, p, h, o);
            }
        }
    }

#pragma unroll
    for (int off = 32; off > 0; off >>= 1) dmax = max(dmax, (uint32_t)__shfl_xor((int)dmax, off, 64));     // non-negative floats order like uints

    // ---- fold the 4 waves of the workgroup into one slab (fixed order -> deterministic), then one store ------
    __syncthreads();                                        // everyone is done with the transpose images
    float *red = reinterpret_cast<float *>(smem + kShAligned);          // kSlabStride floats, reuses the images
    // per-lane partial sums that still need a cross-lane reduction
#pragma unroll
    for (int t = 0; t < 16; ++t) dw3[t] = wave_sum32(dw3[t]);          // over the 32 points of a lane half
    db0 += __shfl_xor(db0, 32, 64);                                     // the two lane halves hold 16 points each
    db1 += __shfl_xor(db1, 32, 64);
    db2 += __shfl_xor(db2, 32, 64);
    db3 = wave_sum32(db3) ;
    db3 += __shfl_xor(db3, 32, 64);
    for (uint32_t w = 0; w < (blockDim.x >> 6); ++w) {
        if (wib == w) {
            const bool first = w == 0;
#pragma unroll
            for (uint32_t t = 0; t < 16; ++t) {
                const uint32_t o = slot_row(t, h);
                const uint32_t i0 = kW0 + o * 32u + n, i1 = kW1 + o * 32u + n, i2 = kW2 + o * 64u + n;
                red[i0] = (first ? 0.0f : red[i0]) + dW0[t];
                red[i1] = (first ? 0.0f : red[i1]) + dW1[t];
                red[i2] = (first ? 0.0f : red[i2]) + dW2a[t];
                red[i2 + 32u] = (first ? 0.0f : red[i2 + 32u]) + dW2b[t];
                if (n == 0) red[kW3 + o] = (first ? 0.0f : red[kW3 + o]) + dw3[t];
            }
            if (h == 0) {                                    // db*: lane n holds the sum for feature n
                red[kB0 + n] = (first ? 0.0f : red[kB0 + n]) + db0;
                red[kB1 + n] = (first ? 0.0f : red[kB1 + n]) + db1;
                red[kB2 + n] = (first ? 0.0f : red[kB2 + n]) + db2;
            }
            if (lane == 0) {
                red[kB3] = (first ? 0.0f : red[kB3]) + db3;
                red[kSlabLoss] = (first ? 0.0f : red[kSlabLoss]) + loss_part;
                red[kSlabGmax] = __uint_as_float(first ? dmax : max(dmax, __float_as_uint(red[kSlabGmax])));
            }
        }
        __syncthreads();
    }
    float *slab = slabs + (size_t)blockIdx.x * kSlabStride;
    for (uint32_t i = threadIdx.x; i <= kSlabGmax; i += blockDim.x) slab[i] = red[i];
}

// ---- 3b: MLP backward on 16-point tiles (bf16 mode, C = 2; field_mlp16.h) ----------------------------------------
// Same outputs as mlp_backward_kernel (dfeat, one dW slab per workgroup, max |dfeat|) at two or more waves per SIMD.
__global__ void __launch_bounds__(256, 3)                     // 168 VGPRs (11 spilled dwords): three waves per SIMD
mlp16_backward_kernel(const uint16_t *__restrict__ feat, const float *__restrict__ mlp, SrcRays src,
                      const float *__restrict__ grad_acc, LossInputs loss, uint16_t *__restrict__ dfeat, float *__restrict__ slabs,
                      uint32_t n_rays, uint32_t B, int act, uint32_t parts, uint32_t *__restrict__ clear_words) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
#ifdef NAF_MLP_STAMPS        // diagnostic builds (tools/mlp_stamps.py): shader-clock stamps of the kernel's phases, written behind the slab's payload
    long long stamp_[6];
    const long long wall0_ = wall_clock64();
#define NAF_STAMP(i) stamp_[i] = clock64()
#else
#define NAF_STAMP(i)
#endif
    NAF_STAMP(0);
    if (clear_words != nullptr && blockIdx.x == 0u && threadIdx.x < kClearWords) clear_words[threadIdx.x] = 0u;      // see StepExtras
    const uint32_t lane = threadIdx.x & 63u, c = lane & 15u, g = lane >> 4, wib = threadIdx.x >> 6;
    // wave-uniform values are made scalar explicitly: the ray record, its depths range and d acc then live in SGPRs
    const uint32_t wave = __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) >> 6), n_waves = (gridDim.x * blockDim.x) >> 6;
    constexpr uint32_t kShAligned = (Mlp16Shared::kBytes + 15u) & ~15u;
    constexpr uint32_t kImg = 16u * 64u;                                           // bytes per transpose image
    unsigned char *imgG = smem + kShAligned + wib * 3u * kImg;                     // gradient tile G3 / G2 / G1
    unsigned char *imgX = imgG + kImg;                                             // input tile X0
    unsigned char *imgH = imgX + kImg;                                             // hidden tile H2 / H1
    float *zbuf = reinterpret_cast<float *>(smem + kShAligned + 4u * 3u * kImg) + wib * kMaxSamplesLds;
    const bool use_zbuf = src.S <= kMaxSamplesLds;

    // weight-gradient accumulators: tile (o, q) covers outputs 16o.. and inputs 16q..; lane (col, grp), register i
    // <-> dW[out = 16 o + 4 grp + i][in = 16 q + col]
    const f32x4v zero4 = {0.0f, 0.0f, 0.0f, 0.0f};
    f32x4v dW0[2][2], dW1[2][2], dW2[2][4];
#pragma unroll
    for (int o = 0; o < 2; ++o) {
#pragma unroll
        for (int q = 0; q < 2; ++q) { dW0[o][q] = zero4; dW1[o][q] = zero4; }
#pragma unroll
        for (int q = 0; q < 4; ++q) dW2[o][q] = zero4;
    }
    float db[3][2] = {{0.0f, 0.0f}, {0.0f, 0.0f}, {0.0f, 0.0f}};                 // per lane: output 16o + (lane & 15), its 4 points
    float db3 = 0.0f, loss_part = 0.0f;
    uint32_t dmax = 0u;                                      // see mlp_backward_kernel
    f32x4v dw3lo = zero4, dw3hi = zero4;

    // A work item is one ray, or one of `parts` consecutive tile ranges of a ray when there are fewer rays than resident waves
    // (1 024-ray steps: three waves per ray, four tiles each; the backward of a sample needs nothing of its ray but d acc, so
    // the ranges are independent): more waves per SIMD to hide the latency of the dependent MFMA chain behind.
    const uint32_t S = src.S, tiles = (S + 15u) / 16u;
    const uint32_t n_items = n_rays * parts;
    auto ray_of = [&](uint32_t item) { return item / parts; };
    auto first_tile = [&](uint32_t part) { return part * tiles / parts; };
    Feat16Raw ahead;
    if (wave < n_items) load_feat16(feat, B, ray_of(wave) * S + min(16u * first_tile(wave - ray_of(wave) * parts) + c, S - 1u), g, ahead);
    // the weight fragments are built AFTER the first tile's features have been requested: the two round trips overlap (the build
    // ends with the workgroup barrier that makes the fragments visible)
    Mlp16Shared::build_staged<8>(smem, mlp, reinterpret_cast<float *>(smem + kShAligned));      // (the transpose images and depth buffers lend the space)
    NAF_STAMP(1);
    for (uint32_t item = wave; item < n_items; item += n_waves) {
        const uint32_t r = ray_of(item), part = item - r * parts, k_begin = first_tile(part), k_end = first_tile(part + 1u);
        const float *ray = src.rays + (size_t)r * 8;
        const float near = ray[6], far = ray[7];
        const float dnorm = sqrtf(ray[3] * ray[3] + ray[4] * ray[4] + ray[5] * ray[5]);
        float dacc;
        if (loss.target != nullptr) {                        // training step: the loss lives here (wave-uniform arithmetic)
            const float err = loss.acc[r] - loss.target[r], w = loss.weight[r];
            dacc = 2.0f * w * err;
            if (part == 0u) loss_part += w * err * err;                    // once per ray, not once per tile range
        } else dacc = grad_acc[r];
        if (use_zbuf) fill_depths(src, r, near, far, zbuf, lane);

        for (uint32_t k = k_begin; k < k_end; ++k) {
            const uint32_t s = 16u * k + c;
            const bool valid = s < S;
            const uint32_t p = r * S + (valid ? s : S - 1u);
            const Feat16Raw now = ahead;
            {   // the next tile of this item, or the first tile of the wave's next item (a harmless reload at the very end)
                const bool more = k + 1u < k_end;
                const uint32_t next = item + n_waves < n_items ? item + n_waves : item;
                const uint32_t rn = more ? r : ray_of(next);
                const uint32_t sn = more ? s + 16u : 16u * first_tile(next - ray_of(next) * parts) + c;
                load_feat16(feat, B, rn * S + min(sn, S - 1u), g, ahead);
            }
            const bf16x8 x0f = feat16_operand(now);
            // Weight fragments and biases are re-read from LDS for every tile: hoisted out of the loop (which the compiler
            // does when it can prove the addresses loop-invariant) they would pin ~90 registers and halve the occupancy.
            uint32_t tile_tag = 0;
            asm volatile("" : "+v"(tile_tag));
            const unsigned char *wsh = smem + tile_tag;
            Act16 a;
            const float z4 = mlp16_tile_forward(wsh, lane, x0f, a);
            const float sigma = last_act16(act, z4);
            const float gsig = !valid ? 0.0f
                             : dacc * (use_zbuf ? buffered_dist(zbuf, s, S, dnorm) : sample_dist(src, r, s, near, far, dnorm));
            const float g4 = gsig * last_act_grad(act, z4, sigma);

            // output layer: dw3 += g4 * h3, db3 += g4 ; G3 = (w3 g4) * lrelu'(z3)
            f32x4v glo, ghi;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                dw3lo[j] = __fmaf_rn(g4, a.h3lo[j], dw3lo[j]);
                dw3hi[j] = __fmaf_rn(g4, a.h3hi[j], dw3hi[j]);
                glo[j] = a.w3lo[j] * g4 * (a.h3lo[j] > 0.0f ? 1.0f : kLeaky);
                ghi[j] = a.w3hi[j] * g4 * (a.h3hi[j] > 0.0f ? 1.0f : kLeaky);
            }
            if (g == 0u) db3 += g4;

            // layer 2: dW2 = G3 . [X0; H2]^T over the 16 points of the tile
            bf16x8 gf = pack16(glo, ghi);
            tr16_put(imgG, c, g, gf);
            tr16_put(imgX, c, g, x0f);
            tr16_put(imgH, c, g, a.h2f);
            wave_lds_fence<PrecBF16>();
            i16x4v gA[2], xB[2], hB[2];
#pragma unroll
            for (uint32_t o = 0; o < 2; ++o) { gA[o] = tr16_get(imgG, lane, o); xB[o] = tr16_get(imgX, lane, o); hB[o] = tr16_get(imgH, lane, o); }
#pragma unroll
            for (int o = 0; o < 2; ++o) {
                dW2[o][0] = mma16k16(gA[o], xB[0], dW2[o][0]);
                dW2[o][1] = mma16k16(gA[o], xB[1], dW2[o][1]);
                dW2[o][2] = mma16k16(gA[o], hB[0], dW2[o][2]);
                dW2[o][3] = mma16k16(gA[o], hB[1], dW2[o][3]);
                db[2][o] = add_bf16x4(db[2][o], gA[o]);
            }
            wave_lds_fence<PrecBF16>();

            // back through layer 2 (skip layer): d[input] and d[h2]
            f32x4v dxlo = mma16(Mlp16Shared::frag(wsh, kFW2aT, 0, lane), gf, zero4);
            f32x4v dxhi = mma16(Mlp16Shared::frag(wsh, kFW2aT, 1, lane), gf, zero4);
            f32x4v dhlo = mma16(Mlp16Shared::frag(wsh, kFW2bT, 0, lane), gf, zero4);
            f32x4v dhhi = mma16(Mlp16Shared::frag(wsh, kFW2bT, 1, lane), gf, zero4);
            glo = leaky_grad4_packed(dhlo, a.h2f, 0);                              // G2
            ghi = leaky_grad4_packed(dhhi, a.h2f, 1);
            gf = pack16(glo, ghi);
            tr16_put(imgG, c, g, gf);
            tr16_put(imgH, c, g, a.h1f);
            wave_lds_fence<PrecBF16>();
#pragma unroll
            for (uint32_t o = 0; o < 2; ++o) { gA[o] = tr16_get(imgG, lane, o); hB[o] = tr16_get(imgH, lane, o); }
#pragma unroll
            for (int o = 0; o < 2; ++o) {
                dW1[o][0] = mma16k16(gA[o], hB[0], dW1[o][0]);
                dW1[o][1] = mma16k16(gA[o], hB[1], dW1[o][1]);
                db[1][o] = add_bf16x4(db[1][o], gA[o]);
            }
            wave_lds_fence<PrecBF16>();

            dhlo = mma16(Mlp16Shared::frag(wsh, kFW1T, 0, lane), gf, zero4);
            dhhi = mma16(Mlp16Shared::frag(wsh, kFW1T, 1, lane), gf, zero4);
            glo = leaky_grad4_packed(dhlo, a.h1f, 0);                              // G1
            ghi = leaky_grad4_packed(dhhi, a.h1f, 1);
            gf = pack16(glo, ghi);
            tr16_put(imgG, c, g, gf);
            wave_lds_fence<PrecBF16>();
#pragma unroll
            for (uint32_t o = 0; o < 2; ++o) { gA[o] = tr16_get(imgG, lane, o); xB[o] = tr16_get(imgX, lane, o); }   // X0 again: cheaper than keeping it
#pragma unroll
            for (int o = 0; o < 2; ++o) {
                dW0[o][0] = mma16k16(gA[o], xB[0], dW0[o][0]);
                dW0[o][1] = mma16k16(gA[o], xB[1], dW0[o][1]);
                db[0][o] = add_bf16x4(db[0][o], gA[o]);
            }
            wave_lds_fence<PrecBF16>();

            dxlo = mma16(Mlp16Shared::frag(wsh, kFW0T, 0, lane), gf, dxlo);
            dxhi = mma16(Mlp16Shared::frag(wsh, kFW0T, 1, lane), gf, dxhi);
            if (valid) {
#pragma unroll
                for (int j = 0; j < 4; ++j) dmax = max(dmax, max(__float_as_uint(dxlo[j]) & 0x7fffffffu, __float_as_uint(dxhi[j]) & 0x7fffffffu));
                store_feat16(dfeat, B, p, g, __builtin_bit_cast(uint4, pack16(dxlo, dxhi)));
            }
        }
    }

    NAF_STAMP(2);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) dmax = max(dmax, (uint32_t)__shfl_xor((int)dmax, off, 64));     // non-negative floats order like uints

    // ---- fold the 4 waves of the workgroup into one slab (wave order -> deterministic), then one store ---------------
#pragma unroll
    for (int l = 0; l < 3; ++l)
#pragma unroll
        for (int o = 0; o < 2; ++o) {                                               // over the four point groups of an output
            db[l][o] += __shfl_xor(db[l][o], 16, 64);
            db[l][o] += __shfl_xor(db[l][o], 32, 64);
        }
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) {                                         // over the 16 points of a lane group
#pragma unroll
        for (int j = 0; j < 4; ++j) { dw3lo[j] += __shfl_xor(dw3lo[j], off, 64); dw3hi[j] += __shfl_xor(dw3hi[j], off, 64); }
        db3 += __shfl_xor(db3, off, 64);
    }
    __syncthreads();                                                                // images / depth buffers are dead
    NAF_STAMP(3);
    // ---- fold the 4 waves of the workgroup into one slab, in wave order: ((0 + a0) + a1) + a2) + a3 per entry, as ever --------------
    // tools/mlp_stamps.py put the fold of rounds 2-4 -- the waves took turns adding their 64 + 17 values to one LDS image, `red[i] += x`,
    // a dependent read-add-write round trip each -- at 14 000 of the kernel's 52 900 cycles at the reference's batch (with the pass
    // that cleared the image); batching a turn's reads did not help (the turns stay dependent and three workgroups per CU take them at
    // once), an XOR swizzle against the 4-way bank conflict of the lane groups neither.  Now nobody reads what it has just written: a wave
    // STORES its values into an image of its own (no waits), and all 256 threads add two images entry by entry, coalesced.  There is
    // room for two images -- A over the weight fragments, which are dead too, B over the transpose images -- so waves 0 and 1 write
    // together and waves 2 and 3 follow one by one.  Same operands in the same order: the slab has the same bits.
    float *imgA = reinterpret_cast<float *>(smem), *imgB = reinterpret_cast<float *>(smem + kShAligned);
    static_assert(kShAligned >= (kSlabGmax + 1u) * 4u, "the fragment area holds one slab image");
    auto put = [&](float *img) {
#pragma unroll
        for (uint32_t o = 0; o < 2; ++o)
#pragma unroll
            for (uint32_t i = 0; i < 4; ++i) {
                const uint32_t out = 16u * o + 4u * g + i;
#pragma unroll
                for (uint32_t q = 0; q < 2; ++q) {
                    img[kW0 + out * 32u + 16u * q + c] = dW0[o][q][i];
                    img[kW1 + out * 32u + 16u * q + c] = dW1[o][q][i];
                }
#pragma unroll
                for (uint32_t q = 0; q < 4; ++q) img[kW2 + out * 64u + 16u * q + c] = dW2[o][q][i];
            }
        if (g == 0u) {
#pragma unroll
            for (uint32_t o = 0; o < 2; ++o) {
                img[kB0 + 16u * o + c] = db[0][o];
                img[kB1 + 16u * o + c] = db[1][o];
                img[kB2 + 16u * o + c] = db[2][o];
            }
        }
        if (c == 0u) {
#pragma unroll
            for (uint32_t j = 0; j < 4; ++j) {
                img[kW3 + 4u * g + j] = dw3lo[j];
                img[kW3 + 16u + 4u * g + j] = dw3hi[j];
            }
        }
        if (lane == 0u) {
            img[kB3] = db3;
            img[kSlabLoss] = loss_part;
            img[kSlabGmax] = __uint_as_float(dmax);
        }
    };
    // A (+)= B, entry by entry; the first time A's entries are a wave's raw values and stand for 0 + a0 (the cleared image of old: -0.0
    // becomes +0.0), the last time the sums go to the slab instead.  The maximum of the gradient bit patterns rides along.
    auto combine = [&](bool first_pair, float *to_slab) {
        constexpr uint32_t kPer = (kSlabGmax + 1u + 255u) / 256u;
        float x[kPer], y[kPer];
#pragma unroll
        for (uint32_t u = 0; u < kPer; ++u) {
            const uint32_t i = min(threadIdx.x + 256u * u, kSlabGmax);
            x[u] = imgA[i];
            y[u] = imgB[i];
        }
#pragma unroll
        for (uint32_t u = 0; u < kPer; ++u) {
            const uint32_t i = threadIdx.x + 256u * u;
            if (i > kSlabGmax) break;
            const float lhs = first_pair ? 0.0f + x[u] : x[u];
            const float v = i == kSlabGmax ? __uint_as_float(max(__float_as_uint(x[u]), __float_as_uint(y[u]))) : lhs + y[u];
            if (to_slab != nullptr) to_slab[i] = v; else imgA[i] = v;
        }
    };
    float *slab = slabs + (size_t)blockIdx.x * kSlabStride;
    if (wib == 0u) put(imgA);
    if (wib == 1u) put(imgB);
    __syncthreads();
    combine(true, nullptr);
    __syncthreads();
    if (wib == 2u) put(imgB);
    __syncthreads();
    combine(false, nullptr);
    __syncthreads();
    if (wib == 3u) put(imgB);
    __syncthreads();
    NAF_STAMP(4);
    combine(false, slab);
#ifdef NAF_MLP_STAMPS
    NAF_STAMP(5);
    if (threadIdx.x == 0u) {
        uint32_t *dbg = reinterpret_cast<uint32_t *>(slab) + 4300u;
        for (int i = 0; i < 6; ++i) dbg[i] = (uint32_t)(stamp_[i] - stamp_[0]);
        dbg[6] = (uint32_t)wall0_; dbg[7] = (uint32_t)wall_clock64();
    }
#endif
#undef NAF_STAMP
}

// ---- 5: slabs -> grad_mlp (+=), loss, gradient maximum: slab_reduce_block (mlp_slabs.h), as a launch of its own ----------------
// (training steps on the binned scatter let spare workgroups of scatter_bin_kernel do it instead: run_binned_scatter)
__global__ void __launch_bounds__(1024)
mlp_grad_reduce_kernel(SlabReduce sr) {
    __shared__ float part[kReduceGroups][kReduceParams];
    slab_reduce_block<kReduceGroups>(part, sr, blockIdx.x, threadIdx.x);
}

// ---- host side ----------------------------------------------------------------------------------------------
constexpr uint32_t kBackwardBlocks = 512;   // 2 workgroups of 4 waves per CU; also the number of dW slabs
constexpr uint32_t kBackwardBlocks16 = 768; // the 16-point bf16 kernel fits three workgroups per CU (0.89 -> 0.835 ms at 65 536 rays)

// Raise a kernel's dynamic-LDS limit.  The attribute is per device and the call is a cheap host-side table update, so it
// is simply made before every launch of a kernel that may need more than the default: no cached flag to go stale when the
// same process drives a second GPU or a second thread (the ABI takes a stream per call and keeps no state of its own).
template <typename K>
static int raise_lds_limit(K kernel, uint32_t bytes, const char *who) {
    if (hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess)
        return fail(NAF_ERR_LAUNCH, who);
    return NAF_OK;
}

template <typename P>
static uint32_t forward_lds_bytes() { return ((MlpShared<P>::kBytes + 15u) & ~15u) + 4u * kMaxSamplesLds * 4u; }
template <typename P>
static uint32_t backward_lds_bytes() {
    const uint32_t sh = (MlpShared<P>::kBytes + 15u) & ~15u;
    const uint32_t imgs = ((4u * 3u * 32u * P::kTrPitch * (uint32_t)sizeof(typename P::tr_t)) + 15u) & ~15u;
    return sh + std::max<uint32_t>(imgs + 4u * kMaxSamplesLds * 4u, (kMlpParams + 2u) * 4u);
}

struct Workspace {
    unsigned char *feat, *dfeat;
    float *slabs, *grad_acc;
    unsigned char *regions;      // binned scatter: record blocks [level slot][tile][slots], bucket-sorted per tile
    uint32_t *counts;            //                 run words [level slot][bucket][tile] = start | length << 16
    uint32_t *overflow;          //                 contributions that fell back to atomics (diagnostic counter)
    uint32_t *gmax;              //                 bit pattern of max |feature gradient| of the step (fixed-point scale)
    BinPlan plan;
    bool binned;
    size_t bytes;
};

static Workspace carve(void *base, const naf_render_cfg *cfg, uint64_t n_points) {
    const size_t esz = cfg->mlp_precision == NAF_F32 ? 4 : 2;
    const size_t feat_bytes = ((size_t)n_points * cfg->L * cfg->C * esz + 255) & ~(size_t)255;
    const size_t slab_bytes = (size_t)std::max(kBackwardBlocks, kBackwardBlocks16) * kSlabStride * 4;
    const size_t n_rays_max = (size_t)(n_points / std::max<uint32_t>(cfg->n_samples, 1u)) + 1;
    Workspace w;
    w.feat = (unsigned char *)base;
    w.dfeat = w.feat + feat_bytes;
    w.slabs = (float *)(w.dfeat + feat_bytes);
    w.grad_acc = (float *)((unsigned char *)w.slabs + slab_bytes);
    w.bytes = 2 * feat_bytes + slab_bytes + ((n_rays_max * 4 + 255) & ~(size_t)255) + 512;      // + partial sums of the loss
    w.binned = make_bin_plan(cfg, n_points, &w.plan);
    w.regions = nullptr;
    w.counts = nullptr;
    w.overflow = nullptr;
    w.gmax = nullptr;
    if (w.binned) {
        const size_t n_runs = ((size_t)w.plan.levels_per_pass << w.plan.log2_nb) * w.plan.n_tiles;
        const size_t block_bytes = ((size_t)w.plan.levels_per_pass * w.plan.n_tiles * w.plan.slots * record_bytes(cfg) + 255) & ~(size_t)255;
        w.regions = (unsigned char *)base + w.bytes;
        w.counts = (uint32_t *)(w.regions + block_bytes);
        w.overflow = w.counts + n_runs;                      // [0] total, [1 + level] per level
        w.gmax = w.overflow + 33;
        w.bytes += block_bytes + (((n_runs + 33 + 1) * 4 + 255) & ~(size_t)255);
    }
    return w;
}

static int check_cfg(const naf_render_cfg *cfg, const char *who) {
    if (!cfg) return fail(NAF_ERR_INVALID_ARGUMENT, "render: null cfg");
    if (cfg->L * cfg->C != 32u || !(cfg->C == 1 || cfg->C == 2 || cfg->C == 4 || cfg->C == 8))
        return fail(NAF_ERR_UNSUPPORTED, "fused field: encoder output L*C must be 32 with C in {1,2,4,8} "
                                         "(other shapes run through the unfused operators)");
    if (cfg->mlp_precision != NAF_F32 && cfg->mlp_precision != NAF_BF16)
        return fail(NAF_ERR_UNSUPPORTED, "fused field: mlp_precision must be NAF_F32 or NAF_BF16");
    if (cfg->table_dtype < NAF_F32 || cfg->table_dtype > NAF_BF16) return fail(NAF_ERR_UNSUPPORTED, "fused field: bad table_dtype");
    if (cfg->last_activation < 0 || cfg->last_activation > 3) return fail(NAF_ERR_UNSUPPORTED, "fused field: bad last_activation");
    if (!(cfg->bound > 0.0f)) return fail(NAF_ERR_INVALID_ARGUMENT, "fused field: bound must be > 0");
    if (cfg->flags & ~(NAF_CFG_PER_LEVEL_LAUNCHES | NAF_CFG_EXPLICIT_DEPTHS | NAF_CFG_LEVELS_INTERLEAVED | NAF_CFG_FORWARD_FUSED | NAF_CFG_FUSED_STORE_FEATURES | NAF_CFG_ENCODE_TWO_GATHERS | NAF_CFG_ENCODE_WINDOWS | NAF_CFG_BACKWARD_ONE_WAVE_PER_SIMD | NAF_CFG_ENCODE_LEVEL_MAJOR | NAF_CFG_TEST_TINY_BLOCKS | NAF_CFG_ENCODE_GROUPS_2 | NAF_CFG_ENCODE_GROUPS_4 | NAF_CFG_MIN_BUCKETS_MASK | NAF_CFG_SCATTER_PAIR12 | NAF_CFG_LEVELS_GATHER_PASS)) return fail(NAF_ERR_INVALID_ARGUMENT, "fused field: unknown cfg flag");
    if (cfg->scatter_mode < NAF_SCATTER_AUTO || cfg->scatter_mode > NAF_SCATTER_BINNED)
        return fail(NAF_ERR_INVALID_ARGUMENT, "fused field: scatter_mode must be NAF_SCATTER_AUTO, _ATOMIC or _BINNED");
    (void)who;
    return NAF_OK;
}

template <typename TT, typename P, uint32_t C, typename Src>
static int run_encode(const Src &src, const void *table, const int32_t *offsets, void *feat, uint32_t B, const naf_render_cfg *cfg, hipStream_t s,
                      uint32_t lv_begin = 0u, uint32_t lv_end = ~0u, uint32_t chunk = 0u) {
    // encode_kernel.h; features are written in the MLP's operand precision (P::feat_t), whatever the table stores
    return launch_encode<TT, typename P::feat_t, C, Src>(src, table, offsets, feat, B, cfg->H, cfg->L, cfg->flags, s, lv_begin, lv_end, chunk);
}

template <typename P, uint32_t C, typename Src>
static int dispatch_encode(const Src &src, const void *table, const int32_t *offsets, void *feat, uint32_t B, const naf_render_cfg *cfg, hipStream_t s,
                           uint32_t lv_begin = 0u, uint32_t lv_end = ~0u, uint32_t chunk = 0u) {
    switch (cfg->table_dtype) {
        case NAF_F32: return run_encode<F32, P, C>(src, table, offsets, feat, B, cfg, s, lv_begin, lv_end, chunk);
        case NAF_F16: return run_encode<F16, P, C>(src, table, offsets, feat, B, cfg, s, lv_begin, lv_end, chunk);
        default: return run_encode<BF16, P, C>(src, table, offsets, feat, B, cfg, s, lv_begin, lv_end, chunk);
    }
}

template <typename P, uint32_t C, bool kRays>
static int run_mlp_forward(const void *feat, const float *mlp, const SrcRays &src, float *out, uint32_t n_items, uint32_t B,
                           const naf_render_cfg *cfg, hipStream_t s, float *sigma_out = nullptr, float *depth_out = nullptr,
                           OutMap omap = OutMap{0u, 0u, 0u, 0u}) {
    if constexpr (std::is_same<P, PrecBF16>::value && C == 2) {
        {
            // (depth buffers of the four waves, which also stage the weights in the prologue: Mlp16Shared::build_staged)
            const uint32_t lds16 = ((Mlp16Shared::kBytes + 15u) & ~15u) + std::max<uint32_t>(4u * kMaxSamplesLds * 4u, Mlp16Shared::kStageFloats * 4u);
            if constexpr (kRays) {
                // fewer rays than two waves per SIMD, line integrals only: four waves a ray (mlp16_forward_split_kernel)
                const uint32_t tiles = (src.S + 15u) / 16u;
                if (sigma_out == nullptr && depth_out == nullptr && n_items < 2048u && src.S <= kMaxSamplesLds / 2u && tiles >= 4u &&
                    (cfg->flags & NAF_CFG_BACKWARD_ONE_WAVE_PER_SIMD) == 0u) {
                    ProfScope prof_("mlp_forward_kernel", s);
#ifndef NAF_FWD_SPLIT_GRID
#define NAF_FWD_SPLIT_GRID 1024u      // workgroups at most (four fit a CU); A/B builds override it
#endif
                    hipLaunchKernelGGL(mlp16_forward_split_kernel, dim3(std::max(1u, std::min(n_items, NAF_FWD_SPLIT_GRID))), dim3(256), lds16, s,
                                       (const uint16_t *)feat, mlp, src, out, n_items, B, cfg->last_activation);
                    return check_launch("mlp16_forward_split_kernel");
                }
            }
            const uint64_t waves16 = kRays ? n_items : ((uint64_t)n_items + 15) / 16;
            const uint32_t grid16 = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>((waves16 + 3) / 4, 256u * 8u));
            { ProfScope prof_("mlp_forward_kernel", s); hipLaunchKernelGGL((mlp16_forward_kernel<kRays>), dim3(grid16), dim3(256), lds16, s,
                               (const uint16_t *)feat, mlp, src, out, sigma_out, depth_out, n_items, B, cfg->last_activation, omap); }
            return check_launch("mlp16_forward_kernel");
        }
    }
    auto kern = mlp_forward_kernel<P, C, kRays>;
    const uint32_t lds = forward_lds_bytes<P>();
    const uint64_t waves_needed = kRays ? n_items : ((uint64_t)n_items + 31) / 32;
    const uint32_t grid = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>((waves_needed + 3) / 4, 256u * 8u));
    { ProfScope prof_("mlp_forward_kernel", s); hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, (const typename P::feat_t::store_t *)feat, mlp, src, out, sigma_out,
                       depth_out, n_items, B, cfg->last_activation, omap); }
    return check_launch("mlp_forward_kernel");
}

// What the backward pass hands down so that its small jobs ride on kernels it launches anyway: `clear_words` -- the overflow
// counters of the binned scatter, zeroed by the MLP backward kernel (which always precedes the scatter on the stream; three memset
// launches of ~5 us each were 4 % of the reference-size step); `loss_assign` -- loss_out[0] = loss instead of +=; `deferred` --
// where to leave the description of the slab reduction instead of launching it (the first scatter_bin launch then runs it in spare
// workgroups, beside its own: one dependent launch of ~11 us less per step).
struct StepExtras { uint32_t *clear_words; bool loss_assign; SlabReduce *deferred; hipEvent_t after_backward = nullptr; };      // after_backward: recorded once
                                                                                      // the feature gradients are final, before the slab reduction

static int run_mlp_grad_reduce(const float *slabs, uint32_t n_slabs, float *grad_mlp, float *loss_out, bool with_loss, const MlpAdam *madam,
                               uint32_t *gmax_bits, const StepExtras &ex, hipStream_t s) {
    const MlpAdam none{nullptr, nullptr, nullptr, AdamArgs{}};
    const SlabReduce sr{slabs, n_slabs, grad_mlp, loss_out, !with_loss ? kLossNone : ex.loss_assign ? kLossAssign : kLossAdd,
                        madam != nullptr ? *madam : none, gmax_bits};
    if (ex.deferred != nullptr) { *ex.deferred = sr; return NAF_OK; }
    ProfScope prof_("mlp_grad_reduce_kernel", s);
    hipLaunchKernelGGL(mlp_grad_reduce_kernel, dim3(kSlabReduceBlocks), dim3(kReduceParams * kReduceGroups), 0, s, sr);
    return check_launch("mlp_grad_reduce_kernel");
}

// `loss.target` != nullptr: training step -- d loss / d acc is formed inside the kernel from (acc, target, weight), the loss itself
// travels through the slabs into loss_out (+=).  Otherwise `grad_acc` is the upstream gradient (naf_render_backward).
template <typename P, uint32_t C>
static int run_mlp_backward(const void *feat, const float *mlp, const SrcRays &src, const float *grad_acc, const LossInputs &loss, void *dfeat,
                            float *slabs, uint32_t *gmax_bits, float *grad_mlp, float *loss_out, const MlpAdam *madam, uint32_t n_rays, uint32_t B,
                            const naf_render_cfg *cfg, const StepExtras &ex, hipStream_t s) {
    const bool with_loss = loss.target != nullptr;
    if constexpr (std::is_same<P, PrecBF16>::value && C == 2) {
        {
            const uint32_t sh16 = (Mlp16Shared::kBytes + 15u) & ~15u;
            const uint32_t lds16 = sh16 + std::max<uint32_t>(4u * 3u * 1024u + 4u * kMaxSamplesLds * 4u, (kMlpParams + 2u) * 4u);
            // fewer rays than the chip holds waves of this kernel (3 per SIMD = 3 072): `parts` tile ranges per ray, never more
            // ranges than tiles, so that the resident waves all get one item of equal length where the numbers allow it
            // (1 024 rays x 12 tiles: 3 parts of 4 tiles).  NAF_CFG_BACKWARD_ONE_WAVE_PER_SIMD (diagnostic): the old goal of one wave per SIMD.
            const uint32_t tiles = (cfg->n_samples + 15u) / 16u;
            const uint64_t wave_goal = (cfg->flags & NAF_CFG_BACKWARD_ONE_WAVE_PER_SIMD) != 0u ? 1024u : 4u * kBackwardBlocks16;
            const uint32_t parts = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(std::min<uint64_t>(tiles, 12u), wave_goal / std::max<uint32_t>(n_rays, 1u)));
            const uint32_t grid16 = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(((uint64_t)n_rays * parts + 3) / 4, kBackwardBlocks16));
            { ProfScope prof_("mlp_backward_kernel", s); hipLaunchKernelGGL(mlp16_backward_kernel, dim3(grid16), dim3(256), lds16, s, (const uint16_t *)feat, mlp, src,
                               grad_acc, loss, (uint16_t *)dfeat, slabs, n_rays, B, cfg->last_activation, parts, ex.clear_words); }
            if (int rc = check_launch("mlp16_backward_kernel")) return rc;
            if (ex.after_backward != nullptr && hipEventRecord(ex.after_backward, s) != hipSuccess) return fail(NAF_ERR_LAUNCH, "mlp backward: cannot record the event");
            return run_mlp_grad_reduce(slabs, grid16, grad_mlp, loss_out, with_loss, madam, gmax_bits, ex, s);
        }
    }
    auto kern = mlp_backward_kernel<P, C>;
    const uint32_t lds = backward_lds_bytes<P>();
    if (int rc = raise_lds_limit(kern, lds, "mlp_backward_kernel: cannot raise dynamic LDS limit")) return rc;      // fp32 images need > 64 KiB
    const uint32_t grid = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(((uint64_t)n_rays + 3) / 4, kBackwardBlocks));
    { ProfScope prof_("mlp_backward_kernel", s); hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, (const typename P::feat_t::store_t *)feat, mlp, src, grad_acc, loss,
                       (typename P::feat_t::store_t *)dfeat, slabs, n_rays, B, cfg->last_activation, ex.clear_words); }
    if (int rc = check_launch("mlp_backward_kernel")) return rc;
    if (ex.after_backward != nullptr && hipEventRecord(ex.after_backward, s) != hipSuccess) return fail(NAF_ERR_LAUNCH, "mlp backward: cannot record the event");
    return run_mlp_grad_reduce(slabs, grid, grad_mlp, loss_out, with_loss, madam, gmax_bits, ex, s);
}

// The Adam tail (naf_render_train_adam) needs every reducer launch of the step unsplit, the binned scatter and level-major launches.
static bool adam_tail_possible(const naf_render_cfg *cfg, const Workspace &w) {
    if (!w.binned || per_level_launches(cfg)) return false;
    const uint32_t NB = 1u << w.plan.log2_nb;
    for (uint32_t l0 = 0; l0 < cfg->L; l0 += w.plan.levels_per_pass)
        if (reducer_split(NB, std::min(w.plan.levels_per_pass, cfg->L - l0)) != 1u) return false;
    return true;
}

template <typename P, uint32_t C, typename Rec>
static int run_binned_scatter(const SrcRays &src, const void *dfeat, const int32_t *offsets, float *grad_table, uint32_t B,
                              const naf_render_cfg *cfg, const Workspace &w, uint32_t lv_begin, uint32_t lv_end,
                              const naf_grad_buckets *buckets, hipStream_t s, const AdamTail *adam = nullptr, const SlabReduce *slab_job = nullptr) {
    using FT = typename P::feat_t;
    constexpr uint32_t NT = BinShape<Rec>::kThreads, PTS = BinShape<Rec>::kPoints;
    // levels per bin workgroup: all of them when there are enough tiles to fill the chip several times over (the sample
    // position is evaluated once per point, and stores drain behind the next level: 3.85 -> 3.53 ms at 65 536 rays),
    // four when tiles are scarce (1 024-ray steps: 0.093 -> 0.071 ms)
    constexpr uint32_t kLvMany = 16u, kLvFew = 4u;
    const BinPlan &plan = w.plan;
    constexpr bool kHasBig = sizeof(Rec) <= 12;                 // the 1024-thread shape of pass 1 (make_bin_plan)
    const bool big = kHasBig && plan.tile_points == 2u * NT * PTS;
    if (!big && plan.tile_points != NT * PTS) return fail(NAF_ERR_LAUNCH, "binned scatter: plan / kernel tile mismatch");
    const uint32_t threads = big ? 2u * NT : NT;
    const bool many = plan.n_tiles >= (big ? 768u : 1536u);
    const uint32_t LV = many ? kLvMany : kLvFew;
    auto bin = many ? scatter_bin_kernel<FT, C, SrcRays, Rec, NT, PTS, kLvMany> : scatter_bin_kernel<FT, C, SrcRays, Rec, NT, PTS, kLvFew>;
    if constexpr (kHasBig) {
        if (big) bin = many ? scatter_bin_kernel<FT, C, SrcRays, Rec, 2u * NT, PTS, kLvMany> : scatter_bin_kernel<FT, C, SrcRays, Rec, 2u * NT, PTS, kLvFew>;
    }
    auto red = adam != nullptr ? scatter_reduce_kernel<C, Rec, true> : scatter_reduce_kernel<C, Rec, false>;
    const AdamTail tail = adam != nullptr ? *adam : AdamTail{};
    const uint32_t NB = 1u << plan.log2_nb;
    const uint32_t red_lds = plan.max_local_rows * C * 8u;
    const uint32_t bin_lds = (2u * NB + 4u) * 4u + plan.slots * (uint32_t)sizeof(Rec);
    if (int rc = raise_lds_limit(red, red_lds, "binned scatter: cannot raise dynamic LDS limit (reduce)")) return rc;
    if (int rc = raise_lds_limit(bin, bin_lds, "binned scatter: cannot raise dynamic LDS limit (bin)")) return rc;
    static const char *const bin_names[32] = NAF_LEVEL_NAMES("scatter_bin_kernel_L");
    static const char *const red_names[32] = NAF_LEVEL_NAMES("scatter_reduce_kernel_L");
    const bool per_level = per_level_launches(cfg);
    // pass 1 over the levels [l0, l0 + nl): their records fill the region buffer from level slot 0.  The first launch carries the
    // reduction of the MLP backward's slabs when the caller deferred it (StepExtras): kSlabReduceBlocks workgroups behind the tiles.
    SlabReduce job{};
    if (slab_job != nullptr) job = *slab_job;
    auto launch_bin = [&](uint32_t l0, uint32_t nl) -> int {
        ProfScope prof_(per_level ? level_name(bin_names, l0) : "scatter_bin_kernel", s);
        const uint32_t spare = job.slabs != nullptr ? kSlabReduceBlocks : 0u;
        hipLaunchKernelGGL(bin, dim3(plan.n_tiles + spare, (nl + LV - 1u) / LV), dim3(threads), bin_lds, s, src, (const typename FT::store_t *)dfeat,
                           offsets, grad_table, (Rec *)w.regions, w.counts, w.overflow, B, cfg->H, l0, nl, plan, job, B, 1u);
        job = SlabReduce{};
        return check_launch("scatter_bin_kernel");
    };
    // pass 2 over the level slots [ly0, ly0 + nl) of a bin pass that started at level l0
    auto launch_reduce = [&](uint32_t l0, uint32_t ly0, uint32_t nl) -> int {
        ProfScope prof_(per_level ? level_name(red_names, l0 + ly0) : "scatter_reduce_kernel", s);
        // keep >= ~1024 reducer workgroups in flight: with one or two levels per pass split each bucket's tiles.
        // (A reducer workgroup owns a CU's LDS, so 256 run at a time: 512 or more unsplit ones already come in full rounds.)
        const uint32_t n_split = reducer_split(NB, nl);
        if (adam != nullptr && n_split != 1u) return fail(NAF_ERR_LAUNCH, "binned scatter: the Adam tail needs unsplit reducer launches");
        hipLaunchKernelGGL(red, dim3(NB, nl, n_split), dim3(1024), red_lds, s, (const Rec *)w.regions, w.counts, offsets,
                           grad_table, w.gmax, l0, ly0, cfg->H, plan, tail);
        return check_launch("scatter_reduce_kernel");
    };
    if (buckets != nullptr && !per_level && plan.levels_per_pass >= cfg->L && lv_begin == 0u && lv_end == cfg->L) {
        // data parallel, and the records of all levels fit one pass: bin ONCE (the sample position is evaluated once per
        // point and the stores of a level drain behind the next, exactly as in the single-GPU step), then finish the table
        // bucket by bucket -- each bucket's event fires as soon as its rows are final, and its all-reduce overlaps the
        // reduction of the buckets that follow.
        if (int rc = launch_bin(0u, cfg->L)) return rc;
        for (uint32_t b = 0; b < buckets->n_buckets; ++b) {
            if (int rc = launch_reduce(0u, buckets->level_begin[b], buckets->level_end[b] - buckets->level_begin[b])) return rc;
            if (buckets->ready[b] != nullptr && hipEventRecord((hipEvent_t)buckets->ready[b], s) != hipSuccess)
                return fail(NAF_ERR_LAUNCH, "render_train: cannot record a bucket event");
        }
        return NAF_OK;
    }
    for (uint32_t l0 = lv_begin; l0 < lv_end; l0 += plan.levels_per_pass) {
        const uint32_t nl = std::min(plan.levels_per_pass, lv_end - l0);
        if (int rc = launch_bin(l0, nl)) return rc;
        if (int rc = launch_reduce(l0, 0u, nl)) return rc;
    }
    return NAF_OK;
}

// The same launch plan with the kernels of scatter_v2.h (8-byte records; the canonical two-channel bf16 shape).
static int run_binned_scatter2(const SrcRays &src, const void *dfeat, const int32_t *offsets, float *grad_table, uint32_t B,
                               const naf_render_cfg *cfg, const Workspace &w, uint32_t lv_begin, uint32_t lv_end,
                               const naf_grad_buckets *buckets, hipStream_t s, const AdamTail *adam = nullptr, const SlabReduce *slab_job = nullptr,
                               DrawJob *next_draw = nullptr, const GradBlocks *from_blocks = nullptr) {
#ifndef NAF_V2_LV_FEW
#define NAF_V2_LV_FEW 4u      // levels per bin workgroup when tiles are scarce.  A/B builds (tools/build_variant.sh) override it: 2 gains 2 us at
                              // 512 rays and loses 9 at 4 096, 8 the other way round (profiles/round4_ab_reducer_nt_loads_and_levels_per_bin_workgroup.jsonl)
#endif
    constexpr uint32_t NT = 512u, PTS = 2u, kLvMany = 16u, kLvFew = NAF_V2_LV_FEW;
    const BinPlan &plan = w.plan;
    const bool big = plan.tile_points == 2u * NT * PTS;
    if (!big && plan.tile_points != NT * PTS) return fail(NAF_ERR_LAUNCH, "binned scatter: plan / kernel tile mismatch");
    const uint32_t threads = big ? 2u * NT : NT;
    const bool many = plan.n_tiles >= (big ? 768u : 1536u);
    const uint32_t LV = many ? kLvMany : kLvFew;
    // 64 buckets per level (every table up to 2^19 rows per level): the variant whose waves scan the bucket counters themselves
    const bool nb64 = plan.log2_nb == 6u;
    auto bin = big ? (many ? scatter_bin2_kernel<2u * NT, kLvMany, 0u> : scatter_bin2_kernel<2u * NT, kLvFew, 0u>)
                   : nb64 ? (many ? scatter_bin2_kernel<NT, kLvMany, 6u> : scatter_bin2_kernel<NT, kLvFew, 6u>)
                          : (many ? scatter_bin2_kernel<NT, kLvMany, 0u> : scatter_bin2_kernel<NT, kLvFew, 0u>);
    if (from_blocks != nullptr)      // level-parallel: `dfeat` = the all-to-all's blocks, read in place (scatter_v2.h, GradBlocks)
        bin = big ? (many ? scatter_bin2_kernel<2u * NT, kLvMany, 0u, true> : scatter_bin2_kernel<2u * NT, kLvFew, 0u, true>)
                  : nb64 ? (many ? scatter_bin2_kernel<NT, kLvMany, 6u, true> : scatter_bin2_kernel<NT, kLvFew, 6u, true>)
                         : (many ? scatter_bin2_kernel<NT, kLvMany, 0u, true> : scatter_bin2_kernel<NT, kLvFew, 0u, true>);
    const GradBlocks gb = from_blocks != nullptr ? *from_blocks : GradBlocks{};
    const bool fast = adam != nullptr && adam->lp != nullptr;        // tables with a 16-bit shadow: adam_math.h
    auto red = adam == nullptr ? scatter_reduce2_kernel<false, false> : fast ? scatter_reduce2_kernel<true, true> : scatter_reduce2_kernel<true, false>;
    const AdamTail tail = adam != nullptr ? *adam : AdamTail{};
    const uint32_t NB = 1u << plan.log2_nb;
    const uint32_t red_lds = plan.max_local_rows * 2u * 8u;
    const uint32_t bin_lds = (3u * NB + 4u) * 4u + (plan.slots + 1u) * (uint32_t)sizeof(PairFx) + side_list_capacity(plan.slots) * 12u;      // counters, staging, side list
    if (int rc = raise_lds_limit(red, red_lds, "binned scatter: cannot raise dynamic LDS limit (reduce)")) return rc;
    if (int rc = raise_lds_limit(bin, bin_lds, "binned scatter: cannot raise dynamic LDS limit (bin)")) return rc;
    static const char *const bin_names[32] = NAF_LEVEL_NAMES("scatter_bin_kernel_L");
    static const char *const red_names[32] = NAF_LEVEL_NAMES("scatter_reduce_kernel_L");
    const bool per_level = per_level_launches(cfg);
    SlabReduce job{};
    if (slab_job != nullptr) job = *slab_job;
    auto launch_bin = [&](uint32_t l0, uint32_t nl) -> int {
        ProfScope prof_(per_level ? level_name(bin_names, l0) : "scatter_bin_kernel", s);
        // the first launch of the step also carries the next step's pixel draw when the caller handed one over (spare workgroups)
        DrawJob dj{};
        if (next_draw != nullptr && next_draw->count != 0u) { dj = *next_draw; next_draw->count = 0u; }      // consumed
        const uint32_t spare = (job.slabs != nullptr ? kSlabReduceBlocks : 0u) + (dj.count + threads - 1u) / threads;
        hipLaunchKernelGGL(bin, dim3(plan.n_tiles + spare, (nl + LV - 1u) / LV), dim3(threads), bin_lds, s, src, (const uint16_t *)dfeat,
                           offsets, grad_table, (PairFx *)w.regions, w.counts, w.overflow, B, cfg->H, l0, nl, plan, job, dj, gb);
        job = SlabReduce{};
        return check_launch("scatter_bin_kernel");
    };
    auto launch_reduce = [&](uint32_t l0, uint32_t ly0, uint32_t nl) -> int {
        ProfScope prof_(per_level ? level_name(red_names, l0 + ly0) : "scatter_reduce_kernel", s);
        const uint32_t n_split = reducer_split(NB, nl);
        if (adam != nullptr && n_split != 1u) return fail(NAF_ERR_LAUNCH, "binned scatter: the Adam tail needs unsplit reducer launches");
        hipLaunchKernelGGL(red, dim3(NB, nl, n_split), dim3(1024), red_lds, s, (const PairFx *)w.regions, w.counts, offsets,
                           grad_table, w.gmax, l0, ly0, cfg->H, plan, tail);
        return check_launch("scatter_reduce_kernel");
    };
    if (buckets != nullptr && !per_level && plan.levels_per_pass >= cfg->L && lv_begin == 0u && lv_end == cfg->L) {
        if (int rc = launch_bin(0u, cfg->L)) return rc;              // data parallel: bin once, finish the table bucket by bucket (see above)
        for (uint32_t b = 0; b < buckets->n_buckets; ++b) {
            if (int rc = launch_reduce(0u, buckets->level_begin[b], buckets->level_end[b] - buckets->level_begin[b])) return rc;
            if (buckets->ready[b] != nullptr && hipEventRecord((hipEvent_t)buckets->ready[b], s) != hipSuccess)
                return fail(NAF_ERR_LAUNCH, "render_train: cannot record a bucket event");
        }
        return NAF_OK;
    }
    for (uint32_t l0 = lv_begin; l0 < lv_end; l0 += plan.levels_per_pass) {
        const uint32_t nl = std::min(plan.levels_per_pass, lv_end - l0);
        if (int rc = launch_bin(l0, nl)) return rc;
        if (int rc = launch_reduce(l0, 0u, nl)) return rc;
    }
    return NAF_OK;
}

// Table-gradient scatter of the levels [lv_begin, lv_end).
template <typename P, uint32_t C>
static int run_hash_backward_levels(const SrcRays &src, const void *dfeat, const int32_t *offsets, float *grad_table, uint32_t B,
                                    const naf_render_cfg *cfg, const Workspace &w, uint32_t lv_begin, uint32_t lv_end, hipStream_t s,
                                    const naf_grad_buckets *buckets = nullptr, const AdamTail *adam = nullptr, const SlabReduce *slab_job = nullptr,
                                    DrawJob *next_draw = nullptr, const GradBlocks *from_blocks = nullptr) {
    using FT = typename P::feat_t;
    if (from_blocks != nullptr && !(w.binned && scatter_v2(cfg))) return fail(NAF_ERR_LAUNCH, "hash backward: only the 8-byte-record scatter reads gradient blocks in place");
    if (w.binned) {
        if (scatter_v2(cfg)) return run_binned_scatter2(src, dfeat, offsets, grad_table, B, cfg, w, lv_begin, lv_end, buckets, s, adam, slab_job, next_draw, from_blocks);
        if (cfg->mlp_precision == NAF_F32) return run_binned_scatter<P, C, PairF32<C>>(src, dfeat, offsets, grad_table, B, cfg, w, lv_begin, lv_end, buckets, s, adam, slab_job);
        return run_binned_scatter<P, C, PairBF16<C>>(src, dfeat, offsets, grad_table, B, cfg, w, lv_begin, lv_end, buckets, s, adam, slab_job);
    }
    if (adam != nullptr) return fail(NAF_ERR_LAUNCH, "hash backward: the Adam tail needs the binned scatter");
    if (per_level_launches(cfg)) {
        static const char *const names[32] = NAF_LEVEL_NAMES("hash_backward_kernel_L");
        for (uint32_t l = lv_begin; l < lv_end; ++l) {
            ProfScope prof_(level_name(names, l), s);
            hipLaunchKernelGGL((hash_backward_kernel<FT, 3, C, SrcRays>), dim3(hash_grid_x(B), 1), dim3(256), 0, s, src,
                               (const typename FT::store_t *)dfeat, offsets, grad_table, B, cfg->L, cfg->H, false, l);
        }
        return check_launch("hash_backward_kernel");
    }
    { ProfScope prof_("hash_backward_kernel", s); hipLaunchKernelGGL((hash_backward_kernel<FT, 3, C, SrcRays>), dim3(hash_grid_x(B), lv_end - lv_begin), dim3(256), 0, s, src,
                       (const typename FT::store_t *)dfeat, offsets, grad_table, B, cfg->L, cfg->H, false, lv_begin); }
    return check_launch("hash_backward_kernel");
}

// All levels, in the order of `buckets` when given (data-parallel training: each bucket's event is recorded on the stream as
// soon as its rows of the gradient table are final, so that the caller can start that bucket's all-reduce while the next
// bucket is still being binned and reduced).
template <typename P, uint32_t C>
static int run_hash_backward(const SrcRays &src, const void *dfeat, const int32_t *offsets, float *grad_table, uint32_t B,
                             const naf_render_cfg *cfg, const Workspace &w, const naf_grad_buckets *buckets, hipStream_t s,
                             const AdamTail *adam = nullptr, const SlabReduce *slab_job = nullptr, DrawJob *next_draw = nullptr) {
    // (the overflow counters were zeroed by the MLP backward kernel of this call: StepExtras)
    if (buckets == nullptr) return run_hash_backward_levels<P, C>(src, dfeat, offsets, grad_table, B, cfg, w, 0u, cfg->L, s, nullptr, adam, slab_job, next_draw);
    if (adam != nullptr) return fail(NAF_ERR_INVALID_ARGUMENT, "render_train: the Adam tail cannot be combined with gradient buckets");
    if (w.binned && !per_level_launches(cfg) && w.plan.levels_per_pass >= cfg->L)      // one bin pass, per-bucket reduction + events
        return run_hash_backward_levels<P, C>(src, dfeat, offsets, grad_table, B, cfg, w, 0u, cfg->L, s, buckets);
    for (uint32_t b = 0; b < buckets->n_buckets; ++b) {
        if (int rc = run_hash_backward_levels<P, C>(src, dfeat, offsets, grad_table, B, cfg, w, buckets->level_begin[b], buckets->level_end[b], s)) return rc;
        if (buckets->ready[b] != nullptr && hipEventRecord((hipEvent_t)buckets->ready[b], s) != hipSuccess)
            return fail(NAF_ERR_LAUNCH, "render_train: cannot record a bucket event");
    }
    return NAF_OK;
}

static SrcRays make_src(const float *rays, const float *t_rand, const naf_render_cfg *cfg) {
    SrcRays s;
    s.rays = rays; s.t_rand = t_rand; s.S = cfg->n_samples; s.perturb = cfg->perturb != 0; s.bound = cfg->bound;
    s.explicit_z = (cfg->flags & NAF_CFG_EXPLICIT_DEPTHS) != 0u;
    s.seed = cfg->seed; s.ray_base = cfg->ray_index_base;
    s.div_magic = (uint32_t)((1ull << 32) / s.S);
    s.lin_step = 1.0f / (float)(s.S - 1u);
    s.rden = 1.0f / (2.0f * s.bound);
    return s;
}

// ---- forward-only calls ----------------------------------------------------------------------------------------------
// They need the [L, n_points, C] features and nothing else of the training workspace -- or, where the single fused kernel
// applies (NAF_CFG_FORWARD_FUSED, the canonical field in bf16 mode), no workspace at all.
static bool forward_fused(const naf_render_cfg *cfg) {
    return (cfg->flags & NAF_CFG_FORWARD_FUSED) != 0u && cfg->mlp_precision == NAF_BF16 && cfg->C == 2u && cfg->L == kFusedLevels;
}
static size_t feature_bytes(const naf_render_cfg *cfg, uint64_t n_points) {
    const size_t esz = cfg->mlp_precision == NAF_F32 ? 4 : 2;
    return ((size_t)n_points * cfg->L * cfg->C * esz + 255) & ~(size_t)255;
}
static size_t forward_workspace_bytes(const naf_render_cfg *cfg, uint64_t n_points) {
    if (forward_fused(cfg) && (cfg->flags & NAF_CFG_FUSED_STORE_FEATURES) == 0u) return 256;
    return feature_bytes(cfg, n_points) + 256;
}

template <typename TT, typename Src>
static int launch_fused_forward(const Src &src, const void *table, const int32_t *offsets, const float *mlp, float *out, float *sigma_out,
                                float *depth_out, void *feat, uint32_t n_items, uint32_t B, const naf_render_cfg *cfg, OutMap omap, hipStream_t s) {
    constexpr bool kRays = std::is_same<Src, SrcRays>::value;
    const uint32_t lds = ((Mlp16Shared::kBytes + 15u) & ~15u) + kFusedLevels * (uint32_t)sizeof(LevelRec) + (kRays ? 4u * kMaxSamplesLds * 4u : 0u);
    const uint64_t waves = kRays ? n_items : ((uint64_t)n_items + 15) / 16;
    const uint32_t grid = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>((waves + 3) / 4, 256u * 8u));
    uint16_t *fout = (cfg->flags & NAF_CFG_FUSED_STORE_FEATURES) != 0u ? (uint16_t *)feat : nullptr;
    { ProfScope prof_("fused_forward_kernel", s); hipLaunchKernelGGL((fused_forward_kernel<TT, Src>), dim3(grid), dim3(256), lds, s, src,
                       (const typename TT::store_t *)table, offsets, mlp, out, sigma_out, depth_out, fout, n_items, B, cfg->H, cfg->last_activation, omap); }
    return check_launch("fused_forward_kernel");
}
template <typename Src>
static int run_fused_forward(const Src &src, const void *table, const int32_t *offsets, const float *mlp, float *out, float *sigma_out,
                             float *depth_out, void *feat, uint32_t n_items, uint32_t B, const naf_render_cfg *cfg, OutMap omap, hipStream_t s) {
    switch (cfg->table_dtype) {
        case NAF_F32: return launch_fused_forward<F32>(src, table, offsets, mlp, out, sigma_out, depth_out, feat, n_items, B, cfg, omap, s);
        case NAF_F16: return launch_fused_forward<F16>(src, table, offsets, mlp, out, sigma_out, depth_out, feat, n_items, B, cfg, omap, s);
        default: return launch_fused_forward<BF16>(src, table, offsets, mlp, out, sigma_out, depth_out, feat, n_items, B, cfg, omap, s);
    }
}

template <typename P, uint32_t C>
static int render_forward_impl(const float *rays, const float *t_rand, const void *emb, const int32_t *offsets, const float *mlp,
                               float *acc, uint32_t n_rays, const naf_render_cfg *cfg, void *ws, hipStream_t s,
                               float *sigma_out = nullptr, float *depth_out = nullptr, bool keep_features = false) {
    const uint32_t B = n_rays * cfg->n_samples;
    const SrcRays src = make_src(rays, t_rand, cfg);
    if (forward_fused(cfg) && !keep_features)              // a training step needs the features for its backward pass
        return run_fused_forward(src, emb, offsets, mlp, acc, sigma_out, depth_out, ws, n_rays, B, cfg, OutMap{0u, 0u, 0u, 0u}, s);
    if (int rc = dispatch_encode<P, C>(src, emb, offsets, ws, B, cfg, s)) return rc;        // the features sit at the workspace base
    return run_mlp_forward<P, C, true>(ws, mlp, src, acc, n_rays, B, cfg, s, sigma_out, depth_out);
}

template <typename P, uint32_t C>
static int render_backward_impl(const float *rays, const float *t_rand, const float *grad_acc, const void *emb, const int32_t *offsets,
                                const float *mlp, float *grad_emb, float *grad_mlp, uint32_t n_rays, const naf_render_cfg *cfg,
                                void *ws, int features_valid, const naf_grad_buckets *buckets, hipStream_t s,
                                const AdamTail *adam = nullptr, bool from_train = false, const LossInputs &loss = LossInputs{nullptr, nullptr, nullptr},
                                float *loss_out = nullptr, const MlpAdam *madam = nullptr, bool loss_assign = false, DrawJob *next_draw = nullptr) {
    const uint32_t B = n_rays * cfg->n_samples;
    const Workspace w = carve(ws, cfg, B);
    const SrcRays src = make_src(rays, t_rand, cfg);
    AdamTail tail;
    if (adam != nullptr) {                                   // the overflow counters live in this call's workspace
        tail = *adam;
        tail.overflow = w.overflow;
        adam = &tail;
    }
    // (a fused forward of the same cfg left no features behind unless it was asked to store them)
    if (!features_valid || (forward_fused(cfg) && (cfg->flags & NAF_CFG_FUSED_STORE_FEATURES) == 0u && !from_train))
        if (int rc = dispatch_encode<P, C>(src, emb, offsets, w.feat, B, cfg, s)) return rc;
    // Single-GPU steps on the binned scatter defer the slab reduction (MLP gradient / Adam, loss, gradient maximum) to spare workgroups
    // of the scatter's first launch; data-parallel steps need the MLP gradient final BEFORE the scatter (its all-reduce starts at
    // `mlp_ready`), per-level diagnostics keep their launches apart.
    SlabReduce job{};
    const bool defer = w.binned && buckets == nullptr && !per_level_launches(cfg);
    const StepExtras ex{w.binned ? w.overflow : nullptr, loss_assign, defer ? &job : nullptr};
    if (int rc = run_mlp_backward<P, C>(w.feat, mlp, src, grad_acc, loss, w.dfeat, w.slabs, w.binned ? w.gmax : nullptr, grad_mlp, loss_out, madam, n_rays, B, cfg,
                                        ex, s)) return rc;
    // grad_mlp (and, in the training entry point, the loss) are final here, before the table scatter starts
    if (buckets != nullptr && buckets->mlp_ready != nullptr && hipEventRecord((hipEvent_t)buckets->mlp_ready, s) != hipSuccess)
        return fail(NAF_ERR_LAUNCH, "render_train: cannot record the MLP-gradient event");
    return run_hash_backward<P, C>(src, w.dfeat, offsets, grad_emb, B, cfg, w, buckets, s, adam, defer ? &job : nullptr, next_draw);
}

template <typename P, uint32_t C>
static int render_train_impl(const float *rays, const float *t_rand, const float *target, const float *ray_weight, const void *emb,
                             const int32_t *offsets, const float *mlp, float *acc, float *grad_emb, float *grad_mlp, float *loss_out,
                             uint32_t n_rays, const naf_render_cfg *cfg, void *ws, const naf_grad_buckets *buckets, hipStream_t s,
                             const AdamTail *adam = nullptr, const MlpAdam *madam = nullptr, bool loss_assign = false, DrawJob *next_draw = nullptr) {
    if (int rc = render_forward_impl<P, C>(rays, t_rand, emb, offsets, mlp, acc, n_rays, cfg, ws, s, nullptr, nullptr, true)) return rc;
    // the masked squared error and its gradient are formed inside the backward kernel (LossInputs); the loss reaches loss_out
    // through the slab reduction that also finishes the MLP gradient
    const LossInputs loss{acc, target, ray_weight};
    return render_backward_impl<P, C>(rays, t_rand, nullptr, emb, offsets, mlp, grad_emb, grad_mlp, n_rays, cfg, ws, 1, buckets, s, adam, true,
                                      loss, loss_out, madam, loss_assign, next_draw);
}

template <typename P, uint32_t C>
static int field_forward_impl(const float *pts, const void *emb, const int32_t *offsets, const float *mlp, float *sigma, uint32_t B,
                              const naf_render_cfg *cfg, void *ws, hipStream_t s) {
    SrcRaw src{pts, cfg->bound};
    if (forward_fused(cfg)) return run_fused_forward(src, emb, offsets, mlp, sigma, nullptr, nullptr, ws, B, B, cfg, OutMap{0u, 0u, 0u, 0u}, s);
    if (int rc = dispatch_encode<P, C>(src, emb, offsets, ws, B, cfg, s)) return rc;
    SrcRays none{};
    return run_mlp_forward<P, C, false>(ws, mlp, none, sigma, B, B, cfg, s);
}

template <typename P, uint32_t C>
static int field_forward_grid_impl(const SrcGrid &src, const void *emb, const int32_t *offsets, const float *mlp, float *sigma, uint32_t B,
                                   const naf_render_cfg *cfg, void *ws, hipStream_t s) {
    const OutMap omap{src.n[0], src.n[1], src.n[2], src.first};
    if (forward_fused(cfg)) return run_fused_forward(src, emb, offsets, mlp, sigma, nullptr, nullptr, ws, B, B, cfg, omap, s);
    if (int rc = dispatch_encode<P, C>(src, emb, offsets, ws, B, cfg, s)) return rc;
    SrcRays none{};
    return run_mlp_forward<P, C, false>(ws, mlp, none, sigma, B, B, cfg, s, nullptr, nullptr, omap);
}

// ---- level-parallel training (naf_levels_*, naf_hip.h): each rank owns a range of levels ----------------------------------
// Feature gradients arrive from the all-to-all as one block per source rank, [rank][owned level][that rank's points][C]; the
// scatter wants [level][all points][C].  One pass moves them there (16 bytes per lane where the block length allows it), takes
// the maximum |gradient| the fixed-point reducer scales by (the MLP backward of a single-GPU step hands that over; here it ran on
// other GPUs) -- as the bit pattern of the fp32 value, so that NaN / Inf poison the step as they do there.
template <typename T> __device__ __forceinline__ uint32_t abs_bits(T raw);
template <> __device__ __forceinline__ uint32_t abs_bits<uint16_t>(uint16_t raw) { return ((uint32_t)raw << 16) & 0x7fffffffu; }      // bf16
template <> __device__ __forceinline__ uint32_t abs_bits<uint32_t>(uint32_t raw) { return raw & 0x7fffffffu; }                        // fp32

template <typename T, bool kVec>
__global__ void __launch_bounds__(256)
levels_gather_kernel(const unsigned char *__restrict__ blocks, size_t block_stride, T *__restrict__ dst, uint32_t n_ranks, uint32_t nl, uint32_t run,
                     uint32_t lv_begin, uint32_t *__restrict__ gmax_bits) {
    // run = elements of one (rank, level): points of a rank x C.  dst[((lv_begin + l) * n_ranks + r) * run + i] = block_r[l * run + i];
    // blockIdx.y = r * nl + l, the x dimension strides over the run
    constexpr uint32_t kPer = kVec ? 16u / sizeof(T) : 1u;
    const uint32_t r = blockIdx.y / nl, l = blockIdx.y - r * nl, units = run / kPer;
    const T *src = reinterpret_cast<const T *>(blocks + (size_t)r * block_stride) + (size_t)l * run;
    T *out = dst + ((size_t)(lv_begin + l) * n_ranks + r) * run;
    uint32_t m = 0u;
    const uint32_t stride = gridDim.x * blockDim.x;
    if constexpr (kVec) {
        constexpr uint32_t kFlight = 4u;                            // 16-byte loads in flight per lane
        for (uint32_t u0 = blockIdx.x * blockDim.x + threadIdx.x; u0 < units; u0 += kFlight * stride) {
            uint4 q[kFlight];
#pragma unroll
            for (uint32_t j = 0; j < kFlight; ++j) q[j] = reinterpret_cast<const uint4 *>(src)[min(u0 + j * stride, units - 1u)];
#pragma unroll
            for (uint32_t j = 0; j < kFlight; ++j) {
                if (u0 + j * stride < units) reinterpret_cast<uint4 *>(out)[u0 + j * stride] = q[j];
                const uint32_t w[4] = {q[j].x, q[j].y, q[j].z, q[j].w};
#pragma unroll
                for (uint32_t k = 0; k < 4; ++k) {
                    if constexpr (sizeof(T) == 2) m = max(m, max(abs_bits<uint16_t>((uint16_t)(w[k] & 0xffffu)), abs_bits<uint16_t>((uint16_t)(w[k] >> 16))));
                    else m = max(m, abs_bits<uint32_t>(w[k]));
                }
            }
        }
    } else {
        for (uint32_t u = blockIdx.x * blockDim.x + threadIdx.x; u < units; u += stride) {
            const T v = src[u];
            out[u] = v;
            m = max(m, abs_bits<T>(v));
        }
    }
#pragma unroll
    for (uint32_t off = 32u; off != 0u; off >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, (int)off));
    __shared__ uint32_t part[4];
    if ((threadIdx.x & 63u) == 0u) part[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0u) {
        m = max(max(part[0], part[1]), max(part[2], part[3]));
        if (m != 0u) atomicMax(gmax_bits, m);                    // one per workgroup
    }
}

static bool adam_tail_possible_levels(const naf_render_cfg *cfg, const Workspace &w, uint32_t lv_begin, uint32_t lv_end) {
    if (!w.binned || per_level_launches(cfg)) return false;
    const uint32_t NB = 1u << w.plan.log2_nb;
    for (uint32_t l0 = lv_begin; l0 < lv_end; l0 += w.plan.levels_per_pass)
        if (reducer_split(NB, std::min(w.plan.levels_per_pass, lv_end - l0)) != 1u) return false;
    return true;
}

template <typename P, uint32_t C>
static int levels_encode_impl(const float *rays, const float *t_rand, const void *emb, const int32_t *offsets, void *features, uint32_t n_rays,
                              const naf_render_cfg *cfg, uint32_t lv_begin, uint32_t lv_end, uint32_t n_ranks, hipStream_t s) {
    const uint32_t B = n_rays * cfg->n_samples;
    const SrcRays src = make_src(rays, t_rand, cfg);
    return dispatch_encode<P, C>(src, emb, offsets, features, B, cfg, s, lv_begin, lv_end, B / n_ranks);
}

template <typename P, uint32_t C>
static int levels_field_impl(const float *rays, const float *t_rand, const float *target, const float *ray_weight, const void *features,
                             const float *mlp, float *acc, void *feature_grads, float *grad_mlp, float *loss_out, uint32_t n_rays,
                             const naf_render_cfg *cfg, void *ws, hipEvent_t grads_ready, hipStream_t s) {
    const uint32_t B = n_rays * cfg->n_samples;
    const Workspace w = carve(ws, cfg, B);
    const SrcRays src = make_src(rays, t_rand, cfg);
    if (int rc = run_mlp_forward<P, C, true>(features, mlp, src, acc, n_rays, B, cfg, s)) return rc;
    const LossInputs loss{acc, target, ray_weight};
    const StepExtras ex{nullptr, true, nullptr, grads_ready};
    return run_mlp_backward<P, C>(features, mlp, src, nullptr, loss, feature_grads, w.slabs, nullptr, grad_mlp, loss_out, nullptr, n_rays, B, cfg, ex, s);
}

template <typename P, uint32_t C>
static int levels_scatter_impl(const float *rays, const float *t_rand, const void *blocks, size_t block_stride, uint32_t n_ranks,
                               const int32_t *offsets, float *grad_emb, uint32_t n_rays, const naf_render_cfg *cfg, uint32_t lv_begin,
                               uint32_t lv_end, void *ws, const AdamTail *adam, int *adam_applied, hipStream_t s) {
    using T = typename RawWord<sizeof(typename P::feat_t::store_t)>::type;          // the gradients as raw 16- or 32-bit words
    const uint32_t B = n_rays * cfg->n_samples, nl = lv_end - lv_begin;
    const Workspace w = carve(ws, cfg, B);
    const SrcRays src = make_src(rays, t_rand, cfg);
    const uint32_t run = B / n_ranks * C;                                        // elements of one (rank, level)
    uint32_t *gmax = w.binned ? w.gmax : reinterpret_cast<uint32_t *>(w.grad_acc);      // (the atomic scatter has no use for it)
    if (w.binned) {
        if (hipMemsetAsync(w.overflow, 0, 34 * sizeof(uint32_t), s) != hipSuccess) return fail(NAF_ERR_LAUNCH, "levels_scatter: memset failed");      // counters + maximum
    } else if (hipMemsetAsync(gmax, 0, sizeof(uint32_t), s) != hipSuccess) return fail(NAF_ERR_LAUNCH, "levels_scatter: memset failed");
    AdamTail tail;
    const bool fuse = adam != nullptr && adam_tail_possible_levels(cfg, w, lv_begin, lv_end);
    if (fuse) { tail = *adam; tail.overflow = w.overflow; }
    if (adam_applied != nullptr) *adam_applied = fuse ? 1 : 0;
    // The canonical shape (two bf16 channels, 8-byte records) reads the blocks in place: pass 1 finds a point's gradient inside its
    // rank's block and takes the maximum on its way (scatter_v2.h, GradBlocks).  Other shapes, the fp32 parity mode and
    // NAF_CFG_LEVELS_GATHER_PASS keep the pass below.
    if constexpr (sizeof(T) == 2u && C == 2u) {
        const bool in_place = w.binned && scatter_v2(cfg) && (cfg->flags & NAF_CFG_LEVELS_GATHER_PASS) == 0u && block_stride % 4u == 0u &&
                              ((uintptr_t)blocks & 3u) == 0u && (uint64_t)n_ranks * (block_stride / 4u) < (1ull << 32) && B / n_ranks >= 2u;
        if (in_place) {
            const GradBlocks gb = make_grad_blocks(B / n_ranks, (uint32_t)(block_stride / 4u), lv_begin, gmax);
            return run_hash_backward_levels<P, C>(src, blocks, offsets, grad_emb, B, cfg, w, lv_begin, lv_end, s, nullptr, fuse ? &tail : nullptr, nullptr, nullptr, &gb);
        }
    }
    const bool vec = (run * sizeof(T)) % 16u == 0u && block_stride % 16u == 0u && ((uintptr_t)blocks & 15u) == 0u && ((uintptr_t)w.dfeat & 15u) == 0u;
    if ((uint64_t)n_ranks * nl > 65535u) return fail(NAF_ERR_INVALID_ARGUMENT, "levels_scatter: too many (rank, level) blocks");
    const uint32_t units = vec ? run / (uint32_t)(16u / sizeof(T)) : run;
    // ~512 workgroups in all: each ends with ONE atomic on the same word, and same-address atomics retire ~12 ns apart
    const uint32_t gx = std::max<uint32_t>(1u, std::min<uint32_t>((units + 1023u) / 1024u, std::max<uint32_t>(1u, 512u / (n_ranks * nl))));
    {
        ProfScope prof_("levels_gather_kernel", s);
        if (vec) hipLaunchKernelGGL((levels_gather_kernel<T, true>), dim3(gx, n_ranks * nl), dim3(256), 0, s, (const unsigned char *)blocks, block_stride, (T *)w.dfeat, n_ranks, nl, run, lv_begin, gmax);
        else hipLaunchKernelGGL((levels_gather_kernel<T, false>), dim3(gx, n_ranks * nl), dim3(256), 0, s, (const unsigned char *)blocks, block_stride, (T *)w.dfeat, n_ranks, nl, run, lv_begin, gmax);
    }
    if (int rc = check_launch("levels_gather_kernel")) return rc;
    return run_hash_backward_levels<P, C>(src, w.dfeat, offsets, grad_emb, B, cfg, w, lv_begin, lv_end, s, nullptr, fuse ? &tail : nullptr, nullptr);
}

#define NAF_DISPATCH_PC(FN, ...)                                                                      \
    do {                                                                                              \
        if (cfg->mlp_precision == NAF_F32) {                                                          \
            switch (cfg->C) {                                                                         \
                case 1: return FN<PrecF32, 1>(__VA_ARGS__);                                           \
                case 2: return FN<PrecF32, 2>(__VA_ARGS__);                                           \
                case 4: return FN<PrecF32, 4>(__VA_ARGS__);                                           \
                default: return FN<PrecF32, 8>(__VA_ARGS__);                                          \
            }                                                                                         \
        }                                                                                             \
        switch (cfg->C) {                                                                             \
            case 1: return FN<PrecBF16, 1>(__VA_ARGS__);                                              \
            case 2: return FN<PrecBF16, 2>(__VA_ARGS__);                                              \
            case 4: return FN<PrecBF16, 4>(__VA_ARGS__);                                              \
            default: return FN<PrecBF16, 8>(__VA_ARGS__);                                             \
        }                                                                                             \
    } while (0)

}  // namespace naf

using namespace naf;

#ifdef NAF_REDUCE_STAMPS
extern "C" int naf_debug_reduce_stamps(uint32_t *host, uint32_t n_words) {      // diagnostic builds only (tools/reduce_stamps.py)
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(naf::g_reduce_stamps), (size_t)std::min<uint32_t>(n_words, 2048u * 8u) * 4u) == hipSuccess ? 0 : 1;
}
#endif

extern "C" int naf_scatter_overflow_count(const naf_render_cfg *cfg, uint64_t n_points, const void *workspace, uint32_t *count_host) {
    if (!cfg || !workspace || !count_host) return fail(NAF_ERR_INVALID_ARGUMENT, "scatter_overflow_count: null pointer");
    const Workspace w = carve(const_cast<void *>(workspace), cfg, n_points);
    *count_host = 0;
    if (!w.binned) return NAF_OK;
    if (hipMemcpy(count_host, w.overflow, 4, hipMemcpyDeviceToHost) != hipSuccess) return fail(NAF_ERR_LAUNCH, "scatter_overflow_count: copy failed");
    return NAF_OK;
}

extern "C" int naf_scatter_overflow_levels(const naf_render_cfg *cfg, uint64_t n_points, const void *workspace, uint32_t *counts_host) {
    if (!cfg || !workspace || !counts_host) return fail(NAF_ERR_INVALID_ARGUMENT, "scatter_overflow_levels: null pointer");
    const Workspace w = carve(const_cast<void *>(workspace), cfg, n_points);
    std::memset(counts_host, 0, 32 * sizeof(uint32_t));
    if (!w.binned) return NAF_OK;
    if (hipMemcpy(counts_host, w.overflow + 1, 32 * 4, hipMemcpyDeviceToHost) != hipSuccess) return fail(NAF_ERR_LAUNCH, "scatter_overflow_levels: copy failed");
    return NAF_OK;
}

extern "C" size_t naf_render_workspace_bytes(const naf_render_cfg *cfg, uint64_t n_points) {
    if (!cfg) return 0;
    return carve(nullptr, cfg, n_points).bytes;
}

extern "C" size_t naf_forward_workspace_bytes(const naf_render_cfg *cfg, uint64_t n_points) {
    if (!cfg) return 0;
    return forward_workspace_bytes(cfg, n_points);
}

static int check_points(uint64_t n_points) {
    if (n_points >= (1ull << 31)) return fail(NAF_ERR_INVALID_ARGUMENT, "fused field: more than 2^31 points per call; split the batch");
    return NAF_OK;
}
// NAF_CFG_EXPLICIT_DEPTHS makes t_rand the [n_rays, n_samples] depths themselves: it cannot be absent then
static int check_depths(const naf_render_cfg *cfg, const float *t_rand) {
    if ((cfg->flags & NAF_CFG_EXPLICIT_DEPTHS) != 0u && t_rand == nullptr)
        return fail(NAF_ERR_INVALID_ARGUMENT, "render: NAF_CFG_EXPLICIT_DEPTHS needs the depths in t_rand");
    return NAF_OK;
}

extern "C" int naf_render_forward(const float *rays, const float *t_rand, const void *embeddings, const int32_t *offsets,
                                  const float *mlp, float *acc, uint32_t n_rays, const naf_render_cfg *cfg, void *workspace,
                                  void *stream) {
    if (int rc = check_cfg(cfg, "render_forward")) return rc;
    if (int rc = check_depths(cfg, t_rand)) return rc;
    if (n_rays != 0 && (!rays || !embeddings || !offsets || !mlp || !acc || !workspace)) return fail(NAF_ERR_INVALID_ARGUMENT, "render_forward: null pointer");
    if (cfg->n_samples < 2) return fail(NAF_ERR_INVALID_ARGUMENT, "render_forward: n_samples must be >= 2");
    if (int rc = check_points((uint64_t)n_rays * cfg->n_samples)) return rc;
    if (n_rays == 0) return NAF_OK;
    NAF_DISPATCH_PC(render_forward_impl, rays, t_rand, embeddings, offsets, mlp, acc, n_rays, cfg, workspace, (hipStream_t)stream);
}

extern "C" int naf_render_forward_samples(const float *rays, const float *t_rand, const void *embeddings, const int32_t *offsets,
                                          const float *mlp, float *acc, float *sigma, float *optical_depth, uint32_t n_rays,
                                          const naf_render_cfg *cfg, void *workspace, void *stream) {
    if (int rc = check_cfg(cfg, "render_forward_samples")) return rc;
    if (int rc = check_depths(cfg, t_rand)) return rc;
    if (n_rays != 0 && (!rays || !embeddings || !offsets || !mlp || !acc || !workspace)) return fail(NAF_ERR_INVALID_ARGUMENT, "render_forward_samples: null pointer");
    if (cfg->n_samples < 2) return fail(NAF_ERR_INVALID_ARGUMENT, "render_forward_samples: n_samples must be >= 2");
    if (int rc = check_points((uint64_t)n_rays * cfg->n_samples)) return rc;
    if (n_rays == 0) return NAF_OK;
    NAF_DISPATCH_PC(render_forward_impl, rays, t_rand, embeddings, offsets, mlp, acc, n_rays, cfg, workspace, (hipStream_t)stream, sigma,
                    optical_depth);
}

extern "C" int naf_render_backward(const float *rays, const float *t_rand, const float *grad_acc, const void *embeddings,
                                   const int32_t *offsets, const float *mlp, float *grad_embeddings, float *grad_mlp,
                                   uint32_t n_rays, const naf_render_cfg *cfg, void *workspace, int features_valid, void *stream) {
    if (int rc = check_cfg(cfg, "render_backward")) return rc;
    if (int rc = check_depths(cfg, t_rand)) return rc;
    if (n_rays != 0 && (!rays || !grad_acc || !embeddings || !offsets || !mlp || !grad_embeddings || !grad_mlp || !workspace))
        return fail(NAF_ERR_INVALID_ARGUMENT, "render_backward: null pointer");
    if (cfg->n_samples < 2) return fail(NAF_ERR_INVALID_ARGUMENT, "render_backward: n_samples must be >= 2");
    if (int rc = check_points((uint64_t)n_rays * cfg->n_samples)) return rc;
    if (n_rays == 0) return NAF_OK;
    NAF_DISPATCH_PC(render_backward_impl, rays, t_rand, grad_acc, embeddings, offsets, mlp, grad_embeddings, grad_mlp, n_rays, cfg,
                    workspace, features_valid, nullptr, (hipStream_t)stream);
}

static int check_buckets(const naf_grad_buckets *b, uint32_t L) {
    if (b->n_buckets == 0 || b->n_buckets > NAF_MAX_GRAD_BUCKETS) return fail(NAF_ERR_INVALID_ARGUMENT, "render_train_bucketed: n_buckets must be in [1, 16]");
    uint32_t seen = 0;                                       // every level exactly once (L <= 32 on the fused path)
    for (uint32_t i = 0; i < b->n_buckets; ++i) {
        if (b->level_begin[i] >= b->level_end[i] || b->level_end[i] > L) return fail(NAF_ERR_INVALID_ARGUMENT, "render_train_bucketed: empty or out-of-range bucket");
        for (uint32_t l = b->level_begin[i]; l < b->level_end[i]; ++l) {
            if (seen & (1u << l)) return fail(NAF_ERR_INVALID_ARGUMENT, "render_train_bucketed: buckets overlap");
            seen |= 1u << l;
        }
    }
    if (seen != (L >= 32u ? 0xffffffffu : (1u << L) - 1u)) return fail(NAF_ERR_INVALID_ARGUMENT, "render_train_bucketed: buckets do not cover all levels");
    return NAF_OK;
}

static int render_train_entry(const float *rays, const float *t_rand, const float *target, const float *ray_weight,
                              const void *embeddings, const int32_t *offsets, const float *mlp, float *acc,
                              float *grad_embeddings, float *grad_mlp, float *loss_out, uint32_t n_rays,
                              const naf_render_cfg *cfg, void *workspace, const naf_grad_buckets *buckets, void *stream,
                              const AdamTail *adam = nullptr, const MlpAdam *madam = nullptr, bool loss_assign = false, DrawJob *next_draw = nullptr) {
    if (int rc = check_cfg(cfg, "render_train")) return rc;
    if (int rc = check_depths(cfg, t_rand)) return rc;
    if (n_rays != 0 && (!rays || !target || !ray_weight || !embeddings || !offsets || !mlp || !acc || !grad_embeddings || !grad_mlp || !workspace))
        return fail(NAF_ERR_INVALID_ARGUMENT, "render_train: null pointer");
    if (cfg->n_samples < 2) return fail(NAF_ERR_INVALID_ARGUMENT, "render_train: n_samples must be >= 2");
    if (int rc = check_points((uint64_t)n_rays * cfg->n_samples)) return rc;
    if (buckets != nullptr)
        if (int rc = check_buckets(buckets, cfg->L)) return rc;
    if (n_rays == 0) {                                       // an empty shard still has to signal its (zero) gradients as final
        if (loss_assign && loss_out != nullptr && hipMemsetAsync(loss_out, 0, 4, (hipStream_t)stream) != hipSuccess) return fail(NAF_ERR_LAUNCH, "render_train: memset failed");
        if (buckets != nullptr) {
            if (buckets->mlp_ready && hipEventRecord((hipEvent_t)buckets->mlp_ready, (hipStream_t)stream) != hipSuccess) return fail(NAF_ERR_LAUNCH, "render_train: event");
            for (uint32_t b = 0; b < buckets->n_buckets; ++b)
                if (buckets->ready[b] && hipEventRecord((hipEvent_t)buckets->ready[b], (hipStream_t)stream) != hipSuccess) return fail(NAF_ERR_LAUNCH, "render_train: event");
        }
        return NAF_OK;
    }
    NAF_DISPATCH_PC(render_train_impl, rays, t_rand, target, ray_weight, embeddings, offsets, mlp, acc, grad_embeddings, grad_mlp,
                    loss_out, n_rays, cfg, workspace, buckets, (hipStream_t)stream, adam, madam, loss_assign, next_draw);
}

extern "C" int naf_render_train(const float *rays, const float *t_rand, const float *target, const float *ray_weight,
                                const void *embeddings, const int32_t *offsets, const float *mlp, float *acc,
                                float *grad_embeddings, float *grad_mlp, float *loss_out, uint32_t n_rays,
                                const naf_render_cfg *cfg, void *workspace, void *stream) {
    return render_train_entry(rays, t_rand, target, ray_weight, embeddings, offsets, mlp, acc, grad_embeddings, grad_mlp, loss_out,
                              n_rays, cfg, workspace, nullptr, stream);
}

extern "C" int naf_render_train_bucketed(const float *rays, const float *t_rand, const float *target, const float *ray_weight,
                                         const void *embeddings, const int32_t *offsets, const float *mlp, float *acc,
                                         float *grad_embeddings, float *grad_mlp, float *loss_out, uint32_t n_rays,
                                         const naf_render_cfg *cfg, void *workspace, const naf_grad_buckets *buckets, void *stream) {
    if (!buckets) return fail(NAF_ERR_INVALID_ARGUMENT, "render_train_bucketed: null buckets");
    return render_train_entry(rays, t_rand, target, ray_weight, embeddings, offsets, mlp, acc, grad_embeddings, grad_mlp, loss_out,
                              n_rays, cfg, workspace, buckets, stream);
}

static_assert(kAdamLpF16 == NAF_F16 && kAdamLpBF16 == NAF_BF16, "adam_math.h mirrors naf_dtype");

static int render_train_adam_impl(const float *rays, const float *t_rand, const float *target, const float *ray_weight,
                                  const void *embeddings, const int32_t *offsets, const float *mlp, float *acc,
                                  float *grad_embeddings, float *grad_mlp, float *loss_out, uint32_t n_rays,
                                  const naf_render_cfg *cfg, void *workspace, const naf_table_adam *adam, void *stream, DrawJob *next_draw) {
    if (!adam) return fail(NAF_ERR_INVALID_ARGUMENT, "render_train_adam: null adam");
    if (!adam->param || !adam->exp_avg || !adam->exp_avg_sq || !grad_embeddings) return fail(NAF_ERR_INVALID_ARGUMENT, "render_train_adam: null pointer");
    if (adam->step == 0) return fail(NAF_ERR_INVALID_ARGUMENT, "render_train_adam: step is 1-based");
    if (adam->param_lp != nullptr && adam->lp_dtype != NAF_F16 && adam->lp_dtype != NAF_BF16)
        return fail(NAF_ERR_UNSUPPORTED, "render_train_adam: lp_dtype must be NAF_F16 or NAF_BF16 when param_lp is given");
    if (((uintptr_t)adam->param | (uintptr_t)adam->exp_avg | (uintptr_t)adam->exp_avg_sq | (uintptr_t)grad_embeddings) & 15u)
        return fail(NAF_ERR_INVALID_ARGUMENT, "render_train_adam: buffers must be 16-byte aligned");
    if (int rc = check_cfg(cfg, "render_train_adam")) return rc;
    AdamTail tail;
    tail.param = adam->param; tail.m = adam->exp_avg; tail.v = adam->exp_avg_sq;
    tail.lp = adam->param_lp; tail.lp_dtype = adam->lp_dtype; tail.overflow = nullptr;
    tail.a = make_adam_args(adam->lr, adam->beta1, adam->beta2, adam->eps, adam->step, adam->grad_scale);
    // the MLP's own update rides on the slab reduction when the caller hands over its state (an empty batch still has to step it)
    MlpAdam madam{adam->mlp_param, adam->mlp_exp_avg, adam->mlp_exp_avg_sq, tail.a};
    const MlpAdam *mp = nullptr;
    if (adam->mlp_param != nullptr) {
        if (!adam->mlp_exp_avg || !adam->mlp_exp_avg_sq || !grad_mlp) return fail(NAF_ERR_INVALID_ARGUMENT, "render_train_adam: null MLP optimiser state");
        if (adam->mlp_param != mlp) return fail(NAF_ERR_INVALID_ARGUMENT, "render_train_adam: mlp_param must be the parameter block the step reads (`mlp`)");
        if (n_rays != 0) mp = &madam;
        else if (int rc = launch_adam(adam->mlp_param, adam->mlp_exp_avg, adam->mlp_exp_avg_sq, grad_mlp, nullptr, 0, NAF_MLP_PARAMS, tail.a, true, (hipStream_t)stream)) return rc;
    }
    const uint64_t n_points = (uint64_t)n_rays * cfg->n_samples;
    if (n_rays != 0 && workspace != nullptr && n_points < (1ull << 31) && adam_tail_possible(cfg, carve(workspace, cfg, n_points)))
        return render_train_entry(rays, t_rand, target, ray_weight, embeddings, offsets, mlp, acc, grad_embeddings, grad_mlp, loss_out,
                                  n_rays, cfg, workspace, nullptr, stream, &tail, mp, true, next_draw);
    // small batches (atomic scatter), split reducer launches, per-level diagnostics, empty batches: the two passes one after the other
    if (int rc = render_train_entry(rays, t_rand, target, ray_weight, embeddings, offsets, mlp, acc, grad_embeddings, grad_mlp, loss_out,
                                    n_rays, cfg, workspace, nullptr, stream, nullptr, mp, true)) return rc;
    return launch_adam(adam->param, adam->exp_avg, adam->exp_avg_sq, grad_embeddings, adam->param_lp, adam->lp_dtype, adam->n, tail.a,
                       true, (hipStream_t)stream);
}

extern "C" int naf_render_train_adam(const float *rays, const float *t_rand, const float *target, const float *ray_weight,
                                     const void *embeddings, const int32_t *offsets, const float *mlp, float *acc,
                                     float *grad_embeddings, float *grad_mlp, float *loss_out, uint32_t n_rays,
                                     const naf_render_cfg *cfg, void *workspace, const naf_table_adam *adam, void *stream) {
    return render_train_adam_impl(rays, t_rand, target, ray_weight, embeddings, offsets, mlp, acc, grad_embeddings, grad_mlp, loss_out, n_rays,
                                  cfg, workspace, adam, stream, nullptr);
}

// ... and the same step carrying the pixel draw of the NEXT one (naf_hip.h): in spare workgroups of pass 1 of the binned scatter where
// the step runs it (the canonical bf16 shape), as a launch of its own behind the step everywhere else -- same results either way.
extern "C" int naf_render_train_adam_draw(const float *rays, const float *t_rand, const float *target, const float *ray_weight,
                                          const void *embeddings, const int32_t *offsets, const float *mlp, float *acc,
                                          float *grad_embeddings, float *grad_mlp, float *loss_out, uint32_t n_rays,
                                          const naf_render_cfg *cfg, void *workspace, const naf_table_adam *adam,
                                          const naf_next_draw *next, void *stream) {
    if (next == nullptr)
        return render_train_adam_impl(rays, t_rand, target, ray_weight, embeddings, offsets, mlp, acc, grad_embeddings, grad_mlp, loss_out, n_rays,
                                      cfg, workspace, adam, stream, nullptr);
    if (next->rays == rays || (next->target != nullptr && next->target == target))
        return fail(NAF_ERR_INVALID_ARGUMENT, "render_train_adam_draw: the next step's rays / targets must not be the buffers this step reads");
    DrawJob job;
    if (int rc = make_draw_job(&next->draw, next->poses, next->projections, next->pixels, next->target, next->rays, next->first_draw, next->n_draws,
                               next->n_projections, next->det_w, next->det_h, next->du, next->dv, next->ou, next->ov, next->DSD, next->near,
                               next->far, next->parallel, next->seed, &job)) return rc;
    if (int rc = render_train_adam_impl(rays, t_rand, target, ray_weight, embeddings, offsets, mlp, acc, grad_embeddings, grad_mlp, loss_out, n_rays,
                                        cfg, workspace, adam, stream, &job)) return rc;
    return launch_draw(job, (hipStream_t)stream);            // count == 0 when pass 1 took it along
}

/* ---- level-parallel training (naf_hip.h) --------------------------------------------------------------------------------- */
static int check_levels(const naf_render_cfg *cfg, uint32_t lv_begin, uint32_t lv_end, const char *who) {
    if (lv_begin >= lv_end || lv_end > cfg->L) { (void)who; return fail(NAF_ERR_INVALID_ARGUMENT, "levels: empty or out-of-range level range"); }
    return NAF_OK;
}

extern "C" int naf_levels_encode(const float *rays, const float *t_rand, const void *embeddings, const int32_t *offsets, void *features,
                                 uint32_t n_rays, uint32_t n_ranks, const naf_render_cfg *cfg, uint32_t level_begin, uint32_t level_end,
                                 void *stream) {
    if (int rc = check_cfg(cfg, "levels_encode")) return rc;
    if (int rc = check_depths(cfg, t_rand)) return rc;
    if (int rc = check_levels(cfg, level_begin, level_end, "levels_encode")) return rc;
    if (n_rays != 0 && (!rays || !embeddings || !offsets || !features)) return fail(NAF_ERR_INVALID_ARGUMENT, "levels_encode: null pointer");
    if (cfg->n_samples < 2) return fail(NAF_ERR_INVALID_ARGUMENT, "levels_encode: n_samples must be >= 2");
    if (int rc = check_points((uint64_t)n_rays * cfg->n_samples)) return rc;
    if (n_rays == 0) return NAF_OK;
    if (n_ranks == 0 || n_rays % n_ranks != 0) return fail(NAF_ERR_INVALID_ARGUMENT, "levels_encode: every rank must contribute the same number of rays");
    NAF_DISPATCH_PC(levels_encode_impl, rays, t_rand, embeddings, offsets, features, n_rays, cfg, level_begin, level_end, n_ranks, (hipStream_t)stream);
}

extern "C" int naf_levels_field_step(const float *rays, const float *t_rand, const float *target, const float *ray_weight, const void *features,
                                     const float *mlp, float *acc, void *feature_grads, float *grad_mlp, float *loss_out, uint32_t n_rays,
                                     const naf_render_cfg *cfg, void *workspace, void *grads_ready, void *stream) {
    if (int rc = check_cfg(cfg, "levels_field_step")) return rc;
    if (int rc = check_depths(cfg, t_rand)) return rc;
    if (n_rays == 0) {
        if (grads_ready != nullptr && hipEventRecord((hipEvent_t)grads_ready, (hipStream_t)stream) != hipSuccess) return fail(NAF_ERR_LAUNCH, "levels_field_step: event");
        return NAF_OK;
    }
    if (!rays || !target || !ray_weight || !features || !mlp || !acc || !feature_grads || !grad_mlp || !loss_out || !workspace)
        return fail(NAF_ERR_INVALID_ARGUMENT, "levels_field_step: null pointer");
    if (cfg->n_samples < 2) return fail(NAF_ERR_INVALID_ARGUMENT, "levels_field_step: n_samples must be >= 2");
    if (int rc = check_points((uint64_t)n_rays * cfg->n_samples)) return rc;
    NAF_DISPATCH_PC(levels_field_impl, rays, t_rand, target, ray_weight, features, mlp, acc, feature_grads, grad_mlp, loss_out, n_rays, cfg,
                    workspace, (hipEvent_t)grads_ready, (hipStream_t)stream);
}

extern "C" int naf_levels_scatter(const float *rays, const float *t_rand, const void *grad_blocks, size_t block_stride_bytes, uint32_t n_ranks,
                                  const int32_t *offsets, float *grad_embeddings, uint32_t n_rays, const naf_render_cfg *cfg,
                                  uint32_t level_begin, uint32_t level_end, void *workspace, const naf_table_adam *adam, int *adam_applied,
                                  void *stream) {
    if (adam_applied != nullptr) *adam_applied = 0;
    if (int rc = check_cfg(cfg, "levels_scatter")) return rc;
    if (int rc = check_depths(cfg, t_rand)) return rc;
    if (int rc = check_levels(cfg, level_begin, level_end, "levels_scatter")) return rc;
    if (n_rays == 0) return NAF_OK;
    if (!rays || !grad_blocks || !offsets || !grad_embeddings || !workspace) return fail(NAF_ERR_INVALID_ARGUMENT, "levels_scatter: null pointer");
    if (n_ranks == 0 || n_rays % n_ranks != 0) return fail(NAF_ERR_INVALID_ARGUMENT, "levels_scatter: every rank must contribute the same number of rays");
    if (cfg->n_samples < 2) return fail(NAF_ERR_INVALID_ARGUMENT, "levels_scatter: n_samples must be >= 2");
    if (int rc = check_points((uint64_t)n_rays * cfg->n_samples)) return rc;
    const size_t esz = cfg->mlp_precision == NAF_F32 ? 4 : 2;
    if (block_stride_bytes < (size_t)(level_end - level_begin) * (n_rays / n_ranks) * cfg->n_samples * cfg->C * esz || block_stride_bytes % esz != 0)
        return fail(NAF_ERR_INVALID_ARGUMENT, "levels_scatter: block stride smaller than a rank's block");
    AdamTail tail;
    const AdamTail *tp = nullptr;
    if (adam != nullptr) {
        if (!adam->param || !adam->exp_avg || !adam->exp_avg_sq) return fail(NAF_ERR_INVALID_ARGUMENT, "levels_scatter: null optimiser state");
        if (adam->step == 0) return fail(NAF_ERR_INVALID_ARGUMENT, "levels_scatter: step is 1-based");
        if (adam->param_lp != nullptr && adam->lp_dtype != NAF_F16 && adam->lp_dtype != NAF_BF16)
            return fail(NAF_ERR_UNSUPPORTED, "levels_scatter: lp_dtype must be NAF_F16 or NAF_BF16 when param_lp is given");
        if (((uintptr_t)adam->param | (uintptr_t)adam->exp_avg | (uintptr_t)adam->exp_avg_sq | (uintptr_t)grad_embeddings) & 15u)
            return fail(NAF_ERR_INVALID_ARGUMENT, "levels_scatter: buffers must be 16-byte aligned");
        tail.param = adam->param; tail.m = adam->exp_avg; tail.v = adam->exp_avg_sq;
        tail.lp = adam->param_lp; tail.lp_dtype = adam->lp_dtype; tail.overflow = nullptr;
        tail.a = make_adam_args(adam->lr, adam->beta1, adam->beta2, adam->eps, adam->step, adam->grad_scale);
        tp = &tail;
    }
    NAF_DISPATCH_PC(levels_scatter_impl, rays, t_rand, grad_blocks, block_stride_bytes, n_ranks, offsets, grad_embeddings, n_rays, cfg, level_begin,
                    level_end, workspace, tp, adam_applied, (hipStream_t)stream);
}

extern "C" int naf_field_forward(const float *pts, const void *embeddings, const int32_t *offsets, const float *mlp,
                                 float *sigma, uint32_t B, const naf_render_cfg *cfg, void *workspace, void *stream) {
    if (int rc = check_cfg(cfg, "field_forward")) return rc;
    if (B != 0 && (!pts || !embeddings || !offsets || !mlp || !sigma || !workspace)) return fail(NAF_ERR_INVALID_ARGUMENT, "field_forward: null pointer");
    if (int rc = check_points(B)) return rc;
    if (B == 0) return NAF_OK;
    NAF_DISPATCH_PC(field_forward_impl, pts, embeddings, offsets, mlp, sigma, B, cfg, workspace, (hipStream_t)stream);
}

extern "C" int naf_field_forward_grid(const double *start, const double *stop, const uint32_t *dims, const void *embeddings,
                                      const int32_t *offsets, const float *mlp, float *sigma, const naf_render_cfg *cfg,
                                      void *workspace, size_t workspace_bytes, void *stream) {
    if (int rc = check_cfg(cfg, "field_forward_grid")) return rc;
    if (!start || !stop || !dims || !embeddings || !offsets || !mlp || !sigma || !workspace)
        return fail(NAF_ERR_INVALID_ARGUMENT, "field_forward_grid: null pointer");
    SrcGrid src;
    uint64_t total = 1;
    for (int d = 0; d < 3; ++d) {
        if (dims[d] == 0) return fail(NAF_ERR_INVALID_ARGUMENT, "field_forward_grid: empty axis");
        // the reference raises for points outside [-bound, bound] (hashgrid.py:122-123); a grid is checked at its ends
        if (!(std::fabs(start[d]) <= (double)cfg->bound) || !(std::fabs(stop[d]) <= (double)cfg->bound))
            return fail(NAF_ERR_INVALID_ARGUMENT, "field_forward_grid: grid exceeds [-bound, bound]");
        src.start[d] = start[d];
        src.stop[d] = stop[d];
        src.step[d] = dims[d] > 1 ? (stop[d] - start[d]) / (double)(dims[d] - 1) : 0.0;      // numpy.linspace's step
        src.n[d] = dims[d];
        total *= dims[d];                                    // < 2^31 * 2^32 per step: checked before it can wrap
        if (int rc = check_points(total)) return rc;
    }
    src.bound = cfg->bound;
    // The grid is walked in ranges of its traversal order that fit the caller's workspace (every point is evaluated on its own,
    // so the ranges give the same bits as one call): a 1024^3 query needs 64 GiB of features at once, or any smaller buffer.
    const size_t per_point = forward_workspace_bytes(cfg, 1u << 20) > 256 ? (size_t)cfg->L * cfg->C * (cfg->mlp_precision == NAF_F32 ? 4u : 2u) : 0u;
    uint64_t chunk = total;
    if (per_point != 0u) {
        if (workspace_bytes < 512u + 1024u * per_point) return fail(NAF_ERR_INVALID_ARGUMENT, "field_forward_grid: workspace too small (see naf_forward_workspace_bytes)");
        chunk = std::min<uint64_t>(total, ((workspace_bytes - 512u) / per_point) & ~(uint64_t)1023u);
    }
    for (uint64_t first = 0; first < total; first += chunk) {
        src.first = (uint32_t)first;
        const uint32_t B = (uint32_t)std::min<uint64_t>(chunk, total - first);
        const int rc = [&]() -> int { NAF_DISPATCH_PC(field_forward_grid_impl, src, embeddings, offsets, mlp, sigma, B, cfg, workspace, (hipStream_t)stream); }();
        if (rc != NAF_OK) return rc;
    }
    return NAF_OK;
}
