// hash_kernels.h -- templated gfx950 kernels of the multi-resolution hash-grid encoder.
//
// Shared by hash_encode.hip (stand-alone operator, reference ABI) and render_fused.hip (points generated on the
// fly from rays).  Follows reference src/encoder/hashencoder/src/hashencoder.cu:
//   forward  kernel_grid           :77-198      backward  kernel_grid_backward :201-272
//
// MI355X design notes
//   * one lane = one (point, level); a wave covers 64 consecutive points, so consecutive samples of a ray sit in
//     neighbouring lanes and share cache lines on the coarse levels;
//   * level-major grid (blockIdx.y = level): at any moment an XCD works on one level, whose slice (<= 4 MB fp32 /
//     2 MB 16-bit at T=2^19) is what its 4 MB L2 has to hold;
//   * the level regime (dense / wrapped-dense / hash, mask vs modulo) is decoded once per wave on the scalar unit;
//   * C-wide table rows are fetched with one 4/8/16-byte load; all 2^D gathers of a lane are issued back to back
//     before the first use, so every lane keeps 2^D independent misses in flight;
//   * [L,B,C] features are written as fully coalesced C*sizeof(T)*64-byte rows per wave.
#pragma once

#include <cstdlib>

#include "naf_device.h"

namespace naf {

// ---- where the points come from -----------------------------------------------------------------------
// (a) caller-supplied coordinates already in [0,1]  (reference kernel contract, hashencoder.cu:383)
template <uint32_t D>
struct SrcUnit {
    static constexpr bool kInRange = false;     // the reference kernel stays in bounds for any input (index % hashmap_size)
    const float *__restrict__ x;
    uint32_t n = 0;                             // points behind x (only sample_spacing() looks at it)
    __device__ __forceinline__ void get(uint32_t b, float (&out)[D]) const {
#pragma unroll
        for (uint32_t d = 0; d < D; ++d) out[d] = x[(size_t)b * D + d];
    }
    // distance between the first two points (for the locality heuristic of the binned scatter: consecutive samples of a ray)
    __device__ __forceinline__ float sample_spacing() const {
        if (n < 2u) return 1.0f;
        float m = 0.0f;
#pragma unroll
        for (uint32_t d = 0; d < D; ++d) m = fmaxf(m, fabsf(x[D + d] - x[d]));
        return m;
    }
};

// Correctly rounded x / d for a wave-uniform divisor with r = RN(1/d): q0 = x*r, q = fma(fma(-q0, d, x), r, q0)
// (Markstein).  Checked exhaustively against IEEE division for every float x in (0, 2*bound] and the bounds the
// configs use; three instructions instead of the ~12 of the generic division sequence.
__device__ __forceinline__ float div_exact(float x, float d, float r) {
    const float q0 = x * r;
    return __fmaf_rn(__fmaf_rn(-q0, d, x), r, q0);
}

// (b) raw coordinates in [-bound, bound]: the (x+size)/(2 size) of hashgrid.py:125 is applied in registers
struct SrcRaw {
    static constexpr bool kInRange = false;
    const float *__restrict__ pts;
    float bound;
    __device__ __forceinline__ void get(uint32_t b, float (&out)[3]) const {
        const float denom = 2.0f * bound, rden = 1.0f / denom;
#pragma unroll
        for (uint32_t d = 0; d < 3; ++d) out[d] = div_exact(pts[(size_t)b * 3 + d] + bound, denom, rden);
    }
};

// (b') points of a regular grid, generated instead of read: axis k has n[k] values np.linspace(start[k], stop[k], n[k])
//      (tigre.py:388-400 builds the voxel grid that way, in float64, and the dataset casts it to float32).  The traversal is
//      axis 0 FASTEST, then axis 2, then axis 1: the 64 points of a wave are neighbours along x, whose rows share 64-row
//      blocks on the hashed levels (r = x ^ h(y,z)), so their gathers hit the same few lines -- the volume query of
//      train.py:246-250 reads the table through L1 instead of missing it with every corner.
struct SrcGrid {
    static constexpr bool kInRange = false;
    double start[3], step[3], stop[3];
    uint32_t n[3];
    float bound;
    uint32_t first;                    // a launch covers the points [first, first + B) of the traversal (chunked queries)
    // internal point index -> grid indices (axis 0 fastest, then 2, then 1)
    __device__ __forceinline__ void index(uint32_t b, uint32_t (&i)[3]) const {
        b += first;
        i[0] = b % n[0];
        const uint32_t rest = b / n[0];
        i[2] = rest % n[2];
        i[1] = rest / n[2];
    }
    // position in the caller's [n0, n1, n2] array (axis 2 fastest, 'ij' meshgrid order)
    __device__ __forceinline__ size_t flat(uint32_t b) const {
        uint32_t i[3];
        index(b, i);
        return ((size_t)i[0] * n[1] + i[1]) * n[2] + i[2];
    }
    __device__ __forceinline__ void get(uint32_t b, float (&out)[3]) const {
        uint32_t i[3];
        index(b, i);
        const float denom = 2.0f * bound, rden = 1.0f / denom;
#pragma unroll
        for (uint32_t d = 0; d < 3; ++d) {
            // numpy.linspace: start + i * step in float64 (a multiply and an add), the last value is `stop` itself
            const double v = i[d] + 1u == n[d] && n[d] > 1u ? stop[d] : __dadd_rn(__dmul_rn((double)i[d], step[d]), start[d]);
            out[d] = div_exact((float)v + bound, denom, rden);
        }
    }
};

// (c) sample s of ray r (b = r*S + s): stratified depth, point on ray, clamp, normalise
//     (render.py:87-105 + hashgrid.py:125) -- nothing [B,3]-sized ever touches HBM.
struct SrcRays {
    static constexpr bool kInRange = true;       // get() clamps to +-(bound - 1e-6) like render.py:104-105; a NaN stays NaN and lands in cell 0 (locate)
    const float *__restrict__ rays;    // [n_rays, 8]
    const float *__restrict__ t_rand;  // [n_rays, S] or nullptr
    uint32_t S;
    bool perturb;
    bool explicit_z;                   // t_rand holds the depths themselves (NAF_CFG_EXPLICIT_DEPTHS: the fine pass)
    float bound;
    uint64_t seed;
    uint32_t ray_base;
    uint32_t div_magic;                // floor(2^32 / S): b / S without an integer division (host-filled, make_src)
    float lin_step;                    // 1 / (S - 1), the step of torch.linspace(0, 1, S)
    float rden;                        // 1 / (2 bound)

    __device__ __forceinline__ void split(uint32_t b, uint32_t &r, uint32_t &s) const {
        r = __umulhi(b, div_magic);    // floor(b * floor(2^32/S) / 2^32) is floor(b/S) or one less (b < 2^31)
        s = b - r * S;
        if (s >= S) { s -= S; r += 1u; }
    }
    __device__ __forceinline__ float lin(uint32_t i) const {
        return (i < S / 2u) ? (float)i * lin_step : __fmaf_rn(-(float)(S - 1u - i), lin_step, 1.0f);
    }
    __device__ __forceinline__ float base(float near, float far, uint32_t i) const {
        const float t = lin(i);
        return near * (1.0f - t) + far * t;
    }
    __device__ __forceinline__ float depth(uint32_t r, uint32_t s, float near, float far) const {
        if (explicit_z) return t_rand[(size_t)r * S + s];                  // wave-uniform branch
        const float z = base(near, far, s);
        if (!perturb) return z;
        const float u = t_rand ? t_rand[(size_t)r * S + s] : jitter(seed, ray_base + r, s);
        const float lower = s == 0u ? z : 0.5f * (z + base(near, far, s - 1u));
        const float upper = s + 1u == S ? z : 0.5f * (base(near, far, s + 1u) + z);
        return lower + (upper - lower) * u;
    }
    // distance between consecutive samples of the first ray in [0,1] coordinates (for locality heuristics only)
    __device__ __forceinline__ float sample_spacing() const {
        const float dn = sqrtf(rays[3] * rays[3] + rays[4] * rays[4] + rays[5] * rays[5]);
        return (rays[7] - rays[6]) * dn / ((float)S * 2.0f * bound);
    }
    __device__ __forceinline__ void get(uint32_t b, float (&out)[3]) const {
        uint32_t r, s;
        split(b, r, s);
        const float4 *ray = reinterpret_cast<const float4 *>(rays + (size_t)r * 8);
        const float4 a = ray[0], c = ray[1];                       // o.xyz d.x | d.yz near far
        position(a, c, depth(r, s, c.z, c.w), out);
    }
    // the point at depth z on the ray record (a, c): o + d z, clamped, normalised to [0, 1]
    __device__ __forceinline__ void position(const float4 &a, const float4 &c, float z, float (&out)[3]) const {
        const float lim = bound - 1e-6f, denom = 2.0f * bound;
        const float o[3] = {a.x, a.y, a.z}, dir[3] = {a.w, c.x, c.y};
#pragma unroll
        for (uint32_t d = 0; d < 3; ++d) {
            float p = o[d] + dir[d] * z;
            p = p < -lim ? -lim : p;                   // torch.clamp (render.py:104-105): a NaN position stays NaN and poisons its
            p = p > lim ? lim : p;                     // ray like the reference; fminf / fmaxf would turn it into the clamp value
            out[d] = div_exact(p + bound, denom, rden);
        }
    }
};

// ---- forward ------------------------------------------------------------------------------------------
template <typename T, uint32_t D, uint32_t C, typename Src>
__global__ void __launch_bounds__(256)
hash_forward_kernel(Src src, const typename T::store_t *__restrict__ table, const int32_t *__restrict__ offsets,
                    typename T::store_t *__restrict__ outputs, uint32_t B, uint32_t L, uint32_t H, bool blc_layout,
                    typename T::store_t *__restrict__ dy_dx, int jac_mode, uint32_t level_base = 0) {
    using S = typename T::store_t;
    const uint32_t level = level_base + blockIdx.y;
    const LevelMeta m = make_level_meta<D>(offsets, level, H);
    const S *__restrict__ grid = table + (size_t)m.offset * C;

    dispatch_mode<Src::kInRange>(m.mode, [&](auto mode_tag) {
    constexpr uint32_t MODE = decltype(mode_tag)::value;
    for (uint32_t b = blockIdx.x * blockDim.x + threadIdx.x; b < B; b += gridDim.x * blockDim.x) {
        float x[D];
        src.get(b, x);
        float frac[D];
        uint32_t pg[D];
        locate<D>(x, m.scale, frac, pg);

        float w[1u << D];
        float v[1u << D][C];
#pragma unroll
        for (uint32_t c = 0; c < (1u << D); ++c) {
            uint32_t pl[D];
            w[c] = corner<D>(c, frac, pg, pl);
            load_vec<T, C>(grid + (size_t)grid_row<MODE, D>(m, pl) * C, v[c]);
        }
        float acc[C];
#pragma unroll
        for (uint32_t ch = 0; ch < C; ++ch) acc[ch] = 0.0f;
#pragma unroll
        for (uint32_t c = 0; c < (1u << D); ++c)
#pragma unroll
            for (uint32_t ch = 0; ch < C; ++ch) acc[ch] = __fmaf_rn(w[c], v[c][ch], acc[ch]);

        S *out = blc_layout ? outputs + ((size_t)b * L + level) * C : outputs + ((size_t)level * B + b) * C;
        store_vec<T, C>(out, acc);

        if (dy_dx != nullptr) {   // hashencoder.cu:153-197, [B,L,D,C]
            // jac_mode NAF_GRAD_INPUTS_EXACT: d out / d x[gd] = scale * sum over the corners of the other dimensions of
            //   w * (right - left);  NAF_GRAD_INPUTS_REFERENCE: what the reference stores -- no scale (:164-165) and the
            //   other dimensions picked with `nd > gd` (:170), which leaves one coordinate unset for gd < D-1 (base corner here).
            const bool exact = jac_mode != 2;
            S *dst = dy_dx + ((size_t)b * L + level) * D * C;
#pragma unroll
            for (uint32_t gd = 0; gd < D; ++gd) {
                float g[C];
#pragma unroll
                for (uint32_t ch = 0; ch < C; ++ch) g[ch] = 0.0f;
#pragma unroll
                for (uint32_t c = 0; c < (1u << (D - 1)); ++c) {
                    float wc = 1.0f;
                    uint32_t pl[D];
#pragma unroll
                    for (uint32_t d = 0; d < D; ++d) pl[d] = pg[d];
#pragma unroll
                    for (uint32_t nd = 0; nd < D - 1; ++nd) {
                        const uint32_t d = exact ? (nd >= gd ? nd + 1 : nd) : (nd > gd ? nd + 1 : nd);
                        // (a select over compile-time candidates: d is one of nd, nd + 1)
                        const float f = d == nd ? frac[nd] : frac[nd + 1];
                        const bool up = (c >> nd) & 1u;
                        wc *= up ? f : 1.0f - f;
                        if (d == nd) pl[nd] = pg[nd] + (up ? 1u : 0u);
                        else pl[nd + 1] = pg[nd + 1] + (up ? 1u : 0u);
                    }
                    float lo[C], hi[C];
                    pl[gd] = pg[gd];
                    load_vec<T, C>(grid + (size_t)grid_row<MODE, D>(m, pl) * C, lo);
                    pl[gd] = pg[gd] + 1u;
                    load_vec<T, C>(grid + (size_t)grid_row<MODE, D>(m, pl) * C, hi);
#pragma unroll
                    for (uint32_t ch = 0; ch < C; ++ch) g[ch] = __fmaf_rn(wc, hi[ch] - lo[ch], g[ch]);
                }
                if (exact) {
#pragma unroll
                    for (uint32_t ch = 0; ch < C; ++ch) g[ch] *= m.scale;
                }
                store_vec<T, C>(dst + gd * C, g);
            }
        }
    }
    });
}

// ---- backward -----------------------------------------------------------------------------------------
// One lane = one (point, level); all C channels of a corner are added by the same lane with back-to-back
// atomics so the C floats of a row land in one 4*C-byte segment.  fp32 accumulation whatever the table type.
template <typename T, uint32_t D, uint32_t C, typename Src>
__global__ void __launch_bounds__(256)
hash_backward_kernel(Src src, const typename T::store_t *__restrict__ grad, const int32_t *__restrict__ offsets,
                     float *__restrict__ grad_table, uint32_t B, uint32_t L, uint32_t H, bool blc_layout,
                     uint32_t level_base = 0) {
    const uint32_t level = level_base + blockIdx.y;
    const LevelMeta m = make_level_meta<D>(offsets, level, H);
    float *__restrict__ gg = grad_table + (size_t)m.offset * C;

    dispatch_mode<Src::kInRange>(m.mode, [&](auto mode_tag) {
    constexpr uint32_t MODE = decltype(mode_tag)::value;
    for (uint32_t b = blockIdx.x * blockDim.x + threadIdx.x; b < B; b += gridDim.x * blockDim.x) {
        float x[D];
        src.get(b, x);
        float frac[D];
        uint32_t pg[D];
        locate<D>(x, m.scale, frac, pg);
        float g[C];
        load_vec<T, C>(blc_layout ? grad + ((size_t)b * L + level) * C : grad + ((size_t)level * B + b) * C, g);
#pragma unroll
        for (uint32_t c = 0; c < (1u << D); ++c) {
            uint32_t pl[D];
            const float w = corner<D>(c, frac, pg, pl);
            float *dst = gg + (size_t)grid_row<MODE, D>(m, pl) * C;
#pragma unroll
            for (uint32_t ch = 0; ch < C; ++ch) atomicAdd(dst + ch, w * g[ch]);
        }
    }
    });
}

// naf_render_cfg.flags & NAF_CFG_PER_LEVEL_LAUNCHES splits the level-major launches into one launch per level so that
// naf_profile_collect() reports a per-level time (diagnostics only).
static inline const char *level_name(const char *const (&names)[32], uint32_t l) { return names[l < 32 ? l : 31]; }
#define NAF_LEVEL_NAMES(P) { P "00", P "01", P "02", P "03", P "04", P "05", P "06", P "07", P "08", P "09", P "10", P "11", \
                             P "12", P "13", P "14", P "15", P "16", P "17", P "18", P "19", P "20", P "21", P "22", P "23", \
                             P "24", P "25", P "26", P "27", P "28", P "29", P "30", P "31" }

static inline uint32_t hash_grid_x(uint32_t B) {
    return (uint32_t)std::min<uint64_t>(((uint64_t)B + 255) / 256, 1u << 20);
}

}  // namespace naf
