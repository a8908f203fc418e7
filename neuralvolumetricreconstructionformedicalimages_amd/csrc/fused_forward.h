// fused_forward.h -- forward-only NAF field in ONE kernel (gfx950): sample position -> hash-grid gathers -> sigma-MLP (MFMA)
// -> line integral / sigma, with the [L, B, C] features never leaving the registers.
//
// Replaces, for calls that need no gradients (eval_step of train.py:235-250: the projection render and the volume query),
// the pair encode_kernel + mlp16_forward_kernel and their 64 B/point feature round trip through HBM
// (hashencoder.cu:77-198 + network.py:34-58 + render.py:192-201).
//
// Shape: the canonical NAF field in bf16 mode -- 16 levels x 2 channels, MLP operands in bf16 (field_mlp16.h).  One wave works
// on a tile of 16 points.  Lane (c = l & 15, g = l >> 4) owns the eight layer-0 operand values of point c, which for C = 2 are
// the features of levels 2g, 2g+1, 8+2g, 9+2g: the lane gathers exactly those four levels of its point (4 x 8 corners = 32
// independent gathers in flight per lane, the same number the stand-alone encoder keeps in flight) and the packed bf16 results
// ARE its MFMA operand -- no LDS staging, no cross-lane movement between gather and matrix core.
// Because the four lane groups of a wave walk different levels, the level regime (dense / wrapped dense / hash, mask or
// modulo) is per-lane data here, not a wave-uniform branch: rows are computed with both combiners and selected, masks are
// all-ones where the level needs none, and the reference's `index % hashmap_size` (hashencoder.cu:74) stays behind a compare
// that is never taken for in-range points.  Rows, weights (product order of hashencoder.cu:122-133), the fp32 fma chain and the
// single rounding to bf16 are those of encode_kernel, so the results are bit-identical to the two-kernel path.
#pragma once

#include "field_mlp16.h"
#include "hash_kernels.h"

namespace naf {

struct LevelRec {                      // one per level, in LDS (32 bytes)
    uint32_t offset, size;             // first row / rows of the level
    uint32_t k1, k2;                   // multipliers of dimensions 1 and 2: the hash primes, or the (uint32-wrapped) dense strides
    uint32_t mask;                     // size - 1 where the modulo is a mask, all ones otherwise
    uint32_t hashed;
    float scale;
    uint32_t pad;
};

__device__ __forceinline__ void build_level_recs(LevelRec *recs, const int32_t *__restrict__ offsets, uint32_t L, uint32_t H) {
    for (uint32_t l = threadIdx.x; l < L; l += blockDim.x) {
        const LevelMeta m = make_level_meta<3>(offsets, l, H);
        const bool hashed = is_hash_mode(m.mode);
        LevelRec r;
        r.offset = m.offset;
        r.size = m.size;
        r.k1 = hashed ? kPrime1 : m.stride1;
        r.k2 = hashed ? kPrime2 : m.stride2;
        r.mask = (m.mode == kDenseMask || m.mode == kHashMask) ? m.size - 1u : 0xffffffffu;
        r.hashed = hashed ? 1u : 0u;
        r.scale = m.scale;
        r.pad = 0u;
        recs[l] = r;
    }
}

// rows and weights of the 8 corners of the cell of x on one level; issues the gathers of its four x-neighbour pairs (PairWindow)
template <typename TT>
__device__ __forceinline__ void gather_level(const LevelRec &r, const float (&x)[3], const typename TT::store_t *__restrict__ table,
                                             uint32_t table_rows, float (&w)[8], PairWindow<TT, 2> (&pw)[4]) {
    float frac[3];
    uint32_t pg[3];
    locate<3>(x, r.scale, frac, pg);
    const uint32_t t1 = pg[1] * r.k1, t2 = pg[2] * r.k2;
    const uint32_t term[3][2] = {{pg[0], pg[0] + 1u}, {t1, t1 + r.k1}, {t2, t2 + r.k2}};
    const typename TT::store_t *__restrict__ grid = table + (size_t)r.offset * 2u;
    const uint32_t safe_last = table_rows - PairWindow<TT, 2>::kWin - r.offset;
    uint32_t row[8];
#pragma unroll
    for (uint32_t c = 0; c < 8; ++c) {
        float wc = 1.0f;
#pragma unroll
        for (uint32_t d = 0; d < 3; ++d) wc *= ((c >> d) & 1u) ? frac[d] : 1.0f - frac[d];
        const uint32_t a = term[0][c & 1u], b = term[1][(c >> 1) & 1u], e = term[2][(c >> 2) & 1u];
        w[c] = wc;
        row[c] = (r.hashed ? (a ^ b ^ e) : (a + b + e)) & r.mask;
    }
    // `index % hashmap_size` (hashencoder.cu:74) where the mask has not done it: one branch per level, never taken for points
    // inside [0, 1] on the levels of the shipped configurations (their modulo is a mask or the identity)
    uint32_t top = row[0];
#pragma unroll
    for (uint32_t c = 1; c < 8; ++c) top = max(top, row[c]);
    if (top >= r.size) {
#pragma unroll
        for (uint32_t c = 0; c < 8; ++c) row[c] %= r.size;
    }
#pragma unroll
    for (uint32_t j = 0; j < 4; ++j) pw[j].issue(grid, row[2 * j], row[2 * j + 1], safe_last);
}
// trilinear interpolation of the gathered corners (fp32 fma chain in corner order) -> the level's two features as one bf16 pair
template <typename TT>
__device__ __forceinline__ uint32_t finish_level(const float (&w)[8], const PairWindow<TT, 2> (&pw)[4]) {
    float a0 = 0.0f, a1 = 0.0f;
#pragma unroll
    for (uint32_t j = 0; j < 4; ++j) {
        float va[2], vb[2];
        pw[j].finish(va, vb);
        a0 = __fmaf_rn(w[2 * j], va[0], a0);
        a1 = __fmaf_rn(w[2 * j], va[1], a1);
        a0 = __fmaf_rn(w[2 * j + 1], vb[0], a0);
        a1 = __fmaf_rn(w[2 * j + 1], vb[1], a1);
    }
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    bf16x2 pk;
    pk[0] = (__bf16)a0;
    pk[1] = (__bf16)a1;
    return __builtin_bit_cast(uint32_t, pk);
}

// the lane's layer-0 operand for the point at x: levels 2g, 2g+1, 8+2g, 9+2g (feat16 order), two levels (8 windows) at a time
template <typename TT>
__device__ __forceinline__ Feat16Raw gather_point_features(const LevelRec *recs, uint32_t g, const float (&x)[3],
                                                           const typename TT::store_t *__restrict__ table, uint32_t table_rows) {
    Feat16Raw f;
#pragma unroll
    for (uint32_t half = 0; half < 2; ++half) {
        float w[2][8];
        PairWindow<TT, 2> pw[2][4];
#pragma unroll
        for (uint32_t j = 0; j < 2; ++j) gather_level<TT>(recs[8u * half + 2u * g + j], x, table, table_rows, w[j], pw[j]);
#pragma unroll
        for (uint32_t j = 0; j < 2; ++j) f.w[2u * half + j] = finish_level<TT>(w[j], pw[j]);
        __builtin_amdgcn_sched_barrier(0);                   // keep the second half's 8 windows out of the first half's registers
    }
    return f;
}

__device__ __forceinline__ void store_point_features(uint16_t *__restrict__ feat, uint32_t B, uint32_t p, uint32_t g, const Feat16Raw &f) {
    uint32_t *f32 = reinterpret_cast<uint32_t *>(feat);
    f32[(size_t)(2u * g) * B + p] = f.w[0];
    f32[(size_t)(2u * g + 1u) * B + p] = f.w[1];
    f32[(size_t)(8u + 2u * g) * B + p] = f.w[2];
    f32[(size_t)(9u + 2u * g) * B + p] = f.w[3];
}

constexpr uint32_t kFusedLevels = 16;

}  // namespace naf
