// field_mlp16.h -- the NAF sigma-MLP on 16-point tiles (v_mfma_f32_16x16x32_bf16 / v_mfma_f32_16x16x16_bf16), bf16 mode.
//
// Same network and the same transposed formulation as field_mlp.h (Y[out, pt] = W[out, in] . X[in, pt]), but a tile is
// 16 points instead of 32: a lane then carries 8 instead of 16 values per activation matrix, the kernels need about half
// the registers and two to four waves fit a SIMD instead of one or two -- the 32-point kernels run at ~10 cycles per
// instruction because nothing hides their dependent MFMA / LDS / VALU chain.
//
// Lane l = (c = l & 15, g = l >> 4) owns, for point c of the tile, the eight features
//     feat(g, j) = 4 g + j            (j = 0..3)
//                = 16 + 4 g + (j - 4) (j = 4..7)
// which is (i) what the two 16x16 accumulators of a layer hold for that lane (C/D map: col = l & 15, row = 4 (l >> 4) + reg,
// one MFMA per half of the 32 outputs), (ii) the k order of the B operand of the next layer (lane holds k = 8 g + j), so
// activations chain from accumulator to operand with a bf16 pack and no data movement, and (iii) for C = 2 exactly four
// (level, channel pair) dwords of the [L, B, C] feature tensor: levels 2g, 2g+1, 8+2g, 9+2g -- the bf16 features ARE the
// layer-0 operand, no conversion.  Weight fragments (A operands) are stored in LDS in that k order.
//
// Weight gradients contract over the 16 points of a tile: v_mfma_f32_16x16x16_bf16 with operands read back from a
// [point][feature] LDS image by ds_read_b64_tr_b16 (lane 4q+p of a 16-lane group supplies row q, columns 4p..4p+3; lane i
// receives column i of the four rows = four points of one feature: the k = 4 (l >> 4) + j operand order of that MFMA).
#pragma once

#include "field_mlp.h"

namespace naf {

typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef short i16x4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t feat16(uint32_t g, uint32_t j) { return j < 4u ? 4u * g + j : 12u + 4u * g + j; }

// LDS block shared by the waves of a workgroup: [kNumWFrag][2 output halves][64 lanes] x 16 B, then b0 b1 b2 w3 (32 floats
// each) and b3.
struct Mlp16Shared {
    static constexpr uint32_t kHalfBytes = 64u * 16u;
    static constexpr uint32_t kFragBytes = 2u * kHalfBytes;
    static constexpr uint32_t kBiasOff = kNumWFrag * kFragBytes;
    static constexpr uint32_t kBytes = kBiasOff + (4u * 32u + 4u) * 4u;

    // Every 256-thread workgroup builds its own copy at kernel start, and at the reference's batch size (1 024 rays: a few tiles per
    // wave) that prologue is a visible share of the MLP kernels.  The element index of W(f)[m][k] is base + m * sm + k * sk for
    // every fragment -- no branch on f -- so a wave issues the loads of ALL its (fragment, half) blocks back to back and pays one
    // L2 round trip instead of one per block (kFrags = 4: two blocks per wave, 8: four).
    template <uint32_t kFrags>
    static __device__ __forceinline__ void build(unsigned char *lds, const float *__restrict__ mlp) {
        const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;          // blockDim.x == 256 (four waves)
        const uint32_t r = lane & 15u, g = lane >> 4;
        constexpr uint32_t kIter = 2u * kFrags / 4u;
        float v[kIter][8];
#pragma unroll
        for (uint32_t it = 0; it < kIter; ++it) {
            const uint32_t fo = wave + 4u * it, f = fo >> 1, o = fo & 1u;
            const uint32_t layer = f & 3u;                                         // 0: W0, 1: W1, 2: W2[:, :32], 3: W2[:, 32:]
            const uint32_t base = layer == 0u ? kW0 : layer == 1u ? kW1 : layer == 2u ? kW2 : kW2 + 32u;
            const uint32_t pitch = layer >= 2u ? 64u : 32u;
            const bool transposed = f >= kFW0T;
            const uint32_t sm = transposed ? 1u : pitch, sk = transposed ? pitch : 1u;
            const uint32_t m = 16u * o + r;
#pragma unroll
            for (uint32_t j = 0; j < 8; ++j) v[it][j] = mlp[base + m * sm + feat16(g, j) * sk];
        }
        float *bias = reinterpret_cast<float *>(lds + kBiasOff);
        float bv = 0.0f;
        if (threadIdx.x < 129u) {
            const uint32_t k = threadIdx.x >> 5, j = threadIdx.x & 31u;
            bv = mlp[k == 0 ? kB0 + j : k == 1 ? kB1 + j : k == 2 ? kB2 + j : k == 3 ? kW3 + j : kB3];
        }
#pragma unroll
        for (uint32_t it = 0; it < kIter; ++it) {
            const uint32_t fo = wave + 4u * it, f = fo >> 1, o = fo & 1u;
            bf16x8 pk;
#pragma unroll
            for (uint32_t j = 0; j < 8; ++j) pk[j] = (__bf16)v[it][j];
            reinterpret_cast<bf16x8 *>(lds + f * kFragBytes + o * kHalfBytes)[lane] = pk;
        }
        if (threadIdx.x < 129u) bias[threadIdx.x] = bv;
        __syncthreads();
    }
    static __device__ __forceinline__ bf16x8 frag(const unsigned char *lds, uint32_t f, uint32_t o, uint32_t lane) {
        return reinterpret_cast<const bf16x8 *>(lds + f * kFragBytes + o * kHalfBytes)[lane];
    }
    // the lane's eight entries feat(g, 0..7) of vector k (0:b0 1:b1 2:b2 3:w3), as two accumulator-shaped halves
    static __device__ __forceinline__ void vec8(const unsigned char *lds, uint32_t k, uint32_t g, f32x4v &lo, f32x4v &hi) {
        const float *b = reinterpret_cast<const float *>(lds + kBiasOff) + k * 32u;
        lo = *reinterpret_cast<const f32x4v *>(b + 4u * g);
        hi = *reinterpret_cast<const f32x4v *>(b + 16u + 4u * g);
    }
    static __device__ __forceinline__ float b3(const unsigned char *lds) { return reinterpret_cast<const float *>(lds + kBiasOff)[128]; }
};

__device__ __forceinline__ f32x4v mma16(bf16x8 a, bf16x8 b, f32x4v c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }

__device__ __forceinline__ bf16x8 pack16(const f32x4v &lo, const f32x4v &hi) {
    bf16x8 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) { v[j] = (__bf16)lo[j]; v[4 + j] = (__bf16)hi[j]; }
    return v;
}
__device__ __forceinline__ f32x4v leaky4(const f32x4v &z) {
    f32x4v h;
#pragma unroll
    for (int j = 0; j < 4; ++j) h[j] = fmaxf(z[j], kLeaky * z[j]);
    return h;
}

// Hidden activations of one tile (fp32, the lane's eight features as two accumulator halves) and their operand forms.
struct Act16 {
    f32x4v h1lo, h1hi, h2lo, h2hi, h3lo, h3hi;
    f32x4v w3lo, w3hi;                 // output-layer weights at the lane's features (read per tile, not kept across tiles)
    bf16x8 h1f, h2f;
};

// Forward of one 16-point tile.  x0f: the lane's layer-0 operand (its eight bf16 features).  Returns z4, the pre-activation
// of the output unit for point c = lane & 15 (all four lane groups hold the same value).
__device__ __forceinline__ float mlp16_tile_forward(const unsigned char *shared, uint32_t lane, const bf16x8 &x0f, Act16 &a) {
    const uint32_t g = lane >> 4;
    f32x4v blo, bhi;
    Mlp16Shared::vec8(shared, 0, g, blo, bhi);
    f32x4v zlo = mma16(Mlp16Shared::frag(shared, kFW0, 0, lane), x0f, blo);
    f32x4v zhi = mma16(Mlp16Shared::frag(shared, kFW0, 1, lane), x0f, bhi);
    a.h1lo = leaky4(zlo); a.h1hi = leaky4(zhi);
    a.h1f = pack16(a.h1lo, a.h1hi);
    Mlp16Shared::vec8(shared, 1, g, blo, bhi);
    zlo = mma16(Mlp16Shared::frag(shared, kFW1, 0, lane), a.h1f, blo);
    zhi = mma16(Mlp16Shared::frag(shared, kFW1, 1, lane), a.h1f, bhi);
    a.h2lo = leaky4(zlo); a.h2hi = leaky4(zhi);
    a.h2f = pack16(a.h2lo, a.h2hi);
    Mlp16Shared::vec8(shared, 2, g, blo, bhi);
    zlo = mma16(Mlp16Shared::frag(shared, kFW2a, 0, lane), x0f, blo);          // skip connection: cat([input, h2])
    zhi = mma16(Mlp16Shared::frag(shared, kFW2a, 1, lane), x0f, bhi);
    zlo = mma16(Mlp16Shared::frag(shared, kFW2b, 0, lane), a.h2f, zlo);
    zhi = mma16(Mlp16Shared::frag(shared, kFW2b, 1, lane), a.h2f, zhi);
    a.h3lo = leaky4(zlo); a.h3hi = leaky4(zhi);
    Mlp16Shared::vec8(shared, 3, g, a.w3lo, a.w3hi);                            // w3 at the lane's features
    float part = 0.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j) part = __fmaf_rn(a.w3lo[j], a.h3lo[j], part);
#pragma unroll
    for (int j = 0; j < 4; ++j) part = __fmaf_rn(a.w3hi[j], a.h3hi[j], part);
    part += __shfl_xor(part, 16, 64);
    part += __shfl_xor(part, 32, 64);
    return part + Mlp16Shared::b3(shared);
}

// The lane's layer-0 operand for C = 2: four dwords of the [L, B, 2] bf16 feature tensor (levels 2g, 2g+1, 8+2g, 9+2g).
struct Feat16Raw { uint32_t w[4]; };
__device__ __forceinline__ void load_feat16(const uint16_t *__restrict__ feat, uint32_t B, uint32_t p, uint32_t g, Feat16Raw &raw) {
    const uint32_t *f32 = reinterpret_cast<const uint32_t *>(feat);            // one dword = (channel 0, channel 1) of a level
    raw.w[0] = f32[(size_t)(2u * g) * B + p];
    raw.w[1] = f32[(size_t)(2u * g + 1u) * B + p];
    raw.w[2] = f32[(size_t)(8u + 2u * g) * B + p];
    raw.w[3] = f32[(size_t)(9u + 2u * g) * B + p];
}
__device__ __forceinline__ bf16x8 feat16_operand(const Feat16Raw &raw) {
    const uint4 v = make_uint4(raw.w[0], raw.w[1], raw.w[2], raw.w[3]);
    return __builtin_bit_cast(bf16x8, v);
}


// ---- backward helpers --------------------------------------------------------------------------------------------
// Transpose image of one 16-point tile: [point 0..15][32 features] bf16, 64-byte rows.  Lane (c, g) writes its two packed
// 4-feature groups (features 4g.. and 16+4g..) as 8-byte chunks 'g' and '4 + g' of row c; chunk k of row c sits at position
// k ^ ((c >> 1) & 7), which spreads one write instruction over all banks.
__device__ __forceinline__ void tr16_put(unsigned char *img, uint32_t c, uint32_t g, const bf16x8 &packed) {
    const uint4 v = __builtin_bit_cast(uint4, packed);
    const uint32_t sw = (c >> 1) & 7u;
    unsigned char *row = img + c * 64u;
    *reinterpret_cast<uint2 *>(row + 8u * (g ^ sw)) = make_uint2(v.x, v.y);
    *reinterpret_cast<uint2 *>(row + 8u * ((4u + g) ^ sw)) = make_uint2(v.z, v.w);
}
// Operand of v_mfma_f32_16x16x16_bf16 with k = points: lane (i = l & 15, gp = l >> 4) gets feature 16 * half + i of points
// 4 gp .. 4 gp + 3.  ds_read_b64_tr_b16: lane 4q+p of a 16-lane group supplies the address of row q, columns 4p..4p+3 and
// receives column i of the four rows.  All 64 lanes must be active.
__device__ __forceinline__ i16x4v tr16_get(const unsigned char *img, uint32_t lane, uint32_t half) {
    typedef __attribute__((address_space(3))) i16x4v lds_i16x4v;
    const uint32_t i = lane & 15u, gp = lane >> 4, q = i >> 2, p = i & 3u;
    const uint32_t row = 4u * gp + q, chunk = 4u * half + p;
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_i16x4v *)(img + row * 64u + 8u * (chunk ^ ((row >> 1) & 7u))));
}
__device__ __forceinline__ f32x4v mma16k16(i16x4v a, i16x4v b, f32x4v c) { return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0); }
__device__ __forceinline__ float sum_bf16x4(i16x4v v) {
    float s = 0.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j) s += __uint_as_float((uint32_t)(uint16_t)v[j] << 16);
    return s;
}
// derivative mask taken from the PACKED activation (its sign survives the bf16 rounding): the fp32 copies of h1 / h2 need
// not stay live through the backward chain.  `half` selects elements 0..3 or 4..7 of the operand.
__device__ __forceinline__ f32x4v leaky_grad4_packed(const f32x4v &d, const bf16x8 &hf, uint32_t half) {
    const uint4 w = __builtin_bit_cast(uint4, hf);
    const uint32_t w0 = half ? w.z : w.x, w1 = half ? w.w : w.y;
    f32x4v g;
    g[0] = d[0] * (__uint_as_float(w0 << 16) > 0.0f ? 1.0f : kLeaky);
    g[1] = d[1] * (__uint_as_float(w0 & 0xffff0000u) > 0.0f ? 1.0f : kLeaky);
    g[2] = d[2] * (__uint_as_float(w1 << 16) > 0.0f ? 1.0f : kLeaky);
    g[3] = d[3] * (__uint_as_float(w1 & 0xffff0000u) > 0.0f ? 1.0f : kLeaky);
    return g;
}
__device__ __forceinline__ f32x4v leaky_grad4(const f32x4v &d, const f32x4v &h) {
    f32x4v g;
#pragma unroll
    for (int j = 0; j < 4; ++j) g[j] = d[j] * (h[j] > 0.0f ? 1.0f : kLeaky);
    return g;
}

}  // namespace naf
