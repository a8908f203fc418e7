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
    // The same fragments through a staged copy of the three weight matrices.  build()'s loads are gathers straight from global memory
    // -- 16 or 32 per lane, a different cache line per lane of a row group, the transposed fragments strided by a whole row -- and
    // tools/mlp_stamps.py put them at 13 500 of the backward kernel's 52 900 cycles at the reference's batch (three workgroups per
    // CU queue on the same address path).  Here the 4 096 weights arrive as 16 coalesced dwords per thread in ONE round trip, land in
    // a padded LDS image (row pitch 33 / 65 floats: the rows of a fragment's 16 lanes fall into different banks) and the gathers run
    // on the LDS.  Same values, same conversion: the fragments are bit-identical.  `stage`: kStageFloats floats of LDS that nothing
    // else uses until the barrier at the end (the kernels lend the depth buffers / transpose images, which they fill later).
    static constexpr uint32_t kStW0 = 0u, kStW1 = 32u * 33u, kStW2 = 2u * 32u * 33u, kStageFloats = 2u * 32u * 33u + 32u * 65u;
    template <uint32_t kFrags>
    static __device__ __forceinline__ void build_staged(unsigned char *lds, const float *__restrict__ mlp, float *stage) {
        const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;          // blockDim.x == 256 (four waves)
        const uint32_t r = lane & 15u, g = lane >> 4;
        float w[16];
#pragma unroll
        for (uint32_t u = 0; u < 16; ++u) {                                        // u < 4: W0, u < 8: W1, else W2 (256 threads x 4 = a matrix of 1 024)
            const uint32_t i = threadIdx.x + 256u * (u & 3u) + (u >= 8u ? 1024u * ((u - 8u) >> 2) : 0u);
            w[u] = mlp[(u < 4u ? kW0 : u < 8u ? kW1 : kW2) + i];
        }
        float *bias = reinterpret_cast<float *>(lds + kBiasOff);
        float bv = 0.0f;
        if (threadIdx.x < 129u) {
            const uint32_t k = threadIdx.x >> 5, j = threadIdx.x & 31u;
            bv = mlp[k == 0 ? kB0 + j : k == 1 ? kB1 + j : k == 2 ? kB2 + j : k == 3 ? kW3 + j : kB3];
        }
#pragma unroll
        for (uint32_t u = 0; u < 16; ++u) {
            const uint32_t i = threadIdx.x + 256u * (u & 3u) + (u >= 8u ? 1024u * ((u - 8u) >> 2) : 0u);
            if (u < 8u) stage[(u < 4u ? kStW0 : kStW1) + (i >> 5) * 33u + (i & 31u)] = w[u];
            else stage[kStW2 + (i >> 6) * 65u + (i & 63u)] = w[u];
        }
        if (threadIdx.x < 129u) bias[threadIdx.x] = bv;
        __syncthreads();
        constexpr uint32_t kIter = 2u * kFrags / 4u;
#pragma unroll
        for (uint32_t it = 0; it < kIter; ++it) {
            const uint32_t fo = wave + 4u * it, f = fo >> 1, o = fo & 1u;
            const uint32_t layer = f & 3u;                                         // 0: W0, 1: W1, 2: W2[:, :32], 3: W2[:, 32:]
            const uint32_t base = layer == 0u ? kStW0 : layer == 1u ? kStW1 : layer == 2u ? kStW2 : kStW2 + 32u;
            const uint32_t pitch = layer >= 2u ? 65u : 33u;
            const bool transposed = f >= kFW0T;
            const uint32_t sm = transposed ? 1u : pitch, sk = transposed ? pitch : 1u;
            const uint32_t m = 16u * o + r;
            bf16x8 pk;
#pragma unroll
            for (uint32_t j = 0; j < 8; ++j) pk[j] = (__bf16)stage[base + m * sm + feat16(g, j) * sk];
            reinterpret_cast<bf16x8 *>(lds + f * kFragBytes + o * kHalfBytes)[lane] = pk;
        }
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
// max(z, 0.01 z) == LeakyReLU(z).  fmaxf() on an MFMA result costs a third instruction per value (the compiler quiets a possible
// signalling NaN with v_max z, z first): 24 of the 119 vector instructions of a forward tile.  The v_max is therefore spelled out.
// Its second operand is the product the compiler schedules itself -- behind the wait states the MFMA result needs -- so the
// hand-written instruction can never read the accumulator early (the hazard recogniser does not look into inline asm; see
// leaky_inplace in field_mlp.h, where the operand did not depend on such a product).
__device__ __forceinline__ float max_plain(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ f32x4v leaky4(const f32x4v &z) {
    f32x4v h;
#pragma unroll
    for (int j = 0; j < 4; ++j) h[j] = max_plain(z[j], kLeaky * z[j]);
    return h;
}

// Hidden activations of one tile (fp32, the lane's eight features as two accumulator halves) and their operand forms.
struct Act16 {
    f32x4v h1lo, h1hi, h2lo, h2hi, h3lo, h3hi;
    f32x4v w3lo, w3hi;                 // output-layer weights at the lane's features (read per tile, not kept across tiles)
    bf16x8 h1f, h2f;
};

// Sum over the four 16-lane groups of a wave, result in every lane: v_permlane16_swap / v_permlane32_swap exchange rows (half waves)
// between two registers in one vector instruction each -- __shfl_xor goes through ds_bpermute (four address instructions and an
// LDS round trip per step, twice in the dependent chain of every tile).  Same operands in the same order as the xor-shuffle form:
// the result has the same bits.
__device__ __forceinline__ float sum_lane_groups(float v) {
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// Final activation of the bf16-mode kernels: last_act, the function the fp32 parity mode uses -- the bf16 mode differs from it in the
// rounding of MFMA operands and in nothing else.  NAF_MLP16_FAST_SIGMOID (A/B builds, tools/build_variant.sh) takes the hardware's exp2
// and reciprocal instead of expf's range reduction and an IEEE division (6 instead of 29 vector instructions per tile: forward
// 0.249 -> 0.242 ms, backward 0.695 -> 0.677 ms at 65 536 rays, nothing measurable at 1 024).  Not the default: the change is two ulp
// of sigma, and on the reference's 75 000-step schedule that is enough to send ONE of four seeds down another trajectory (final volume
// PSNR of seed 0: 39.78 -> 37.77 dB, seeds 1-3 within +-0.25 dB: profiles/round4_seed_study_mlp_numerics.jsonl) -- a 2 dB lottery ticket
// is a poor trade for 0.4 % of a large-batch step.  The backward recomputes sigma with this same function.
__device__ __forceinline__ float last_act16(int kind, float z) {
#ifdef NAF_MLP16_FAST_SIGMOID
    if (kind == 0) return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(z * -1.44269504088896341f));
#endif
    return last_act(kind, z);
}

// Where a tile's forward takes its weights from.  FromLds re-reads every fragment and bias vector per tile (the backward kernel:
// hoisted they would pin ~90 registers there); InRegs holds the lane's share of all four layers -- 12 fragments, 3 bias vectors, w3,
// b3: 81 registers -- for the forward kernel, whose tile loop then touches the LDS for nothing but the depths.  (The compiler used to
// hoist FromLds' reads out of that loop by itself; it stops doing so as soon as the loop body contains an asm statement, which it
// cannot prove to return, so the choice is made explicit.)
struct Mlp16FromLds {
    const unsigned char *shared;
    uint32_t lane;
    __device__ __forceinline__ bf16x8 frag(uint32_t f, uint32_t o) const { return Mlp16Shared::frag(shared, f, o, lane); }
    __device__ __forceinline__ void vec8(uint32_t k, f32x4v &lo, f32x4v &hi) const { Mlp16Shared::vec8(shared, k, lane >> 4, lo, hi); }
    __device__ __forceinline__ float b3() const { return Mlp16Shared::b3(shared); }
};
struct Mlp16InRegs {
    bf16x8 w[4][2];                    // kFW0, kFW1, kFW2a, kFW2b x output half
    f32x4v vlo[4], vhi[4];             // b0 b1 b2 w3 at the lane's features
    float bias3;
    __device__ __forceinline__ void load(const unsigned char *shared, uint32_t lane) {
        static_assert(kFW0 == 0 && kFW1 == 1 && kFW2a == 2 && kFW2b == 3, "forward fragments come first");
#pragma unroll
        for (uint32_t f = 0; f < 4; ++f)
#pragma unroll
            for (uint32_t o = 0; o < 2; ++o) w[f][o] = Mlp16Shared::frag(shared, f, o, lane);
#pragma unroll
        for (uint32_t k = 0; k < 4; ++k) Mlp16Shared::vec8(shared, k, lane >> 4, vlo[k], vhi[k]);
        bias3 = Mlp16Shared::b3(shared);
    }
    __device__ __forceinline__ bf16x8 frag(uint32_t f, uint32_t o) const { return w[f][o]; }
    __device__ __forceinline__ void vec8(uint32_t k, f32x4v &lo, f32x4v &hi) const { lo = vlo[k]; hi = vhi[k]; }
    __device__ __forceinline__ float b3() const { return bias3; }
};

// Forward of one 16-point tile.  x0f: the lane's layer-0 operand (its eight bf16 features).  Returns z4, the pre-activation
// of the output unit for point c = lane & 15 (all four lane groups hold the same value).
template <typename W>
__device__ __forceinline__ float mlp16_tile_forward(const W &wt, const bf16x8 &x0f, Act16 &a) {
    f32x4v blo, bhi;
    wt.vec8(0, blo, bhi);
    f32x4v zlo = mma16(wt.frag(kFW0, 0), x0f, blo);
    f32x4v zhi = mma16(wt.frag(kFW0, 1), x0f, bhi);
    a.h1lo = leaky4(zlo); a.h1hi = leaky4(zhi);
    a.h1f = pack16(a.h1lo, a.h1hi);
    wt.vec8(1, blo, bhi);
    zlo = mma16(wt.frag(kFW1, 0), a.h1f, blo);
    zhi = mma16(wt.frag(kFW1, 1), a.h1f, bhi);
    a.h2lo = leaky4(zlo); a.h2hi = leaky4(zhi);
    a.h2f = pack16(a.h2lo, a.h2hi);
    wt.vec8(2, blo, bhi);
    zlo = mma16(wt.frag(kFW2a, 0), x0f, blo);                                  // skip connection: cat([input, h2])
    zhi = mma16(wt.frag(kFW2a, 1), x0f, bhi);
    zlo = mma16(wt.frag(kFW2b, 0), a.h2f, zlo);
    zhi = mma16(wt.frag(kFW2b, 1), a.h2f, zhi);
    a.h3lo = leaky4(zlo); a.h3hi = leaky4(zhi);
    wt.vec8(3, a.w3lo, a.w3hi);                                                // w3 at the lane's features
    float part = 0.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j) part = __fmaf_rn(a.w3lo[j], a.h3lo[j], part);
#pragma unroll
    for (int j = 0; j < 4; ++j) part = __fmaf_rn(a.w3hi[j], a.h3hi[j], part);
    return sum_lane_groups(part) + wt.b3();
}
__device__ __forceinline__ float mlp16_tile_forward(const unsigned char *shared, uint32_t lane, const bf16x8 &x0f, Act16 &a) {
    return mlp16_tile_forward(Mlp16FromLds{shared, lane}, x0f, a);
}

// The lane's layer-0 operand for C = 2: four dwords of the [L, B, 2] bf16 feature tensor (levels 2g, 2g+1, 8+2g, 9+2g).
struct Feat16Raw { uint32_t w[4]; };
// The four addresses are ONE per-lane address (level 2g, point p) plus the wave-uniform distances B, 8B, 9B.  Written as four
// independent index expressions the compiler keeps four loop-invariant 64-bit bases per lane alive across the tile loop (and as
// many again for the gradient stores): 16 registers, which the backward kernel paid for with spills whose reloads wait for
// vmcnt(0) -- i.e. for the prefetch of the next tile -- in the middle of every tile.  The empty asm hides the sum from the
// reassociation that would rebuild those bases.
__device__ __forceinline__ size_t feat16_lane_index(uint32_t B, uint32_t p, uint32_t g) {
    size_t i = (size_t)(2u * g) * B + p;                                      // level 2g, point p; one dword = (channel 0, channel 1)
    asm("" : "+v"(i));
    return i;
}
__device__ __forceinline__ void load_feat16(const uint16_t *__restrict__ feat, uint32_t B, uint32_t p, uint32_t g, Feat16Raw &raw) {
    const uint32_t *f32 = reinterpret_cast<const uint32_t *>(feat);
    const size_t i = feat16_lane_index(B, p, g);
    raw.w[0] = f32[i];
    raw.w[1] = f32[i + B];
    raw.w[2] = f32[i + (size_t)B * 8u];
    raw.w[3] = f32[i + (size_t)B * 9u];
}
__device__ __forceinline__ void store_feat16(uint16_t *__restrict__ dfeat, uint32_t B, uint32_t p, uint32_t g, const uint4 &v) {
    uint32_t *d32 = reinterpret_cast<uint32_t *>(dfeat);
    const size_t i = feat16_lane_index(B, p, g);
    d32[i] = v.x;
    d32[i + B] = v.y;
    d32[i + (size_t)B * 8u] = v.z;
    d32[i + (size_t)B * 9u] = v.w;
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
// acc + the four bf16 values of an operand fragment (bias gradients).  NAF_MLP16_DOT2_BIAS_SUM (A/B builds): two v_dot2c_f32_bf16
// against (1, 1) instead of four unpacks and four adds -- 28 instructions per tile less, backward 0.703 -> 0.694 ms at 65 536 rays;
// the sums then round in another order, which is all it takes to move seed 0's trajectory (see last_act16), so it is not the default.
__device__ __forceinline__ float add_bf16x4(float acc, i16x4v v) {
    const uint2 w = __builtin_bit_cast(uint2, v);
#ifdef NAF_MLP16_DOT2_BIAS_SUM
    const uint32_t ones = 0x3f803f80u;
    asm("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(acc) : "v"(w.x), "v"(ones));
    asm("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(acc) : "v"(w.y), "v"(ones));
    return acc;
#else
    float s = 0.0f;
    s += __uint_as_float(w.x << 16); s += __uint_as_float(w.x & 0xffff0000u);
    s += __uint_as_float(w.y << 16); s += __uint_as_float(w.y & 0xffff0000u);
    return acc + s;
#endif
}
// derivative mask taken from the PACKED activation (its sign survives the bf16 rounding): the fp32 copies of h1 / h2 need
// not stay live through the backward chain.  `half` selects elements 0..3 or 4..7 of the operand.
__device__ __forceinline__ f32x4v leaky_grad4_packed(const f32x4v &d, const bf16x8 &hf, uint32_t half) {
    const uint4 w = __builtin_bit_cast(uint4, hf);
    const uint32_t w0 = half ? w.z : w.x, w1 = half ? w.w : w.y;
    f32x4v g;
    // the odd element sits in the high half: it is positive exactly when the whole word, read as a signed integer, exceeds 0xffff
    // (sign clear, high half non-zero) -- one compare instead of a mask and a compare
    g[0] = d[0] * (__uint_as_float(w0 << 16) > 0.0f ? 1.0f : kLeaky);
    g[1] = d[1] * ((int32_t)w0 > 0xffff ? 1.0f : kLeaky);
    g[2] = d[2] * (__uint_as_float(w1 << 16) > 0.0f ? 1.0f : kLeaky);
    g[3] = d[3] * ((int32_t)w1 > 0xffff ? 1.0f : kLeaky);
    return g;
}
__device__ __forceinline__ f32x4v leaky_grad4(const f32x4v &d, const f32x4v &h) {
    f32x4v g;
#pragma unroll
    for (int j = 0; j < 4; ++j) g[j] = d[j] * (h[j] > 0.0f ? 1.0f : kLeaky);
    return g;
}

}  // namespace naf
