// field_mlp.h -- the NAF sigma-MLP on CDNA4 matrix cores, forward and backward, one wave = 32 points per tile.
//
// Network (reference src/network/network.py:6-58 with config/*.yaml:7-12): in 32 -> 32 -> 32 -> [cat(in,h)=64] -> 32 -> 1,
// LeakyReLU(0.01) between layers, sigmoid / leaky-relu / tanh / identity at the end.
//
// Everything is computed TRANSPOSED: a layer is  Y[out, pt] = W[out, in] . X[in, pt] + b,  so the 32 points of a
// tile are the MFMA N dimension (one point per lane & 31) and features are the M / K dimensions.  The 32x32 fp32
// accumulator of v_mfma_f32_32x32x* then has "point on the lane, feature rows in the 16 registers", which is exactly
// the B-operand layout of the next layer's MFMA: the whole chain runs in registers with no LDS traffic and no
// cross-lane movement.  Lane l = (n = l & 31, h = l >> 5) owns, for point n, the 16 feature rows
//     row(t, h) = (t & 3) + 8 * (t >> 2) + 4 * h,   t = 0..15
// ("slot t").  All operands are expressed as "16 fp32 values in slot order":
//     B operand  (activations): value t = X[row(t,h)][n]
//     A operand  (weights):     value t = W[m = l & 31][row(t,h)]        (or W^T for the backward chain)
// PrecBF16 packs slots 0-7 / 8-15 into the two K=16 steps of v_mfma_f32_32x32x16_bf16 (fp32 accumulate);
// PrecF32 feeds slot t to step t of v_mfma_f32_32x32x2_f32, which is bit-for-bit a k-ordered fmaf chain.
//
// Only the weight gradients dW = G . X^T contract over POINTS, i.e. need "feature on the lane, points in the
// registers": G and X tiles take one trip through a per-wave LDS image (written [feature][point], read back
// 8 points at a time) and feed the same MFMA.  Weight fragments live in LDS in fragment-ready order
// (one conflict-free ds_read per lane per K step), built once per workgroup.
#pragma once

#include "naf_device.h"

namespace naf {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// parameter block layout (include/naf_hip.h, NAF_MLP_PARAMS)
constexpr uint32_t kW0 = 0, kB0 = 1024, kW1 = 1056, kB1 = 2080, kW2 = 2112, kB2 = 4160, kW3 = 4192, kB3 = 4224;
constexpr uint32_t kMlpParams = 4225;
constexpr float kLeaky = 0.01f;

__device__ __forceinline__ uint32_t slot_row(uint32_t t, uint32_t h) { return (t & 3u) + 8u * (t >> 2) + 4u * h; }

// Weight fragments kept in LDS: index of the 8 A-operand matrices
enum WFrag : uint32_t { kFW0 = 0, kFW1, kFW2a, kFW2b, kFW0T, kFW1T, kFW2aT, kFW2bT, kNumWFrag };

// value t of lane (m,h) for fragment f, read from the fp32 parameter block
__device__ __forceinline__ float wfrag_value(const float *__restrict__ mlp, uint32_t f, uint32_t m, uint32_t h, uint32_t t) {
    const uint32_t r = slot_row(t, h);
    switch (f) {
        case kFW0:   return mlp[kW0 + m * 32u + r];
        case kFW1:   return mlp[kW1 + m * 32u + r];
        case kFW2a:  return mlp[kW2 + m * 64u + r];
        case kFW2b:  return mlp[kW2 + m * 64u + 32u + r];
        case kFW0T:  return mlp[kW0 + r * 32u + m];
        case kFW1T:  return mlp[kW1 + r * 32u + m];
        case kFW2aT: return mlp[kW2 + r * 64u + m];
        default:     return mlp[kW2 + r * 64u + 32u + m];
    }
}

// ------------------------------------------------------------------------------------------------------
struct PrecBF16 {
    using feat_t = BF16;                      // storage of feature / feature-gradient tensors
    using tr_t = __bf16;                      // element of the LDS transpose image
    static constexpr uint32_t kTrPitch = 32;  // elements per image row: [point][32 features], swizzled 8-byte chunks
    static constexpr uint32_t kWFragBytes = 2 * 64 * 16;   // per fragment: 2 K-steps x 64 lanes x 8 bf16
    struct Frag { bf16x8 v[2]; };

    static __device__ __forceinline__ Frag pack(const float (&x)[16]) {
        Frag f;
#pragma unroll
        for (int j = 0; j < 8; ++j) { f.v[0][j] = (__bf16)x[j]; f.v[1][j] = (__bf16)x[8 + j]; }
        return f;
    }
    static __device__ __forceinline__ f32x16 mma(const Frag &a, const Frag &b, f32x16 acc) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v[0], b.v[0], acc, 0, 0, 0);
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v[1], b.v[1], acc, 0, 0, 0);
    }
    // build fragment f in LDS (called by all lanes of one wave)
    static __device__ __forceinline__ void build_wfrag(unsigned char *lds, const float *__restrict__ mlp, uint32_t f, uint32_t lane) {
        float x[16];
#pragma unroll
        for (uint32_t t = 0; t < 16; ++t) x[t] = wfrag_value(mlp, f, lane & 31u, lane >> 5, t);
        const Frag fr = pack(x);
        bf16x8 *dst = reinterpret_cast<bf16x8 *>(lds + f * kWFragBytes);
        dst[lane] = fr.v[0];
        dst[64 + lane] = fr.v[1];
    }
    static __device__ __forceinline__ Frag load_wfrag(const unsigned char *lds, uint32_t f, uint32_t lane) {
        const bf16x8 *src = reinterpret_cast<const bf16x8 *>(lds + f * kWFragBytes);
        Frag fr;
        fr.v[0] = src[lane];
        fr.v[1] = src[64 + lane];
        return fr;
    }
    // Transpose image [point][feature], 64-byte rows, read back with ds_read_b64_tr_b16 (cdna_hip_programming.md T10).
    // A lane writes its four packed 4-slot groups -- slots 4g..4g+3 are features 8g+4h..8g+4h+3 of its point -- as four
    // 8-byte chunks; chunk c of row n sits at position c ^ ((n >> 1) & 7): the 64 writes of one group then spread over
    // all 32 write banks (2-way, the minimum for 512 B), and a transposed read takes whole rows, so it stays conflict-free.
    // `packed` is pack(v): the forward / backward chain has already converted the tile, nothing is converted again.
    static __device__ __forceinline__ void tr_put(tr_t *img, uint32_t n, uint32_t h, const float (&)[16], const Frag &packed) {
        unsigned char *row = reinterpret_cast<unsigned char *>(img) + n * 64u;
        const uint32_t sw = (n >> 1) & 7u;
        const uint4 lo = __builtin_bit_cast(uint4, packed.v[0]), hi = __builtin_bit_cast(uint4, packed.v[1]);
        *reinterpret_cast<uint2 *>(row + 8u * ((0u + h) ^ sw)) = make_uint2(lo.x, lo.y);
        *reinterpret_cast<uint2 *>(row + 8u * ((2u + h) ^ sw)) = make_uint2(lo.z, lo.w);
        *reinterpret_cast<uint2 *>(row + 8u * ((4u + h) ^ sw)) = make_uint2(hi.x, hi.y);
        *reinterpret_cast<uint2 *>(row + 8u * ((6u + h) ^ sw)) = make_uint2(hi.z, hi.w);
    }
    // Lane (m,h) gets feature m of points 8h..8h+7 (K step 0) and 16+8h..16+8h+7 (K step 1): an operand with K = points.
    // Per 16-lane group the instruction gathers a block of 4 rows (points) x 16 columns (features); lane 4q+p of the
    // group supplies the address of row q, columns 4p..4p+3 and lane i receives column i of the four rows.
    // Must be called by all 64 lanes (EXEC all ones).
    static __device__ __forceinline__ Frag tr_load(const tr_t *img, uint32_t m, uint32_t h) {
        typedef short v4i16 __attribute__((ext_vector_type(4)));
        typedef __attribute__((address_space(3))) v4i16 lds_v4i16;
        const uint32_t q = (m >> 2) & 3u, p = m & 3u, chunk = 4u * ((m >> 4) & 1u) + p;
        const unsigned char *base = reinterpret_cast<const unsigned char *>(img);
        Frag fr;
#pragma unroll
        for (uint32_t s = 0; s < 2; ++s) {
            v4i16 part[2];
#pragma unroll
            for (uint32_t u = 0; u < 2; ++u) {
                const uint32_t r = 16u * s + 8u * h + 4u * u + q;
                part[u] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4i16 *)(base + r * 64u + 8u * (chunk ^ ((r >> 1) & 7u))));
            }
            typedef short v8i16 __attribute__((ext_vector_type(8)));
            const v8i16 both = __builtin_shufflevector(part[0], part[1], 0, 1, 2, 3, 4, 5, 6, 7);
            fr.v[s] = __builtin_bit_cast(bf16x8, both);
        }
        return fr;
    }
    static __device__ __forceinline__ float frag_sum(const Frag &f) {
        float s = 0.0f;
#pragma unroll
        for (int j = 0; j < 8; ++j) s += (float)f.v[0][j] + (float)f.v[1][j];
        return s;
    }
};

struct PrecF32 {
    using feat_t = F32;
    using tr_t = float;
    static constexpr uint32_t kTrPitch = 36;
    static constexpr uint32_t kWFragBytes = 16 * 64 * 4;   // 16 K-steps x 64 lanes x fp32
    struct Frag { float v[16]; };

    static __device__ __forceinline__ Frag pack(const float (&x)[16]) {
        Frag f;
#pragma unroll
        for (int t = 0; t < 16; ++t) f.v[t] = x[t];
        return f;
    }
    static __device__ __forceinline__ f32x16 mma(const Frag &a, const Frag &b, f32x16 acc) {
#pragma unroll
        for (int t = 0; t < 16; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.v[t], b.v[t], acc, 0, 0, 0);
        return acc;
    }
    static __device__ __forceinline__ void build_wfrag(unsigned char *lds, const float *__restrict__ mlp, uint32_t f, uint32_t lane) {
        float *dst = reinterpret_cast<float *>(lds + f * kWFragBytes);
#pragma unroll
        for (uint32_t t = 0; t < 16; ++t) dst[t * 64u + lane] = wfrag_value(mlp, f, lane & 31u, lane >> 5, t);
    }
    static __device__ __forceinline__ Frag load_wfrag(const unsigned char *lds, uint32_t f, uint32_t lane) {
        const float *src = reinterpret_cast<const float *>(lds + f * kWFragBytes);
        Frag fr;
#pragma unroll
        for (uint32_t t = 0; t < 16; ++t) fr.v[t] = src[t * 64u + lane];
        return fr;
    }
    // transpose image: row = feature, col = point (element-wise stores; fp32 has no transposing LDS read)
    static __device__ __forceinline__ void tr_put(tr_t *img, uint32_t n, uint32_t h, const float (&v)[16], const Frag &) {
#pragma unroll
        for (uint32_t t = 0; t < 16; ++t) img[slot_row(t, h) * kTrPitch + n] = v[t];
    }
    static __device__ __forceinline__ Frag tr_load(const tr_t *img, uint32_t m, uint32_t h) {
        Frag fr;
        const f32x4 *p0 = reinterpret_cast<const f32x4 *>(img + m * kTrPitch + 8u * h);
        const f32x4 *p1 = reinterpret_cast<const f32x4 *>(img + m * kTrPitch + 16u + 8u * h);
        const f32x4 a = p0[0], b = p0[1], c = p1[0], d = p1[1];
#pragma unroll
        for (int j = 0; j < 4; ++j) { fr.v[j] = a[j]; fr.v[4 + j] = b[j]; fr.v[8 + j] = c[j]; fr.v[12 + j] = d[j]; }
        return fr;
    }
    static __device__ __forceinline__ float frag_sum(const Frag &f) {
        float s = 0.0f;
#pragma unroll
        for (int t = 0; t < 16; ++t) s += f.v[t];
        return s;
    }
};

// ------------------------------------------------------------------------------------------------------
// Small per-workgroup LDS block shared by all waves: weight fragments + biases + w3.
template <typename P>
struct MlpShared {
    static constexpr uint32_t kFragBytes = kNumWFrag * P::kWFragBytes;
    static constexpr uint32_t kBiasOff = kFragBytes;               // b0 b1 b2 w3 : 4 x 32 floats, then b3
    static constexpr uint32_t kBytes = kFragBytes + (4 * 32 + 4) * 4;

    // n_frags = 4 for forward-only kernels, 8 with the transposed set
    static __device__ __forceinline__ void build(unsigned char *lds, const float *__restrict__ mlp, uint32_t n_frags) {
        const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
        for (uint32_t f = wave; f < n_frags; f += n_waves) P::build_wfrag(lds, mlp, f, lane);
        float *bias = reinterpret_cast<float *>(lds + kBiasOff);
        for (uint32_t i = threadIdx.x; i < 129u; i += blockDim.x) {
            const uint32_t k = i >> 5, j = i & 31u;
            const uint32_t src = k == 0 ? kB0 + j : k == 1 ? kB1 + j : k == 2 ? kB2 + j : k == 3 ? kW3 + j : kB3;
            bias[i] = mlp[src];
        }
        __syncthreads();
    }
    // 16 values in slot order of vector k (0:b0 1:b1 2:b2 3:w3) for lane-half h
    static __device__ __forceinline__ void slot_vector(const unsigned char *lds, uint32_t k, uint32_t h, float (&out)[16]) {
        const float *b = reinterpret_cast<const float *>(lds + kBiasOff) + k * 32u;
#pragma unroll
        for (uint32_t q = 0; q < 4; ++q) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(b + 8u * q + 4u * h);
#pragma unroll
            for (uint32_t i = 0; i < 4; ++i) out[4 * q + i] = v[i];
        }
    }
    static __device__ __forceinline__ float b3(const unsigned char *lds) { return reinterpret_cast<const float *>(lds + kBiasOff)[128]; }
};

__device__ __forceinline__ f32x16 splat_slots(const float (&x)[16]) {
    f32x16 v;
#pragma unroll
    for (int t = 0; t < 16; ++t) v[t] = x[t];
    return v;
}
__device__ __forceinline__ void leaky_inplace(f32x16 &z, float (&out)[16]) {
    // slope < 1: max(z, 0.01 z) == LeakyReLU(z): v_mul + (canonicalise +) v_max.  Do NOT replace the fmaxf by an inline-asm
    // v_max_f32 to drop the canonicalise: the hazard recogniser does not look into inline asm, and in some instantiations
    // the read came too soon after the MFMA / v_accvgpr_read that produced z -- values off by ~5e-3, caught by the
    // fp32 parity tests of the C = 4 / C = 8 shapes.
#pragma unroll
    for (int t = 0; t < 16; ++t) out[t] = fmaxf(z[t], kLeaky * z[t]);
}

// final activation and its derivative expressed through the OUTPUT y (network.py:23-32)
__device__ __forceinline__ float last_act(int kind, float z) {
    switch (kind) {
        case 0: return 1.0f / (1.0f + expf(-z));
        case 1: return z > 0.0f ? z : kLeaky * z;
        case 2: return tanhf(z);
        default: return z;
    }
}
__device__ __forceinline__ float last_act_grad(int kind, float z, float y) {
    switch (kind) {
        case 0: return y * (1.0f - y);
        case 1: return z > 0.0f ? 1.0f : kLeaky;
        case 2: return 1.0f - y * y;
        default: return 1.0f;
    }
}

// Forward of one 32-point tile.  x0: the lane's 16 input features (slot order).  Returns z4 (pre-activation of the
// output unit) for point n = lane & 31 (both lane halves hold the same value).  h1/h2/h3 are the hidden activations
// in slot order (needed by the backward pass; the forward-only caller ignores them and they are optimised away).
template <typename P>
__device__ __forceinline__ float mlp_tile_forward(const unsigned char *shared, uint32_t lane, const float (&x0)[16],
                                                  typename P::Frag &x0f, float (&h1)[16], float (&h2)[16], float (&h3)[16]) {
    using Sh = MlpShared<P>;
    const uint32_t h = lane >> 5;
    float b[16];
    x0f = P::pack(x0);
    Sh::slot_vector(shared, 0, h, b);
    f32x16 z = P::mma(P::load_wfrag(shared, kFW0, lane), x0f, splat_slots(b));
    leaky_inplace(z, h1);
    Sh::slot_vector(shared, 1, h, b);
    z = P::mma(P::load_wfrag(shared, kFW1, lane), P::pack(h1), splat_slots(b));
    leaky_inplace(z, h2);
    Sh::slot_vector(shared, 2, h, b);
    z = P::mma(P::load_wfrag(shared, kFW2a, lane), x0f, splat_slots(b));          // skip connection: cat([input, h2])
    z = P::mma(P::load_wfrag(shared, kFW2b, lane), P::pack(h2), z);
    leaky_inplace(z, h3);
    Sh::slot_vector(shared, 3, h, b);                                              // w3 in slot order
    float part = 0.0f;
#pragma unroll
    for (int t = 0; t < 16; ++t) part = __fmaf_rn(b[t], h3[t], part);
    return part + __shfl_xor(part, 32, 64) + Sh::b3(shared);
}

}  // namespace naf
