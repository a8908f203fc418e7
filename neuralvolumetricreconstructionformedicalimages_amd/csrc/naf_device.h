// naf_device.h -- device-side helpers shared by the gfx950 kernels of libnaf_hip.so.
//
// Index / position arithmetic follows reference src/encoder/hashencoder/src/hashencoder.cu:36-74,99-111
// bit for bit (uint32 wrap-around of the dense stride included).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

namespace naf {

constexpr uint32_t kPrime1 = 19349663u;
constexpr uint32_t kPrime2 = 83492791u;

// How `index % hashmap_size` is evaluated for a level; decided once per level on the scalar unit.
enum IndexMode : uint32_t {
    kDenseNoMod = 0,  // dense strides, largest reachable index < T  -> the modulo is the identity
    kDenseMask = 1,   // dense (possibly uint32-wrapped) strides, T is a power of two -> mask
    kDenseMod = 2,    // dense, general T -> real modulo
    kHashMask = 3,    // xor-prime hash, T is a power of two -> mask
    kHashMod = 4,     // xor-prime hash, general T
    kDenseGuard = 5   // kDenseNoMod for coordinates nobody has range-checked: `idx % T` (hashencoder.cu:74) is the identity
                      // for x in [0,1] only, so the modulo is kept behind a compare (one v_cmp per corner, never taken in range)
};

__host__ __device__ constexpr bool is_hash_mode(uint32_t mode) { return mode == kHashMask || mode == kHashMod; }

struct LevelMeta {
    uint32_t offset;  // first row of the level in the table
    uint32_t size;    // T_l  rows in the level
    uint32_t stride1; // uint32-wrapped stride of dim 1  ((res+1)   mod 2^32)
    uint32_t stride2; // uint32-wrapped stride of dim 2  ((res+1)^2 mod 2^32)
    uint32_t mode;    // IndexMode
    float scale;      // 2^l * H - 1           (hashencoder.cu:99)
};

// All inputs are wave-uniform; the compiler keeps this on the scalar ALU.
template <uint32_t D>
__device__ __forceinline__ LevelMeta make_level_meta(const int32_t *__restrict__ offsets, uint32_t level, uint32_t H) {
    LevelMeta m;
    m.offset = (uint32_t)offsets[level];
    m.size = (uint32_t)offsets[level + 1] - m.offset;
    m.scale = exp2f((float)level) * (float)H - 1.0f;
    const uint32_t res = (uint32_t)ceilf(m.scale) + 1u;          // hashencoder.cu:100
    // replay the stride loop of get_grid_index (hashencoder.cu:59-64)
    uint32_t stride = 1;
    uint64_t max_index = 0;  // largest index reachable without wrap, for the no-mod proof
    bool wrapped = false;
    uint32_t strides[3] = {1u, 0u, 0u};
#pragma unroll
    for (uint32_t d = 0; d < D; ++d) {
        if (stride <= m.size) {
            strides[d] = stride;
            max_index += (uint64_t)res * stride;
            const uint64_t wide = (uint64_t)stride * (res + 1u);
            wrapped = wrapped || (wide >> 32) != 0;
            stride = (uint32_t)wide;
        } else {
            strides[d] = 0u;  // dims past the cut are not accumulated (only matters when hashed anyway)
        }
    }
    m.stride1 = strides[1];
    m.stride2 = D > 2 ? strides[2] : 0u;
    const bool hashed = stride > m.size;
    const bool pow2 = (m.size & (m.size - 1u)) == 0u;
    // Levels with 2^l * H >= 2^31 (only reachable with L > 27 at H = 16) overflow the uint32 resolution -- the float -> uint32
    // conversions saturate on this hardware as they do in the reference's CUDA build, and `res` wraps to 0 -- so the
    // no-modulo proof below does not hold for them: they always take a mask / modulo path (in bounds by construction).
    const bool exact = m.scale < 2147483648.0f;
    if (hashed) m.mode = pow2 ? kHashMask : kHashMod;
    else if (exact && !wrapped && max_index < (uint64_t)m.size) m.mode = kDenseNoMod;
    else m.mode = pow2 ? kDenseMask : kDenseMod;
    return m;
}

// hashencoder.cu:55-74 with the level regime as a compile-time constant.  Returns the row (not multiplied by C).
template <uint32_t MODE, uint32_t D>
__device__ __forceinline__ uint32_t grid_row(const LevelMeta &m, const uint32_t (&p)[D]) {
    uint32_t idx;
    if constexpr (is_hash_mode(MODE)) {
        idx = p[0];                         // prime 1
        idx ^= p[1] * kPrime1;
        if constexpr (D > 2) idx ^= p[2] * kPrime2;
    } else {
        idx = p[0] + p[1] * m.stride1;
        if constexpr (D > 2) idx += p[2] * m.stride2;
    }
    if constexpr (MODE == kDenseMask || MODE == kHashMask) idx &= (m.size - 1u);
    else if constexpr (MODE == kDenseGuard) { if (idx >= m.size) idx %= m.size; }
    else if constexpr (MODE != kDenseNoMod) idx %= m.size;
    return idx;
}

// Runs `body(std::integral_constant<uint32_t, MODE>{})` for the (wave-uniform) regime of the level: one scalar
// branch per kernel instead of one per corner, and straight-line gather code inside.
// kInRange = the point source guarantees coordinates in [0,1] (SrcRays clamps like render.py:104-105); every other source
// (caller-supplied coordinates) gets kDenseGuard instead of kDenseNoMod, i.e. the reference's in-bounds behaviour for ANY input.
template <bool kInRange, typename Body>
__device__ __forceinline__ void dispatch_mode(uint32_t mode, Body &&body) {
    switch (mode) {
        case kDenseNoMod:
            if constexpr (kInRange) body(std::integral_constant<uint32_t, kDenseNoMod>{});
            else body(std::integral_constant<uint32_t, kDenseGuard>{});
            break;
        case kDenseMask:  body(std::integral_constant<uint32_t, kDenseMask>{}); break;
        case kDenseMod:   body(std::integral_constant<uint32_t, kDenseMod>{}); break;
        case kHashMask:   body(std::integral_constant<uint32_t, kHashMask>{}); break;
        default:          body(std::integral_constant<uint32_t, kHashMod>{}); break;
    }
}

// hashencoder.cu:106-111.  nvcc contracts x*scale+0.5 into an FMA; we ask for it explicitly.
// The reference subtracts the CONVERTED cell index, pos -= (float)pos_grid, and its float -> uint32 conversion saturates
// (negative and NaN -> 0, >= 2^32 -> 2^32-1, exactly what v_cvt_u32_f32 does).  For 0 <= pos < 2^32 that is pos - floor(pos);
// outside (coordinates nobody range-checked, or levels whose resolution exceeds uint32) it is not, and the extra
// conversion per dimension keeps every kernel bit-identical to the reference there too.
template <uint32_t D>
__device__ __forceinline__ void locate(const float (&x)[D], float scale, float (&frac)[D], uint32_t (&pg)[D]) {
#pragma unroll
    for (uint32_t d = 0; d < D; ++d) {
        const float pos = __fmaf_rn(x[d], scale, 0.5f);
        pg[d] = (uint32_t)floorf(pos);
        frac[d] = pos - (float)pg[d];
    }
}

// corner weight / coordinates, d = 0..D-1 product order as in hashencoder.cu:122-133
template <uint32_t D>
__device__ __forceinline__ float corner(uint32_t c, const float (&frac)[D], const uint32_t (&pg)[D], uint32_t (&pl)[D]) {
    float w = 1.0f;
#pragma unroll
    for (uint32_t d = 0; d < D; ++d) {
        if ((c >> d) & 1u) { w *= frac[d]; pl[d] = pg[d] + 1u; }
        else               { w *= 1.0f - frac[d]; pl[d] = pg[d]; }
    }
    return w;
}

// All 2^D corners of a cell at once: weights in the reference's product order and rows from shared per-dimension terms.
// (p+1)*k == p*k + k in uint32 arithmetic, so each dimension costs one multiply (a quarter-rate instruction on CDNA)
// instead of one per corner; the rows are bit-identical to grid_row() on the corner coordinates.
template <uint32_t MODE, uint32_t D>
__device__ __forceinline__ void cell_corners(const LevelMeta &m, const float (&frac)[D], const uint32_t (&pg)[D],
                                             float (&w)[1u << D], uint32_t (&row)[1u << D]) {
    uint32_t term[D][2];
#pragma unroll
    for (uint32_t d = 0; d < D; ++d) {
        uint32_t k;
        if constexpr (is_hash_mode(MODE)) k = d == 0 ? 1u : d == 1 ? kPrime1 : kPrime2;
        else k = d == 0 ? 1u : d == 1 ? m.stride1 : m.stride2;
        term[d][0] = d == 0 ? pg[0] : pg[d] * k;
        term[d][1] = term[d][0] + k;
    }
#pragma unroll
    for (uint32_t c = 0; c < (1u << D); ++c) {
        float wc = 1.0f;
        uint32_t idx = 0u;
#pragma unroll
        for (uint32_t d = 0; d < D; ++d) {
            const uint32_t bit = (c >> d) & 1u;
            wc *= bit ? frac[d] : 1.0f - frac[d];
            if constexpr (is_hash_mode(MODE)) idx ^= term[d][bit];
            else idx += term[d][bit];
        }
        if constexpr (MODE == kDenseMask || MODE == kHashMask) idx &= (m.size - 1u);
        else if constexpr (MODE == kDenseGuard) { if (idx >= m.size) idx %= m.size; }
        else if constexpr (MODE != kDenseNoMod) idx %= m.size;
        w[c] = wc;
        row[c] = idx;
    }
}

// ---- storage types ---------------------------------------------------------------------------
struct F32 { using store_t = float; };
struct F16 { using store_t = _Float16; };
struct BF16 { using store_t = uint16_t; };

__device__ __forceinline__ float bf16_to_f32(uint16_t v) { return __uint_as_float((uint32_t)v << 16); }
__device__ __forceinline__ uint16_t f32_to_bf16(float f) {   // round to nearest even: one v_cvt_pk_bf16_f32 on gfx950
    return __builtin_bit_cast(uint16_t, (__bf16)f);
}

template <typename T> struct Conv;
template <> struct Conv<F32> {
    static __device__ __forceinline__ float load(const float *p) { return *p; }
    static __device__ __forceinline__ void store(float *p, float v) { *p = v; }
};
template <> struct Conv<F16> {
    static __device__ __forceinline__ float load(const _Float16 *p) { return (float)*p; }
    static __device__ __forceinline__ void store(_Float16 *p, float v) { *p = (_Float16)v; }
};
template <> struct Conv<BF16> {
    static __device__ __forceinline__ float load(const uint16_t *p) { return bf16_to_f32(*p); }
    static __device__ __forceinline__ void store(uint16_t *p, float v) { *p = f32_to_bf16(v); }
};

// Vector load/store of C consecutive table elements as fp32 (C*sizeof(store_t) is 2..32 bytes, naturally aligned).
template <typename T, uint32_t C>
__device__ __forceinline__ void load_vec(const typename T::store_t *p, float (&v)[C]) {
    using S = typename T::store_t;
    constexpr uint32_t bytes = C * sizeof(S);
    if constexpr (bytes == 2) {
        v[0] = Conv<T>::load(p);
    } else if constexpr (bytes == 4) {
        const uint32_t raw = *reinterpret_cast<const uint32_t *>(p);
        if constexpr (sizeof(S) == 4) v[0] = __uint_as_float(raw);
        else { S e[2]; __builtin_memcpy(e, &raw, 4); v[0] = Conv<T>::load(&e[0]); v[1] = Conv<T>::load(&e[1]); }
    } else if constexpr (bytes == 8) {
        const uint2 raw = *reinterpret_cast<const uint2 *>(p);
        S e[8 / sizeof(S)]; __builtin_memcpy(e, &raw, 8);
#pragma unroll
        for (uint32_t c = 0; c < C; ++c) v[c] = Conv<T>::load(&e[c]);
    } else if constexpr (bytes == 16) {
        const uint4 raw = *reinterpret_cast<const uint4 *>(p);
        S e[16 / sizeof(S)]; __builtin_memcpy(e, &raw, 16);
#pragma unroll
        for (uint32_t c = 0; c < C; ++c) v[c] = Conv<T>::load(&e[c]);
    } else {  // 32 bytes: two 16-byte halves
        const uint4 r0 = reinterpret_cast<const uint4 *>(p)[0];
        const uint4 r1 = reinterpret_cast<const uint4 *>(p)[1];
        S e[32 / sizeof(S)]; __builtin_memcpy(e, &r0, 16); __builtin_memcpy(e + 16 / sizeof(S), &r1, 16);
#pragma unroll
        for (uint32_t c = 0; c < C; ++c) v[c] = Conv<T>::load(&e[c]);
    }
}

template <typename T, uint32_t C>
__device__ __forceinline__ void store_vec(typename T::store_t *p, const float (&v)[C]) {
    using S = typename T::store_t;
    constexpr uint32_t bytes = C * sizeof(S);
    S e[C];
    if constexpr (sizeof(S) == 2 && C % 2 == 0 && !__is_same(S, _Float16)) {   // bf16: one v_cvt_pk_bf16_f32 per pair
        typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (uint32_t c = 0; c < C; c += 2) {
            bf16x2 pk;
            pk[0] = (__bf16)v[c];
            pk[1] = (__bf16)v[c + 1];
            __builtin_memcpy(&e[c], &pk, 4);
        }
    } else {
#pragma unroll
        for (uint32_t c = 0; c < C; ++c) Conv<T>::store(&e[c], v[c]);
    }
    if constexpr (bytes == 2) { *p = e[0]; }
    else if constexpr (bytes == 4) { uint32_t raw; __builtin_memcpy(&raw, e, 4); *reinterpret_cast<uint32_t *>(p) = raw; }
    else if constexpr (bytes == 8) { uint2 raw; __builtin_memcpy(&raw, e, 8); *reinterpret_cast<uint2 *>(p) = raw; }
    else if constexpr (bytes == 16) { uint4 raw; __builtin_memcpy(&raw, e, 16); *reinterpret_cast<uint4 *>(p) = raw; }
    else {
        uint4 r0, r1; __builtin_memcpy(&r0, e, 16); __builtin_memcpy(&r1, e + 16 / sizeof(S), 16);
        reinterpret_cast<uint4 *>(p)[0] = r0; reinterpret_cast<uint4 *>(p)[1] = r1;
    }
}

// A load whose conversion to fp32 is deferred: the raw words can cross a barrier / a block of stores before anything waits
// for them (s_waitcnt is placed at the first use, and vmcnt retires in order, so issue order matters).
template <typename T, uint32_t C>
struct RawVec {
    static constexpr uint32_t kBytes = C * sizeof(typename T::store_t);
    uint32_t w[(kBytes + 3) / 4];
};
template <typename T, uint32_t C>
__device__ __forceinline__ void raw_load(const typename T::store_t *p, RawVec<T, C> &r) {
    constexpr uint32_t bytes = RawVec<T, C>::kBytes;
    if constexpr (bytes == 2) r.w[0] = *reinterpret_cast<const uint16_t *>(p);
    else if constexpr (bytes == 4) r.w[0] = *reinterpret_cast<const uint32_t *>(p);
    else if constexpr (bytes == 8) { const uint2 v = *reinterpret_cast<const uint2 *>(p); r.w[0] = v.x; r.w[1] = v.y; }
    else {
#pragma unroll
        for (uint32_t i = 0; i < bytes / 16; ++i) {
            const uint4 v = reinterpret_cast<const uint4 *>(p)[i];
            r.w[4 * i] = v.x; r.w[4 * i + 1] = v.y; r.w[4 * i + 2] = v.z; r.w[4 * i + 3] = v.w;
        }
    }
}
template <typename T, uint32_t C>
__device__ __forceinline__ void raw_unpack(const RawVec<T, C> &r, float (&v)[C]) {
    using S = typename T::store_t;
    S e[C];
    __builtin_memcpy(e, r.w, C * sizeof(S));
#pragma unroll
    for (uint32_t c = 0; c < C; ++c) v[c] = Conv<T>::load(&e[c]);
}

// ---- x-neighbour corner pairs through one 16-byte window ------------------------------------------------------------------
// The corners (x, x+1) of a cell sit in rows r and r' with r ^ r' = x ^ (x+1) = 1, 3, 7, ... on hashed levels (the prime of
// dimension 0 is 1, hashencoder.cu:36-52) and r' = r + 1 on dense ones: 83 % of the pairs are less than four rows apart.  The
// texture path of a CU (TCP) spends one tag lookup per lane and instruction whatever the access width (counter evidence:
// profiles/round3_encode_cache_counters.md: encode_kernel sits at ~0.9 TCP accesses per clock and CU with L2 at 40 % of its
// bandwidth), so ONE 4-byte-aligned dwordx4 at min(r, r') replaces two dword gathers for those pairs; the far row of the rest
// comes from a second, exec-masked load.  Same rows, same values, same arithmetic afterwards -- only fewer L1 accesses.
typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));

template <typename T, uint32_t C>
struct PairWindow {
    static constexpr uint32_t kRowBytes = C * (uint32_t)sizeof(typename T::store_t);
    static constexpr bool kUsable = kRowBytes == 4u || kRowBytes == 8u;
    static constexpr uint32_t kDw = kUsable ? kRowBytes / 4u : 1u;       // dwords per row
    static constexpr uint32_t kWin = 4u / kDw;                            // rows per window
    u32x4_a4 win;
    uint32_t far[kDw];
    uint32_t d, lo_off;                // window index of the higher / lower row
    bool swap;                         // the first corner is the higher row

    // `safe_last`: the last row of this level at which a window still ends inside the table (>= the level's size for every
    // level but the last, so the clamp only ever moves windows at the very end of the allocation)
    __device__ __forceinline__ void issue(const typename T::store_t *__restrict__ grid, uint32_t ra, uint32_t rb, uint32_t safe_last) {
        const uint32_t lo = min(ra, rb), hi = max(ra, rb), base = min(lo, safe_last);
        swap = ra > rb;
        d = hi - base;
        lo_off = lo - base;
        const uint32_t *words = reinterpret_cast<const uint32_t *>(grid);
        win = *reinterpret_cast<const u32x4_a4 *>(words + (size_t)base * kDw);
#pragma unroll
        for (uint32_t i = 0; i < kDw; ++i) far[i] = 0u;
        if (d >= kWin) {
#pragma unroll
            for (uint32_t i = 0; i < kDw; ++i) far[i] = words[(size_t)hi * kDw + i];
        }
    }
    __device__ __forceinline__ void row_at(uint32_t i, uint32_t (&out)[kDw]) const {
        if constexpr (kDw == 1u) out[0] = i == 1u ? win.y : i == 2u ? win.z : i == 3u ? win.w : win.x;
        else { out[0] = i == 1u ? win.z : win.x; out[1] = i == 1u ? win.w : win.y; }
    }
    // values of the first (a) and second (b) corner as fp32
    __device__ __forceinline__ void finish(float (&a)[C], float (&b)[C]) const {
        RawVec<T, C> lo, hi;
#pragma unroll
        for (uint32_t i = 0; i < kDw; ++i) lo.w[i] = i == 0u ? win.x : win.y;
        if (lo_off != 0u) row_at(lo_off, lo.w);                            // only within kWin rows of the end of the table
        if (d < kWin) row_at(d, hi.w);
        else {
#pragma unroll
            for (uint32_t i = 0; i < kDw; ++i) hi.w[i] = far[i];
        }
        RawVec<T, C> ra, rb;
#pragma unroll
        for (uint32_t i = 0; i < kDw; ++i) { ra.w[i] = swap ? hi.w[i] : lo.w[i]; rb.w[i] = swap ? lo.w[i] : hi.w[i]; }
        raw_unpack<T, C>(ra, a);
        raw_unpack<T, C>(rb, b);
    }
};

// Workgroup barrier that orders LDS traffic only.  __syncthreads() carries a workgroup-scope release fence, i.e.
// s_waitcnt vmcnt(0): every wave would sit out the round trip of the global stores it has just issued.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Value of lane (l - kShift) of the same 16-lane row, 0 for the first kShift lanes of a row: a DPP row_shr modifier on the
// consuming VALU instruction, no LDS crossbar traffic.  Every lane of the wave must execute it (convergent).
template <uint32_t kShift>
__device__ __forceinline__ uint32_t dpp_row_shr(uint32_t v) {
    static_assert(kShift >= 1 && kShift <= 15, "row_shr:1..15");
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x110 + kShift, 0xf, 0xf, true);
}
template <uint32_t kShift>
__device__ __forceinline__ float dpp_row_shr(float v) { return __uint_as_float(dpp_row_shr<kShift>(__float_as_uint(v))); }

// ---- counter-based jitter (used when the caller passes no t_rand) --------------------------------
// Two rounds of the "lowbias32" integer finaliser over (seed, global ray index, sample); 24 bits -> [0,1).
// 32-bit only on purpose: it is re-evaluated by every kernel that needs the sample position.
__device__ __host__ __forceinline__ uint32_t mix32(uint32_t h) {
    h ^= h >> 16; h *= 0x7feb352du;
    h ^= h >> 15; h *= 0x846ca68bu;
    h ^= h >> 16;
    return h;
}
__device__ __host__ __forceinline__ float jitter(uint64_t seed, uint32_t ray, uint32_t sample) {
    uint32_t h = mix32((uint32_t)seed ^ (ray * 0x9e3779b1u));
    h = mix32(h ^ (uint32_t)(seed >> 32) ^ (sample * 0x85ebca77u) ^ 0x27d4eb2fu);
    return (float)(h >> 8) * (1.0f / 16777216.0f);
}

// ---- stratified depth of sample s on a ray (render.py:87-100) ------------------------------------
// torch.linspace(0,1,S) on CPU/GPU evaluates start + i*step for i < S/2 and end - (S-1-i)*step above.
__device__ __forceinline__ float lin_t(uint32_t i, uint32_t S) {
    const float step = 1.0f / (float)(S - 1u);
    // the upper half is a fused multiply-subtract in torch's CPU and CUDA builds (checked against torch.linspace)
    return (i < S / 2u) ? (float)i * step : __fmaf_rn(-(float)(S - 1u - i), step, 1.0f);
}
__device__ __forceinline__ float base_z(float near, float far, uint32_t i, uint32_t S) {
    const float t = lin_t(i, S);
    return near * (1.0f - t) + far * t;      // two roundings like the reference (no fma)
}
// jittered depth; u is the uniform for this sample (ignored when !perturb)
__device__ __forceinline__ float sample_z(float near, float far, uint32_t i, uint32_t S, bool perturb, float u) {
    const float z = base_z(near, far, i, S);
    if (!perturb) return z;
    const float lower = i == 0u ? z : 0.5f * (z + base_z(near, far, i - 1u, S));
    const float upper = i + 1u == S ? z : 0.5f * (base_z(near, far, i + 1u, S) + z);
    return lower + (upper - lower) * u;
}

}  // namespace naf
