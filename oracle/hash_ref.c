/*
 * oracle/hash_ref.c -- TEST INFRASTRUCTURE ONLY (never linked into or called by the product).
 *
 * Scalar CPU restatement of the multi-resolution hash-grid encoder the reference ships as CUDA
 * (reference: src/encoder/hashencoder/src/hashencoder.cu).  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load the library built from this file.
 *
 * Parity status: the reference has no tests or golden vectors for this path and its CUDA source
 * cannot be built in this image, so this restatement is pinned by
 *   (1) hand-derived integer known-answer tests for every index regime (tests/test_index_kat.py),
 *   (2) agreement with an independently written vectorised numpy restatement (oracle/hashgrid_ref.py),
 *   (3) analytic properties (constant table, trilinear reproduction, weight partition of unity).
 *
 * Floating point: the CUDA build contracts a*b+c into FMA (nvcc default -fmad=true), so the position
 * and the accumulation are written with fmaf() here; weights are plain fp32 products in d = 0,1,2 order.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#define NAF_MAX_D 3
#define NAF_MAX_C 8

/* hashencoder.cu:36-52  (primes for d = 0,1,2; D <= 3 here) */
static uint32_t oracle_fast_hash(uint32_t D, const uint32_t *pg) {
    static const uint32_t primes[3] = {1u, 19349663u, 83492791u};
    uint32_t h = 0;
    for (uint32_t d = 0; d < D; ++d) h ^= pg[d] * primes[d];
    return h;
}

/* hashencoder.cu:55-74.  uint32 arithmetic, stride wraps mod 2^32. */
uint32_t naf_oracle_grid_index(uint32_t D, uint32_t C, uint32_t ch, uint32_t hashmap_size,
                               uint32_t resolution, const uint32_t *pg) {
    uint32_t stride = 1, index = 0;
    for (uint32_t d = 0; d < D && stride <= hashmap_size; ++d) {
        index += pg[d] * stride;
        stride *= (resolution + 1u);
    }
    if (stride > hashmap_size) index = oracle_fast_hash(D, pg);
    return (index % hashmap_size) * C + ch;
}

/* float -> uint32 as the reference's CUDA build converts (cvt.rzi.u32.f32): saturating, NaN -> 0.  In C the cast is
 * undefined outside [0, 2^32), so the rule is spelled out. */
static uint32_t cuda_f2u(float f) {
    if (!(f > 0.0f)) return 0u;                 /* negative, zero, NaN */
    if (f >= 4294967296.0f) return 0xffffffffu;
    return (uint32_t)f;
}

/* hashencoder.cu:99-100 */
static void level_geometry(uint32_t level, uint32_t H, float *scale, uint32_t *resolution) {
    *scale = exp2f((float)level) * (float)H - 1.0f;
    *resolution = cuda_f2u(ceilf(*scale)) + 1u;
}

/* hashencoder.cu:106-111 */
static void locate(uint32_t D, const float *x, float scale, float *frac, uint32_t *pg) {
    for (uint32_t d = 0; d < D; ++d) {
        float p = fmaf(x[d], scale, 0.5f);
        float fl = floorf(p);
        pg[d] = cuda_f2u(fl);
        frac[d] = p - (float)pg[d];
    }
}

/* hashencoder.cu:77-198.  outputs is level-major [L,B,C]; dy_dx is [B,L,D,C]. */
void naf_oracle_hash_encode_forward(const float *inputs, const float *embeddings, const int32_t *offsets,
                                    float *outputs, uint32_t B, uint32_t D, uint32_t C, uint32_t L,
                                    uint32_t H, int calc_grad_inputs, float *dy_dx) {
#pragma omp parallel for schedule(static)
    for (int64_t lb = 0; lb < (int64_t)L * B; ++lb) {
        const uint32_t level = (uint32_t)(lb / B), b = (uint32_t)(lb % B);
        const float *grid = embeddings + (size_t)(uint32_t)offsets[level] * C;
        const uint32_t hashmap_size = (uint32_t)(offsets[level + 1] - offsets[level]);
        float scale; uint32_t resolution;
        level_geometry(level, H, &scale, &resolution);
        float frac[NAF_MAX_D]; uint32_t pg[NAF_MAX_D];
        locate(D, inputs + (size_t)b * D, scale, frac, pg);

        float acc[NAF_MAX_C];
        for (uint32_t c = 0; c < C; ++c) acc[c] = 0.0f;
        for (uint32_t corner = 0; corner < (1u << D); ++corner) {
            float w = 1.0f; uint32_t pl[NAF_MAX_D];
            for (uint32_t d = 0; d < D; ++d) {
                if ((corner & (1u << d)) == 0) { w *= 1.0f - frac[d]; pl[d] = pg[d]; }
                else                            { w *= frac[d];        pl[d] = pg[d] + 1u; }
            }
            const uint32_t idx = naf_oracle_grid_index(D, C, 0, hashmap_size, resolution, pl);
            for (uint32_t c = 0; c < C; ++c) acc[c] = fmaf(w, grid[idx + c], acc[c]);
        }
        float *out = outputs + ((size_t)level * B + b) * C;
        for (uint32_t c = 0; c < C; ++c) out[c] = acc[c];

        if (calc_grad_inputs) {            /* hashencoder.cu:153-197 */
            /* calc_grad_inputs == 1: the exact derivative, scale * sum_w (right - left) over the corners of the OTHER
             * dimensions.  == 2: what the reference stores: no scale (:164-165), other dimensions picked
             * with `nd > gd` (:170) -- for gd < D-1 one coordinate is then never assigned (the CUDA code reads an
             * uninitialised register); the base corner pg[d] stands in for it. */
            const int exact = calc_grad_inputs != 2;
            float *dst = dy_dx + ((size_t)b * L + level) * D * C;
            for (uint32_t gd = 0; gd < D; ++gd) {
                float g[NAF_MAX_C];
                for (uint32_t c = 0; c < C; ++c) g[c] = 0.0f;
                for (uint32_t corner = 0; corner < (1u << (D - 1)); ++corner) {
                    float w = 1.0f; uint32_t pl[NAF_MAX_D];
                    for (uint32_t d = 0; d < D; ++d) pl[d] = pg[d];
                    for (uint32_t nd = 0; nd < D - 1; ++nd) {
                        const uint32_t d = exact ? (nd >= gd ? nd + 1 : nd) : (nd > gd ? nd + 1 : nd);
                        if ((corner & (1u << nd)) == 0) { w *= 1.0f - frac[d]; pl[d] = pg[d]; }
                        else                             { w *= frac[d];        pl[d] = pg[d] + 1u; }
                    }
                    pl[gd] = pg[gd];
                    const uint32_t il = naf_oracle_grid_index(D, C, 0, hashmap_size, resolution, pl);
                    pl[gd] = pg[gd] + 1u;
                    const uint32_t ir = naf_oracle_grid_index(D, C, 0, hashmap_size, resolution, pl);
                    for (uint32_t c = 0; c < C; ++c) g[c] = fmaf(w, grid[ir + c] - grid[il + c], g[c]);
                }
                for (uint32_t c = 0; c < C; ++c) dst[gd * C + c] = exact ? g[c] * scale : g[c];
            }
        }
    }
}

/* hashencoder.cu:201-272 + 275-298.  grad is [B, L*C]; grad_embeddings (+=) is [sum T_l, C];
 * serial accumulation in (level, b, corner) order -> deterministic. */
void naf_oracle_hash_encode_backward(const float *grad, const float *inputs, const float *embeddings,
                                     const int32_t *offsets, float *grad_embeddings, uint32_t B,
                                     uint32_t D, uint32_t C, uint32_t L, uint32_t H,
                                     int calc_grad_inputs, const float *dy_dx, float *grad_inputs) {
    (void)embeddings;
#pragma omp parallel for schedule(static)      /* levels own disjoint table slices -> race free */
    for (int64_t lv = 0; lv < (int64_t)L; ++lv) {
        const uint32_t level = (uint32_t)lv;
        float *gg = grad_embeddings + (size_t)(uint32_t)offsets[level] * C;
        const uint32_t hashmap_size = (uint32_t)(offsets[level + 1] - offsets[level]);
        float scale; uint32_t resolution;
        level_geometry(level, H, &scale, &resolution);
        for (uint32_t b = 0; b < B; ++b) {
            float frac[NAF_MAX_D]; uint32_t pg[NAF_MAX_D];
            locate(D, inputs + (size_t)b * D, scale, frac, pg);
            const float *g = grad + (size_t)b * L * C + (size_t)level * C;
            for (uint32_t corner = 0; corner < (1u << D); ++corner) {
                float w = 1.0f; uint32_t pl[NAF_MAX_D];
                for (uint32_t d = 0; d < D; ++d) {
                    if ((corner & (1u << d)) == 0) { w *= 1.0f - frac[d]; pl[d] = pg[d]; }
                    else                            { w *= frac[d];        pl[d] = pg[d] + 1u; }
                }
                const uint32_t idx = naf_oracle_grid_index(D, C, 0, hashmap_size, resolution, pl);
                for (uint32_t c = 0; c < C; ++c) gg[idx + c] += w * g[c];
            }
        }
    }
    if (calc_grad_inputs) {
        for (uint32_t t = 0; t < B * D; ++t) {
            const uint32_t b = t / D, d = t - b * D;
            const float *g = grad + (size_t)b * L * C;
            const float *j = dy_dx + (size_t)b * L * D * C;
            float s = grad_inputs[t];
            for (uint32_t l = 0; l < L; ++l)
                for (uint32_t c = 0; c < C; ++c) s += g[l * C + c] * j[l * D * C + d * C + c];
            grad_inputs[t] = s;
        }
    }
}

/* Convenience for KATs: indices + weights of the 2^D corners of one point at one level. */
void naf_oracle_corners(const float *x, uint32_t D, uint32_t C, uint32_t level, uint32_t H,
                        uint32_t hashmap_size, uint32_t *idx_out, float *w_out) {
    float scale; uint32_t resolution;
    level_geometry(level, H, &scale, &resolution);
    float frac[NAF_MAX_D]; uint32_t pg[NAF_MAX_D];
    locate(D, x, scale, frac, pg);
    for (uint32_t corner = 0; corner < (1u << D); ++corner) {
        float w = 1.0f; uint32_t pl[NAF_MAX_D];
        for (uint32_t d = 0; d < D; ++d) {
            if ((corner & (1u << d)) == 0) { w *= 1.0f - frac[d]; pl[d] = pg[d]; }
            else                            { w *= frac[d];        pl[d] = pg[d] + 1u; }
        }
        idx_out[corner] = naf_oracle_grid_index(D, C, 0, hashmap_size, resolution, pl);
        w_out[corner] = w;
    }
}
