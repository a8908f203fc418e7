// draw_device.h -- device side of the data path of a training step: ray generation (tigre.py:402-456, 463-528) and the valid-pixel
// draw (tigre.py:354-372).  Shared by render_ops.hip (naf_generate_rays, naf_draw_scan_rays) and scatter_v2.h, whose pass-1 launch can
// carry the draw of the NEXT step in spare workgroups (naf_render_train_adam_draw): the draw is 6 us of latency in a launch of four
// workgroups, and nothing in a step depends on the next step's pixels.
#pragma once

#include "naf_device.h"
#include "../../include/naf_hip.h"

namespace naf {

struct RayGeo {
    uint32_t W, H;          // detector columns / rows (nDetector[0], nDetector[1])
    float du, dv;           // pixel pitch  (dDetector)
    float ou, ov;           // detector offset (offDetector)
    float DSD;
    float near, far;        // tigre.py:575-586
    int parallel;           // 0 cone, 1 parallel
};

__device__ __forceinline__ void make_ray(const float *__restrict__ poses, uint64_t flat, const RayGeo &g, float4 *out) {
    const uint64_t per_proj = (uint64_t)g.W * g.H;
    const uint32_t proj = (uint32_t)(flat / per_proj);
    const uint32_t rem = (uint32_t)(flat - (uint64_t)proj * per_proj);
    const uint32_t row = rem / g.W, col = rem - row * g.W;
    const float *P = poses + (size_t)proj * 12;                 // 3x4 row-major [R | t]
    // tigre.py:423-429: uu along columns, vv along rows
    const float uu = ((float)col + 0.5f - (float)g.W / 2.0f) * g.du + g.ou;
    const float vv = ((float)row + 0.5f - (float)g.H / 2.0f) * g.dv + g.ov;
    float o[3], d[3];
    if (!g.parallel) {                                          // tigre.py:434-437
        const float dx = uu / g.DSD, dy = vv / g.DSD;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            d[k] = P[4 * k + 0] * dx + P[4 * k + 1] * dy + P[4 * k + 2];
            o[k] = P[4 * k + 3];
        }
    } else {                                                    // tigre.py:438-447
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            d[k] = P[4 * k + 2];
            o[k] = P[4 * k + 0] * uu + P[4 * k + 1] * vv + P[4 * k + 3];
        }
    }
    out[0] = make_float4(o[0], o[1], o[2], d[0]);
    out[1] = make_float4(d[1], d[2], g.near, g.far);
}

// ---- G6: the data side of a training step on the device (reference src/dataset/tigre.py:354-372) ------------------
// `np.random.choice(valid, n_rays, replace=False)` + three fancy-indexing gathers per item become ONE launch: draw index i
// is sent through a keyed bijection of [0, n_valid) (a 4-round Feistel network on the smallest even-width bit field that
// covers n_valid, restricted to the range by cycle walking), so the first n outputs are n DISTINCT uniformly chosen
// entries of the valid-pixel list -- no sort, no host synchronisation, and ranks of a data-parallel job that share the
// seed can each take a slice of the same draw.  The thread then gathers the measured value and generates the ray.
struct ScanDraw {
    uint32_t n_segments, per_segment;
    const int64_t *valid[NAF_MAX_DRAW_SEGMENTS];
    uint32_t n_valid[NAF_MAX_DRAW_SEGMENTS];
};

__device__ __forceinline__ uint32_t feistel_permute(uint32_t i, uint32_t n, uint32_t half_bits, uint64_t seed) {
    const uint32_t mask = (1u << half_bits) - 1u;
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    uint32_t x = i;
    do {                                                     // the walk stays inside the cycle of i: still a bijection on [0, n)
        uint32_t l = x >> half_bits, r = x & mask;
#pragma unroll
        for (uint32_t round = 0; round < 4u; ++round) {
            const uint32_t f = mix32(r ^ (round & 1u ? k1 : k0) ^ (0x9e3779b9u * (round + 1u))) & mask;
            const uint32_t t = l ^ f;
            l = r;
            r = t;
        }
        x = (l << half_bits) | r;
    } while (x >= n);
    return x;
}

// One draw job = the arguments of naf_draw_scan_rays, validated on the host (make_draw_job, render_ops.hip).  count == 0: nothing to do.
struct DrawJob {
    ScanDraw draw;
    const float *poses, *projections;
    int64_t *pixels;
    float *target, *rays;
    uint32_t first, count;
    RayGeo g;
    uint64_t seed, n_pixels;
};

// draw t of the job: distinct valid pixel -> measured value -> ray
__device__ __forceinline__ void draw_one(const DrawJob &j, uint32_t t) {
    const uint32_t i = j.first + t;                             // global draw index: segment-major
    const uint32_t seg = i / j.draw.per_segment, k = i - seg * j.draw.per_segment;
    const uint32_t n = j.draw.n_valid[seg];
    uint32_t half_bits = 1u;
    while ((1ull << (2u * half_bits)) < (uint64_t)n) ++half_bits;
    const uint32_t idx = feistel_permute(k, n, half_bits, j.seed + 0x632be59bd9b4e019ull * (seg + 1u));
    const int64_t flat = j.draw.valid[seg][idx];
    if (j.pixels) j.pixels[t] = flat;
    if ((uint64_t)flat >= j.n_pixels) {                         // a list entry outside the scan (the reference's fancy index would raise):
        const float nan = __builtin_nanf("");                   // nothing is read through it, the ray and its value are NaN
        if (j.target) j.target[t] = nan;
        reinterpret_cast<float4 *>(j.rays + (size_t)t * 8)[0] = make_float4(nan, nan, nan, nan);
        reinterpret_cast<float4 *>(j.rays + (size_t)t * 8)[1] = make_float4(nan, nan, j.g.near, j.g.far);
        return;
    }
    if (j.target) j.target[t] = j.projections[flat];
    make_ray(j.poses, (uint64_t)flat, j.g, reinterpret_cast<float4 *>(j.rays + (size_t)t * 8));
}

// host side (render_ops.hip): argument checks of naf_draw_scan_rays -> a DrawJob; the stand-alone launch
int make_draw_job(const naf_scan_draw *draw, const float *poses, const float *projections, int64_t *pixels, float *target, float *rays,
                  uint32_t first_draw, uint32_t n_draws, uint32_t n_projections, uint32_t det_w, uint32_t det_h, float du, float dv, float ou,
                  float ov, float DSD, float near, float far, int parallel, uint64_t seed, DrawJob *job);
int launch_draw(const DrawJob &job, hipStream_t stream);

}  // namespace naf
