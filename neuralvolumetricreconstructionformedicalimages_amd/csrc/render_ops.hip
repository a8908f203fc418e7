// render_ops.hip -- stand-alone ray-march operators for gfx950: stratified sampling, the attenuation line
// integral and its backward, the encoder range check.  They back the drop-in `render()` surface
// (reference src/render/render.py:82-212); the fused training path lives in render_fused.hip.
#include "naf_device.h"
#include "naf_host.h"

namespace naf {

// render.py:87-105.  One lane = one sample; a ray's 8 floats are fetched once per lane through L1 (broadcast).
__global__ void __launch_bounds__(256)
sample_rays_kernel(const float *__restrict__ rays, const float *__restrict__ t_rand, float *__restrict__ z_vals,
                   float *__restrict__ pts, uint32_t n_rays, uint32_t S, bool perturb, float bound, uint64_t seed,
                   uint32_t ray_base) {
    const uint64_t total = (uint64_t)n_rays * S;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t r = (uint32_t)(i / S), s = (uint32_t)(i - (uint64_t)r * S);
        const float *ray = rays + (size_t)r * 8;
        const float near = ray[6], far = ray[7];
        float u = 0.0f;
        if (perturb) u = t_rand ? t_rand[i] : jitter(seed, ray_base + r, s);
        const float z = sample_z(near, far, s, S, perturb, u);
        z_vals[i] = z;
        const float lim = bound - 1e-6f;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            float p = ray[d] + ray[3 + d] * z;          // mul then add, as torch evaluates it
            p = fminf(fmaxf(p, -lim), lim);
            pts[i * 3 + d] = p;
        }
    }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

__device__ __forceinline__ float dist_at(const float *__restrict__ z, uint32_t s, uint32_t S, float dnorm) {
    const float d = (s + 1u < S) ? z[s + 1] - z[s] : 1e-10f;     // render.py:192-193
    return d * dnorm;                                              // render.py:194
}

// render.py:192-201: one wave per ray, lanes stride the samples, wave reduction of sigma*dist.
__global__ void __launch_bounds__(256)
integrate_forward_kernel(const float *__restrict__ sigma, const float *__restrict__ z_vals,
                         const float *__restrict__ rays, float *__restrict__ acc, uint32_t n_rays, uint32_t S) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t n_waves = (gridDim.x * blockDim.x) >> 6;
    for (uint32_t r = wave; r < n_rays; r += n_waves) {
        const float *ray = rays + (size_t)r * 8;
        const float dnorm = sqrtf(ray[3] * ray[3] + ray[4] * ray[4] + ray[5] * ray[5]);
        const float *z = z_vals + (size_t)r * S;
        float part = 0.0f;
        for (uint32_t s = lane; s < S; s += 64u) part += sigma[(size_t)r * S + s] * dist_at(z, s, S, dnorm);
        part = wave_sum(part);
        if (lane == 0) acc[r] = part;
    }
}

__global__ void __launch_bounds__(256)
integrate_backward_kernel(const float *__restrict__ grad_acc, const float *__restrict__ z_vals,
                          const float *__restrict__ rays, float *__restrict__ grad_sigma, uint32_t n_rays, uint32_t S) {
    const uint64_t total = (uint64_t)n_rays * S;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t r = (uint32_t)(i / S), s = (uint32_t)(i - (uint64_t)r * S);
        const float *ray = rays + (size_t)r * 8;
        const float dnorm = sqrtf(ray[3] * ray[3] + ray[4] * ray[4] + ray[5] * ray[5]);
        grad_sigma[i] = grad_acc[r] * dist_at(z_vals + (size_t)r * S, s, S, dnorm);
    }
}

// hashgrid.py:122-125 without the host round trips: normalise to [0,1] and flag[0] = out-of-range seen,
// flag[1]/flag[2] = min/max as order-preserving ints (so the ValueError can print the range like the reference).
__device__ __forceinline__ int32_t ordered_int(float f) {
    const int32_t i = __float_as_int(f);
    return i >= 0 ? i : i ^ 0x7fffffff;
}

__global__ void __launch_bounds__(256)
normalize_inputs_kernel(const float *__restrict__ x, uint64_t n, float size, float *__restrict__ out01,
                        int32_t *__restrict__ flag) {
    const float lo = -size, hi = size, denom = 2.0f * size;
    float mn = INFINITY, mx = -INFINITY;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const float v = x[i];
        mn = fminf(mn, v);
        mx = fmaxf(mx, v);
        if (v != v) { mn = v; mx = v; }
        if (out01) out01[i] = (v + size) / denom;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        mn = fminf(mn, __shfl_xor(mn, off, 64));
        mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    }
    if ((threadIdx.x & 63u) == 0 && !(mn > mx)) {
        if (!(mn >= lo) || !(mx <= hi)) atomicOr(flag, 1);
        atomicMin(flag + 1, ordered_int(mn));
        atomicMax(flag + 2, ordered_int(mx));
    }
}

static uint32_t grid_for(uint64_t items, uint32_t per_block, uint32_t cap = 256u * 32u) {
    return (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>((items + per_block - 1) / per_block, cap));
}

}  // namespace naf

using namespace naf;

extern "C" int naf_sample_rays(const float *rays, const float *t_rand, float *z_vals, float *pts, uint32_t n_rays,
                               uint32_t n_samples, int perturb, float bound, uint64_t seed, uint32_t ray_index_base,
                               void *stream) {
    if (!rays || !z_vals || !pts) return fail(NAF_ERR_INVALID_ARGUMENT, "sample_rays: null pointer");
    if (n_samples < 2) return fail(NAF_ERR_INVALID_ARGUMENT, "sample_rays: n_samples must be >= 2");
    if (n_rays == 0) return NAF_OK;
    const uint64_t total = (uint64_t)n_rays * n_samples;
    { ProfScope prof_("sample_rays_kernel", (hipStream_t)stream); hipLaunchKernelGGL(sample_rays_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, rays, t_rand,
                       z_vals, pts, n_rays, n_samples, perturb != 0, bound, seed, ray_index_base); }
    return check_launch("sample_rays_kernel");
}

extern "C" int naf_integrate_forward(const float *sigma, const float *z_vals, const float *rays, float *acc,
                                     uint32_t n_rays, uint32_t n_samples, void *stream) {
    if (!sigma || !z_vals || !rays || !acc) return fail(NAF_ERR_INVALID_ARGUMENT, "integrate_forward: null pointer");
    if (n_rays == 0) return NAF_OK;
    { ProfScope prof_("integrate_forward_kernel", (hipStream_t)stream); hipLaunchKernelGGL(integrate_forward_kernel, dim3(grid_for(n_rays, 4)), dim3(256), 0, (hipStream_t)stream, sigma,
                       z_vals, rays, acc, n_rays, n_samples); }
    return check_launch("integrate_forward_kernel");
}

extern "C" int naf_integrate_backward(const float *grad_acc, const float *z_vals, const float *rays, float *grad_sigma,
                                      uint32_t n_rays, uint32_t n_samples, void *stream) {
    if (!grad_acc || !z_vals || !rays || !grad_sigma) return fail(NAF_ERR_INVALID_ARGUMENT, "integrate_backward: null pointer");
    if (n_rays == 0) return NAF_OK;
    const uint64_t total = (uint64_t)n_rays * n_samples;
    { ProfScope prof_("integrate_backward_kernel", (hipStream_t)stream); hipLaunchKernelGGL(integrate_backward_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       grad_acc, z_vals, rays, grad_sigma, n_rays, n_samples); }
    return check_launch("integrate_backward_kernel");
}

extern "C" int naf_normalize_inputs(const float *x, uint64_t n, float size, float *out01, int32_t *flag, void *stream) {
    if (!x || !flag) return fail(NAF_ERR_INVALID_ARGUMENT, "normalize_inputs: null pointer");
    if (n == 0) return NAF_OK;
    { ProfScope prof_("normalize_inputs_kernel", (hipStream_t)stream); hipLaunchKernelGGL(normalize_inputs_kernel, dim3(grid_for(n, 1024, 2048)), dim3(256), 0, (hipStream_t)stream, x, n,
                       size, out01, flag); }
    return check_launch("normalize_inputs_kernel");
}

// ---- G3/G4: on-the-fly ray generation (reference src/dataset/tigre.py:402-456, 463-528) --------------------------
// The reference precomputes rays[N,H,W,8] for every pixel of every projection (419 MB at 50x512^2, 24 GB at
// 720x1024^2); here a ray is 32 bytes produced on demand from its pose and pixel.
namespace naf {

struct RayGeo {
    uint32_t W, H;          // detector columns / rows (nDetector[0], nDetector[1])
    float du, dv;           // pixel pitch  (dDetector)
    float ou, ov;           // detector offset (offDetector)
    float DSD;
    float near, far;        // tigre.py:575-586
    int parallel;           // 0 cone, 1 parallel
};

__global__ void __launch_bounds__(256)
generate_rays_kernel(const float *__restrict__ poses, const int64_t *__restrict__ pixels, int64_t first_pixel,
                     float *__restrict__ rays, uint64_t n, RayGeo g) {
    const uint64_t per_proj = (uint64_t)g.W * g.H;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t flat = pixels ? (uint64_t)pixels[i] : (uint64_t)first_pixel + i;
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
        float4 *out = reinterpret_cast<float4 *>(rays + i * 8);
        out[0] = make_float4(o[0], o[1], o[2], d[0]);
        out[1] = make_float4(d[1], d[2], g.near, g.far);
    }
}

}  // namespace naf

extern "C" int naf_generate_rays(const float *poses, const int64_t *pixels, int64_t first_pixel, float *rays, uint64_t n,
                                 uint32_t n_projections, uint32_t det_w, uint32_t det_h, float du, float dv, float ou,
                                 float ov, float DSD, float near, float far, int parallel, void *stream) {
    if (!poses || !rays) return fail(NAF_ERR_INVALID_ARGUMENT, "generate_rays: null pointer");
    if (det_w == 0 || det_h == 0 || n_projections == 0) return fail(NAF_ERR_INVALID_ARGUMENT, "generate_rays: empty detector");
    if (((uintptr_t)rays) & 15u) return fail(NAF_ERR_INVALID_ARGUMENT, "generate_rays: rays must be 16-byte aligned");
    if (!pixels && (first_pixel < 0 || (uint64_t)first_pixel + n > (uint64_t)n_projections * det_w * det_h))
        return fail(NAF_ERR_INVALID_ARGUMENT, "generate_rays: pixel range outside the scan");
    if (n == 0) return NAF_OK;
    RayGeo g{det_w, det_h, du, dv, ou, ov, DSD, near, far, parallel};
    { ProfScope prof_("generate_rays_kernel", (hipStream_t)stream); hipLaunchKernelGGL(generate_rays_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, poses, pixels,
                       first_pixel, rays, n, g); }
    return check_launch("generate_rays_kernel");
}
