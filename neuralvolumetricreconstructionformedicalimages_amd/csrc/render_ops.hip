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
    hipLaunchKernelGGL(sample_rays_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, rays, t_rand,
                       z_vals, pts, n_rays, n_samples, perturb != 0, bound, seed, ray_index_base);
    return check_launch("sample_rays_kernel");
}

extern "C" int naf_integrate_forward(const float *sigma, const float *z_vals, const float *rays, float *acc,
                                     uint32_t n_rays, uint32_t n_samples, void *stream) {
    if (!sigma || !z_vals || !rays || !acc) return fail(NAF_ERR_INVALID_ARGUMENT, "integrate_forward: null pointer");
    if (n_rays == 0) return NAF_OK;
    hipLaunchKernelGGL(integrate_forward_kernel, dim3(grid_for(n_rays, 4)), dim3(256), 0, (hipStream_t)stream, sigma,
                       z_vals, rays, acc, n_rays, n_samples);
    return check_launch("integrate_forward_kernel");
}

extern "C" int naf_integrate_backward(const float *grad_acc, const float *z_vals, const float *rays, float *grad_sigma,
                                      uint32_t n_rays, uint32_t n_samples, void *stream) {
    if (!grad_acc || !z_vals || !rays || !grad_sigma) return fail(NAF_ERR_INVALID_ARGUMENT, "integrate_backward: null pointer");
    if (n_rays == 0) return NAF_OK;
    const uint64_t total = (uint64_t)n_rays * n_samples;
    hipLaunchKernelGGL(integrate_backward_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       grad_acc, z_vals, rays, grad_sigma, n_rays, n_samples);
    return check_launch("integrate_backward_kernel");
}

extern "C" int naf_normalize_inputs(const float *x, uint64_t n, float size, float *out01, int32_t *flag, void *stream) {
    if (!x || !flag) return fail(NAF_ERR_INVALID_ARGUMENT, "normalize_inputs: null pointer");
    if (n == 0) return NAF_OK;
    hipLaunchKernelGGL(normalize_inputs_kernel, dim3(grid_for(n, 1024, 2048)), dim3(256), 0, (hipStream_t)stream, x, n,
                       size, out01, flag);
    return check_launch("normalize_inputs_kernel");
}
