// render_ops.hip -- stand-alone ray-march operators for gfx950: stratified sampling, the attenuation line
// integral and its backward, the encoder range check.  They back the drop-in `render()` surface
// (reference src/render/render.py:82-212); the fused training path lives in render_fused.hip.
#include <cmath>
#include <cstring>

#include "naf_device.h"
#include "naf_host.h"
#include "draw_device.h"

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
        if (pts == nullptr) continue;                   // depths only (engine.sample_depths)
        const float lim = bound - 1e-6f;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            float p = ray[d] + ray[3 + d] * z;          // mul then add, as torch evaluates it
            p = p < -lim ? -lim : p;                    // torch.clamp: a NaN position stays NaN (fminf / fmaxf would drop it)
            p = p > lim ? lim : p;
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

// ---- coarse -> fine resampling (render.py:113-126 + raw2outputs weights :203-206 + sample_pdf :215-247) ----------------
// The reference runs ~25 ATen kernels per chunk for this (abs/cat/max/div, cumsum, searchsorted, two gathers over an expanded
// [n, N_fine, S] view, where, sort of the concatenation).  Here: one small pass for the chunk-wide maximum of the weights,
// then ONE wave per ray does everything in LDS -- coarse depths, interval weights, their inclusive WAVE PREFIX SUM (the
// cdf), inverse-transform sampling by binary search, and a bitonic sort of the merged depths.

// weights = |sigma[s] - sigma[s-1]| (1e-10 for the first sample); max over the whole chunk as the bit pattern of a
// non-negative float (orders like an unsigned integer).  The caller initialises *max_bits with the bits of 1e-10f.
__global__ void __launch_bounds__(256)
fine_weight_max_kernel(const float *__restrict__ sigma, uint64_t total, uint32_t S, uint32_t *__restrict__ max_bits) {
    float m = 0.0f;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t s = (uint32_t)(i % S);
        if (s != 0u) m = fmaxf(m, fabsf(sigma[i] - sigma[i - 1]));
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    // one atomic per WORKGROUP, not per wave: same-address atomics retire one at a time (~12 ns each on MI355X)
    __shared__ float wave_max[4];
    if ((threadIdx.x & 63u) == 0u) wave_max[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0u) {
        m = fmaxf(fmaxf(wave_max[0], wave_max[1]), fmaxf(wave_max[2], wave_max[3]));
        if (m > 0.0f) atomicMax(max_bits, __float_as_uint(m));
    }
}

__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// LDS per wave: z[S] | cdf[S-1 (+1)] | merged[P], P = power of two >= S + N_fine.
__global__ void __launch_bounds__(256)
fine_depths_kernel(const float *__restrict__ rays, const float *__restrict__ t_rand, const float *__restrict__ sigma,
                   const uint32_t *__restrict__ max_bits, const float *__restrict__ u_rand, float *__restrict__ z_out,
                   float *__restrict__ weights_out, uint32_t n_rays, uint32_t S, uint32_t NF, uint32_t P, bool perturb, bool det,
                   uint64_t seed, uint32_t ray_base) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t lane = threadIdx.x & 63u, wib = threadIdx.x >> 6;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = (gridDim.x * blockDim.x) >> 6;
    float *zc = reinterpret_cast<float *>(smem) + (size_t)wib * (2u * S + P);
    float *cdf = zc + S;                                    // S - 1 entries: one per bin edge (mid-point of two samples)
    float *merged = cdf + S;
    const float wmax = __uint_as_float(*max_bits);
    const uint32_t M = S - 1u;                              // bin edges; M - 1 = S - 2 intervals carry the weights 1 .. S-2
    for (uint32_t r = wave; r < n_rays; r += n_waves) {
        const float *ray = rays + (size_t)r * 8;
        const float near = ray[6], far = ray[7];
        for (uint32_t s = lane; s < S; s += 64u) {
            float uu = 0.0f;
            if (perturb) uu = t_rand ? t_rand[(size_t)r * S + s] : jitter(seed, ray_base + r, s);
            const float z = sample_z(near, far, s, S, perturb, uu);
            zc[s] = z;
            merged[s] = z;
        }
        // interval weights (+1e-5, render.py:217) and their sum
        const float *sg = sigma + (size_t)r * S;
        float part = 0.0f;
        for (uint32_t s = lane; s < S; s += 64u) {
            const float w = s == 0u ? 1e-10f / wmax : fabsf(sg[s] - sg[s - 1u]) / wmax;
            if (weights_out) weights_out[(size_t)r * S + s] = w;
            if (s >= 1u && s + 1u < S) part += w + 1e-5f;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);
        const float total = part;
        // cdf[k] = sum_{i < k} pdf[i], pdf[i] = (w[i+1] + 1e-5) / total: inclusive wave prefix sum, 64 intervals at a time
        float carry = 0.0f;
        if (lane == 0u) cdf[0] = 0.0f;
        for (uint32_t base = 0; base + 1u < M; base += 64u) {
            const uint32_t i = base + lane;                 // interval index, < M - 1
            const bool in = i + 1u < M;
            float v = in ? (fabsf(sg[i + 1u] - sg[i]) / wmax + 1e-5f) / total : 0.0f;
#pragma unroll
            for (uint32_t d = 1; d < 64u; d <<= 1) {
                const float up = __shfl_up(v, d, 64);
                if (lane >= d) v += up;
            }
            v += carry;
            if (in) cdf[i + 1u] = v;
            carry = __shfl(v, 63, 64);
        }
        wave_lds_sync();
        // inverse-transform samples
        for (uint32_t j = lane; j < NF; j += 64u) {
            float uq;
            if (det) uq = NF > 1u ? lin_t(j, NF) : 0.0f;     // torch.linspace(0, 1, N_fine); a single step is [0]
            else uq = u_rand ? u_rand[(size_t)r * NF + j] : jitter(seed ^ 0x9e3779b97f4a7c15ull, ray_base + r, j);
            uint32_t lo = 0u, hi = M;                        // searchsorted(cdf, u, right=True): entries <= u
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if (cdf[mid] <= uq) lo = mid + 1u; else hi = mid;
            }
            const uint32_t below = lo == 0u ? 0u : lo - 1u, above = lo < M ? lo : M - 1u;
            const float c0 = cdf[below], c1 = cdf[above];
            float denom = c1 - c0;
            if (denom < 1e-5f) denom = 1.0f;
            const float t = (uq - c0) / denom;
            const float b0 = 0.5f * (zc[below + 1u] + zc[below]), b1 = 0.5f * (zc[above + 1u] + zc[above]);
            merged[S + j] = b0 + t * (b1 - b0);
        }
        for (uint32_t i = S + NF + lane; i < P; i += 64u) merged[i] = INFINITY;
        wave_lds_sync();
        // bitonic sort of merged[0, P), ascending
        for (uint32_t k = 2u; k <= P; k <<= 1) {
            for (uint32_t j = k >> 1; j > 0u; j >>= 1) {
                for (uint32_t t = lane; t < (P >> 1); t += 64u) {
                    const uint32_t i = ((t & ~(j - 1u)) << 1) | (t & (j - 1u)), q = i | j;     // the pair (i, i + j)
                    const float a = merged[i], b = merged[q];
                    const bool up = (i & k) == 0u;
                    if ((a > b) == up) { merged[i] = b; merged[q] = a; }
                }
                wave_lds_sync();
            }
        }
        for (uint32_t i = lane; i < S + NF; i += 64u) z_out[(size_t)r * (S + NF) + i] = merged[i];
        wave_lds_sync();
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
    if (n_rays != 0 && (!rays || !z_vals)) return fail(NAF_ERR_INVALID_ARGUMENT, "sample_rays: null pointer");
    if (n_samples < 2) return fail(NAF_ERR_INVALID_ARGUMENT, "sample_rays: n_samples must be >= 2");
    if (n_rays == 0) return NAF_OK;
    const uint64_t total = (uint64_t)n_rays * n_samples;
    { ProfScope prof_("sample_rays_kernel", (hipStream_t)stream); hipLaunchKernelGGL(sample_rays_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, rays, t_rand,
                       z_vals, pts, n_rays, n_samples, perturb != 0, bound, seed, ray_index_base); }
    return check_launch("sample_rays_kernel");
}

extern "C" int naf_integrate_forward(const float *sigma, const float *z_vals, const float *rays, float *acc,
                                     uint32_t n_rays, uint32_t n_samples, void *stream) {
    if (n_rays != 0 && (!sigma || !z_vals || !rays || !acc)) return fail(NAF_ERR_INVALID_ARGUMENT, "integrate_forward: null pointer");
    if (n_rays == 0) return NAF_OK;
    { ProfScope prof_("integrate_forward_kernel", (hipStream_t)stream); hipLaunchKernelGGL(integrate_forward_kernel, dim3(grid_for(n_rays, 4)), dim3(256), 0, (hipStream_t)stream, sigma,
                       z_vals, rays, acc, n_rays, n_samples); }
    return check_launch("integrate_forward_kernel");
}

extern "C" int naf_integrate_backward(const float *grad_acc, const float *z_vals, const float *rays, float *grad_sigma,
                                      uint32_t n_rays, uint32_t n_samples, void *stream) {
    if (n_rays != 0 && (!grad_acc || !z_vals || !rays || !grad_sigma)) return fail(NAF_ERR_INVALID_ARGUMENT, "integrate_backward: null pointer");
    if (n_rays == 0) return NAF_OK;
    const uint64_t total = (uint64_t)n_rays * n_samples;
    { ProfScope prof_("integrate_backward_kernel", (hipStream_t)stream); hipLaunchKernelGGL(integrate_backward_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       grad_acc, z_vals, rays, grad_sigma, n_rays, n_samples); }
    return check_launch("integrate_backward_kernel");
}

extern "C" int naf_fine_depths(const float *rays, const float *t_rand, const float *sigma, const float *u, float *z_out,
                               float *weights_out, uint32_t n_rays, uint32_t n_samples, uint32_t n_fine, int perturb, int det,
                               uint64_t seed, uint32_t ray_index_base, void *scratch, void *stream) {
    if (n_rays != 0 && (!rays || !sigma || !z_out || !scratch)) return fail(NAF_ERR_INVALID_ARGUMENT, "fine_depths: null pointer");
    if (n_samples < 3 || n_fine == 0) return fail(NAF_ERR_INVALID_ARGUMENT, "fine_depths: needs n_samples >= 3 and n_fine >= 1");
    if (n_samples > 1024u || n_samples + n_fine > 2048u)
        return fail(NAF_ERR_UNSUPPORTED, "fine_depths: at most 1024 coarse and 2048 merged depths per ray (LDS-resident sort)");
    if (n_rays == 0) return NAF_OK;
    hipStream_t s = (hipStream_t)stream;
    const float floor_w = 1e-10f;                            // the first sample's weight (render.py:204) bounds the maximum from below
    uint32_t bits;
    std::memcpy(&bits, &floor_w, 4);
    if (hipMemsetD32Async((hipDeviceptr_t)scratch, (int)bits, 1, s) != hipSuccess) return fail(NAF_ERR_LAUNCH, "fine_depths: memset failed");
    const uint64_t total = (uint64_t)n_rays * n_samples;
    { ProfScope prof_("fine_weight_max_kernel", s); hipLaunchKernelGGL(fine_weight_max_kernel, dim3(grid_for(total, 1024, 1024)), dim3(256), 0, s, sigma, total, n_samples, (uint32_t *)scratch); }
    if (int rc = check_launch("fine_weight_max_kernel")) return rc;
    uint32_t P = 2;
    while (P < n_samples + n_fine) P <<= 1;
    const uint32_t lds = 4u * (2u * n_samples + P) * 4u;
    if (lds > (64u << 10) &&
        hipFuncSetAttribute((const void *)fine_depths_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return fail(NAF_ERR_LAUNCH, "fine_depths: cannot raise dynamic LDS limit");
    { ProfScope prof_("fine_depths_kernel", s); hipLaunchKernelGGL(fine_depths_kernel, dim3(grid_for(n_rays, 4)), dim3(256), lds, s, rays, t_rand, sigma, (const uint32_t *)scratch, u, z_out,
                       weights_out, n_rays, n_samples, n_fine, P, perturb != 0, det != 0, seed, ray_index_base); }
    return check_launch("fine_depths_kernel");
}

extern "C" int naf_normalize_inputs(const float *x, uint64_t n, float size, float *out01, int32_t *flag, void *stream) {
    if (n != 0 && (!x || !flag)) return fail(NAF_ERR_INVALID_ARGUMENT, "normalize_inputs: null pointer");
    if (n == 0) return NAF_OK;
    { ProfScope prof_("normalize_inputs_kernel", (hipStream_t)stream); hipLaunchKernelGGL(normalize_inputs_kernel, dim3(grid_for(n, 1024, 2048)), dim3(256), 0, (hipStream_t)stream, x, n,
                       size, out01, flag); }
    return check_launch("normalize_inputs_kernel");
}

// ---- G3/G4: on-the-fly ray generation (reference src/dataset/tigre.py:402-456, 463-528) --------------------------
// The reference precomputes rays[N,H,W,8] for every pixel of every projection (419 MB at 50x512^2, 24 GB at
// 720x1024^2); here a ray is 32 bytes produced on demand from its pose and pixel.
namespace naf {

__global__ void __launch_bounds__(256)
generate_rays_kernel(const float *__restrict__ poses, const int64_t *__restrict__ pixels, int64_t first_pixel,
                     float *__restrict__ rays, uint64_t n, RayGeo g, uint64_t n_pixels) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t flat = pixels ? (uint64_t)pixels[i] : (uint64_t)first_pixel + i;
        if (flat >= n_pixels) {                                 // list entry outside the scan (negative ones included): a NaN ray, no pose read
            const float nan = __builtin_nanf("");
            reinterpret_cast<float4 *>(rays + i * 8)[0] = make_float4(nan, nan, nan, nan);
            reinterpret_cast<float4 *>(rays + i * 8)[1] = make_float4(nan, nan, g.near, g.far);
            continue;
        }
        make_ray(poses, flat, g, reinterpret_cast<float4 *>(rays + i * 8));
    }
}

__global__ void __launch_bounds__(256)
draw_scan_rays_kernel(DrawJob job) {
    for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < job.count; t += gridDim.x * blockDim.x) draw_one(job, t);
}

}  // namespace naf

int naf::make_draw_job(const naf_scan_draw *draw, const float *poses, const float *projections, int64_t *pixels, float *target, float *rays,
                       uint32_t first_draw, uint32_t n_draws, uint32_t n_projections, uint32_t det_w, uint32_t det_h, float du, float dv,
                       float ou, float ov, float DSD, float near, float far, int parallel, uint64_t seed, DrawJob *job) {
    if (!draw || (n_draws != 0 && (!poses || !rays))) return fail(NAF_ERR_INVALID_ARGUMENT, "draw_scan_rays: null pointer");
    if (n_draws != 0 && target && !projections) return fail(NAF_ERR_INVALID_ARGUMENT, "draw_scan_rays: target without projections");
    if (draw->n_segments == 0 || draw->n_segments > NAF_MAX_DRAW_SEGMENTS || draw->rays_per_segment == 0)
        return fail(NAF_ERR_INVALID_ARGUMENT, "draw_scan_rays: n_segments must be in [1, 16] and rays_per_segment > 0");
    if (det_w == 0 || det_h == 0 || n_projections == 0) return fail(NAF_ERR_INVALID_ARGUMENT, "draw_scan_rays: empty detector");
    if (n_draws != 0 && (((uintptr_t)rays) & 15u)) return fail(NAF_ERR_INVALID_ARGUMENT, "draw_scan_rays: rays must be 16-byte aligned");
    const uint64_t total = (uint64_t)draw->n_segments * draw->rays_per_segment;
    if (total > 0xffffffffull) return fail(NAF_ERR_INVALID_ARGUMENT, "draw_scan_rays: more than 2^32 - 1 draws per call");
    if ((uint64_t)first_draw + n_draws > total) return fail(NAF_ERR_INVALID_ARGUMENT, "draw_scan_rays: draw range outside n_segments * rays_per_segment");
    ScanDraw &d = job->draw;
    d.n_segments = draw->n_segments;
    d.per_segment = draw->rays_per_segment;
    for (uint32_t j = 0; j < NAF_MAX_DRAW_SEGMENTS; ++j) {
        const bool used = j < draw->n_segments;
        d.valid[j] = used ? draw->valid[j] : nullptr;
        d.n_valid[j] = used ? draw->n_valid[j] : 0u;
        if (used && !draw->valid[j]) return fail(NAF_ERR_INVALID_ARGUMENT, "draw_scan_rays: null valid-pixel list");
        // the reference's np.random.choice(..., replace=False) raises the same way (tigre.py:357)
        if (used && draw->n_valid[j] < draw->rays_per_segment)
            return fail(NAF_ERR_INVALID_ARGUMENT, "Cannot take a larger sample than population when 'replace=False'");
    }
    job->poses = poses; job->projections = projections; job->pixels = pixels; job->target = target; job->rays = rays;
    job->first = first_draw; job->count = n_draws;
    job->g = RayGeo{det_w, det_h, du, dv, ou, ov, DSD, near, far, parallel};
    job->seed = seed; job->n_pixels = (uint64_t)n_projections * det_w * det_h;
    return NAF_OK;
}

int naf::launch_draw(const DrawJob &job, hipStream_t stream) {
    if (job.count == 0) return NAF_OK;
    { ProfScope prof_("draw_scan_rays_kernel", stream);
      hipLaunchKernelGGL(draw_scan_rays_kernel, dim3(grid_for(job.count, 256)), dim3(256), 0, stream, job); }
    return check_launch("draw_scan_rays_kernel");
}

extern "C" int naf_draw_scan_rays(const naf_scan_draw *draw, const float *poses, const float *projections, int64_t *pixels,
                                  float *target, float *rays, uint32_t first_draw, uint32_t n_draws, uint32_t n_projections,
                                  uint32_t det_w, uint32_t det_h, float du, float dv, float ou, float ov, float DSD, float near,
                                  float far, int parallel, uint64_t seed, void *stream) {
    DrawJob job;
    if (int rc = make_draw_job(draw, poses, projections, pixels, target, rays, first_draw, n_draws, n_projections, det_w, det_h, du, dv, ou, ov,
                               DSD, near, far, parallel, seed, &job)) return rc;
    return launch_draw(job, (hipStream_t)stream);
}

extern "C" int naf_generate_rays(const float *poses, const int64_t *pixels, int64_t first_pixel, float *rays, uint64_t n,
                                 uint32_t n_projections, uint32_t det_w, uint32_t det_h, float du, float dv, float ou,
                                 float ov, float DSD, float near, float far, int parallel, void *stream) {
    if (n != 0 && (!poses || !rays)) return fail(NAF_ERR_INVALID_ARGUMENT, "generate_rays: null pointer");
    if (det_w == 0 || det_h == 0 || n_projections == 0) return fail(NAF_ERR_INVALID_ARGUMENT, "generate_rays: empty detector");
    if (((uintptr_t)rays) & 15u) return fail(NAF_ERR_INVALID_ARGUMENT, "generate_rays: rays must be 16-byte aligned");
    if (!pixels && (first_pixel < 0 || (uint64_t)first_pixel + n > (uint64_t)n_projections * det_w * det_h))
        return fail(NAF_ERR_INVALID_ARGUMENT, "generate_rays: pixel range outside the scan");
    if (n == 0) return NAF_OK;
    RayGeo g{det_w, det_h, du, dv, ou, ov, DSD, near, far, parallel};
    { ProfScope prof_("generate_rays_kernel", (hipStream_t)stream); hipLaunchKernelGGL(generate_rays_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, poses, pixels,
                       first_pixel, rays, n, g, (uint64_t)n_projections * det_w * det_h); }
    return check_launch("generate_rays_kernel");
}
