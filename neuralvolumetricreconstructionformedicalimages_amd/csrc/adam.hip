// adam.hip -- fused Adam update for gfx950 (torch.optim.Adam semantics as used by reference
// src/trainer.py:54: betas=(0.9,0.999), eps=1e-8, no weight decay, no amsgrad).
//
// One pass over (param, exp_avg, exp_avg_sq, grad): 16 B/lane vector accesses, optional low-precision
// shadow copy of the parameters (16-bit hash tables keep an fp32 master) and optional in-place zeroing of
// the gradient, so the 57 MB (T=2^19) / 422 MB (T=2^22) table is streamed exactly once per step.
#include "adam_math.h"
#include "naf_device.h"
#include "naf_host.h"

#include <cmath>

namespace naf {

template <int LP>   // 0 none, 1 f16, 2 bf16
__global__ void __launch_bounds__(256)
adam_kernel(float *__restrict__ param, float *__restrict__ m, float *__restrict__ v, float *__restrict__ grad,
            void *__restrict__ param_lp, uint64_t n, AdamArgs a, bool zero_grad) {
    constexpr bool kFast = LP != 0;        // a 16-bit shadow is what the kernels read: adam_math.h
    const uint64_t n4 = n / 4;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (uint64_t)gridDim.x * blockDim.x) {
        float4 p = reinterpret_cast<float4 *>(param)[i];
        float4 mm = reinterpret_cast<float4 *>(m)[i];
        float4 vv = reinterpret_cast<float4 *>(v)[i];
        const float4 g = reinterpret_cast<float4 *>(grad)[i];
        adam_one<kFast>(p.x, mm.x, vv.x, g.x, a);
        adam_one<kFast>(p.y, mm.y, vv.y, g.y, a);
        adam_one<kFast>(p.z, mm.z, vv.z, g.z, a);
        adam_one<kFast>(p.w, mm.w, vv.w, g.w, a);
        reinterpret_cast<float4 *>(param)[i] = p;
        reinterpret_cast<float4 *>(m)[i] = mm;
        reinterpret_cast<float4 *>(v)[i] = vv;
        if (zero_grad) reinterpret_cast<float4 *>(grad)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if constexpr (LP == 1) {
            _Float16 h[4] = {(_Float16)p.x, (_Float16)p.y, (_Float16)p.z, (_Float16)p.w};
            uint2 raw; __builtin_memcpy(&raw, h, 8);
            reinterpret_cast<uint2 *>(param_lp)[i] = raw;
        } else if constexpr (LP == 2) {
            uint16_t h[4] = {f32_to_bf16(p.x), f32_to_bf16(p.y), f32_to_bf16(p.z), f32_to_bf16(p.w)};
            uint2 raw; __builtin_memcpy(&raw, h, 8);
            reinterpret_cast<uint2 *>(param_lp)[i] = raw;
        }
    }
    // tail (n not a multiple of 4): handled by the first few lanes of block 0
    if (blockIdx.x == 0 && threadIdx.x < (n & 3u)) {
        const uint64_t i = n4 * 4 + threadIdx.x;
        float p = param[i], mm = m[i], vv = v[i];
        adam_one<kFast>(p, mm, vv, grad[i], a);
        param[i] = p; m[i] = mm; v[i] = vv;
        if (zero_grad) grad[i] = 0.0f;
        if constexpr (LP == 1) reinterpret_cast<_Float16 *>(param_lp)[i] = (_Float16)p;
        else if constexpr (LP == 2) reinterpret_cast<uint16_t *>(param_lp)[i] = f32_to_bf16(p);
    }
}

}  // namespace naf

using namespace naf;

AdamArgs naf::make_adam_args(float lr, float beta1, float beta2, float eps, uint32_t step, float grad_scale) {
    AdamArgs a;
    a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.grad_scale = grad_scale;
    a.bias1 = (float)(1.0 - std::pow((double)beta1, (double)step));
    a.bias2_sqrt = (float)std::sqrt(1.0 - std::pow((double)beta2, (double)step));
    a.step_size = a.lr / a.bias1;
    a.inv_bias2_sqrt = 1.0f / a.bias2_sqrt;
    return a;
}

int naf::launch_adam(float *param, float *exp_avg, float *exp_avg_sq, float *grad, void *param_lp, int lp_dtype, uint64_t n,
                     const AdamArgs &a, bool zero_grad, hipStream_t s) {
    if (n == 0) return NAF_OK;
    const uint32_t grid = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>((n / 4 + 255) / 256, 256u * 16u));
    if (!param_lp) { ProfScope prof_("adam_kernel", s); hipLaunchKernelGGL(adam_kernel<0>, dim3(grid), dim3(256), 0, s, param, exp_avg, exp_avg_sq, grad, nullptr, n, a, zero_grad); }
    else if (lp_dtype == NAF_F16) { ProfScope prof_("adam_kernel", s); hipLaunchKernelGGL(adam_kernel<1>, dim3(grid), dim3(256), 0, s, param, exp_avg, exp_avg_sq, grad, param_lp, n, a, zero_grad); }
    else if (lp_dtype == NAF_BF16) { ProfScope prof_("adam_kernel", s); hipLaunchKernelGGL(adam_kernel<2>, dim3(grid), dim3(256), 0, s, param, exp_avg, exp_avg_sq, grad, param_lp, n, a, zero_grad); }
    else return fail(NAF_ERR_UNSUPPORTED, "adam_step: lp_dtype must be NAF_F16 or NAF_BF16 when param_lp is given");
    return check_launch("adam_kernel");
}

extern "C" int naf_adam_step(float *param, float *exp_avg, float *exp_avg_sq, float *grad, void *param_lp, int lp_dtype,
                             uint64_t n, float lr, float beta1, float beta2, float eps, uint32_t step, float grad_scale,
                             int zero_grad, void *stream) {
    if (n != 0 && (!param || !exp_avg || !exp_avg_sq || !grad)) return fail(NAF_ERR_INVALID_ARGUMENT, "adam_step: null pointer");
    if (step == 0) return fail(NAF_ERR_INVALID_ARGUMENT, "adam_step: step is 1-based");
    // (fewer than four elements take the scalar tail of the kernel: the ragged head of a row range that starts inside a 16-byte group)
    if (n >= 4 && (((uintptr_t)param | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq | (uintptr_t)grad) & 15u))
        return fail(NAF_ERR_INVALID_ARGUMENT, "adam_step: buffers must be 16-byte aligned");
    return launch_adam(param, exp_avg, exp_avg_sq, grad, param_lp, lp_dtype, n, make_adam_args(lr, beta1, beta2, eps, step, grad_scale),
                       zero_grad != 0, (hipStream_t)stream);
}
