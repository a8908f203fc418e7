// adam_math.h -- the Adam update of one element, shared by adam_kernel (adam.hip) and by the tail of the gradient
// reducer (scatter_binned.h), which applies it to the table rows it has just finished (naf_render_train_adam): both
// compile the same expression with -ffp-contract=off, so the two routes give the same bits.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace naf {

struct AdamArgs {
    float lr, beta1, beta2, eps, bias1, bias2_sqrt, grad_scale;
    float step_size, inv_bias2_sqrt;       // lr / bias1 and 1 / bias2_sqrt, rounded once on the host (fast form only)
};

// kFast = false: torch's operation sequence with correctly rounded sqrt and divisions (fp32 parity mode, the MLP, any table without a
// 16-bit shadow).  kFast = true: the same update with one hardware reciprocal and the hardware square root (1 ulp each) in place
// of the two IEEE divisions and the corrected root -- 24 fewer vector instructions per element, which matters where the update is
// issue-bound (the tail of the gradient reducer at small batches: profiles/round3_wave_state.md).  Used exactly where the table the
// kernels READ is a 16-bit shadow of the master: its 8 or 11 mantissa bits swallow a 1-ulp difference in the fp32 master many times
// over.  Both routes to a table update (reducer tail, adam_kernel) pick the form by the same rule, so they still give the same bits.
template <bool kFast = false>
__device__ __forceinline__ float adam_one(float &p, float &m, float &v, float g, const AdamArgs &a) {
    g *= a.grad_scale;
    m = m + (g - m) * (1.0f - a.beta1);                    // torch: exp_avg.lerp_(grad, 1-beta1)
    v = v * a.beta2 + (1.0f - a.beta2) * g * g;            // torch: exp_avg_sq.mul_(b2).addcmul_(g,g,1-b2)
    if constexpr (kFast) {
        const float denom = __builtin_amdgcn_sqrtf(v) * a.inv_bias2_sqrt + a.eps;      // (a denormal v is far below eps^2 either way)
        p = p - a.step_size * (m * __builtin_amdgcn_rcpf(denom));
    } else {
        const float denom = sqrtf(v) / a.bias2_sqrt + a.eps;   // torch: (sqrt(v)/sqrt(bias2)).add_(eps)
        p = p - (a.lr / a.bias1) * (m / denom);                // torch: param.addcdiv_(m, denom, -lr/bias1)
    }
    return p;
}

// What the reducer needs to finish a table row with its Adam update instead of writing the gradient out.
constexpr int kAdamLpF16 = 1, kAdamLpBF16 = 2;   // naf_dtype codes of a 16-bit shadow table (checked where naf_hip.h is visible)

struct AdamTail {
    float *param, *m, *v;          // fp32 master table and its moments, [rows, C] like the table
    void *lp;                      // 16-bit shadow of the table (what the gathers read) or nullptr
    int lp_dtype;                  // NAF_F16 / NAF_BF16 when lp != nullptr
    const uint32_t *overflow;      // [1 + level]: contributions of that level that pass 1 added to the gradient table with atomics
    AdamArgs a;
};

// torch computes the bias corrections in double on the host (torch/optim/adam.py _single_tensor_adam)
AdamArgs make_adam_args(float lr, float beta1, float beta2, float eps, uint32_t step, float grad_scale);
// adam_kernel over n elements (16-byte aligned buffers): the launch behind naf_adam_step
int launch_adam(float *param, float *exp_avg, float *exp_avg_sq, float *grad, void *param_lp, int lp_dtype, uint64_t n,
                const AdamArgs &a, bool zero_grad, hipStream_t s);

}  // namespace naf
