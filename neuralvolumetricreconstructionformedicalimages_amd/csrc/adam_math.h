// adam_math.h -- the Adam update of one element, shared by adam_kernel (adam.hip) and by the tail of the gradient
// reducer (scatter_binned.h), which applies it to the table rows it has just finished (naf_render_train_adam): both
// compile the same expression with -ffp-contract=off, so the two routes give the same bits.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace naf {

struct AdamArgs {
    float lr, beta1, beta2, eps, bias1, bias2_sqrt, grad_scale;
    float step_size, inv_bias2_sqrt;      // lr / bias1 and 1 / bias2_sqrt (host, make_adam_args): the operands of the kFast form
};

// kFast = false (fp32 tables: the parity mode): torch's operation sequence with correctly rounded sqrt and divisions -- the optimiser
// follows torch to the rounding there.
// kFast = true (tables that keep a 16-bit shadow: the value every kernel reads is the update rounded to 8 or 11 significant bits, so
// an ulp of the fp32 master is three orders of magnitude below what the step can resolve anyway): the same update with the hardware
// square root and reciprocal (1 ulp each) and fused multiply-adds -- 11 vector instructions per element instead of ~45.  The
// reducer's Adam tail is issue-bound at the reference's batch (17 M of its 26 M vector instructions per launch were this function,
// profiles/round3_wave_state.md).  Round 3 measured this form once and dropped it because it moved the FIRST crossing of 35 dB of a
// curve that oscillates by +-1.5 dB there; on the reference's full schedule (bench.py full_schedule) the final volume PSNR is the
// figure that counts, and it does not move (profiles/round4_*).
template <bool kFast = false>
__device__ __forceinline__ float adam_one(float &p, float &m, float &v, float g, const AdamArgs &a) {
    g *= a.grad_scale;
    if constexpr (kFast) {
        m = __builtin_fmaf(g - m, 1.0f - a.beta1, m);
        v = __builtin_fmaf((1.0f - a.beta2) * g, g, v * a.beta2);
        const float denom = __builtin_fmaf(__builtin_amdgcn_sqrtf(v), a.inv_bias2_sqrt, a.eps);
        p = __builtin_fmaf(-a.step_size * m, __builtin_amdgcn_rcpf(denom), p);
    } else {
        m = m + (g - m) * (1.0f - a.beta1);                    // torch: exp_avg.lerp_(grad, 1-beta1)
        v = v * a.beta2 + (1.0f - a.beta2) * g * g;            // torch: exp_avg_sq.mul_(b2).addcmul_(g,g,1-b2)
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
