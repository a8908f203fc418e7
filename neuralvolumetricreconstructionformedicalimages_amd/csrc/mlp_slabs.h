// mlp_slabs.h -- the weight-gradient slabs of the MLP backward kernels and their reduction (gfx950).
//
// Every workgroup of mlp_backward_kernel / mlp16_backward_kernel writes one slab: its sums of the 4 225 weight and bias gradients
// (layout of the parameter block), its share of the loss, and the bit pattern of its largest |feature gradient|.  slab_reduce_block
// adds the slabs up in a FIXED order (deterministic) -- as a launch of its own (mlp_grad_reduce_kernel) or inside spare workgroups
// of scatter_bin_kernel, which follows on the stream and needs none of its results -- and finishes the sums: grad_mlp (+=) or the
// MLP's Adam update, the loss, and the gradient maximum that scales the fixed-point reducer of the binned scatter.
#pragma once

#include "adam_math.h"
#include "field_mlp.h"

namespace naf {

constexpr uint32_t kSlabStride = 4352;
constexpr uint32_t kSlabLoss = kMlpParams;         // slab entry behind the parameter block: the workgroup's share of the loss
// ... and behind that the bit pattern of the workgroup's max |feature gradient| (the fixed-point scale of the binned scatter).  It
// used to be one atomicMax per WAVE on a single word: same-address atomics retire one at a time (~12 ns each, MI355X_MICROARCH.md
// "fanin"), 1 024 .. 3 072 of them were 12 .. 37 us of a 53 us kernel at 1 024 rays.  The slab reduction takes the maximum instead.
constexpr uint32_t kSlabGmax = kMlpParams + 1u;

// ... and the reduction applies the MLP's Adam update to the sums it has just formed (naf_render_train_adam with mlp_param set)
// instead of writing them out for one more launch to read back.
struct MlpAdam { float *param, *m, *v; AdamArgs a; };

enum LossMode : int { kLossNone = 0, kLossAdd = 1, kLossAssign = 2 };       // loss_out[0] untouched / += / = the step's loss

struct SlabReduce {
    const float *slabs;          // nullptr: nothing to do
    uint32_t n_slabs;
    float *grad_mlp, *loss_out;
    int loss_mode;
    MlpAdam adam;                // param == nullptr: grad_mlp += sums
    uint32_t *gmax_out;          // bit pattern of max |feature gradient| of the step (nullptr: not wanted)
};

// One block = kReduceParams consecutive slab entries.  The slabs are split into kReduceGroups interleaved groups (group v adds
// slabs v, v + 32, ... in four independent chains), the 32 partial sums are combined in group order: the result does not depend
// on how many threads do the work (kGroups real groups of 32 threads walk the 32 virtual ones).  Entry kSlabLoss is the loss,
// entry kSlabGmax the gradient maximum (integer maximum of the bit patterns: non-negative floats order like uints, Inf / NaN win).
constexpr uint32_t kReduceParams = 32, kReduceGroups = 32;
constexpr uint32_t kSlabReduceBlocks = (kSlabGmax + 1u + kReduceParams - 1u) / kReduceParams;

template <uint32_t kGroups>
__device__ __forceinline__ void slab_reduce_block(float (*part)[kReduceParams], const SlabReduce &sr, uint32_t block, uint32_t tid) {
    static_assert(kReduceGroups % kGroups == 0u, "real groups must divide the virtual ones");
    const float *__restrict__ slabs = sr.slabs;
    const uint32_t n_slabs = sr.n_slabs;
    const uint32_t j = tid % kReduceParams, g = tid / kReduceParams;
    const uint32_t i = block * kReduceParams + j;
    for (uint32_t vg = g; vg < kReduceGroups; vg += kGroups) {
        if (i == kSlabGmax) {
            uint32_t m = 0u;
            for (uint32_t k = vg; k < n_slabs; k += kReduceGroups) m = max(m, __float_as_uint(slabs[(size_t)k * kSlabStride + i]));
            part[vg][j] = __uint_as_float(m);
        } else {
            float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
            if (i <= kSlabLoss) {
                uint32_t k = vg;
#pragma unroll 2
                for (; k + 3u * kReduceGroups < n_slabs; k += 4u * kReduceGroups) {
                    s0 += slabs[(size_t)(k + 0u * kReduceGroups) * kSlabStride + i];
                    s1 += slabs[(size_t)(k + 1u * kReduceGroups) * kSlabStride + i];
                    s2 += slabs[(size_t)(k + 2u * kReduceGroups) * kSlabStride + i];
                    s3 += slabs[(size_t)(k + 3u * kReduceGroups) * kSlabStride + i];
                }
                for (; k < n_slabs; k += kReduceGroups) s0 += slabs[(size_t)k * kSlabStride + i];
            }
            part[vg][j] = (s0 + s1) + (s2 + s3);
        }
    }
    __syncthreads();
    if (g == 0 && i == kSlabGmax) {
        uint32_t m = 0u;
#pragma unroll
        for (uint32_t q = 0; q < kReduceGroups; ++q) m = max(m, __float_as_uint(part[q][j]));
        if (sr.gmax_out != nullptr) *sr.gmax_out = m;
    } else if (g == 0 && i <= kSlabLoss) {
        float total = 0.0f;
#pragma unroll
        for (uint32_t q = 0; q < kReduceGroups; ++q) total += part[q][j];
        if (i == kSlabLoss) {
            if (sr.loss_mode != kLossNone && sr.loss_out != nullptr) sr.loss_out[0] = sr.loss_mode == kLossAssign ? total : sr.loss_out[0] + total;
        } else if (sr.adam.param != nullptr) {
            const float gsum = sr.grad_mlp[i] + total;       // whatever the caller had accumulated there (+=), as adam_kernel would see it
            float p = sr.adam.param[i], m = sr.adam.m[i], v = sr.adam.v[i];
            adam_one(p, m, v, gsum, sr.adam.a);
            sr.adam.param[i] = p; sr.adam.m[i] = m; sr.adam.v[i] = v;
            sr.grad_mlp[i] = 0.0f;
        } else sr.grad_mlp[i] += total;
    }
}

}  // namespace naf
