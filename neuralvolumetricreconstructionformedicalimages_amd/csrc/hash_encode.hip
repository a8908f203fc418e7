// hash_encode.hip -- multi-resolution hash-grid encoder for gfx950 (MI355X), stand-alone operator form.
//
// Replaces reference src/encoder/hashencoder/src/hashencoder.cu (host entries :373-428, launch wrappers :301-369)
// behind the C ABI of include/naf_hip.h.  The kernels themselves are in hash_kernels.h.
#include <algorithm>

#include "naf_host.h"
#include "hash_kernels.h"

namespace naf {

// hashencoder.cu:275-298: grad_inputs[b,d] += sum_{l,c} grad[b,l,c] * dy_dx[b,l,d,c]
template <typename T, uint32_t D, uint32_t C>
__global__ void __launch_bounds__(256)
input_backward_kernel(const typename T::store_t *__restrict__ grad, const typename T::store_t *__restrict__ dy_dx,
                      float *__restrict__ grad_inputs, uint32_t B, uint32_t L, bool blc_layout) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * D) return;
    const uint32_t b = t / D, d = t - b * D;
    float s = grad_inputs[t];
    for (uint32_t l = 0; l < L; ++l) {
        float g[C], j[C];
        load_vec<T, C>(blc_layout ? grad + ((size_t)b * L + l) * C : grad + ((size_t)l * B + b) * C, g);
        load_vec<T, C>(dy_dx + (((size_t)b * L + l) * D + d) * C, j);
#pragma unroll
        for (uint32_t ch = 0; ch < C; ++ch) s += g[ch] * j[ch];
    }
    grad_inputs[t] = s;
}

template <typename T, uint32_t D, uint32_t C>
static int launch_forward(const float *inputs, const void *emb, const int32_t *offsets, void *out, uint32_t B, uint32_t L,
                          uint32_t H, bool blc, void *dy_dx, int jac_mode, hipStream_t s) {
    using S = typename T::store_t;
    { ProfScope prof_("hash_forward_kernel", s); hipLaunchKernelGGL((hash_forward_kernel<T, D, C, SrcUnit<D>>), dim3(hash_grid_x(B), L), dim3(256), 0, s,
                       SrcUnit<D>{inputs}, (const S *)emb, offsets, (S *)out, B, L, H, blc, (S *)dy_dx, jac_mode); }
    return check_launch("hash_forward_kernel");
}

template <typename T, uint32_t D, uint32_t C>
static int launch_backward(const void *grad, const float *inputs, const int32_t *offsets, float *gtab, uint32_t B,
                           uint32_t L, uint32_t H, bool blc, const void *dy_dx, float *ginp, hipStream_t s) {
    using S = typename T::store_t;
    { ProfScope prof_("hash_backward_kernel", s); hipLaunchKernelGGL((hash_backward_kernel<T, D, C, SrcUnit<D>>), dim3(hash_grid_x(B), L), dim3(256), 0, s,
                       SrcUnit<D>{inputs}, (const S *)grad, offsets, gtab, B, L, H, blc); }
    int rc = check_launch("hash_backward_kernel");
    if (rc != NAF_OK || dy_dx == nullptr) return rc;
    { ProfScope prof_("input_backward_kernel", s); hipLaunchKernelGGL((input_backward_kernel<T, D, C>), dim3(((uint64_t)B * D + 255) / 256), dim3(256), 0, s,
                       (const S *)grad, (const S *)dy_dx, ginp, B, L, blc); }
    return check_launch("input_backward_kernel");
}

#define NAF_DISPATCH_DC(T, FN, ...)                                                        \
    switch (D * 16 + C) {                                                                  \
        case 2 * 16 + 1: return FN<T, 2, 1>(__VA_ARGS__);                                  \
        case 2 * 16 + 2: return FN<T, 2, 2>(__VA_ARGS__);                                  \
        case 2 * 16 + 4: return FN<T, 2, 4>(__VA_ARGS__);                                  \
        case 2 * 16 + 8: return FN<T, 2, 8>(__VA_ARGS__);                                  \
        case 3 * 16 + 1: return FN<T, 3, 1>(__VA_ARGS__);                                  \
        case 3 * 16 + 2: return FN<T, 3, 2>(__VA_ARGS__);                                  \
        case 3 * 16 + 4: return FN<T, 3, 4>(__VA_ARGS__);                                  \
        case 3 * 16 + 8: return FN<T, 3, 8>(__VA_ARGS__);                                  \
        default: break;                                                                    \
    }

}  // namespace naf

using namespace naf;

static int check_dims(uint32_t D, uint32_t C) {
    // the reference throws this text for unsupported C *and* D (hashencoder.cu:310,324)
    if (!(D == 2 || D == 3) || !(C == 1 || C == 2 || C == 4 || C == 8))
        return fail(NAF_ERR_UNSUPPORTED, "GridEncoding: C must be 1, 2, 4, or 8.");
    return NAF_OK;
}

extern "C" int naf_hash_encode_forward(const float *inputs, const void *embeddings, const int32_t *offsets,
                                       void *outputs, uint32_t B, uint32_t D, uint32_t C, uint32_t L, uint32_t H,
                                       int calc_grad_inputs, void *dy_dx, int dtype, int out_layout, void *stream) {
    if (B != 0 && (!inputs || !embeddings || !offsets || !outputs)) return fail(NAF_ERR_INVALID_ARGUMENT, "hash_encode_forward: null pointer");
    if (calc_grad_inputs < NAF_GRAD_INPUTS_NONE || calc_grad_inputs > NAF_GRAD_INPUTS_REFERENCE)
        return fail(NAF_ERR_INVALID_ARGUMENT, "hash_encode_forward: calc_grad_inputs must be NAF_GRAD_INPUTS_NONE, _EXACT or _REFERENCE");
    if (calc_grad_inputs && !dy_dx) return fail(NAF_ERR_INVALID_ARGUMENT, "hash_encode_forward: calc_grad_inputs without dy_dx");
    if (L == 0 || L > 65535u) return fail(NAF_ERR_INVALID_ARGUMENT, "hash_encode_forward: L must be in [1, 65535]");
    if (int rc = check_dims(D, C)) return rc;
    if (B == 0) return NAF_OK;
    const bool blc = out_layout == NAF_LAYOUT_BLC;
    void *jac = calc_grad_inputs ? dy_dx : nullptr;
    hipStream_t s = (hipStream_t)stream;
    switch (dtype) {
        case NAF_F32: NAF_DISPATCH_DC(F32, launch_forward, inputs, embeddings, offsets, outputs, B, L, H, blc, jac, calc_grad_inputs, s); break;
        case NAF_F16: NAF_DISPATCH_DC(F16, launch_forward, inputs, embeddings, offsets, outputs, B, L, H, blc, jac, calc_grad_inputs, s); break;
        case NAF_BF16: NAF_DISPATCH_DC(BF16, launch_forward, inputs, embeddings, offsets, outputs, B, L, H, blc, jac, calc_grad_inputs, s); break;
        default: break;
    }
    return fail(NAF_ERR_UNSUPPORTED, "hash_encode_forward: dtype must be NAF_F32, NAF_F16 or NAF_BF16");
}

extern "C" int naf_hash_encode_backward(const void *grad, const float *inputs, const void *embeddings,
                                        const int32_t *offsets, float *grad_embeddings, uint32_t B, uint32_t D,
                                        uint32_t C, uint32_t L, uint32_t H, int calc_grad_inputs, const void *dy_dx,
                                        float *grad_inputs, int dtype, int grad_layout, void *stream) {
    (void)embeddings;   // kept for signature parity with hashencoder.h:14; the scatter does not read the table
    if (B != 0 && (!grad || !inputs || !offsets || !grad_embeddings)) return fail(NAF_ERR_INVALID_ARGUMENT, "hash_encode_backward: null pointer");
    if (calc_grad_inputs && (!dy_dx || !grad_inputs)) return fail(NAF_ERR_INVALID_ARGUMENT, "hash_encode_backward: calc_grad_inputs without dy_dx/grad_inputs");
    if (L == 0 || L > 65535u) return fail(NAF_ERR_INVALID_ARGUMENT, "hash_encode_backward: L must be in [1, 65535]");
    if (int rc = check_dims(D, C)) return rc;
    if (B == 0) return NAF_OK;
    const bool blc = grad_layout == NAF_LAYOUT_BLC;
    const void *jac = calc_grad_inputs ? dy_dx : nullptr;
    hipStream_t s = (hipStream_t)stream;
    switch (dtype) {
        case NAF_F32: NAF_DISPATCH_DC(F32, launch_backward, grad, inputs, offsets, grad_embeddings, B, L, H, blc, jac, grad_inputs, s); break;
        case NAF_F16: NAF_DISPATCH_DC(F16, launch_backward, grad, inputs, offsets, grad_embeddings, B, L, H, blc, jac, grad_inputs, s); break;
        case NAF_BF16: NAF_DISPATCH_DC(BF16, launch_backward, grad, inputs, offsets, grad_embeddings, B, L, H, blc, jac, grad_inputs, s); break;
        default: break;
    }
    return fail(NAF_ERR_UNSUPPORTED, "hash_encode_backward: dtype must be NAF_F32, NAF_F16 or NAF_BF16");
}
