// hash_encode.hip -- multi-resolution hash-grid encoder for gfx950 (MI355X), stand-alone operator form.
//
// Replaces reference src/encoder/hashencoder/src/hashencoder.cu (host entries :373-428, launch wrappers :301-369)
// behind the C ABI of include/naf_hip.h.  The kernels themselves are in hash_kernels.h.
#include <algorithm>

#include "naf_host.h"
#include "hash_kernels.h"
#include "encode_kernel.h"
#include "scatter_host.h"

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
    // Three dimensions, no input gradients -- what NAF asks of the operator (hashgrid.py:131, calc_grad_inputs False): the encoder of the
    // training path (encode_kernel.h: 16-byte window gathers, several points per lane, XCD groups for small batches), bit-identical to
    // the one-point-per-lane kernel below, which keeps the dy_dx variants and D = 2.
    if constexpr (D == 3) {
        if (dy_dx == nullptr) return launch_encode<T, T, C, SrcUnit<3>>(SrcUnit<3>{inputs}, emb, offsets, out, B, H, L, 0u, s, 0u, ~0u, 0u, blc);
    }
    { ProfScope prof_("hash_forward_kernel", s); hipLaunchKernelGGL((hash_forward_kernel<T, D, C, SrcUnit<D>>), dim3(hash_grid_x(B), L), dim3(256), 0, s,
                       SrcUnit<D>{inputs}, (const S *)emb, offsets, (S *)out, B, L, H, blc, (S *)dy_dx, jac_mode); }
    return check_launch("hash_forward_kernel");
}

template <typename T, uint32_t D, uint32_t C>
static int launch_input_backward(const void *grad, const void *dy_dx, float *ginp, uint32_t B, uint32_t L, bool blc, hipStream_t s) {
    using S = typename T::store_t;
    { ProfScope prof_("input_backward_kernel", s); hipLaunchKernelGGL((input_backward_kernel<T, D, C>), dim3(((uint64_t)B * D + 255) / 256), dim3(256), 0, s,
                       (const S *)grad, (const S *)dy_dx, ginp, B, L, blc); }
    return check_launch("input_backward_kernel");
}

template <typename T, uint32_t D, uint32_t C>
static int launch_backward(const void *grad, const float *inputs, const int32_t *offsets, float *gtab, uint32_t B,
                           uint32_t L, uint32_t H, bool blc, const void *dy_dx, float *ginp, hipStream_t s) {
    using S = typename T::store_t;
    { ProfScope prof_("hash_backward_kernel", s); hipLaunchKernelGGL((hash_backward_kernel<T, D, C, SrcUnit<D>>), dim3(hash_grid_x(B), L), dim3(256), 0, s,
                       SrcUnit<D>{inputs}, (const S *)grad, offsets, gtab, B, L, H, blc); }
    int rc = check_launch("hash_backward_kernel");
    if (rc != NAF_OK || dy_dx == nullptr) return rc;
    return launch_input_backward<T, D, C>(grad, dy_dx, ginp, B, L, blc, s);
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

// ---- backward with a workspace: the binned scatter of the training path behind the operator's signature --------------------------
// One global float atomic per corner and channel (hash_backward_kernel, the reference's scheme: hashencoder.cu:257-269) runs at the
// memory side's request rate on MI355X -- 50.7 of 52.9 ms of a 16 384-ray step.  A caller that lends a workspace gets
// the two-pass scatter instead (scatter_binned.h): grad_embeddings += the same sums, formed in a fixed order.
template <typename T>
__global__ void __launch_bounds__(256)
grad_absmax_kernel(const typename T::store_t *__restrict__ grad, uint64_t n, uint32_t *__restrict__ gmax_bits) {
    // bit pattern of max |grad| (non-negative floats order like uints; Inf / NaN win and poison the reducer's sums)
    uint32_t m = 0u;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        m = max(m, __float_as_uint(Conv<T>::load(grad + i)) & 0x7fffffffu);
#pragma unroll
    for (uint32_t off = 32u; off != 0u; off >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, (int)off));
    __shared__ uint32_t part[4];
    if ((threadIdx.x & 63u) == 0u) part[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0u) {
        m = max(max(part[0], part[1]), max(part[2], part[3]));
        if (m != 0u) atomicMax(gmax_bits, m);
    }
}

// the plan of the stand-alone scatter: the training path's, asked for 12-byte / fp32 pair records (the 8-byte form of scatter_v2.h
// assumes range-checked coordinates) and a record buffer of at most kWsBudgetBytes per pass
constexpr size_t kWsBudgetBytes = (size_t)8 << 30;
static bool ws_plan(uint32_t B, uint32_t C, uint32_t L, uint32_t log2T, int dtype, naf_render_cfg *cfg, BinPlan *plan) {
    *cfg = naf_render_cfg{};
    cfg->C = C; cfg->L = L; cfg->log2_hashmap_size = log2T; cfg->scatter_mode = NAF_SCATTER_BINNED;
    cfg->mlp_precision = dtype == NAF_F32 ? NAF_F32 : NAF_BF16;
    cfg->flags = NAF_CFG_SCATTER_PAIR12;
    if (!(C == 2u || C == 4u) || !make_bin_plan(cfg, B, plan)) return false;
    const size_t per_level = (size_t)plan->n_tiles * plan->slots * record_bytes(cfg);
    plan->levels_per_pass = (uint32_t)std::min<size_t>(plan->levels_per_pass, std::max<size_t>(1, kWsBudgetBytes / per_level));
    return true;
}
struct WsLayout { unsigned char *regions; uint32_t *counts, *overflow, *gmax; size_t bytes; };
static WsLayout ws_carve(void *base, const naf_render_cfg *cfg, const BinPlan &plan) {
    const size_t n_runs = ((size_t)plan.levels_per_pass << plan.log2_nb) * plan.n_tiles;
    const size_t block_bytes = ((size_t)plan.levels_per_pass * plan.n_tiles * plan.slots * record_bytes(cfg) + 255) & ~(size_t)255;
    WsLayout w;
    w.regions = (unsigned char *)base;
    w.counts = (uint32_t *)(w.regions + block_bytes);
    w.overflow = w.counts + n_runs;                              // [0] total, [1 + level] per level
    w.gmax = w.overflow + 33;
    w.bytes = block_bytes + (((n_runs + 33 + 1) * 4 + 255) & ~(size_t)255);
    return w;
}

template <typename T, uint32_t C>
static int launch_backward_ws(const void *grad, const float *inputs, const int32_t *offsets, float *gtab, uint32_t B, uint32_t L, uint32_t H,
                              bool blc, const naf_render_cfg *cfg, const BinPlan &plan, const WsLayout &w, hipStream_t s) {
    using S = typename T::store_t;
    using Rec = typename std::conditional<std::is_same<T, F32>::value, PairF32<C>, PairBF16<C>>::type;
    constexpr uint32_t NT = BinShape<Rec>::kThreads, PTS = BinShape<Rec>::kPoints, LV = 4u;
    constexpr bool kHasBig = sizeof(Rec) <= 12;
    const bool big = kHasBig && plan.tile_points == 2u * NT * PTS;
    if (!big && plan.tile_points != NT * PTS) return fail(NAF_ERR_LAUNCH, "hash_encode_backward_ws: plan / kernel tile mismatch");
    const SrcUnit<3> src{inputs, B};
    if (hipMemsetAsync(w.overflow, 0, 34 * sizeof(uint32_t), s) != hipSuccess) return fail(NAF_ERR_LAUNCH, "hash_encode_backward_ws: memset failed");
    const uint64_t n = (uint64_t)B * L * C;
    { ProfScope prof_("grad_absmax_kernel", s); hipLaunchKernelGGL((grad_absmax_kernel<T>), dim3((uint32_t)std::min<uint64_t>((n + 255) / 256, 4096)), dim3(256), 0, s,
                       (const S *)grad, n, w.gmax); }
    auto bin = scatter_bin_kernel<T, C, SrcUnit<3>, Rec, NT, PTS, LV>;
    if constexpr (kHasBig) { if (big) bin = scatter_bin_kernel<T, C, SrcUnit<3>, Rec, 2u * NT, PTS, LV>; }
    auto red = scatter_reduce_kernel<C, Rec, false>;
    const uint32_t NB = 1u << plan.log2_nb, threads = big ? 2u * NT : NT;
    const uint32_t red_lds = plan.max_local_rows * C * 8u, bin_lds = (2u * NB + 4u) * 4u + plan.slots * (uint32_t)sizeof(Rec);
    if (hipFuncSetAttribute((const void *)red, hipFuncAttributeMaxDynamicSharedMemorySize, (int)red_lds) != hipSuccess ||
        hipFuncSetAttribute((const void *)bin, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bin_lds) != hipSuccess)
        return fail(NAF_ERR_LAUNCH, "hash_encode_backward_ws: cannot raise the dynamic LDS limit");
    const uint32_t sl = blc ? 1u : B, sb = blc ? L : 1u;
    for (uint32_t l0 = 0; l0 < L; l0 += plan.levels_per_pass) {
        const uint32_t nl = std::min(plan.levels_per_pass, L - l0);
        { ProfScope prof_("scatter_bin_kernel", s); hipLaunchKernelGGL(bin, dim3(plan.n_tiles, (nl + LV - 1u) / LV), dim3(threads), bin_lds, s, src, (const S *)grad, offsets, gtab,
                           (Rec *)w.regions, w.counts, w.overflow, B, H, l0, nl, plan, SlabReduce{}, sl, sb); }
        if (int rc = check_launch("scatter_bin_kernel")) return rc;
        { ProfScope prof_("scatter_reduce_kernel", s); hipLaunchKernelGGL(red, dim3(NB, nl, reducer_split(NB, nl)), dim3(1024), red_lds, s, (const Rec *)w.regions, w.counts, offsets, gtab,
                           w.gmax, l0, 0u, H, plan, AdamTail{}); }
        if (int rc = check_launch("scatter_reduce_kernel")) return rc;
    }
    (void)cfg;
    return NAF_OK;
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

extern "C" size_t naf_hash_encode_workspace_bytes(uint32_t B, uint32_t D, uint32_t C, uint32_t L, uint32_t log2_hashmap_size, int dtype) {
    naf_render_cfg cfg;
    BinPlan plan;
    if (D != 3u || B < kBinMinPoints || L == 0u || L > 32u || !ws_plan(B, C, L, log2_hashmap_size, dtype, &cfg, &plan)) return 0;
    return ws_carve(nullptr, &cfg, plan).bytes;
}

extern "C" int naf_hash_encode_backward_ws(const void *grad, const float *inputs, const void *embeddings, const int32_t *offsets,
                                           float *grad_embeddings, uint32_t B, uint32_t D, uint32_t C, uint32_t L, uint32_t H,
                                           int calc_grad_inputs, const void *dy_dx, float *grad_inputs, int dtype, int grad_layout,
                                           uint32_t log2_hashmap_size, void *workspace, size_t workspace_bytes, void *stream) {
    naf_render_cfg cfg;
    BinPlan plan;
    const size_t need = naf_hash_encode_workspace_bytes(B, D, C, L, log2_hashmap_size, dtype);
    // shapes the binned scatter does not cover (D = 2, C = 1 or 8, small batches), or no workspace: the atomic scatter
    if (need == 0 || workspace == nullptr || workspace_bytes < need || !ws_plan(B, C, L, log2_hashmap_size, dtype, &cfg, &plan))
        return naf_hash_encode_backward(grad, inputs, embeddings, offsets, grad_embeddings, B, D, C, L, H, calc_grad_inputs, dy_dx, grad_inputs,
                                        dtype, grad_layout, stream);
    if (!grad || !inputs || !offsets || !grad_embeddings) return fail(NAF_ERR_INVALID_ARGUMENT, "hash_encode_backward_ws: null pointer");
    if (calc_grad_inputs && (!dy_dx || !grad_inputs)) return fail(NAF_ERR_INVALID_ARGUMENT, "hash_encode_backward_ws: calc_grad_inputs without dy_dx/grad_inputs");
    if (((uintptr_t)workspace & 255u) != 0u) return fail(NAF_ERR_INVALID_ARGUMENT, "hash_encode_backward_ws: the workspace must be 256-byte aligned");
    const bool blc = grad_layout == NAF_LAYOUT_BLC;
    hipStream_t s = (hipStream_t)stream;
    const WsLayout w = ws_carve(workspace, &cfg, plan);
    int rc = NAF_ERR_UNSUPPORTED;
    switch (dtype * 16 + (int)C) {
        case NAF_F32 * 16 + 2: rc = launch_backward_ws<F32, 2>(grad, inputs, offsets, grad_embeddings, B, L, H, blc, &cfg, plan, w, s); break;
        case NAF_F32 * 16 + 4: rc = launch_backward_ws<F32, 4>(grad, inputs, offsets, grad_embeddings, B, L, H, blc, &cfg, plan, w, s); break;
        case NAF_F16 * 16 + 2: rc = launch_backward_ws<F16, 2>(grad, inputs, offsets, grad_embeddings, B, L, H, blc, &cfg, plan, w, s); break;
        case NAF_F16 * 16 + 4: rc = launch_backward_ws<F16, 4>(grad, inputs, offsets, grad_embeddings, B, L, H, blc, &cfg, plan, w, s); break;
        case NAF_BF16 * 16 + 2: rc = launch_backward_ws<BF16, 2>(grad, inputs, offsets, grad_embeddings, B, L, H, blc, &cfg, plan, w, s); break;
        case NAF_BF16 * 16 + 4: rc = launch_backward_ws<BF16, 4>(grad, inputs, offsets, grad_embeddings, B, L, H, blc, &cfg, plan, w, s); break;
        default: return fail(NAF_ERR_UNSUPPORTED, "hash_encode_backward_ws: dtype must be NAF_F32, NAF_F16 or NAF_BF16");
    }
    if (rc != NAF_OK || !calc_grad_inputs) return rc;
    switch (dtype) {                                              // the input gradient does not depend on how the table gradient was formed
        case NAF_F32: NAF_DISPATCH_DC(F32, launch_input_backward, grad, dy_dx, grad_inputs, B, L, blc, s); break;
        case NAF_F16: NAF_DISPATCH_DC(F16, launch_input_backward, grad, dy_dx, grad_inputs, B, L, blc, s); break;
        default: NAF_DISPATCH_DC(BF16, launch_input_backward, grad, dy_dx, grad_inputs, B, L, blc, s); break;
    }
    return fail(NAF_ERR_UNSUPPORTED, "hash_encode_backward_ws: unsupported D / C");
}
