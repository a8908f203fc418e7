/*
 * naf_hip.h -- C ABI of libnaf_hip.so, the MI355X (gfx950) native NAF hot path.
 *
 * This is the drop-in boundary for the reference's native operator module `_hash_encoder`
 * (reference: src/encoder/hashencoder/src/bindings.cpp:5-8, hashencoder.h:13-14) plus the fused
 * field / ray-march / optimiser entry points that replace the ATen kernels behind
 * src/render/render.py:82-212, src/network/network.py:34-58 and src/trainer.py:54-58,134-142.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless stated otherwise;
 *   - the caller allocates everything, the callee writes in place (reference ownership model,
 *     hashgrid.py:30-35,59-64); "(+=)" marks buffers the callee accumulates into (caller zeroes them);
 *   - `stream` is a hipStream_t passed as void* (NULL = legacy default stream, which is what the
 *     reference uses, hashencoder.cu:306,336);
 *   - every function returns NAF_OK (0) or a negative naf_status; naf_last_error() returns a
 *     thread-local human readable message for the last failure on the calling thread;
 *   - launches are asynchronous; no function synchronises or allocates device memory;
 *   - the library keeps NO mutable process-wide state besides the opt-in profiler (naf_profile_*): everything a call
 *     depends on is in its arguments, so it is re-entrant across threads, streams and devices.
 */
#ifndef NAF_HIP_H
#define NAF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum naf_status {
    NAF_OK = 0,
    NAF_ERR_INVALID_ARGUMENT = -1, /* null pointer, zero size, misaligned pointer ...            */
    NAF_ERR_UNSUPPORTED = -2,      /* reference: std::runtime_error("GridEncoding: C must be 1, 2, 4, or 8.") hashencoder.cu:310,324 */
    NAF_ERR_LAUNCH = -3            /* hipGetLastError() != hipSuccess after the launch            */
} naf_status;

/* storage type of tables / features; arithmetic is always fp32 (hashencoder.cu:107-143) */
typedef enum naf_dtype { NAF_F32 = 0, NAF_F16 = 1, NAF_BF16 = 2 } naf_dtype;

/* layout of per-point feature tensors */
typedef enum naf_layout {
    NAF_LAYOUT_LBC = 0, /* [L, B, C]  level-major, what kernel_grid writes (hashencoder.cu:95)        */
    NAF_LAYOUT_BLC = 1  /* [B, L*C]   what _hash_encode.forward returns after permute (hashgrid.py:40) */
} naf_layout;

const char *naf_last_error(void);
int naf_abi_version(void);

/* Optional per-kernel timing for bench.py: when enabled, every kernel launched through this library is bracketed
 * by a pair of HIP events on ITS launch stream.  naf_profile_collect() synchronises them and writes one text line
 * per kernel ("<kernel> <launches> <total_ms>\n") into `buf`, then clears the log.  Up to 8192 launches are kept. */
int naf_profile_enable(int on);
int naf_profile_collect(char *buf, size_t buflen);

/* ------------------------------------------------------------------------------------------------
 * E6  hash_encode_forward   (replaces hashencoder.cu:373-396 / hashencoder.h:13)
 *   inputs      f32  [B, D]   in [0,1]
 *   embeddings  dtype [sum_l T_l, C]
 *   offsets     i32  [L+1]    (device)
 *   outputs     dtype, layout `out_layout`
 *   dy_dx       dtype [B, L, D, C]  written iff calc_grad_inputs != 0 (may be NULL otherwise)
 * D in {2,3}, C in {1,2,4,8} else NAF_ERR_UNSUPPORTED.
 * calc_grad_inputs: NAF_GRAD_INPUTS_NONE, NAF_GRAD_INPUTS_EXACT (d feature / d x for x in [0,1], including the level
 * scale 2^l*H-1 that the chain rule needs), or NAF_GRAD_INPUTS_REFERENCE (what hashencoder.cu:153-197 stores: the scale
 * factor is commented out there, :164-165, and the loop picks the interpolated dimensions with `nd > gd`, :170, so for
 * gd < D-1 one coordinate is never set -- the reference reads an uninitialised register there; this mode uses the base
 * corner for it, the only defined reading).
 */
#define NAF_GRAD_INPUTS_NONE 0
#define NAF_GRAD_INPUTS_EXACT 1
#define NAF_GRAD_INPUTS_REFERENCE 2
int naf_hash_encode_forward(const float *inputs, const void *embeddings, const int32_t *offsets, void *outputs,
                            uint32_t B, uint32_t D, uint32_t C, uint32_t L, uint32_t H, int calc_grad_inputs,
                            void *dy_dx, int dtype, int out_layout, void *stream);

/* E7/E8  hash_encode_backward  (replaces hashencoder.cu:398-428 / hashencoder.h:14)
 *   grad             dtype, layout `grad_layout`
 *   grad_embeddings  f32 [sum_l T_l, C]  (+=)   -- always fp32 (the reference accumulates in scalar_t)
 *   grad_inputs      f32 [B, D]          (+=)   iff calc_grad_inputs != 0
 */
int naf_hash_encode_backward(const void *grad, const float *inputs, const void *embeddings, const int32_t *offsets,
                             float *grad_embeddings, uint32_t B, uint32_t D, uint32_t C, uint32_t L, uint32_t H,
                             int calc_grad_inputs, const void *dy_dx, float *grad_inputs, int dtype, int grad_layout,
                             void *stream);

/* E7 with a workspace.  The reference scheme above -- one global float atomic per corner and channel (hashencoder.cu:257-269) --
 * runs at the memory side's request rate on MI355X (50.7 of 52.9 ms of a 16 384-ray step).  A caller that lends a
 * workspace gets the two-pass binned scatter of the training path instead: same contract (grad_embeddings += the same sums,
 * fp32; grad_inputs as above), the sums formed in a fixed order (bit-reproducible), no allocation inside the library.
 *   naf_hash_encode_workspace_bytes: bytes `naf_hash_encode_backward_ws` needs for this shape, or 0 when the shape is one the
 *       binned scatter does not cover (D = 2, C = 1 or 8, fewer than 2^13 points): then -- and whenever `workspace` is NULL or too
 *       small -- naf_hash_encode_backward_ws IS naf_hash_encode_backward.
 *   log2_hashmap_size: log2 of the largest level (encoder hyper-parameter, hashgrid.py:96); the levels described by `offsets`
 *       must not exceed it.   workspace: device memory, 256-byte aligned, contents undefined before and after the call.       */
size_t naf_hash_encode_workspace_bytes(uint32_t B, uint32_t D, uint32_t C, uint32_t L, uint32_t log2_hashmap_size, int dtype);
int naf_hash_encode_backward_ws(const void *grad, const float *inputs, const void *embeddings, const int32_t *offsets,
                                float *grad_embeddings, uint32_t B, uint32_t D, uint32_t C, uint32_t L, uint32_t H,
                                int calc_grad_inputs, const void *dy_dx, float *grad_inputs, int dtype, int grad_layout,
                                uint32_t log2_hashmap_size, void *workspace, size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------------------------------
 * R2  stratified sampling along rays  (replaces the ATen ops of render.py:87-105)
 *   rays    f32 [n_rays, 8]  = origin(3) direction(3) near far   (tigre.py:248-255)
 *   t_rand  f32 [n_rays, S]  jitter in [0,1) or NULL; with NULL and perturb != 0 the jitter is the
 *           counter-based generator naf_jitter(seed, ray, sample) documented in DESIGN.md
 *   z_vals  f32 [n_rays, S]  out
 *   pts     f32 [n_rays, S, 3] out, clamped to +-(bound - 1e-6); NULL: depths only
 */
int naf_sample_rays(const float *rays, const float *t_rand, float *z_vals, float *pts, uint32_t n_rays,
                    uint32_t n_samples, int perturb, float bound, uint64_t seed, uint32_t ray_index_base,
                    void *stream);

/* R5  coarse -> fine resampling in one pass (replaces raw2outputs' weights + sample_pdf + sort, render.py:113-126,203-247):
 *   sigma        f32 [n_rays, S]        coarse network output per sample (naf_render_forward_samples)
 *   t_rand       as in naf_sample_rays: the jitter the COARSE pass used (its depths are recomputed here)
 *   u            f32 [n_rays, n_fine]   uniforms of the inverse-transform sampling, NULL with det != 0 (evenly spaced
 *                                       quantiles, render.py:228) or to use the counter-based generator
 *   z_out        f32 [n_rays, S+n_fine] the coarse depths and the new samples, sorted ascending (render.py:123)
 *   weights_out  f32 [n_rays, S]        optional: the normalised weights ("weights0" of the render dict); the division is by
 *                                       the maximum over the WHOLE call, like the reference's chunk
 *   scratch      >= 4 bytes of device memory
 * One wave per ray: the cdf is an inclusive wave prefix sum, the merge a bitonic sort in LDS.  S <= 1024, S + n_fine <= 2048. */
int naf_fine_depths(const float *rays, const float *t_rand, const float *sigma, const float *u, float *z_out, float *weights_out,
                    uint32_t n_rays, uint32_t n_samples, uint32_t n_fine, int perturb, int det, uint64_t seed,
                    uint32_t ray_index_base, void *scratch, void *stream);

/* G3/G4  ray generation (replaces the precomputed rays[N,H,W,8] of tigre.py:247-255,402-456,463-528)
 *   poses   f32 [n_projections, 3, 4]  = [R | t] of angle2pose (tigre.py:530-572), cast to fp32 like torch.Tensor(pose)
 *   pixels  i64 [n] flat pixel index  proj*H*W + row*W + col, or NULL for the dense range first_pixel .. first_pixel+n-1
 *   rays    f32 [n, 8] out (16-byte aligned): origin, direction (cone: un-normalised, tigre.py:434-437), near, far
 */
int naf_generate_rays(const float *poses, const int64_t *pixels, int64_t first_pixel, float *rays, uint64_t n,
                      uint32_t n_projections, uint32_t det_w, uint32_t det_h, float du, float dv, float ou, float ov,
                      float DSD, float near, float far, int parallel, void *stream);

/* G6  the data side of a training step (replaces np.random.choice(..., replace=False) + the fancy-indexing gathers of
 * TIGREDataset.__getitem__, tigre.py:354-372): for each of `n_segments` projections, `rays_per_segment` DISTINCT entries of
 * its list of valid pixels (flat indices proj*H*W + row*W + col whose measured value is non-zero) are drawn uniformly at
 * random through a keyed bijection of [0, n_valid) -- draw index i -> valid[perm_seed(i)] -- then the measured value is
 * gathered and the ray generated, all in one launch and without any host round trip.
 *   pixels  i64 [n_draws] out, optional      target  f32 [n_draws] out, optional (= projections[pixel])
 *   rays    f32 [n_draws, 8] out
 * [first_draw, first_draw + n_draws) is a slice of the n_segments * rays_per_segment draws (segment-major): ranks of a
 * data-parallel job that pass the same seed each take their slice of ONE draw.  A list shorter than rays_per_segment is
 * refused with the reference's message ("Cannot take a larger sample than population when 'replace=False'"). */
#define NAF_MAX_DRAW_SEGMENTS 16
typedef struct naf_scan_draw {
    uint32_t n_segments;
    uint32_t rays_per_segment;
    const int64_t *valid[NAF_MAX_DRAW_SEGMENTS];   /* device pointers */
    uint32_t n_valid[NAF_MAX_DRAW_SEGMENTS];
} naf_scan_draw;
int naf_draw_scan_rays(const naf_scan_draw *draw, const float *poses, const float *projections, int64_t *pixels, float *target,
                       float *rays, uint32_t first_draw, uint32_t n_draws, uint32_t n_projections, uint32_t det_w, uint32_t det_h,
                       float du, float dv, float ou, float ov, float DSD, float near, float far, int parallel, uint64_t seed,
                       void *stream);

/* R4  line integral acc = sum_s sigma_s * dist_s  (render.py:192-201), and its backward.
 *   sigma f32 [n_rays, S]; z_vals f32 [n_rays, S]; rays f32 [n_rays, 8]; acc f32 [n_rays]
 *   backward: grad_sigma[r,s] = grad_acc[r] * dist[r,s]
 */
int naf_integrate_forward(const float *sigma, const float *z_vals, const float *rays, float *acc, uint32_t n_rays,
                          uint32_t n_samples, void *stream);
int naf_integrate_backward(const float *grad_acc, const float *z_vals, const float *rays, float *grad_sigma,
                           uint32_t n_rays, uint32_t n_samples, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Fused field + ray march for the canonical NAF network (network.py:6-58 with num_layers=4,
 * hidden_dim=32, skips=[2], out_dim=1, in_dim = L*C = 32; config/NAME.yaml:7-19).
 *
 * MLP parameter block `mlp` (f32, 4225 values, row-major like nn.Linear.weight):
 *   W0[32][32] b0[32] W1[32][32] b1[32] W2[32][64] b2[32] W3[1][32] b3[1]
 * `grad_mlp` (+=) has the same layout.
 */
#define NAF_MLP_PARAMS 4225

typedef struct naf_render_cfg {
    uint32_t n_samples;      /* S (render.py:87)                                                  */
    int32_t perturb;         /* stratified jitter on/off (render.py:94-100)                       */
    float bound;             /* network bound (render.py:103, hashgrid.py:125)                    */
    uint32_t L, C, H;        /* encoder levels / level_dim / base_resolution (hashgrid.py:78-86)  */
    int32_t table_dtype;     /* naf_dtype of `embeddings`                                         */
    int32_t mlp_precision;   /* NAF_F32: exact fp32 MFMA (parity mode); NAF_BF16: bf16 MFMA, fp32 accumulate */
    int32_t last_activation; /* 0 sigmoid, 1 leaky-relu, 2 tanh, 3 none (network.py:23-32)        */
    uint64_t seed;           /* jitter seed when t_rand == NULL                                   */
    uint32_t ray_index_base; /* global index of ray 0 of this call (jitter stream is per global ray) */
    uint32_t log2_hashmap_size; /* hashgrid.py:81 (sizes the binned gradient scatter; 0 = unknown -> plain atomics) */
    int32_t scatter_mode;    /* how naf_render_backward / _train scatter the table gradient (NAF_SCATTER_*); part of the
                                cfg because the workspace layout depends on it                                      */
    uint32_t flags;          /* NAF_CFG_* bits                                                                     */
} naf_render_cfg;

#define NAF_SCATTER_AUTO 0   /* binned two-pass scatter from 2^13 points per call, plain fp32 atomics below          */
#define NAF_SCATTER_ATOMIC 1 /* always fp32 atomics: the reference's scheme (hashencoder.cu:257-269)                 */
#define NAF_SCATTER_BINNED 2 /* always the binned scatter                                                            */

#define NAF_CFG_PER_LEVEL_LAUNCHES 1u /* diagnostics: one launch per level instead of level-major grids, so that
                                         naf_profile_collect() reports per-level times (changes the workspace size) */
#define NAF_CFG_EXPLICIT_DEPTHS 2u    /* the `t_rand` argument of the naf_render_* entry points holds the sample DEPTHS
                                         z[n_rays, S] themselves (the fine pass renders at the merged, sorted depths of
                                         naf_fine_depths, render.py:123-126); `perturb` is ignored                    */
#define NAF_CFG_LEVELS_INTERLEAVED 4u /* diagnostics: the encoder walks all levels of a point tile at once instead of one
                                         level at a time chip-wide -- same results, the cache behaviour of a kernel that
                                         gathers every level of a tile (what a single fused gather+MLP kernel would see) */

#define NAF_CFG_FORWARD_FUSED 8u      /* forward-only entry points (naf_render_forward, _forward_samples, naf_field_forward, _grid):
                                         gathers + MLP + line integral in ONE kernel where the shape allows it (16 levels x 2
                                         channels, bf16 MLP operands); the [L, B, C] features then never reach HBM and the calls
                                         need no workspace (naf_forward_workspace_bytes).  Bit-identical to the two-kernel path. */
#define NAF_CFG_ENCODE_TWO_GATHERS 32u /* diagnostics: the encoder fetches the two x-neighbour corners of a cell with two gathers
                                         at every batch size instead of one 16-byte window (same results; A/B timing only) */
#define NAF_CFG_BACKWARD_ONE_WAVE_PER_SIMD 128u /* diagnostics: the MLP backward splits rays into tile ranges only up to one wave per SIMD (rounds 1-2) instead of three, and the bf16 MLP forward keeps one wave per ray at small batches (no mlp16_forward_split_kernel) */
#define NAF_CFG_ENCODE_LEVEL_MAJOR 256u /* diagnostics: the encoder never splits the XCDs into groups (see NAF_CFG_ENCODE_GROUPS_*)       */
#define NAF_CFG_ENCODE_WINDOWS 64u     /* diagnostics: the 16-byte window form at every batch size (default: below 600 000 points per
                                          call and for fp32 tables; two gathers with four points per lane in flight above)        */
#define NAF_CFG_FUSED_STORE_FEATURES 16u /* diagnostics: the fused kernel also stores the features it computed              */
#define NAF_CFG_ENCODE_GROUPS_2 512u     /* the encoder splits the eight XCDs into 2 groups that take alternate levels, so that a level's
                                            slice of the table is pulled through four L2s instead of eight (encode_kernel); without a
                                            flag the batch size decides: 4 groups below 160 000 points per call, 2 below 500 000    */
#define NAF_CFG_ENCODE_GROUPS_4 1024u    /* ... into 4 groups (levels mod 4); both flags: 8 groups, one level per XCD at a time       */
#define NAF_CFG_MIN_BUCKETS_SHIFT 13u    /* bits 13-14: the binned scatter uses at least 64 << value row buckets per level (default: as few as the
                                            reducer's LDS allows, 64 at T = 2^19).  A level-parallel rank that owns two or four levels asks
                                            for 256 / 128, so that its reducer launch has 512 workgroups that each own their rows (no split
                                            launches, the Adam tail applies) -- naf_levels_scatter                                        */
#define NAF_CFG_MIN_BUCKETS_MASK (3u << NAF_CFG_MIN_BUCKETS_SHIFT)
#define NAF_CFG_SCATTER_PAIR12 2048u     /* diagnostics: the binned scatter of the canonical shape (two bf16 channels) keeps the 12-byte pair
                                            records and the kernels of rounds 2-3 (scatter_binned.h) instead of scatter_v2.h's 8-byte ones */
#define NAF_CFG_LEVELS_GATHER_PASS 32768u /* diagnostics: naf_levels_scatter re-orders the gradient blocks into [level][point][C] with a pass of
                                            its own (rounds 3-4) even where pass 1 of the scatter can read them in place (two bf16
                                            channels) -- same results bit for bit; A/B timing and tests                              */
#define NAF_CFG_TEST_TINY_BLOCKS 4096u   /* tests: the record blocks of pass 1 hold a quarter of a tile's records, so that most
                                            records take the overflow route (counted global atomics) and the reducer's Adam tail has
                                            spilled contributions to fold in                                                        */


/* Diagnostic (synchronous, host result): number of gradient contributions of the LAST binned backward on this
 * workspace that did not fit their bucket stream and were applied with plain atomics instead (still correct). */
int naf_scatter_overflow_count(const naf_render_cfg *cfg, uint64_t n_points, const void *workspace, uint32_t *count_host);

/* The same per level: counts_host[32] (host memory), entry l = records of level l that fell back to atomics. */
int naf_scatter_overflow_levels(const naf_render_cfg *cfg, uint64_t n_points, const void *workspace, uint32_t *counts_host);

/* Workspace size in bytes for naf_render_* / naf_field_forward over `n_points` points (= n_rays * n_samples for the
 * render entry points): feature and feature-gradient tensors [L, n_points, C] plus the MLP-gradient slabs. */
size_t naf_render_workspace_bytes(const naf_render_cfg *cfg, uint64_t n_points);

/* Workspace size in bytes for the FORWARD-ONLY entry points (naf_render_forward, naf_render_forward_samples,
 * naf_field_forward, naf_field_forward_grid) over `n_points` points: the [L, n_points, C] feature tensor and nothing else --
 * 64 B per point in bf16 mode, 128 B in fp32 mode, against ~1.1 KB per point of the training layout -- or 256 bytes when
 * NAF_CFG_FORWARD_FUSED applies.  A workspace of naf_render_workspace_bytes() is always large enough as well. */
size_t naf_forward_workspace_bytes(const naf_render_cfg *cfg, uint64_t n_points);

/* Forward only (eval, train.py:235-239): acc[r] = sum_s sigma(pts[r,s]) * dist[r,s]. */
int naf_render_forward(const float *rays, const float *t_rand, const void *embeddings, const int32_t *offsets,
                       const float *mlp, float *acc, uint32_t n_rays, const naf_render_cfg *cfg, void *workspace,
                       void *stream);

/* Forward with the per-sample quantities the coarse -> fine pass consumes (render.py:113-126, 203-211):
 *   sigma          f32 [n_rays, S]  network output at every sample (NULL: not wanted)
 *   optical_depth  f32 [n_rays, S]  running line integral tau[r,s] = sum_{s' <= s} sigma[r,s'] * dist[r,s'] -- an inclusive
 *                                   wave prefix sum over the samples of a ray; tau[r,S-1] == acc[r] (NULL: not wanted) */
int naf_render_forward_samples(const float *rays, const float *t_rand, const void *embeddings, const int32_t *offsets,
                               const float *mlp, float *acc, float *sigma, float *optical_depth, uint32_t n_rays,
                               const naf_render_cfg *cfg, void *workspace, void *stream);

/* Backward of naf_render_forward for an arbitrary upstream gradient grad_acc[r] = dLoss/dacc[r]
 * (what autograd hands to the renderer): grad_embeddings (+=), grad_mlp (+=).
 * features_valid != 0 promises that `workspace` still holds the features naf_render_forward wrote for the SAME
 * rays / t_rand / parameters; with 0 they are recomputed first.
 */
int naf_render_backward(const float *rays, const float *t_rand, const float *grad_acc, const void *embeddings,
                        const int32_t *offsets, const float *mlp, float *grad_embeddings, float *grad_mlp, uint32_t n_rays,
                        const naf_render_cfg *cfg, void *workspace, int features_valid, void *stream);

/* Training step body (train.py:69-127 + trainer.py:134-142 minus the optimiser):
 *   acc = render(rays); loss = sum_r weight[r] * (acc[r] - target[r])^2  with weight[r] = mask/len
 *   (the caller encodes the reference's masked chunk means in `ray_weight`);
 *   grad_embeddings (+=), grad_mlp (+=), loss_out[0] (+=).
 */
int naf_render_train(const float *rays, const float *t_rand, const float *target, const float *ray_weight,
                     const void *embeddings, const int32_t *offsets, const float *mlp, float *acc,
                     float *grad_embeddings, float *grad_mlp, float *loss_out, uint32_t n_rays,
                     const naf_render_cfg *cfg, void *workspace, void *stream);

/* The same step for data-parallel training (one process per GPU): the levels of the table are scattered in
 * the order of `buckets`, and each bucket's event is recorded on `stream` as soon as its rows of grad_embeddings are
 * final, so the caller can all-reduce that slice (RCCL, on another stream that waits for the event) while the next
 * bucket is still being binned and reduced.  `mlp_ready` is recorded once grad_mlp and loss_out are final, which is
 * before the table scatter starts.  Events are hipEvent_t handles owned by the caller; NULL entries are skipped.
 * The buckets must be disjoint level ranges that cover [0, L).  Numerically identical to naf_render_train. */
#define NAF_MAX_GRAD_BUCKETS 16
typedef struct naf_grad_buckets {
    uint32_t n_buckets;
    uint32_t level_begin[NAF_MAX_GRAD_BUCKETS];
    uint32_t level_end[NAF_MAX_GRAD_BUCKETS];
    void *ready[NAF_MAX_GRAD_BUCKETS];
    void *mlp_ready;
} naf_grad_buckets;
int naf_render_train_bucketed(const float *rays, const float *t_rand, const float *target, const float *ray_weight,
                              const void *embeddings, const int32_t *offsets, const float *mlp, float *acc,
                              float *grad_embeddings, float *grad_mlp, float *loss_out, uint32_t n_rays,
                              const naf_render_cfg *cfg, void *workspace, const naf_grad_buckets *buckets, void *stream);

/* Field query sigma(x) for a point list (volume query, train.py:246-250): pts f32 [B,3] in [-bound,bound]. */
int naf_field_forward(const float *pts, const void *embeddings, const int32_t *offsets, const float *mlp,
                      float *sigma, uint32_t B, const naf_render_cfg *cfg, void *workspace, void *stream);

/* The same on a regular grid that is generated instead of read (the volume query of train.py:246-250 on the voxel grid
 * of tigre.py:388-400): axis k holds dims[k] values numpy.linspace(start[k], stop[k], dims[k]) (float64, cast to float32
 * like the dataset does), sigma is [dims[0], dims[1], dims[2]] in 'ij' order.  Bit-identical to naf_field_forward on the
 * materialised grid; the kernel walks the grid with axis 0 fastest, which turns most gathers of the hashed levels into L1
 * hits.  start / stop / dims are HOST arrays of three; the grid must lie inside [-bound, bound].
 * `workspace_bytes` is the size of `workspace`: the grid is evaluated in as many ranges of its traversal as that buffer needs
 * (at least naf_forward_workspace_bytes(cfg, 1024); same bits for every split), so a 1024^3 query (foot_50) runs in any budget. */
int naf_field_forward_grid(const double *start, const double *stop, const uint32_t *dims, const void *embeddings,
                           const int32_t *offsets, const float *mlp, float *sigma, const naf_render_cfg *cfg, void *workspace,
                           size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------------------------------
 * T4  Adam (torch.optim.Adam semantics, trainer.py:54: lr, betas=(0.9,0.999), eps=1e-8, no weight decay,
 * amsgrad off).  `step` is the 1-based step count.  If param_lp != NULL a low-precision copy of the
 * updated parameters (naf_dtype lp_dtype) is written too (16-bit tables keep an fp32 master).
 * If zero_grad != 0 the gradient buffer is cleared in the same pass.  param / exp_avg / exp_avg_sq / grad must be 16-byte
 * aligned when n >= 4 (fewer elements are stepped one by one: the ragged head of a row range).
 */
/* One training step of the TABLE in one call (single-GPU steps): naf_render_train whose gradient reducer applies the Adam
 * update to the table rows it has just finished instead of writing their gradient out for naf_adam_step to read back and
 * clear -- the 57 MB (T=2^19) gradient table is then neither written, re-read nor zeroed.  Same results, bit for bit, as
 * naf_render_train followed by naf_adam_step(param, exp_avg, exp_avg_sq, grad_embeddings, param_lp, lp_dtype, n, ...,
 * zero_grad = 1): `grad_embeddings` must be all zero on entry and is all zero on return; the MLP gradient is written as usual
 * (the caller steps the 4 225 MLP parameters with naf_adam_step, or hands their state over in `adam`); loss_out[0] is
 * OVERWRITTEN with this step's loss (a whole optimiser step has nothing to accumulate into; it spares the caller a clear).  Batches that take the atomic scatter
 * (< 2^13 points) or split reducer launches run the two passes one after the other inside the call.
 * `embeddings` is what the kernels gather from (the 16-bit shadow `param_lp` in 16-bit mode, `param` itself in fp32 mode). */
typedef struct naf_table_adam {
    float *param;          /* fp32 master table [rows, C] */
    float *exp_avg;        /* Adam moments, same shape */
    float *exp_avg_sq;
    void *param_lp;        /* 16-bit shadow table or NULL */
    int32_t lp_dtype;      /* NAF_F16 / NAF_BF16 when param_lp != NULL */
    uint64_t n;            /* rows * C */
    float lr, beta1, beta2, eps;
    uint32_t step;         /* 1-based */
    float grad_scale;      /* gradient multiplier (1 unless the loss was scaled) */
    /* Optional: the MLP's Adam state (4 225 fp32 values each, same hyper-parameters and step).  With mlp_param set (it must be
     * the `mlp` block the call reads) the reduction of the weight-gradient slabs applies the MLP's update itself: grad_mlp is
     * consumed (+= semantics: whatever it held is added first) and left zero, bit for bit what naf_adam_step(mlp ..., zero_grad
     * = 1) after the call would have done.  NULL: grad_mlp is written as usual and the caller steps the MLP. */
    float *mlp_param, *mlp_exp_avg, *mlp_exp_avg_sq;
} naf_table_adam;
int naf_render_train_adam(const float *rays, const float *t_rand, const float *target, const float *ray_weight,
                          const void *embeddings, const int32_t *offsets, const float *mlp, float *acc, float *grad_embeddings,
                          float *grad_mlp, float *loss_out, uint32_t n_rays, const naf_render_cfg *cfg, void *workspace,
                          const naf_table_adam *adam, void *stream);

/* The same step carrying the pixel draw of the NEXT step (G6, naf_draw_scan_rays): nothing in a step depends on the next step's
 * pixels, and the draw is 6 us of latency in a launch of four workgroups, so the step takes it along in spare workgroups of the
 * scatter's first launch (the canonical two-channel bf16 shape on the binned scatter; every other shape runs it as a launch of its
 * own behind the step).  `next` holds the arguments of naf_draw_scan_rays; its rays / target buffers must not be the ones this step
 * reads (double-buffer them).  NULL: exactly naf_render_train_adam. */
typedef struct naf_next_draw {
    naf_scan_draw draw;
    const float *poses, *projections;
    int64_t *pixels;           /* optional */
    float *target;             /* optional */
    float *rays;
    uint32_t first_draw, n_draws, n_projections, det_w, det_h;
    float du, dv, ou, ov, DSD, near, far;
    int32_t parallel;
    uint64_t seed;
} naf_next_draw;
int naf_render_train_adam_draw(const float *rays, const float *t_rand, const float *target, const float *ray_weight,
                               const void *embeddings, const int32_t *offsets, const float *mlp, float *acc, float *grad_embeddings,
                               float *grad_mlp, float *loss_out, uint32_t n_rays, const naf_render_cfg *cfg, void *workspace,
                               const naf_table_adam *adam, const naf_next_draw *next, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Level-parallel training for small steps on several GPUs (one process per GPU; no counterpart in the reference, which has no
 * distributed code -- it shards trainer.py:134-142 around train.py:48-135 like the data-parallel step does, with the same result).
 * Rank k of N owns the levels [k L/N, (k+1) L/N) of the table: their rows, Adam moments and 16-bit shadow.  A step is
 *   1. naf_levels_encode       the owned levels of EVERY rank's sample points      -> features [rank][owned levels][its points][C]
 *      all-to-all (RCCL, xGMI point to point): every rank receives all L levels of ITS points, [L][own points][C]
 *   2. naf_levels_field_step   MLP forward, loss, MLP backward on the own rays      -> feature gradients [L][own points][C],
 *                              grad_mlp (+=), loss_out (+=)   (summed over ranks by a 17 KB all-reduce)
 *      all-to-all back: every rank receives the gradients of its levels for every rank's points, one block per source rank
 *   3. naf_levels_scatter      table-gradient scatter of the owned levels over all points, Adam on their rows
 * Per rank and step 2 x (N-1)/N x points x L x C x 2 B cross the links (25 MB at the reference's 1 024 rays x 192 samples) instead
 * of the (N-1)/N x (57 + 28.5) MB a gradient exchange of the T = 2^19 table needs whatever the batch, and the optimiser pass shrinks
 * to 1/N of the table.  Above ~3 000 rays per GPU and step the gradient exchange is the smaller one (engine.py picks).
 * `rays` of calls 1 and 3 are all ranks' rays in rank order, cfg->ray_index_base the global index of the first of them, so that
 * every point gets the jitter the rank that renders it uses.  Features are bf16 (mlp_precision NAF_BF16) or fp32 (NAF_F32). */

/* features: one block per destination rank, [n_ranks][level_end - level_begin][B / n_ranks][C] (what the all-to-all sends as it is);
 * n_rays = all ranks' rays (a multiple of n_ranks), B = n_rays * n_samples points. */
int naf_levels_encode(const float *rays, const float *t_rand, const void *embeddings, const int32_t *offsets, void *features,
                      uint32_t n_rays, uint32_t n_ranks, const naf_render_cfg *cfg, uint32_t level_begin, uint32_t level_end,
                      void *stream);

/* features / feature_grads: [L][B][C] of the n_rays own rays.  acc[n_rays] is written; grad_mlp (+=) as in naf_render_train,
 * loss_out[0] is OVERWRITTEN with this rank's share of the loss.  `workspace`: naf_render_workspace_bytes(cfg, B).
 * `grads_ready` (hipEvent_t owned by the caller, or NULL) is recorded on `stream` as soon as feature_grads are final -- before the
 * reduction that finishes grad_mlp and loss_out -- so that the all-to-all of the gradients can start behind it on another stream. */
int naf_levels_field_step(const float *rays, const float *t_rand, const float *target, const float *ray_weight, const void *features,
                          const float *mlp, float *acc, void *feature_grads, float *grad_mlp, float *loss_out, uint32_t n_rays,
                          const naf_render_cfg *cfg, void *workspace, void *grads_ready, void *stream);

/* grad_blocks: n_ranks blocks `block_stride_bytes` apart, block r = [level_end - level_begin][B / n_ranks][C] feature gradients of
 * rank r's points (what the all-to-all delivers); n_rays = all ranks' rays (a multiple of n_ranks), `workspace`:
 * naf_render_workspace_bytes(cfg, B).  grad_embeddings (+=) receives the gradient of the owned levels' rows -- unless `adam` is
 * given and the reducer can apply the update itself (as in naf_render_train_adam; *adam_applied = 1): then the rows of the owned
 * levels in adam->param / exp_avg / exp_avg_sq / param_lp are stepped and grad_embeddings stays zero.  With *adam_applied = 0
 * the caller steps those rows with naf_adam_step.  adam->mlp_* are ignored (the MLP is replicated: its gradient is all-reduced).
 * With two bf16 channels (the canonical shape) the blocks are read in place by pass 1 of the scatter -- they must stay untouched until
 * the call's kernels have run; other shapes copy them into the workspace first (NAF_CFG_LEVELS_GATHER_PASS: always). */
int naf_levels_scatter(const float *rays, const float *t_rand, const void *grad_blocks, size_t block_stride_bytes, uint32_t n_ranks,
                       const int32_t *offsets, float *grad_embeddings, uint32_t n_rays, const naf_render_cfg *cfg,
                       uint32_t level_begin, uint32_t level_end, void *workspace, const naf_table_adam *adam, int *adam_applied,
                       void *stream);

int naf_adam_step(float *param, float *exp_avg, float *exp_avg_sq, float *grad, void *param_lp, int lp_dtype,
                  uint64_t n, float lr, float beta1, float beta2, float eps, uint32_t step, float grad_scale,
                  int zero_grad, void *stream);

/* E2  encoder input stage (hashgrid.py:122-125) without host round trips:
 *   out01[i] = (x[i] + size) / (2*size)   (IEEE fp32 add and divide, as torch evaluates it on the host)
 *   flag[0] |= 1 if any x[i] < -size or x[i] > size (or NaN); flag[1], flag[2] = min / max of x as
 *   order-preserving int32 (caller initialises flag to {0, INT32_MAX, INT32_MIN}).
 * out01 may be NULL (range check only).
 */
int naf_normalize_inputs(const float *x, uint64_t n, float size, float *out01, int32_t *flag, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* NAF_HIP_H */
