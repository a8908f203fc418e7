"""Python front-end of the fused ray-march entry points of libnaf_hip.so (include/naf_hip.h):
naf_render_forward / naf_render_backward / naf_render_train / naf_field_forward.

The reference evaluates the same maths as ~40 ATen/cuBLAS launches per 200-ray chunk
(src/render/render.py:82-212, src/network/network.py:34-58); here one call covers the whole ray batch.
"""
from __future__ import annotations

import contextlib
import ctypes

import torch
from torch.autograd import Function

from . import _abi
from .network import LAST_ACTIVATIONS

_workspaces = {}
_feature_gen = {}          # device -> generation counter of the features currently held by the workspace


def _bump(device):
    _feature_gen[device] = _feature_gen.get(device, 0) + 1
    return _feature_gen[device]


_default_scatter_mode = _abi.SCATTER_AUTO
_default_flags = 0
# True: forward-only calls (projection render, volume query) run gathers + MLP in ONE kernel where the shape allows it
# (NAF_CFG_FORWARD_FUSED; same bits, no workspace).  Off by default because it is slower -- measured on MI355X
# (profiles/round3_eval_fused_vs_two_kernels.jsonl): 512 x 512 projection 34.2 ms fused against 12.3 ms with encode_kernel +
# mlp16_forward_kernel, 256^3 grid query 5.0 against 3.4 ms.  A kernel that walks all 16 levels of a tile has the whole table as
# its working set: 62 GB of fabric fetches per call against 0.3 GB for the level-major encoder, whose level stays in L2; the
# feature round trip it avoids is 0.2 ms of that.
forward_fused = False


@contextlib.contextmanager
def scatter_mode(mode, flags=None):
    """Default `naf_render_cfg.scatter_mode` (and optionally `.flags`) of every cfg built inside the block (tests and
    diagnostics compare the binned scatter with the reference's atomics this way).  The value travels in the cfg of each
    call; the library itself keeps no mode."""
    global _default_scatter_mode, _default_flags
    if mode not in (_abi.SCATTER_AUTO, _abi.SCATTER_ATOMIC, _abi.SCATTER_BINNED):
        raise ValueError("scatter mode must be SCATTER_AUTO (0), SCATTER_ATOMIC (1) or SCATTER_BINNED (2)")
    saved = (_default_scatter_mode, _default_flags)
    _default_scatter_mode = mode
    if flags is not None:
        _default_flags = int(flags)
    try:
        yield
    finally:
        _default_scatter_mode, _default_flags = saved


def render_cfg(net, n_samples, perturb, mlp_precision=None, seed=0, ray_index_base=0, scatter=None, flags=None, forward_only=False):
    enc = net.encoder
    if flags is None:
        flags = _default_flags | (_abi.CFG_FORWARD_FUSED if (forward_only and forward_fused) else 0)
    table_dtype = _abi.dtype_code(enc.embeddings.dtype)
    if mlp_precision is None:          # parity mode for fp32 tables, bf16 matrix cores for 16-bit tables
        mlp_precision = _abi.F32 if table_dtype == _abi.F32 else _abi.BF16
    return _abi.RenderCfg(n_samples=int(n_samples), perturb=int(bool(perturb)), bound=float(net.bound),
                          L=enc.num_levels, C=enc.level_dim, H=enc.base_resolution, table_dtype=table_dtype,
                          mlp_precision=int(mlp_precision), last_activation=LAST_ACTIVATIONS[net.last_activation],
                          seed=int(seed) & (2 ** 64 - 1), ray_index_base=int(ray_index_base), log2_hashmap_size=int(enc.log2_hashmap_size),
                          scatter_mode=_default_scatter_mode if scatter is None else int(scatter),
                          flags=int(flags))


def workspace(cfg, n_points, device):
    """Grow-only scratch buffer per device for calls that run a backward pass (features, feature gradients, MLP-gradient slabs,
    scatter records: ~1.1 KB per point)."""
    return _grow(device, int(_abi.lib().naf_render_workspace_bytes(ctypes.byref(cfg), int(n_points))))


# Forward-only calls need the [L, n, C] features and nothing else (64 B per point in bf16 mode, 128 B in fp32 mode; nothing with
# the fused kernel).  Their workspace is capped: a volume query walks its grid in as many ranges as the cap requires
# (`naf_field_forward_grid` does that itself, bit-identically), so the 1024^3 query of foot_50 runs in the same 1 GiB as the
# 256^3 one of chest_50 (train.py:246-250; the training layout would want 196 GiB for it).
FORWARD_WORKSPACE_CAP = 1 << 30


def forward_workspace(cfg, n_points, device, cap=None):
    need = int(_abi.lib().naf_forward_workspace_bytes(ctypes.byref(cfg), int(n_points)))
    floor = int(_abi.lib().naf_forward_workspace_bytes(ctypes.byref(cfg), 1024)) + 512
    return _grow(device, max(floor, min(need, FORWARD_WORKSPACE_CAP if cap is None else int(cap))))


def _grow(device, need):
    buf = _workspaces.get(device)
    if buf is None or buf.numel() < need:
        buf = torch.empty(need, dtype=torch.uint8, device=device)
        _workspaces[device] = buf
    return buf


def forward_points_per_call(cfg, device, cap=None):
    """Points one forward-only call may cover under the workspace cap (callers split larger point lists / ray batches)."""
    per_point = int(_abi.lib().naf_forward_workspace_bytes(ctypes.byref(cfg), 1 << 20)) / float(1 << 20)
    if per_point < 1.0:                                   # fused kernel: no features in HBM
        return (1 << 31) - 1
    return max(1024, int(((FORWARD_WORKSPACE_CAP if cap is None else int(cap)) - 512) / per_point) // 1024 * 1024)


def _offsets(enc, device):
    if enc.offsets.device != device:
        enc.offsets = enc.offsets.to(device)
    return enc.offsets


class _FusedRender(Function):
    """acc[r] = sum_s sigma(pts[r,s]) * dist[r,s]  with gradients for the hash table and the MLP block."""

    @staticmethod
    def forward(ctx, rays, t_rand, embeddings, mlp, offsets, cfg):
        rays = rays.contiguous().float()
        n_rays = rays.shape[0]
        acc = torch.empty(n_rays, device=rays.device, dtype=torch.float32)
        ws = workspace(cfg, n_rays * cfg.n_samples, rays.device)
        mlp = mlp.contiguous()
        emb = embeddings.contiguous()
        _abi.check(_abi.lib().naf_render_forward(_abi.ptr(rays), _abi.ptr(t_rand), _abi.ptr(emb), _abi.ptr(offsets),
                                                 _abi.ptr(mlp), _abi.ptr(acc), n_rays, ctypes.byref(cfg), _abi.ptr(ws),
                                                 _abi.stream_ptr()), "render_forward")
        ctx.save_for_backward(rays, t_rand, emb, mlp, offsets)
        ctx.cfg = cfg
        ctx.gen = _bump(rays.device)
        ctx.ws_ptr = ws.data_ptr()
        return acc

    @staticmethod
    def backward(ctx, grad_acc):
        rays, t_rand, emb, mlp, offsets = ctx.saved_tensors
        cfg = ctx.cfg
        n_rays = rays.shape[0]
        grad_emb = torch.zeros(emb.shape, device=emb.device, dtype=torch.float32)
        grad_mlp = torch.zeros(_abi.MLP_PARAMS, device=emb.device, dtype=torch.float32)
        ws = workspace(cfg, n_rays * cfg.n_samples, rays.device)
        # the features are still in the workspace iff nobody rendered (or re-allocated it) in between
        valid = int(ws.data_ptr() == ctx.ws_ptr and _feature_gen.get(rays.device) == ctx.gen)
        grad_acc = grad_acc.contiguous().float()
        _abi.check(_abi.lib().naf_render_backward(_abi.ptr(rays), _abi.ptr(t_rand), _abi.ptr(grad_acc), _abi.ptr(emb),
                                                  _abi.ptr(offsets), _abi.ptr(mlp), _abi.ptr(grad_emb), _abi.ptr(grad_mlp),
                                                  n_rays, ctypes.byref(cfg), _abi.ptr(ws), valid, _abi.stream_ptr()),
                   "render_backward")
        return None, None, grad_emb.to(emb.dtype), grad_mlp, None, None


def fused_render(rays, net, n_samples, perturb, t_rand=None, seed=0, mlp_precision=None, z_vals=None):
    """Differentiable fused render of `rays` [n,8] through `net` (a fused_supported() DensityNetwork) -> acc [n].
    `z_vals` [n, n_samples]: render at these (sorted) depths instead of the stratified ones -- the fine pass."""
    flags = None
    if z_vals is not None:
        z_vals = z_vals.detach().contiguous().float()      # render.py:121: the fine depths carry no gradient
        if z_vals.shape != (rays.shape[0], n_samples):
            raise ValueError("z_vals must be [n_rays, n_samples]")
        t_rand, perturb, flags = z_vals, False, _default_flags | _abi.CFG_EXPLICIT_DEPTHS
    needs_grad = torch.is_grad_enabled() and (net.encoder.embeddings.requires_grad or any(p.requires_grad for p in net.layers.parameters()))
    if not needs_grad:
        # eval (train.py:235-239 under torch.no_grad()): forward-only call -- the features are not kept, so the single fused
        # kernel applies and the workspace is the features alone (or nothing)
        if flags is not None and forward_fused:
            flags |= _abi.CFG_FORWARD_FUSED
        cfg = render_cfg(net, n_samples, perturb, mlp_precision, seed, flags=flags, forward_only=True)
        rays = rays.contiguous().float()
        n = rays.shape[0]
        acc = torch.empty(n, device=rays.device, dtype=torch.float32)
        step = max(1, forward_points_per_call(cfg, rays.device) // int(n_samples))
        ws = forward_workspace(cfg, min(n, step) * int(n_samples), rays.device)
        enc = net.encoder
        emb, mlp, offs = enc.embeddings.detach().contiguous(), net.packed_mlp().detach().contiguous(), _offsets(enc, rays.device)
        for b in range(0, n, step):                      # rays are independent; the jitter stream is per global ray index
            m = min(step, n - b)
            cfg.ray_index_base = b
            tr = None if t_rand is None else t_rand[b:b + m]
            _abi.check(_abi.lib().naf_render_forward(_abi.ptr(rays[b:b + m]), _abi.ptr(tr), _abi.ptr(emb), _abi.ptr(offs), _abi.ptr(mlp),
                                                     _abi.ptr(acc[b:b + m]), m, ctypes.byref(cfg), _abi.ptr(ws), _abi.stream_ptr()),
                       "render_forward")
        _bump(rays.device)
        return acc
    cfg = render_cfg(net, n_samples, perturb, mlp_precision, seed, flags=flags)
    return _FusedRender.apply(rays, t_rand, net.encoder.embeddings, net.packed_mlp(), _offsets(net.encoder, rays.device), cfg)


@torch.no_grad()
def fine_depths(rays, sigma, n_fine, perturb, t_rand=None, u=None, det=False, seed=0):
    """Coarse -> fine resampling on the device (naf_fine_depths): `sigma` [n, S] of the coarse pass ->
    (z_all [n, S + n_fine] sorted, weights0 [n, S]).  `t_rand` is the jitter the coarse pass used; `u` [n, n_fine] the
    uniforms of the inverse-transform sampling (None with det=True: evenly spaced quantiles)."""
    rays = rays.contiguous().float()
    sigma = sigma.contiguous().float()
    n, S = sigma.shape
    z_all = torch.empty(n, S + n_fine, device=rays.device, dtype=torch.float32)
    weights = torch.empty(n, S, device=rays.device, dtype=torch.float32)
    scratch = torch.empty(1, device=rays.device, dtype=torch.int32)
    _abi.check(_abi.lib().naf_fine_depths(_abi.ptr(rays), _abi.ptr(t_rand), _abi.ptr(sigma), _abi.ptr(u), _abi.ptr(z_all),
                                          _abi.ptr(weights), n, S, int(n_fine), int(bool(perturb)), int(bool(det)),
                                          int(seed) & (2 ** 64 - 1), 0, _abi.ptr(scratch), _abi.stream_ptr()), "fine_depths")
    return z_all, weights


@torch.no_grad()
def render_samples(rays, net, n_samples, perturb, t_rand=None, seed=0, mlp_precision=None, want_sigma=True, want_depth=True):
    """Forward render with the per-sample quantities of the coarse -> fine pass (render.py:113-126, 203-211):
    -> (acc [n], sigma [n, S] or None, optical_depth [n, S] or None).  optical_depth[r, s] is the running line integral
    sum_{s' <= s} sigma * dist (a wave prefix sum in the MLP kernel); its last column equals acc."""
    rays = rays.contiguous().float()
    n = rays.shape[0]
    cfg = render_cfg(net, n_samples, perturb, mlp_precision, seed, forward_only=True)
    ws = forward_workspace(cfg, n * cfg.n_samples, rays.device, cap=1 << 62)
    acc = torch.empty(n, device=rays.device, dtype=torch.float32)
    sigma = torch.empty(n, n_samples, device=rays.device, dtype=torch.float32) if want_sigma else None
    depth = torch.empty(n, n_samples, device=rays.device, dtype=torch.float32) if want_depth else None
    enc = net.encoder
    # every tensor behind a pointer stays referenced until the launch has been issued (a temporary would go back to the caching
    # allocator at once and could be handed to an allocation that is enqueued before the kernel)
    emb, mlp, offs = enc.embeddings.detach().contiguous(), net.packed_mlp().contiguous(), _offsets(enc, rays.device)
    _abi.check(_abi.lib().naf_render_forward_samples(
        _abi.ptr(rays), _abi.ptr(t_rand), _abi.ptr(emb), _abi.ptr(offs), _abi.ptr(mlp), _abi.ptr(acc), _abi.ptr(sigma), _abi.ptr(depth),
        n, ctypes.byref(cfg), _abi.ptr(ws), _abi.stream_ptr()), "render_forward_samples")
    _bump(rays.device)
    return acc, sigma, depth


@torch.no_grad()
def field_query_grid(net, starts, stops, dims, mlp_precision=None, workspace_cap=None):
    """sigma on the regular grid whose axis k is numpy.linspace(starts[k], stops[k], dims[k]) -> [d0, d1, d2]
    (`naf_field_forward_grid`): the volume query of train.py:246-250 without materialising the [n^3, 3] point list, bit-identical
    to `field_query` on it and faster (the kernel walks the grid along x).  The library evaluates the grid in as many ranges
    as the forward workspace (`workspace_cap` bytes, default FORWARD_WORKSPACE_CAP) requires -- same values for any cap."""
    enc = net.encoder
    device = enc.embeddings.device
    dims = [int(v) for v in dims]
    B = dims[0] * dims[1] * dims[2]
    if max(abs(float(v)) for v in list(starts) + list(stops)) > net.bound:
        raise ValueError(f"HashGrid encoder: inputs range [{min(starts)}, {max(stops)}] not in [{-net.bound}, {net.bound}]!")
    cfg = render_cfg(net, 2, False, mlp_precision, forward_only=True)
    ws = forward_workspace(cfg, B, device, workspace_cap)
    sigma = torch.empty(dims, device=device, dtype=torch.float32)
    a, b, d = (ctypes.c_double * 3)(*[float(v) for v in starts]), (ctypes.c_double * 3)(*[float(v) for v in stops]), (ctypes.c_uint32 * 3)(*dims)
    emb, mlp, offs = enc.embeddings.detach().contiguous(), net.packed_mlp().contiguous(), _offsets(enc, device)
    _abi.check(_abi.lib().naf_field_forward_grid(ctypes.byref(a), ctypes.byref(b), ctypes.byref(d), _abi.ptr(emb), _abi.ptr(offs), _abi.ptr(mlp),
                                                 _abi.ptr(sigma), ctypes.byref(cfg), _abi.ptr(ws), ws.numel(), _abi.stream_ptr()), "field_forward_grid")
    _bump(device)
    return sigma


@torch.no_grad()
def field_query(net, pts, mlp_precision=None):
    """sigma(pts) for a point cloud [..., 3] in [-bound, bound] -> [..., 1] (volume query, train.py:246-250)."""
    enc = net.encoder
    flat = pts.reshape(-1, 3).contiguous().float()
    B = flat.shape[0]
    if enc.strict_range:
        enc._normalize(flat, net.bound)          # raises ValueError like hashgrid.py:122-123
    cfg = render_cfg(net, 2, False, mlp_precision, forward_only=True)
    step = forward_points_per_call(cfg, flat.device)
    ws = forward_workspace(cfg, min(B, step), flat.device)
    sigma = torch.empty(B, device=flat.device, dtype=torch.float32)
    mlp = net.packed_mlp().contiguous()
    emb = enc.embeddings.detach().contiguous()
    for b in range(0, B, step):                          # points are independent: the split changes nothing
        n = min(step, B - b)
        _abi.check(_abi.lib().naf_field_forward(_abi.ptr(flat[b:b + n]), _abi.ptr(emb), _abi.ptr(_offsets(enc, flat.device)), _abi.ptr(mlp),
                                                _abi.ptr(sigma[b:b + n]), n, ctypes.byref(cfg), _abi.ptr(ws), _abi.stream_ptr()),
                   "field_forward")
    _bump(flat.device)          # a pending _FusedRender.backward must recompute its features
    return sigma.reshape(list(pts.shape[:-1]) + [1])
