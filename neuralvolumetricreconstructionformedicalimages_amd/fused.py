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


def render_cfg(net, n_samples, perturb, mlp_precision=None, seed=0, ray_index_base=0, scatter=None, flags=None):
    enc = net.encoder
    table_dtype = _abi.dtype_code(enc.embeddings.dtype)
    if mlp_precision is None:          # parity mode for fp32 tables, bf16 matrix cores for 16-bit tables
        mlp_precision = _abi.F32 if table_dtype == _abi.F32 else _abi.BF16
    return _abi.RenderCfg(n_samples=int(n_samples), perturb=int(bool(perturb)), bound=float(net.bound),
                          L=enc.num_levels, C=enc.level_dim, H=enc.base_resolution, table_dtype=table_dtype,
                          mlp_precision=int(mlp_precision), last_activation=LAST_ACTIVATIONS[net.last_activation],
                          seed=int(seed) & (2 ** 64 - 1), ray_index_base=int(ray_index_base), log2_hashmap_size=int(enc.log2_hashmap_size),
                          scatter_mode=_default_scatter_mode if scatter is None else int(scatter),
                          flags=_default_flags if flags is None else int(flags))


def workspace(cfg, n_points, device):
    """Grow-only scratch buffer per device (features, feature gradients, MLP-gradient slabs)."""
    need = int(_abi.lib().naf_render_workspace_bytes(ctypes.byref(cfg), int(n_points)))
    buf = _workspaces.get(device)
    if buf is None or buf.numel() < need:
        buf = torch.empty(need, dtype=torch.uint8, device=device)
        _workspaces[device] = buf
    return buf


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
    cfg = render_cfg(net, n_samples, perturb, mlp_precision, seed)
    ws = workspace(cfg, n * cfg.n_samples, rays.device)
    acc = torch.empty(n, device=rays.device, dtype=torch.float32)
    sigma = torch.empty(n, n_samples, device=rays.device, dtype=torch.float32) if want_sigma else None
    depth = torch.empty(n, n_samples, device=rays.device, dtype=torch.float32) if want_depth else None
    enc = net.encoder
    _abi.check(_abi.lib().naf_render_forward_samples(
        _abi.ptr(rays), _abi.ptr(t_rand), _abi.ptr(enc.embeddings.detach().contiguous()), _abi.ptr(_offsets(enc, rays.device)),
        _abi.ptr(net.packed_mlp().contiguous()), _abi.ptr(acc), _abi.ptr(sigma), _abi.ptr(depth), n, ctypes.byref(cfg), _abi.ptr(ws),
        _abi.stream_ptr()), "render_forward_samples")
    _bump(rays.device)
    return acc, sigma, depth


@torch.no_grad()
def field_query_grid(net, starts, stops, dims, mlp_precision=None):
    """sigma on the regular grid whose axis k is numpy.linspace(starts[k], stops[k], dims[k]) -> [d0, d1, d2]
    (`naf_field_forward_grid`): the volume query of train.py:246-250 without materialising the [n^3, 3] point list, bit-identical
    to `field_query` on it and faster (the kernel walks the grid along x)."""
    enc = net.encoder
    device = enc.embeddings.device
    dims = [int(v) for v in dims]
    B = dims[0] * dims[1] * dims[2]
    if max(abs(float(v)) for v in list(starts) + list(stops)) > net.bound:
        raise ValueError(f"HashGrid encoder: inputs range [{min(starts)}, {max(stops)}] not in [{-net.bound}, {net.bound}]!")
    cfg = render_cfg(net, 2, False, mlp_precision)
    ws = workspace(cfg, B, device)
    sigma = torch.empty(dims, device=device, dtype=torch.float32)
    a, b, d = (ctypes.c_double * 3)(*[float(v) for v in starts]), (ctypes.c_double * 3)(*[float(v) for v in stops]), (ctypes.c_uint32 * 3)(*dims)
    _abi.check(_abi.lib().naf_field_forward_grid(ctypes.byref(a), ctypes.byref(b), ctypes.byref(d), _abi.ptr(enc.embeddings.detach().contiguous()),
                                                 _abi.ptr(_offsets(enc, device)), _abi.ptr(net.packed_mlp().contiguous()), _abi.ptr(sigma),
                                                 ctypes.byref(cfg), _abi.ptr(ws), _abi.stream_ptr()), "field_forward_grid")
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
    cfg = render_cfg(net, 2, False, mlp_precision)
    ws = workspace(cfg, B, flat.device)
    sigma = torch.empty(B, device=flat.device, dtype=torch.float32)
    mlp = net.packed_mlp().contiguous()
    emb = enc.embeddings.detach().contiguous()
    _abi.check(_abi.lib().naf_field_forward(_abi.ptr(flat), _abi.ptr(emb), _abi.ptr(_offsets(enc, flat.device)), _abi.ptr(mlp),
                                            _abi.ptr(sigma), B, ctypes.byref(cfg), _abi.ptr(ws), _abi.stream_ptr()),
               "field_forward")
    _bump(flat.device)          # a pending _FusedRender.backward must recompute its features
    return sigma.reshape(list(pts.shape[:-1]) + [1])
