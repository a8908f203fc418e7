"""Renderer: host-side mirror of reference src/render/render.py over libnaf_hip.so.

Same call surface and return keys as the reference:
    render(rays, net, net_fine, n_samples, n_fine, perturb, netchunk, raw_noise_std, chunk_size=None)
        -> {"acc", "pts", ["tv_loss"], ["acc0", "weights0", "pts0"]}                      (render.py:31-146)
    run_network(inputs, fn, netchunk)                                                     (render.py:148-156)
    raw2outputs(raw, z_vals, rays_d, raw_noise_std)  -> acc, weights                      (render.py:178-212)
    sample_pdf(bins, weights, N_samples, det)                                             (render.py:215-247)
One optional extra keyword, `t_rand`, lets a caller pass the stratified jitter explicitly (the reference draws
torch.rand inside render_chunk, render.py:99) so runs can be reproduced against the oracle.

Execution: when `net` is the canonical NAF network and there is no fine pass the whole chunk is one fused call
(hash gather -> MFMA MLP -> wave-reduced line integral); otherwise sampling and integration are the stand-alone
HIP operators and the network runs through `run_network` exactly like the reference.
"""
from __future__ import annotations

import torch
from torch.autograd import Function

from . import _abi
from .fused import fine_depths, fused_render, render_samples

RETURN_PTS = True        # the reference always returns the sample points; switch off to save n*S*12 bytes
CHECK_NUMERICS = True    # render.py:141-144 prints on NaN/Inf (costs one host sync per chunk)


def _sample(rays, n_samples, perturb, bound, t_rand):
    n = rays.shape[0]
    z = torch.empty(n, n_samples, device=rays.device, dtype=torch.float32)
    pts = torch.empty(n, n_samples, 3, device=rays.device, dtype=torch.float32)
    _abi.check(_abi.lib().naf_sample_rays(_abi.ptr(rays), _abi.ptr(t_rand), _abi.ptr(z), _abi.ptr(pts), n, n_samples,
                                          int(bool(perturb)), float(bound), 0, 0, _abi.stream_ptr()), "sample_rays")
    return z, pts


def noise_line_integral(rays, z_vals, raw_noise_std, noise=None):
    """What `raw_noise_std` adds to a ray's line integral.  render.py:196-201 computes acc = sum_s (sigma_s + noise_s) * dist_s with
    noise ~ N(0, raw_noise_std^2) per sample: the noise enters ADDITIVELY (it is neither a network input nor part of the weights of
    :203-211), so acc = acc_noise_free + sum_s noise_s * dist_s and the gradients of the loss see it only through d loss / d acc.
    The fused kernels therefore stay noise-free and this term -- same distribution, sample for sample, as the reference's -- is
    added to their output (or, for the training engine, subtracted from the target).  `noise`: explicit N(0,1) draws [n, S]."""
    d = z_vals[..., 1:] - z_vals[..., :-1]
    d = torch.cat([d, torch.full_like(d[..., :1], 1e-10)], -1) * torch.norm(rays[..., 3:6], dim=-1, keepdim=True)
    if noise is None:
        noise = torch.randn(z_vals.shape, device=z_vals.device)
    return torch.sum(noise * float(raw_noise_std) * d, dim=-1)


class _Integrate(Function):
    """acc[r] = sum_s sigma[r,s] * dist[r,s]   (render.py:192-201)."""

    @staticmethod
    def forward(ctx, sigma, z_vals, rays):
        sigma = sigma.contiguous().float()
        n, S = sigma.shape
        acc = torch.empty(n, device=sigma.device, dtype=torch.float32)
        _abi.check(_abi.lib().naf_integrate_forward(_abi.ptr(sigma), _abi.ptr(z_vals), _abi.ptr(rays), _abi.ptr(acc), n, S,
                                                    _abi.stream_ptr()), "integrate_forward")
        ctx.save_for_backward(z_vals, rays)
        ctx.shape = (n, S)
        return acc

    @staticmethod
    def backward(ctx, grad_acc):
        z_vals, rays = ctx.saved_tensors
        n, S = ctx.shape
        grad_sigma = torch.empty(n, S, device=grad_acc.device, dtype=torch.float32)
        grad_acc = grad_acc.contiguous().float()
        _abi.check(_abi.lib().naf_integrate_backward(_abi.ptr(grad_acc), _abi.ptr(z_vals), _abi.ptr(rays), _abi.ptr(grad_sigma),
                                                     n, S, _abi.stream_ptr()), "integrate_backward")
        return grad_sigma, None, None


def run_network(inputs, fn, netchunk):
    uvt_flat = torch.reshape(inputs, [-1, inputs.shape[-1]])
    out_flat = torch.cat([fn(uvt_flat[i:i + netchunk]) for i in range(0, uvt_flat.shape[0], netchunk)], 0)
    return out_flat.reshape(list(inputs.shape[:-1]) + [out_flat.shape[-1]])


def raw2outputs(raw, z_vals, rays_d, raw_noise_std=0.0):
    if raw_noise_std > 0.0:
        raw0 = raw[..., 0] + torch.randn(raw[..., 0].shape, device=raw.device) * raw_noise_std
    else:
        raw0 = raw[..., 0]
    if raw.is_cuda:
        rays = torch.zeros(rays_d.shape[0], 8, device=raw.device, dtype=torch.float32)
        rays[:, 3:6] = rays_d
        acc = _Integrate.apply(raw0, z_vals.contiguous().float(), rays)
    else:
        raise RuntimeError("raw2outputs: tensors must live on the GPU (no CPU path)")
    # importance weights of the fine pass (render.py:203-211): |sigma jump| between neighbouring samples (1e-10 for the
    # first), or the network's second output channel; both scaled by the largest weight of the WHOLE chunk (App. A-10)
    n_channels = raw.shape[-1]
    if n_channels == 1:
        sigma = raw[..., 0]
        weights = torch.nn.functional.pad((sigma[:, 1:] - sigma[:, :-1]).abs(), (1, 0), value=1e-10)
    elif n_channels == 2:
        weights = raw[..., 1]
    else:
        raise NotImplementedError("Wrong raw shape")
    return acc, weights / weights.max()


def sample_pdf(bins, weights, N_samples, det=False):
    """Inverse-transform sampling of the piecewise-constant density `weights` over the intervals between `bins`
    (render.py:215-247): [n, M] bins, [n, M-1] weights -> [n, N_samples] depths.  `det` takes evenly spaced quantiles."""
    mass = weights + 1e-5                                            # keeps empty rays samplable
    cdf = torch.cumsum(mass / mass.sum(-1, keepdim=True), -1)
    cdf = torch.nn.functional.pad(cdf, (1, 0))                       # cdf[..., 0] = 0 : one value per bin edge
    rows = list(cdf.shape[:-1])
    if det:
        u = torch.linspace(0.0, 1.0, steps=N_samples, device=cdf.device).expand(rows + [N_samples]).contiguous()
    else:
        u = torch.rand(rows + [N_samples], device=cdf.device)
    upper = torch.searchsorted(cdf, u, right=True)                   # first edge whose cdf exceeds u
    lower = (upper - 1).clamp_min(0)
    upper = upper.clamp_max(cdf.shape[-1] - 1)
    cdf_lo, cdf_hi = cdf.gather(-1, lower), cdf.gather(-1, upper)
    z_lo, z_hi = bins.gather(-1, lower), bins.gather(-1, upper)
    span = cdf_hi - cdf_lo
    span = torch.where(span < 1e-5, torch.ones_like(span), span)     # flat stretch of the cdf: stay on the lower edge
    return z_lo + (u - cdf_lo) / span * (z_hi - z_lo)


def _points(rays, z_vals, bound):
    pts = rays[..., None, :3] + rays[..., None, 3:6] * z_vals[..., :, None]
    return pts.clamp(-(bound - 1e-6), bound - 1e-6)


def render_chunk(rays, net, net_fine, n_samples, n_fine, perturb, netchunk, raw_noise_std, t_rand=None):
    if not rays.is_cuda:
        raise RuntimeError("render: rays must live on the GPU (no CPU path)")
    rays = rays.contiguous().float()
    n_rays = rays.shape[0]
    if perturb and t_rand is None:
        t_rand = torch.rand(n_rays, n_samples, device=rays.device)       # same RNG stream as render.py:99
    if not perturb:
        t_rand = None
    fine = net_fine is not None and n_fine > 0
    can_fuse = getattr(net, "fused_supported", lambda: False)()     # raw_noise_std: additive term on acc (noise_line_integral)
    noisy = float(raw_noise_std) > 0.0
    fused = (not fine) and can_fuse
    # coarse -> fine with both networks in the fused shape: three launches per pass instead of the reference's ATen soup --
    # coarse forward with per-sample sigma, naf_fine_depths (weights, cdf prefix sum, inverse sampling, sort), fine render
    # at the explicit depths.  The fine depths carry no gradient (render.py:121) and train.py's loss only reads "acc", so
    # the coarse pass runs without a graph: "acc0" / "weights0" come back detached here.
    fused_fine = (fine and can_fuse and getattr(net_fine, "fused_supported", lambda: False)() and n_samples >= 3
                  and n_samples <= 1024 and n_samples + n_fine <= 2048)

    z_vals = pts = None
    if RETURN_PTS or noisy or not (fused or fused_fine):
        z_vals, pts = _sample(rays, n_samples, perturb, net.bound, t_rand)
    if fused_fine:
        acc0, sigma, _ = render_samples(rays, net, n_samples, perturb, t_rand=t_rand, want_depth=False)
        det = perturb == 0.0
        u = None if det else torch.rand(n_rays, n_fine, device=rays.device)        # same RNG stream position as render.py:230
        z_all, weights0 = fine_depths(rays, sigma, n_fine, perturb, t_rand=t_rand, u=u, det=det)
        acc = fused_render(rays, net_fine, n_samples + n_fine, False, z_vals=z_all)
        if noisy:                                           # both passes draw their own noise (render.py:115,125)
            acc0 = acc0 + noise_line_integral(rays, z_vals, raw_noise_std)
            acc = acc + noise_line_integral(rays, z_all, raw_noise_std)
        ret = {"acc0": acc0, "weights0": weights0, "pts0": pts, "acc": acc}
        if RETURN_PTS:
            ret["pts"] = _points(rays, z_all, net.bound)
            ret["tv_loss"] = torch.sum(torch.abs(ret["pts"][:, 1:, :] - ret["pts"][:, :-1, :])) * 0.1
        if CHECK_NUMERICS:
            for k in ret:
                if ret[k] is not None and not torch.isfinite(ret[k]).all():
                    print(f"! [Numerical Error] {k} contains nan or inf.")
        return ret
    if fused:
        acc = fused_render(rays, net, n_samples, perturb, t_rand=t_rand)
        if noisy:
            acc = acc + noise_line_integral(rays, z_vals, raw_noise_std)
        weights = None
    else:
        raw = run_network(pts, net, netchunk)
        acc, weights = raw2outputs(raw, z_vals, rays[..., 3:6], raw_noise_std)

    ret = {}
    if fine:
        ret.update(acc0=acc, weights0=weights, pts0=pts)
        z_mid = 0.5 * (z_vals[..., 1:] + z_vals[..., :-1])
        z_samples = sample_pdf(z_mid, weights[..., 1:-1], n_fine, det=(perturb == 0.0)).detach()
        z_vals, _ = torch.sort(torch.cat([z_vals, z_samples], -1), -1)
        pts = _points(rays, z_vals, net.bound)
        raw = run_network(pts, net_fine, netchunk)
        acc, _ = raw2outputs(raw, z_vals, rays[..., 3:6], raw_noise_std)
    ret["acc"] = acc
    if pts is not None:
        ret["pts"] = pts
        ret["tv_loss"] = torch.sum(torch.abs(pts[:, 1:, :] - pts[:, :-1, :])) * 0.1       # render.py:129-131
    if CHECK_NUMERICS:
        for k in ret:
            if ret[k] is not None and not torch.isfinite(ret[k]).all():
                print(f"! [Numerical Error] {k} contains nan or inf.")
    return ret


def render(rays, net, net_fine, n_samples, n_fine, perturb, netchunk, raw_noise_std, chunk_size=None, t_rand=None):
    n_rays = rays.shape[0]
    if chunk_size is None or chunk_size >= n_rays:
        return render_chunk(rays, net, net_fine, n_samples, n_fine, perturb, netchunk, raw_noise_std, t_rand)
    parts = []
    for i in range(0, n_rays, chunk_size):
        tr = None if t_rand is None else t_rand[i:i + chunk_size].contiguous()
        parts.append(render_chunk(rays[i:i + chunk_size], net, net_fine, n_samples, n_fine, perturb, netchunk,
                                  raw_noise_std, tr))
    ret = {"acc": torch.cat([p["acc"] for p in parts], 0)}
    if "pts" in parts[0]:
        ret["pts"] = torch.cat([p["pts"] for p in parts], 0)
    if "acc0" in parts[0]:
        for k in ("acc0", "weights0", "pts0"):
            ret[k] = torch.cat([p[k] for p in parts], 0)
    return ret
