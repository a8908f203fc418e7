"""oracle/render_ref.py -- TEST INFRASTRUCTURE ONLY.

CPU (torch, fp32) restatement of the reference renderer, src/render/render.py:
  * render / chunking ................ render.py:31-79
  * render_chunk (sampling, clamp) ... render.py:82-146
  * run_network ...................... render.py:148-156
  * raw2outputs (sum sigma*delta) .... render.py:178-212
  * sample_pdf ....................... render.py:215-247
Pinned against golden vectors captured from the imported reference module (tests/golden/make_golden.py).
`t_rand` may be passed explicitly so stochastic sampling is reproducible (the reference draws torch.rand).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import torch


def sample_depths(near, far, n_samples, perturb, t_rand=None):
    """render.py:87-100.  near/far: [n,1] -> z_vals [n, n_samples]."""
    t = torch.linspace(0.0, 1.0, steps=n_samples, device=near.device)
    z = near * (1.0 - t) + far * t
    z = z.expand([near.shape[0], n_samples])
    if perturb:
        mids = 0.5 * (z[..., 1:] + z[..., :-1])
        upper = torch.cat([mids, z[..., -1:]], -1)
        lower = torch.cat([z[..., :1], mids], -1)
        if t_rand is None:
            t_rand = torch.rand(z.shape, device=z.device)
        z = lower + (upper - lower) * t_rand
    return z


def points_on_rays(rays, z_vals, bound):
    """render.py:102-105: pts = o + d*z, clamped to +-(bound - 1e-6)."""
    o, d = rays[..., :3], rays[..., 3:6]
    pts = o[..., None, :] + d[..., None, :] * z_vals[..., :, None]
    b = bound - 1e-6
    return pts.clamp(-b, b)


def run_network(inputs, fn, netchunk):
    flat = inputs.reshape(-1, inputs.shape[-1])
    outs = [fn(flat[i:i + netchunk]) for i in range(0, flat.shape[0], netchunk)]
    out = torch.cat(outs, 0)
    return out.reshape(list(inputs.shape[:-1]) + [out.shape[-1]])


def raw2outputs(raw, z_vals, rays_d, raw_noise_std=0.0):
    """render.py:192-206: line integral acc = sum_s sigma_s * dist_s; fine-pass weights."""
    dists = z_vals[..., 1:] - z_vals[..., :-1]
    last = torch.full_like(dists[..., :1], 1e-10)
    dists = torch.cat([dists, last], -1) * torch.norm(rays_d[..., None, :], dim=-1)
    noise = 0.0
    if raw_noise_std > 0.0:
        noise = torch.randn(raw[..., 0].shape, device=raw.device) * raw_noise_std
    acc = torch.sum((raw[..., 0] + noise) * dists, dim=-1)
    if raw.shape[-1] == 1:
        eps = torch.ones_like(raw[:, :1, -1]) * 1e-10
        weights = torch.cat([eps, (raw[:, 1:, -1] - raw[:, :-1, -1]).abs()], dim=-1)
        weights = weights / weights.max()
    elif raw.shape[-1] == 2:
        weights = raw[..., 1] / raw[..., 1].max()
    else:
        raise NotImplementedError("Wrong raw shape")
    return acc, weights


def sample_pdf(bins, weights, n_samples, det=False, u=None):
    """render.py:215-247 inverse-CDF sampling."""
    weights = weights + 1e-5
    pdf = weights / weights.sum(-1, keepdim=True)
    cdf = torch.cumsum(pdf, -1)
    cdf = torch.cat([torch.zeros_like(cdf[..., :1]), cdf], -1)
    if u is None:
        if det:
            u = torch.linspace(0.0, 1.0, steps=n_samples).expand(list(cdf.shape[:-1]) + [n_samples])
        else:
            u = torch.rand(list(cdf.shape[:-1]) + [n_samples])
    u = u.contiguous().to(cdf.device)
    inds = torch.searchsorted(cdf, u, right=True)
    below = (inds - 1).clamp(min=0)
    above = inds.clamp(max=cdf.shape[-1] - 1)
    g = torch.stack([below, above], -1)
    shape = [g.shape[0], g.shape[1], cdf.shape[-1]]
    cdf_g = torch.gather(cdf.unsqueeze(1).expand(shape), 2, g)
    bins_g = torch.gather(bins.unsqueeze(1).expand(shape), 2, g)
    denom = cdf_g[..., 1] - cdf_g[..., 0]
    denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)
    t = (u - cdf_g[..., 0]) / denom
    return bins_g[..., 0] + t * (bins_g[..., 1] - bins_g[..., 0])


def render_chunk(rays, net, net_fine, n_samples, n_fine, perturb, netchunk, raw_noise_std, t_rand=None):
    near, far = rays[..., 6:7], rays[..., 7:]
    z = sample_depths(near, far, n_samples, perturb, t_rand)
    pts = points_on_rays(rays, z, net.bound)
    raw = run_network(pts, net, netchunk)
    acc, weights = raw2outputs(raw, z, rays[..., 3:6], raw_noise_std)
    ret = {}
    if net_fine is not None and n_fine > 0:
        ret.update(acc0=acc, weights0=weights, pts0=pts)
        mid = 0.5 * (z[..., 1:] + z[..., :-1])
        zs = sample_pdf(mid, weights[..., 1:-1], n_fine, det=(perturb == 0.0)).detach()
        z, _ = torch.sort(torch.cat([z, zs], -1), -1)
        pts = points_on_rays(rays, z, net.bound)
        raw = run_network(pts, net_fine, netchunk)
        acc, _ = raw2outputs(raw, z, rays[..., 3:6], raw_noise_std)
    ret.update(acc=acc, pts=pts, tv_loss=0.1 * (pts[:, 1:] - pts[:, :-1]).abs().sum())
    return ret


def render(rays, net, net_fine, n_samples, n_fine, perturb, netchunk, raw_noise_std, chunk_size=None, t_rand=None):
    n = rays.shape[0]
    if chunk_size is None or chunk_size >= n:
        return render_chunk(rays, net, net_fine, n_samples, n_fine, perturb, netchunk, raw_noise_std, t_rand)
    parts = []
    for i in range(0, n, chunk_size):
        tr = None if t_rand is None else t_rand[i:i + chunk_size]
        parts.append(render_chunk(rays[i:i + chunk_size], net, net_fine, n_samples, n_fine, perturb, netchunk,
                                  raw_noise_std, tr))
    out = {"acc": torch.cat([p["acc"] for p in parts], 0), "pts": torch.cat([p["pts"] for p in parts], 0)}
    if "acc0" in parts[0]:
        for k in ("acc0", "weights0", "pts0"):
            out[k] = torch.cat([p[k] for p in parts], 0)
    return out
