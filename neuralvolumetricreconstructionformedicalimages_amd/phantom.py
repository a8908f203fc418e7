"""Synthetic CT data with closed-form projections (the reference ships no data, /root/reference/.MISSING_LARGE_BLOBS).

A phantom is a sum of K rotated ellipsoids with attenuation in (0,1]; both the voxel volume (ground truth for
`get_psnr_3d`) and the line integral along any ray are analytic, so no TIGRE projector
(reference dataGenerator/generateData.py:178,189) is needed.  The pickle-style dict returned by `make_scan` has the
schema `TIGREDataset` reads (src/dataset/tigre.py:227-323, format_data.py:25-58).
"""
from __future__ import annotations

import numpy as np
import torch


def ellipsoid_table(seed=0, K=12, extent=0.128):
    """K ellipsoids inside a cube of half-width `extent` (metres): centres, semi-axes, rotations, densities."""
    rng = np.random.default_rng(seed)
    centres = rng.uniform(-0.45, 0.45, (K, 3)) * extent
    axes = rng.uniform(0.12, 0.45, (K, 3)) * extent
    rots = []
    for _ in range(K):
        q = rng.standard_normal(4)
        q /= np.linalg.norm(q)
        w, x, y, z = q
        rots.append([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    dens = rng.uniform(0.08, 0.22, K)
    # one big body ellipsoid so most rays see something
    centres[0], axes[0], dens[0] = 0.0, np.array([0.85, 0.7, 0.9]) * extent, 0.15
    rots[0] = np.eye(3).tolist()
    return {"c": centres, "a": axes, "R": np.asarray(rots), "rho": dens}


def _t(table, device, dtype):
    return {k: torch.as_tensor(v, device=device, dtype=dtype) for k, v in table.items()}


def line_integrals(rays, table, dtype=torch.float64):
    """Exact  int sigma dl  along rays [n,8] (origin, un-normalised direction, near, far): float32 [n]."""
    t = _t(table, rays.device, dtype)
    o = rays[:, None, :3].to(dtype) - t["c"][None]                    # [n,K,3]
    d = rays[:, None, 3:6].to(dtype).expand_as(o)
    p = torch.einsum("kij,nkj->nki", t["R"], o) / t["a"][None]
    q = torch.einsum("kij,nkj->nki", t["R"], d) / t["a"][None]
    qq, pq, pp = (q * q).sum(-1), (p * q).sum(-1), (p * p).sum(-1)
    disc = pq * pq - qq * (pp - 1.0)
    chord = 2.0 * torch.sqrt(disc.clamp(min=0.0)) / qq                # in units of the ray parameter
    length = chord * rays[:, None, 3:6].to(dtype).norm(dim=-1)
    return (length * t["rho"][None]).sum(-1).float()


def volume(geo, table, device="cpu", slab=16):
    """Ground-truth attenuation at the voxel centres: float32 [n1,n2,n3] (same grid as tigre.py:388-400)."""
    t = _t(table, device, torch.float32)
    n1, n2, n3 = (int(v) for v in geo.nVoxel)
    s = geo.sVoxel / 2 - geo.dVoxel / 2
    ax = [torch.linspace(-float(s[i]), float(s[i]), n, device=device) for i, n in enumerate((n1, n2, n3))]
    out = torch.empty(n1, n2, n3, device=device)
    for i0 in range(0, n1, slab):
        X, Y, Z = torch.meshgrid(ax[0][i0:i0 + slab], ax[1], ax[2], indexing="ij")
        pts = torch.stack([X, Y, Z], -1).reshape(-1, 1, 3) - t["c"][None]
        y = torch.einsum("kij,nkj->nki", t["R"], pts) / t["a"][None]
        inside = ((y * y).sum(-1) <= 1.0).float()
        out[i0:i0 + slab] = (inside * t["rho"][None]).sum(-1).reshape(X.shape)
    return out


def scan_geometry(n_voxel=256, mode="cone", tilt_angle=0):
    """Geometry of SURVEY.md 8(d): the reference's own 256^3 example (src/dataset/tigreWithOwnRay.py:213-227) scaled."""
    k = 256.0 / n_voxel
    data = {
        "DSD": 1500.0, "DSO": 1000.0,
        "nDetector": [2 * n_voxel, 2 * n_voxel], "dDetector": [0.8 * k, 0.8 * k],
        "nVoxel": [n_voxel] * 3, "dVoxel": [1.0 * k] * 3,
        "offOrigin": [0, 0, 0], "offDetector": [0, 0], "accuracy": 0.5, "mode": mode, "filter": None,
    }
    if tilt_angle:
        data["tilt_angle"] = tilt_angle
    return data
