"""Scan dataset: host-side mirror of `TIGREDataset` (reference src/dataset/tigre.py:220-382).

Same constructor (`path, n_rays, type, device`), same item dictionaries (`rays [n,8]`, `projs`, `coords`, `full_proj`)
and the same pickle schema (tigre.py:230-323; written by format_data.py:25-58 / dataGenerator/generateData.py:153-211).
What differs is the MI355X-first data layout: the reference materialises `rays[N,H,W,8]` for every pixel of every
projection on the device (419 MB at 50x512^2, 24 GB at 720x1024^2, tigre.py:247-255); here only the poses [N,3,4]
live in HBM and rays are generated on demand by `naf_generate_rays` for the pixels a step actually uses.  `.rays`
stays available as a lazy, indexable view for code written against the reference.

Unlike the committed reference cone-beam scans work (`get_rays`, tigre.py:434-437) and
`full_proj` is optional (standard NAF pickles do not have it).
"""
from __future__ import annotations

import pickle

import numpy as np
import torch
from torch.utils.data import Dataset

from .geometry import ConeGeometry, RayGenerator, get_near_far, get_voxels


class _LazyRays:
    """`dataset.rays[i]` -> [H, W, 8] like the reference's precomputed tensor, generated when asked for."""

    def __init__(self, raygen):
        self._g = raygen
        self.shape = (raygen.n_projections, raygen.H, raygen.W, 8)

    def __len__(self):
        return self._g.n_projections

    def __getitem__(self, key):
        if isinstance(key, tuple):                       # rays[index, rows, cols]
            index, rows, cols = key
            pix = int(index) * self._g.pixels_per_projection + rows.long() * self._g.W + cols.long()
            pix = pix.to(self._g.device)                 # index tensors may live on the host, like indexing the reference's tensor
            return self._g.rays_for_pixels(pix.reshape(-1)).reshape(list(pix.shape) + [8])
        return self._g.rays_for_projection(int(key)).reshape(self._g.H, self._g.W, 8)


class TIGREDataset(Dataset):
    """TIGRE dataset (`path` may also be an already loaded dict with the pickle schema)."""

    def __init__(self, path, n_rays=1024, type="train", device="cuda", shard=(0, 1), seed=None):
        """`shard = (rank, world)` (data parallel): every rank draws the SAME `n_rays` pixels per item (shared generator
        seed) and keeps its contiguous slice, so the union over ranks is exactly the single-process batch."""
        super().__init__()
        if isinstance(path, dict):
            data = path
        else:
            with open(path, "rb") as handle:
                data = pickle.load(handle)
        if type not in ("train", "val"):
            raise ValueError("type must be 'train' or 'val'")
        self.geo = ConeGeometry(data)
        self.type = type
        self.n_rays = n_rays
        self.device = torch.device(device)
        self.near, self.far = get_near_far(self.geo)

        split = data[type]
        projs = np.asarray(split["projections"])
        self.projs = torch.tensor(projs, dtype=torch.float32, device=self.device)
        self.full_proj = None
        if data.get("full_proj") is not None:
            self.full_proj = torch.tensor(np.asarray(data["full_proj"]), dtype=torch.complex64, device=self.device)
        self.angles = np.asarray(split["angles"], dtype=np.float64).reshape(-1)
        self.raygen = RayGenerator(self.geo, self.angles, self.device)
        self.rays = _LazyRays(self.raygen)
        self.n_samples = int(data["numTrain"] if type == "train" else data["numVal"])
        H, W = self.raygen.H, self.raygen.W
        rows, cols = torch.meshgrid(torch.arange(H, device=self.device), torch.arange(W, device=self.device), indexing="ij")
        self.coords = torch.stack([rows, cols], -1).reshape(-1, 2).float()     # (row, col) like tigre.py:256-276
        self.image = torch.tensor(np.asarray(data["image"]), dtype=torch.float32, device=self.device)
        self._voxels = None
        # valid (non-zero) pixels per projection, found once (the reference recomputes |proj| > 0 per item, tigre.py:356):
        # after the first pass over the scan an item costs no host synchronisation at all
        self._valid = {}
        self._mask = {}                                  # ptycho mask per projection (train.py:59-60 rebuilds it every step)
        self.shard = (int(shard[0]), int(shard[1]))
        if self.shard[1] > 1 and seed is None:
            seed = 0                                     # ranks must agree on the pixel draw
        self._generator = None
        if seed is not None:
            self._generator = torch.Generator(device=self.device)
            self._generator.manual_seed(int(seed))
        # keyed draws of naf_draw_scan_rays: (seed, item counter) -> one key per item; unseeded datasets start from entropy
        self._draw_seed = int(seed) if seed is not None else int(torch.seed() & 0x7FFFFFFFFFFFFFFF)
        self._draw_count = 0

    @property
    def voxels(self):
        if self._voxels is None:
            self._voxels = torch.tensor(get_voxels(self.geo), dtype=torch.float32, device=self.device)
        return self._voxels

    @property
    def voxel_axes(self):
        """(starts, stops, dims) of the voxel grid: axis k is numpy.linspace(starts[k], stops[k], dims[k]) (tigre.py:388-400)."""
        s = self.geo.sVoxel / 2 - self.geo.dVoxel / 2
        return [-float(v) for v in s], [float(v) for v in s], [int(v) for v in self.geo.nVoxel]

    def __len__(self):
        return self.n_samples

    def set_draw_position(self, items_drawn):
        """Continue the keyed pixel draws after `items_drawn` items (a resumed run: Trainer passes its global step)."""
        self._draw_count = int(items_drawn)

    def _valid_pixels(self, index):
        """Pixels of projection `index` with a non-zero measured value (tigre.py:354-356), as indices inside the projection."""
        index = int(index)
        hit = self._valid.get(index)
        if hit is None:
            flat = self.projs[index].reshape(-1)
            hit = self._valid[index] = torch.nonzero(flat.abs() > 0, as_tuple=False).reshape(-1).contiguous()
        return hit

    def draw_item(self, index):
        """One training item drawn entirely on the device (`naf_draw_scan_rays`): n_rays distinct valid pixels of projection
        `index`, their measured values and rays in one launch -- no randperm sort, no host synchronisation once the
        projection's valid list is cached.  With shard = (rank, world) every rank takes its slice of the SAME draw."""
        index = int(index)
        key = ("flat", index)
        valid = self._valid.get(key)
        if valid is None:
            valid = self._valid[key] = (self._valid_pixels(index) + index * self.raygen.pixels_per_projection).contiguous()
        self._draw_count += 1
        seed = (self._draw_seed * 0x9E3779B97F4A7C15 + self._draw_count * 0xD1B54A32D192ED03) & (2 ** 64 - 1)
        first, count = 0, self.n_rays
        if self.shard[1] > 1:
            from .dist import shard_range
            first, end = shard_range(self.n_rays, *self.shard)
            count = end - first
        pixels, target, rays = self.raygen.draw([valid], self.n_rays, seed, projections=self.projs.reshape(-1), first=first, count=count)
        return pixels - index * self.raygen.pixels_per_projection, target, rays

    def ptycho_mask(self, index, threshold=0.007):
        """`get_ptycho_mask(full_proj[index])` (util.py:196-205), computed once per projection."""
        from .utils import get_ptycho_mask
        index = int(index)
        hit = self._mask.get(index)
        if hit is None:
            hit = self._mask[index] = get_ptycho_mask(self.full_proj[index], threshold=threshold)
        return hit

    def sample_pixels(self, index, n_rays=None, generator=None):
        """`n_rays` distinct valid pixels of projection `index` (tigre.py:356-359: choice without replacement),
        drawn on the device."""
        n_rays = self.n_rays if n_rays is None else n_rays
        valid = self._valid_pixels(index)
        if valid.numel() < n_rays:
            raise ValueError("Cannot take a larger sample than population when 'replace=False'")
        if generator is None:
            generator = self._generator
        perm = torch.randperm(valid.numel(), device=self.device, generator=generator)[:n_rays]
        return valid[perm]

    def __getitem__(self, index):
        if self.type == "train":
            pix, projs, rays = self.draw_item(index)
            W = self.raygen.W
            select_coords = torch.stack([pix // W, pix % W], -1)
            out = {"projs": projs, "rays": rays, "coords": select_coords}
            if self.full_proj is not None:
                out["full_proj"] = self.full_proj[index]
                out["mask"] = self.ptycho_mask(index).reshape(-1)[pix]      # extra key: the mask at the sampled pixels
            return out
        return {"projs": self.projs[index], "rays": self.rays[index]}


def synthetic_scan(n_voxel=64, n_train=50, n_val=8, mode="cone", tilt_angle=0, seed=0, device="cpu", full_proj=False,
                   geometry=None, train_angles=None):
    """A complete in-memory scan with the pickle schema, from the analytic phantom (no data ships with the reference).
    Train angles: linspace(0, pi, n+1)[:-1] (generateData.py:175) unless `train_angles` (radians) is given; val angles:
    sorted U(0, pi) with seed 1.  `geometry` overrides the default scanner dict (e.g. the 256 x 356 laminography detector)."""
    from . import phantom
    from .geometry import angle2pose  # noqa: F401  (documented dependency)

    data = dict(geometry) if geometry is not None else phantom.scan_geometry(n_voxel, mode, tilt_angle)
    tilt_angle = data.get("tilt_angle", tilt_angle)
    if train_angles is not None:
        n_train = len(train_angles)
    geo = ConeGeometry(data)
    table = phantom.ellipsoid_table(seed=seed, extent=float(geo.sVoxel[0]) / 2)
    if tilt_angle:                                       # flat sample for laminography
        table["c"][:, 2] *= 0.3
        table["a"][:, 2] *= 0.3
    rng = np.random.RandomState(1)
    splits = {"train": np.linspace(0, np.pi, n_train + 1)[:-1] if train_angles is None else np.asarray(train_angles, dtype=np.float64),
              "val": np.sort(rng.uniform(0, np.pi, n_val))}
    dev = torch.device(device)
    for name, angles in splits.items():
        H, W = int(geo.nDetector[1]), int(geo.nDetector[0])
        if dev.type == "cuda":
            gen = RayGenerator(geo, angles, dev)
            projs = torch.stack([phantom.line_integrals(gen.rays_for_projection(i), table).reshape(H, W) for i in range(len(angles))])
        else:                                             # CPU construction for tests: the oracle-free torch formula
            projs = torch.stack([phantom.line_integrals(_rays_cpu(geo, a), table).reshape(H, W) for a in angles])
        data[name] = {"angles": angles, "projections": projs.cpu().numpy()}
    data["numTrain"], data["numVal"] = n_train, n_val
    data["image"] = phantom.volume(geo, table, device=dev).cpu().numpy()
    if full_proj:
        data["full_proj"] = data["train"]["projections"].astype(np.complex64)
    return data


def _rays_cpu(geo, angle):
    """Host construction of one projection's rays (only used to synthesise data without a GPU)."""
    from .geometry import angle2pose
    W, H = int(geo.nDetector[0]), int(geo.nDetector[1])
    pose = torch.Tensor(angle2pose(geo.DSO, angle, geo.tilt_angle))
    uu = ((torch.arange(W).float() + 0.5 - W / 2) * float(geo.dDetector[0]) + float(geo.offDetector[0]))[None, :].expand(H, W)
    vv = ((torch.arange(H).float() + 0.5 - H / 2) * float(geo.dDetector[1]) + float(geo.offDetector[1]))[:, None].expand(H, W)
    R, t = pose[:3, :3], pose[:3, 3]
    if geo.mode == "cone":
        dirs = torch.stack([uu / geo.DSD, vv / geo.DSD, torch.ones_like(uu)], -1)
        d = dirs @ R.T
        o = t.expand(d.shape)
    else:
        d = R[:, 2].expand(H, W, 3)
        o = torch.stack([uu, vv, torch.zeros_like(uu)], -1) @ R.T + t
    near, far = get_near_far(geo)
    nf = torch.tensor([near, far], dtype=torch.float32).expand(H, W, 2)
    return torch.cat([o, d, nf], -1).reshape(-1, 8)
