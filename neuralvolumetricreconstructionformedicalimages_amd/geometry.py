"""Scan geometry and ray generation: host-side mirror of the geometry half of reference src/dataset/tigre.py.

  * `ConeGeometry(data)`  -- same fields and mm->m conversion as tigre.py:183-217
  * `angle2pose`, `get_near_far`, `get_voxels` -- tigre.py:530-572, 575-586, 388-400 (float64 host maths, init only)
  * `RayGenerator` -- replaces the precomputed `rays[N,H,W,8]` tensor (tigre.py:247-255): poses [N,3,4] live in HBM and
    rays are produced on demand by `naf_generate_rays` for any list of pixels (cone AND parallel/tilted geometry,
    i.e. both `get_rays` tigre.py:402-456 and `get_rays2` :463-528).
"""
from __future__ import annotations

import numpy as np
import torch

from . import _abi


class ConeGeometry(object):
    """Cone beam CT geometry. Lengths are converted from millimetres to metres (tigre.py:183-217)."""

    def __init__(self, data):
        self.DSD = data["DSD"] / 1000
        self.DSO = data["DSO"] / 1000
        self.nDetector = np.array(data["nDetector"])
        self.dDetector = np.array(data["dDetector"]) / 1000
        self.sDetector = self.nDetector * self.dDetector
        self.nVoxel = np.array(data["nVoxel"])
        self.dVoxel = np.array(data["dVoxel"]) / 1000
        self.sVoxel = self.nVoxel * self.dVoxel
        self.offOrigin = np.array(data["offOrigin"]) / 1000
        self.offDetector = np.array(data["offDetector"]) / 1000
        self.accuracy = data.get("accuracy", 0.5)
        self.mode = data["mode"]
        self.filter = data.get("filter")
        self.magnification = 1
        self.tilt_angle = data.get("tilt_angle", 0)      # degrees


def _about_x(phi):
    return np.array([[1.0, 0.0, 0.0], [0.0, np.cos(phi), -np.sin(phi)], [0.0, np.sin(phi), np.cos(phi)]])


def _about_z(phi):
    return np.array([[np.cos(phi), -np.sin(phi), 0.0], [np.sin(phi), np.cos(phi), 0.0], [0.0, 0.0, 1.0]])


def angle2pose(DSO, angle, tilt_angle=0):
    """4x4 source pose for projection `angle` (rad) and laminography `tilt_angle` (deg); tigre.py:530-572.

    rotation = Rz(angle) . Rz(+90 deg) . Rx(-90 deg) . Rx(-tilt)   (the tilt is a clockwise turn about x);
    the source sits at DSO * [cos a, sin a, tan tilt]."""
    tilt = np.radians(tilt_angle)
    pose = np.eye(4)
    pose[:3, :3] = ((_about_z(angle) @ _about_z(np.pi / 2)) @ _about_x(-np.pi / 2)) @ _about_x(-tilt)
    pose[:3, 3] = [DSO * np.cos(angle), DSO * np.sin(angle), DSO * np.tan(tilt)]
    return pose


def get_near_far(geo, tolerance=0.005):
    """tigre.py:575-586: distance window around the xy footprint of the volume (tilt ignored)."""
    corners = [np.linalg.norm([geo.offOrigin[0] + sx * geo.sVoxel[0] / 2, geo.offOrigin[1] + sy * geo.sVoxel[1] / 2])
               for sx in (-1, 1) for sy in (-1, 1)]
    dist_max = np.max(corners)
    near = np.max([0, geo.DSO - dist_max - tolerance])
    far = np.min([geo.DSO * 2, geo.DSO + dist_max + tolerance])
    return near, far


def get_voxels(geo):
    """Voxel-centre coordinates [n1,n2,n3,3] (tigre.py:388-400)."""
    n1, n2, n3 = (int(v) for v in geo.nVoxel)
    s1, s2, s3 = geo.sVoxel / 2 - geo.dVoxel / 2
    xyz = np.meshgrid(np.linspace(-s1, s1, n1), np.linspace(-s2, s2, n2), np.linspace(-s3, s3, n3), indexing="ij")
    return np.asarray(xyz).transpose([1, 2, 3, 0])


class RayGenerator:
    """Poses of a scan on the device + on-demand ray generation for arbitrary pixels."""

    def __init__(self, geo, angles, device):
        self.geo = geo
        self.device = torch.device(device)
        self.n_projections = len(angles)
        self.W, self.H = int(geo.nDetector[0]), int(geo.nDetector[1])
        self.near, self.far = get_near_far(geo)
        if geo.mode not in ("cone", "parallel"):
            raise NotImplementedError("Unknown CT scanner type!")
        poses = np.stack([angle2pose(geo.DSO, a, geo.tilt_angle)[:3, :4] for a in angles])
        self.poses = torch.Tensor(poses).contiguous().to(self.device)       # float64 -> float32 like tigre.py:419

    @property
    def pixels_per_projection(self):
        return self.W * self.H

    def _call(self, pixels, first, n, out):
        g = self.geo
        if out is None:
            out = torch.empty(n, 8, device=self.device, dtype=torch.float32)
        _abi.check(_abi.lib().naf_generate_rays(
            _abi.ptr(self.poses), _abi.ptr(pixels), int(first), _abi.ptr(out), int(n), self.n_projections, self.W, self.H,
            float(g.dDetector[0]), float(g.dDetector[1]), float(g.offDetector[0]), float(g.offDetector[1]), float(g.DSD),
            float(self.near), float(self.far), int(g.mode == "parallel"), _abi.stream_ptr()), "generate_rays")
        return out

    def draw(self, valid_lists, rays_per_list, seed, projections=None, first=0, count=None, rays_out=None, want_pixels=True,
             target_out=None):
        """`rays_per_list` distinct entries of each list in `valid_lists` (int64 device tensors of flat pixel indices),
        drawn on the device by `naf_draw_scan_rays` -> (pixels [n] or None, target [n] or None, rays [n, 8]).
        `first` / `count` select a slice of the len(valid_lists) * rays_per_list draws (data-parallel shards)."""
        g = self.geo
        k = len(valid_lists)
        total = k * int(rays_per_list)
        count = total - first if count is None else int(count)
        st = _abi.ScanDraw()
        st.n_segments, st.rays_per_segment = k, int(rays_per_list)
        for j, v in enumerate(valid_lists):
            if v.dtype != torch.int64 or not v.is_cuda or not v.is_contiguous():
                raise RuntimeError("draw: valid-pixel lists must be contiguous int64 device tensors")
            if v.numel() < rays_per_list:
                raise ValueError("Cannot take a larger sample than population when 'replace=False'")      # tigre.py:357
            st.valid[j], st.n_valid[j] = v.data_ptr(), v.numel()
        rays = rays_out if rays_out is not None else torch.empty(count, 8, device=self.device, dtype=torch.float32)
        pixels = torch.empty(count, device=self.device, dtype=torch.int64) if want_pixels else None
        target = None
        if projections is not None:
            target = target_out if target_out is not None else torch.empty(count, device=self.device, dtype=torch.float32)
        import ctypes
        _abi.check(_abi.lib().naf_draw_scan_rays(
            ctypes.byref(st), _abi.ptr(self.poses), _abi.ptr(projections), _abi.ptr(pixels), _abi.ptr(target), _abi.ptr(rays),
            int(first), count, self.n_projections, self.W, self.H, float(g.dDetector[0]), float(g.dDetector[1]),
            float(g.offDetector[0]), float(g.offDetector[1]), float(g.DSD), float(self.near), float(self.far),
            int(g.mode == "parallel"), int(seed) & (2 ** 64 - 1), _abi.stream_ptr()), "draw_scan_rays")
        return pixels, target, rays

    def plan_draw(self, valid_lists, rays_per_list, seed, projections, rays_out, target_out, first=0, count=None):
        """The arguments of `draw` as a `_abi.NextDraw` (struct naf_next_draw) instead of a launch: `NAFEngine.train_step(...,
        next_draw=plan)` lets step k carry the pixel draw of step k + 1 in spare workgroups of its own launches.  `rays_out` [count, 8]
        and `target_out` [count] must not be the buffers step k reads (double-buffer them).  `plan.launch()` runs it stand-alone."""
        g = self.geo
        k = len(valid_lists)
        count = k * int(rays_per_list) - first if count is None else int(count)
        nd = _abi.NextDraw()
        nd.draw.n_segments, nd.draw.rays_per_segment = k, int(rays_per_list)
        for j, v in enumerate(valid_lists):
            if v.dtype != torch.int64 or not v.is_cuda or not v.is_contiguous():
                raise RuntimeError("draw: valid-pixel lists must be contiguous int64 device tensors")
            if v.numel() < rays_per_list:
                raise ValueError("Cannot take a larger sample than population when 'replace=False'")      # tigre.py:357
            nd.draw.valid[j], nd.draw.n_valid[j] = v.data_ptr(), v.numel()
        nd.poses, nd.projections = self.poses.data_ptr(), projections.data_ptr()
        nd.pixels, nd.target, nd.rays = None, target_out.data_ptr(), rays_out.data_ptr()
        nd.first_draw, nd.n_draws, nd.n_projections, nd.det_w, nd.det_h = int(first), count, self.n_projections, self.W, self.H
        nd.du, nd.dv, nd.ou, nd.ov = float(g.dDetector[0]), float(g.dDetector[1]), float(g.offDetector[0]), float(g.offDetector[1])
        nd.DSD, nd.near, nd.far, nd.parallel = float(g.DSD), float(self.near), float(self.far), int(g.mode == "parallel")
        nd.seed = int(seed) & (2 ** 64 - 1)
        nd._keep = (valid_lists, projections, rays_out, target_out)        # the struct holds raw pointers

        def launch():
            import ctypes
            _abi.check(_abi.lib().naf_draw_scan_rays(
                ctypes.byref(nd.draw), nd.poses, nd.projections, None, nd.target, nd.rays, nd.first_draw, nd.n_draws, nd.n_projections,
                nd.det_w, nd.det_h, nd.du, nd.dv, nd.ou, nd.ov, nd.DSD, nd.near, nd.far, nd.parallel, nd.seed, _abi.stream_ptr()),
                "draw_scan_rays")
        nd.launch = launch
        return nd

    def rays_for_pixels(self, pixels, out=None):
        """pixels: int64 [n] flat indices proj*H*W + row*W + col  ->  rays [n,8]."""
        pixels = pixels.contiguous().to(torch.int64)
        return self._call(pixels, 0, pixels.numel(), out)

    def rays_for_projection(self, index, out=None):
        """All H*W rays of one projection, row-major like `rays[index]` of the reference -> [H*W, 8]."""
        n = self.pixels_per_projection
        return self._call(None, index * n, n, out)
