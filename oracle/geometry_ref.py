"""oracle/geometry_ref.py -- TEST INFRASTRUCTURE ONLY.

CPU restatement of the reference scan geometry / ray generation, src/dataset/tigre.py:
  * ConeGeometry (mm -> m) ............ tigre.py:183-217
  * angle2pose (+ tilt) ............... tigre.py:530-572
  * get_rays (cone and parallel) ...... tigre.py:402-456   (get_rays2 :463-528 is the parallel half)
  * get_near_far ...................... tigre.py:575-586
  * get_voxels ........................ tigre.py:388-400
Pinned against golden vectors captured from the imported reference class (tests/golden/make_golden.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import numpy as np
import torch


class GeometryRef:
    def __init__(self, data):
        mm = 1000.0
        self.DSD = data["DSD"] / mm
        self.DSO = data["DSO"] / mm
        self.nDetector = np.array(data["nDetector"])
        self.dDetector = np.array(data["dDetector"]) / mm
        self.sDetector = self.nDetector * self.dDetector
        self.nVoxel = np.array(data["nVoxel"])
        self.dVoxel = np.array(data["dVoxel"]) / mm
        self.sVoxel = self.nVoxel * self.dVoxel
        self.offOrigin = np.array(data["offOrigin"]) / mm
        self.offDetector = np.array(data["offDetector"]) / mm
        self.accuracy = data.get("accuracy", 0.5)
        self.mode = data["mode"]
        self.filter = data.get("filter")
        self.tilt_angle = data.get("tilt_angle", 0)


def _rot_x(phi):
    c, s = np.cos(phi), np.sin(phi)
    return np.array([[1.0, 0.0, 0.0], [0.0, c, -s], [0.0, s, c]])


def _rot_z(phi):
    c, s = np.cos(phi), np.sin(phi)
    return np.array([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]])


def angle2pose(DSO, angle, tilt_deg=0.0):
    """tigre.py:530-572: rot = Rz(angle) Rz(pi/2) Rx(-pi/2) Rx_cw(tilt); trans = DSO*[cos a, sin a, tan tilt]."""
    tilt = np.radians(tilt_deg)
    rot = _rot_z(angle) @ _rot_z(np.pi / 2) @ _rot_x(-np.pi / 2)
    rot = rot @ _rot_x(-tilt)                       # clockwise about x == Rx(-tilt)
    T = np.eye(4)
    T[:3, :3] = rot
    T[:3, 3] = [DSO * np.cos(angle), DSO * np.sin(angle), DSO * np.tan(tilt)]
    return T


def detector_uv(geo):
    """tigre.py:423-429: uu varies along columns, vv along rows; both [H, W] float32."""
    W, H = int(geo.nDetector[0]), int(geo.nDetector[1])
    cols = torch.linspace(0, W - 1, W)
    rows = torch.linspace(0, H - 1, H)
    uu = ((cols + 0.5 - W / 2) * geo.dDetector[0] + geo.offDetector[0])[None, :].expand(H, W)
    vv = ((rows + 0.5 - H / 2) * geo.dDetector[1] + geo.offDetector[1])[:, None].expand(H, W)
    return uu, vv


def get_rays(angles, geo):
    """-> float32 [N, H, W, 6] (origin, direction); cone dirs are NOT normalised (tigre.py:434-437)."""
    uu, vv = detector_uv(geo)
    out = []
    for a in angles:
        pose = torch.Tensor(angle2pose(geo.DSO, a, geo.tilt_angle))
        R, t = pose[:3, :3], pose[:3, 3]
        if geo.mode == "cone":
            dirs = torch.stack([uu / geo.DSD, vv / geo.DSD, torch.ones_like(uu)], -1)
            d = torch.matmul(R, dirs[..., None]).squeeze(-1)
            o = t.expand(d.shape)
        elif geo.mode == "parallel":
            dirs = torch.stack([torch.zeros_like(uu), torch.zeros_like(uu), torch.ones_like(uu)], -1)
            d = torch.matmul(R, dirs[..., None]).squeeze(-1)
            o = torch.matmul(R, torch.stack([uu, vv, torch.zeros_like(uu)], -1)[..., None]).squeeze(-1) + t.expand(d.shape)
        else:
            raise NotImplementedError("Unknown CT scanner type!")
        out.append(torch.cat([o, d], -1))
    return torch.stack(out, 0)


def get_near_far(geo, tolerance=0.005):
    """tigre.py:575-586 (xy corners only; tilt ignored)."""
    ox, oy = geo.offOrigin[0], geo.offOrigin[1]
    hx, hy = geo.sVoxel[0] / 2, geo.sVoxel[1] / 2
    dmax = max(np.linalg.norm([ox + sx * hx, oy + sy * hy]) for sx in (-1, 1) for sy in (-1, 1))
    near = max(0.0, geo.DSO - dmax - tolerance)
    far = min(geo.DSO * 2, geo.DSO + dmax + tolerance)
    return near, far


def get_voxels(geo):
    """tigre.py:388-400: voxel-centre grid [n1,n2,n3,3], 'ij' indexing."""
    n1, n2, n3 = (int(v) for v in geo.nVoxel)
    s1, s2, s3 = geo.sVoxel / 2 - geo.dVoxel / 2
    g = np.meshgrid(np.linspace(-s1, s1, n1), np.linspace(-s2, s2, n2), np.linspace(-s3, s3, n3), indexing="ij")
    return np.stack(g, -1)
