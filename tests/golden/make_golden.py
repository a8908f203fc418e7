"""tests/golden/make_golden.py -- regenerates tests/golden/*.npz (run in the BUILD container only).

Imports the reference's pure-PyTorch modules from /root/reference (render, network, loss, dataset.tigre,
utils.util) with sys.modules stubs for absent third-party packages (cv2, open3d, torchvision, skimage),
runs them on small seeded inputs and stores inputs + expected outputs.  The reference's hash encoder is
CUDA-only and cannot be imported, so wherever an encoder is needed the oracle's
HashEncoderRef is plugged in as the `encoder` argument of the *reference* DensityNetwork.

The .npz files are data (inputs / expected outputs), never reference source.  The GPU box does not have
/root/reference; tests only read the committed fixtures.

Usage:  python tests/golden/make_golden.py
"""
from __future__ import annotations

import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def import_reference():
    _stub("cv2")
    _stub("open3d")
    _stub("torchvision")
    sk = _stub("skimage")
    sk.metrics = _stub("skimage.metrics", structural_similarity=None)
    sys.path.insert(0, REF)
    sys.path.insert(0, REPO)
    import src.render.render  # noqa: F401
    import src.network.network  # noqa: F401
    import src.loss.loss  # noqa: F401
    import src.dataset.tigre  # noqa: F401
    import src.utils.util  # noqa: F401
    return {k: sys.modules[k] for k in ("src.render.render", "src.network.network", "src.loss.loss",
                                        "src.dataset.tigre", "src.utils.util")}


def geometry_fixture(ref):
    tig = ref["src.dataset.tigre"]
    out = {}
    cases = {
        "cone": dict(DSD=1500.0, DSO=1000.0, nDetector=[8, 6], dDetector=[40.0, 50.0], nVoxel=[6, 5, 4],
                     dVoxel=[40.0, 45.0, 50.0], offOrigin=[0, 0, 0], offDetector=[0, 0], accuracy=0.5, mode="cone",
                     filter=None),
        "cone_off": dict(DSD=1400.0, DSO=900.0, nDetector=[5, 7], dDetector=[60.0, 30.0], nVoxel=[4, 4, 6],
                         dVoxel=[50.0, 50.0, 30.0], offOrigin=[12.0, -7.0, 3.0], offDetector=[4.0, -9.0], accuracy=0.5,
                         mode="cone", filter=None),
        "lamino": dict(DSD=1500.0, DSO=1000.0, nDetector=[7, 5], dDetector=[30.0, 30.0], nVoxel=[7, 7, 3],
                       dVoxel=[30.0, 30.0, 30.0], offOrigin=[0, 0, 0], offDetector=[0, 0], accuracy=0.5,
                       mode="parallel", filter=None, tilt_angle=29),
    }
    angles = np.array([0.0, 0.3, 1.1, 2.0, 3.0, 4.5])
    for name, data in cases.items():
        geo = tig.ConeGeometry(data)
        ds = tig.TIGREDataset.__new__(tig.TIGREDataset)
        ds.geo = geo
        out[f"{name}/angles"] = angles
        for k, v in data.items():
            if k not in ("mode", "filter"):
                out[f"{name}/data/{k}"] = np.asarray(v, dtype=np.float64)
        out[f"{name}/mode"] = np.array(data["mode"])
        out[f"{name}/poses"] = np.stack([ds.angle2pose(geo.DSO, a, geo.tilt_angle) for a in angles])
        out[f"{name}/rays"] = ds.get_rays(angles, geo, "cpu").numpy()
        if data["mode"] == "parallel":
            out[f"{name}/rays2"] = ds.get_rays2(angles, geo, "cpu", 4).numpy()
        out[f"{name}/near_far"] = np.array(ds.get_near_far(geo))
        out[f"{name}/voxels"] = ds.get_voxels(geo)
    np.savez_compressed(os.path.join(HERE, "geometry.npz"), **out)


def lamino_chip_fixture(ref):
    """lamino_chip workload (BASELINE.json configs[2]): the reference's own projection angles (data/angles_real.npy,
    187 values in degrees, converted like format_data.py:13-14), parallel beam tilted by 29 degrees, 256 x 356 detector
    (rows x columns, train.py:54).  Rays of three projections from the reference's get_rays2 (the wired path,
    tigre.py:247,463-528), sub-sampled every 17th row / 19th column to keep the fixture small."""
    tig = ref["src.dataset.tigre"]
    angles_deg = np.load(os.path.join(REF, "data", "angles_real.npy"))
    angles = np.deg2rad(angles_deg.astype(np.float64))
    data = dict(DSD=1500.0, DSO=1000.0, nDetector=[356, 256], dDetector=[1.0, 1.0], nVoxel=[256, 256, 256],
                dVoxel=[1.0, 1.0, 1.0], offOrigin=[0, 0, 0], offDetector=[0, 0], accuracy=0.5, mode="parallel", filter=None,
                tilt_angle=29)
    geo = tig.ConeGeometry(data)
    ds = tig.TIGREDataset.__new__(tig.TIGREDataset)
    ds.geo = geo
    pick = np.array([0, 93, 186])
    rays = ds.get_rays2(angles[pick], geo, "cpu", 4).numpy()                # [3, 256, 356, 6]
    rows, cols = np.arange(0, 256, 17), np.arange(0, 356, 19)
    out = {"angles_deg": angles_deg, "pick": pick, "rows": rows, "cols": cols,
           "rays2": rays[:, rows][:, :, cols], "near_far": np.array(ds.get_near_far(geo)),
           "poses": np.stack([ds.angle2pose(geo.DSO, a, geo.tilt_angle) for a in angles[pick]]), "mode": np.array("parallel")}
    for k, v in data.items():
        if k not in ("mode", "filter"):
            out[f"data/{k}"] = np.asarray(v, dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, "lamino_chip.npz"), **out)


class _IdentityEncoder(torch.nn.Module):
    """Feeds pre-computed features straight through (lets the reference MLP be pinned on its own)."""

    def __init__(self, dim):
        super().__init__()
        self.output_dim = dim

    def forward(self, x, size=1):
        return x


def network_fixture(ref):
    net_mod = ref["src.network.network"]
    out = {}
    for tag, kw in {
        "naf": dict(num_layers=4, hidden_dim=32, skips=[2], out_dim=1, last_activation="sigmoid", bound=0.3),
        "deep": dict(num_layers=6, hidden_dim=16, skips=[3], out_dim=1, last_activation="relu", bound=0.3),
        "tanh": dict(num_layers=3, hidden_dim=8, skips=[], out_dim=1, last_activation="tanh", bound=0.3),
    }.items():
        torch.manual_seed(7)
        net = net_mod.DensityNetwork(_IdentityEncoder(32), **kw)
        x = (torch.randn(37, 32) * 0.5).requires_grad_(True)
        y = net(x)
        gy = torch.randn_like(y)
        y.backward(gy)
        out[f"{tag}/x"] = x.detach().numpy()
        out[f"{tag}/y"] = y.detach().numpy()
        out[f"{tag}/gy"] = gy.numpy()
        out[f"{tag}/gx"] = x.grad.numpy()
        for i, lyr in enumerate(net.layers):
            out[f"{tag}/w{i}"] = lyr.weight.detach().numpy()
            out[f"{tag}/b{i}"] = lyr.bias.detach().numpy()
            out[f"{tag}/gw{i}"] = lyr.weight.grad.numpy()
            out[f"{tag}/gb{i}"] = lyr.bias.grad.numpy()
        out[f"{tag}/skips"] = np.array(kw["skips"], dtype=np.int64)
        out[f"{tag}/last_activation"] = np.array(kw["last_activation"])
    np.savez_compressed(os.path.join(HERE, "network.npz"), **out)


def render_fixture(ref):
    from oracle.hashgrid_ref import HashEncoderRef

    rmod, net_mod = ref["src.render.render"], ref["src.network.network"]
    out = {}
    enc_kw = dict(input_dim=3, num_levels=8, level_dim=2, base_resolution=4, log2_hashmap_size=10)
    torch.manual_seed(3)
    enc = HashEncoderRef(**enc_kw)
    enc.embeddings.data.uniform_(-0.5, 0.5)          # large values so the output actually varies
    net = net_mod.DensityNetwork(enc, bound=0.3, num_layers=4, hidden_dim=32, skips=[2], out_dim=1,
                                 last_activation="sigmoid")
    for k, v in enc_kw.items():
        out[f"enc/{k}"] = np.array(v)
    out["enc/embeddings"] = enc.embeddings.detach().numpy()
    for i, lyr in enumerate(net.layers):
        out[f"net/w{i}"] = lyr.weight.detach().numpy()
        out[f"net/b{i}"] = lyr.bias.detach().numpy()

    # rays of a small cone scan: origin on a circle of radius 1, aimed through the +-0.3 cube (some miss it)
    g = torch.Generator().manual_seed(11)
    n = 24
    ang = torch.rand(n, generator=g) * 6.283
    o = torch.stack([torch.cos(ang), torch.sin(ang), torch.zeros(n)], -1)
    tgt = (torch.rand(n, 3, generator=g) - 0.5) * 0.7
    d = tgt - o
    d = d / d[:, :1].abs().clamp(min=0.2)            # un-normalised like the reference cone rays
    rays = torch.cat([o, d, torch.full((n, 1), 0.55), torch.full((n, 1), 1.45)], -1)
    out["rays"] = rays.numpy()
    S = 20
    for tag, perturb in (("det", False), ("jit", True)):
        torch.manual_seed(5)
        t_rand = torch.rand(n, S)
        torch.manual_seed(5)                           # render_chunk draws exactly this tensor (render.py:99)
        ret = rmod.render(rays, net, None, S, 0, perturb, 4096, 0.0)
        out[f"{tag}/t_rand"] = t_rand.numpy()
        out[f"{tag}/acc"] = ret["acc"].detach().numpy()
        out[f"{tag}/pts"] = ret["pts"].detach().numpy()
        out[f"{tag}/tv_loss"] = ret["tv_loss"].detach().numpy()
        # gradients of the reference chunked loss wrt table + MLP
        net.zero_grad()
        target = torch.linspace(0.05, 0.4, n)
        loss = ((ret["acc"] - target) ** 2).mean()
        loss.backward()
        out[f"{tag}/target"] = target.numpy()
        out[f"{tag}/loss"] = loss.detach().numpy()
        out[f"{tag}/g_embeddings"] = enc.embeddings.grad.numpy().copy()
        for i, lyr in enumerate(net.layers):
            out[f"{tag}/gw{i}"] = lyr.weight.grad.numpy().copy()
            out[f"{tag}/gb{i}"] = lyr.bias.grad.numpy().copy()
    # chunked render must equal the unchunked one (render.py:56-63)
    ret = rmod.render(rays, net, None, S, 0, False, 4096, 0.0, chunk_size=10)
    out["det/acc_chunked"] = ret["acc"].detach().numpy()

    # raw2outputs / sample_pdf on their own
    torch.manual_seed(9)
    raw = torch.rand(6, S, 1)
    z = torch.sort(torch.rand(6, S) + 0.5, -1)[0]
    rd = torch.randn(6, 3)
    acc, w = rmod.raw2outputs(raw, z, rd, 0.0)
    out["r2o/raw"], out["r2o/z"], out["r2o/d"] = raw.numpy(), z.numpy(), rd.numpy()
    out["r2o/acc"], out["r2o/weights"] = acc.numpy(), w.numpy()
    mid = 0.5 * (z[..., 1:] + z[..., :-1])
    out["pdf/bins"], out["pdf/weights"] = mid.numpy(), w[..., 1:-1].numpy()
    out["pdf/samples_det"] = rmod.sample_pdf(mid, w[..., 1:-1], 12, det=True).numpy()

    # coarse + fine pass, deterministic (perturb == 0 -> det sampling), shared encoder (trainer.py:50)
    torch.manual_seed(13)
    net_fine = net_mod.DensityNetwork(enc, bound=0.3, num_layers=4, hidden_dim=32, skips=[2], out_dim=1,
                                      last_activation="sigmoid")
    for i, lyr in enumerate(net_fine.layers):
        out[f"net_fine/w{i}"] = lyr.weight.detach().numpy()
        out[f"net_fine/b{i}"] = lyr.bias.detach().numpy()
    ret = rmod.render(rays, net, net_fine, S, 8, 0.0, 4096, 0.0)
    for k in ("acc", "acc0", "weights0"):
        out[f"fine/{k}"] = ret[k].detach().numpy()
    np.savez_compressed(os.path.join(HERE, "render.npz"), **out)


def loss_metrics_fixture(ref):
    lmod, umod = ref["src.loss.loss"], ref["src.utils.util"]
    out = {}
    torch.manual_seed(21)
    x, y = torch.rand(50), torch.rand(50)
    loss = {"loss": 0.0}
    lmod.calc_mse_loss(loss, x[:30], y[:30])
    lmod.calc_mse_loss(loss, x[30:], y[30:])
    out["mse/x"], out["mse/y"] = x.numpy(), y.numpy()
    out["mse/loss"], out["mse/loss_mse"] = loss["loss"].numpy(), loss["loss_mse"].numpy()

    hr = torch.complex(torch.randn(12, 9) * 0.01, torch.randn(12, 9) * 0.01)
    hr[3:6, 2:5] = 0
    out["mask/hr"] = hr.numpy()
    out["mask/mask"] = umod.get_ptycho_mask(hr.clone(), 0.007).numpy()

    a, b = torch.rand(5, 6, 7), torch.rand(5, 6, 7)
    out["psnr/a"], out["psnr/b"] = a.numpy(), b.numpy()
    out["psnr/psnr_3d"] = np.asarray(umod.get_psnr_3d(a, b))
    out["psnr/mse"] = umod.get_mse(a, b).numpy()
    pa, pb = torch.rand(8, 9), torch.rand(8, 9)
    out["psnr/pa"], out["psnr/pb"] = pa.numpy(), pb.numpy()
    out["psnr/psnr_2d"] = umod.get_psnr(pa.to(torch.complex64), pb.to(torch.complex64)).numpy()
    np.savez_compressed(os.path.join(HERE, "loss_metrics.npz"), **out)


if __name__ == "__main__":
    if not os.path.isdir(REF):
        raise SystemExit("reference tree not present; fixtures can only be regenerated in the build container")
    ref = import_reference()
    todo = {"geometry": geometry_fixture, "network": network_fixture, "render": render_fixture,
            "loss_metrics": loss_metrics_fixture, "lamino_chip": lamino_chip_fixture}
    for name in (sys.argv[1:] or list(todo)):               # optional: names of the fixtures to regenerate
        todo[name](ref)
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))
