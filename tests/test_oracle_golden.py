"""The oracle restatements checked against golden vectors captured from the imported reference modules
(tests/golden/make_golden.py).  fp32 tolerances are stated per test."""
import numpy as np
import torch

from oracle import geometry_ref as G
from oracle import loss_metrics_ref as LM
from oracle import network_ref as N
from oracle import render_ref as R
from oracle.hashgrid_ref import HashEncoderRef


def _geo(g, name):
    data = {k.split("/")[-1]: g[k] for k in g.files if k.startswith(f"{name}/data/")}
    data = {k: (float(v) if v.ndim == 0 else v) for k, v in data.items()}
    data["mode"] = str(g[f"{name}/mode"])
    return G.GeometryRef(data)


def test_geometry_matches_reference(golden):
    g = golden("geometry")
    for name in ("cone", "cone_off", "lamino"):
        geo = _geo(g, name)
        angles = g[f"{name}/angles"]
        poses = np.stack([G.angle2pose(geo.DSO, a, geo.tilt_angle) for a in angles])
        np.testing.assert_allclose(poses, g[f"{name}/poses"], rtol=0, atol=1e-15)
        rays = G.get_rays(angles, geo).numpy()
        np.testing.assert_allclose(rays, g[f"{name}/rays"], rtol=1e-6, atol=1e-7)
        if f"{name}/rays2" in g.files:                     # get_rays2 == parallel half of get_rays
            np.testing.assert_allclose(rays, g[f"{name}/rays2"], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(np.array(G.get_near_far(geo)), g[f"{name}/near_far"], rtol=0, atol=1e-15)
        np.testing.assert_allclose(G.get_voxels(geo), g[f"{name}/voxels"], rtol=0, atol=1e-15)


def _net_params(g, prefix, n):
    ws = [torch.from_numpy(g[f"{prefix}/w{i}"]) for i in range(n)]
    bs = [torch.from_numpy(g[f"{prefix}/b{i}"]) for i in range(n)]
    return ws, bs


def test_mlp_matches_reference(golden):
    g = golden("network")
    for tag, n in (("naf", 4), ("deep", 6), ("tanh", 3)):
        ws, bs = _net_params(g, tag, n)
        for t in ws + bs:
            t.requires_grad_(True)
        x = torch.from_numpy(g[f"{tag}/x"]).requires_grad_(True)
        y = N.mlp_forward(x, ws, bs, tuple(g[f"{tag}/skips"].tolist()), str(g[f"{tag}/last_activation"]))
        np.testing.assert_allclose(y.detach().numpy(), g[f"{tag}/y"], rtol=1e-6, atol=1e-7)
        y.backward(torch.from_numpy(g[f"{tag}/gy"]))
        np.testing.assert_allclose(x.grad.numpy(), g[f"{tag}/gx"], rtol=1e-5, atol=1e-7)
        for i in range(n):
            np.testing.assert_allclose(ws[i].grad.numpy(), g[f"{tag}/gw{i}"], rtol=1e-5, atol=1e-6)
            np.testing.assert_allclose(bs[i].grad.numpy(), g[f"{tag}/gb{i}"], rtol=1e-5, atol=1e-6)


def _render_net(g, prefix="net"):
    enc = HashEncoderRef(**{k: int(g[f"enc/{k}"]) for k in
                            ("input_dim", "num_levels", "level_dim", "base_resolution", "log2_hashmap_size")})
    enc.embeddings.data.copy_(torch.from_numpy(g["enc/embeddings"]))
    net = N.DensityNetworkRef(enc, bound=0.3, num_layers=4, hidden_dim=32, skips=(2,), out_dim=1)
    for i, lyr in enumerate(net.layers):
        lyr.weight.data.copy_(torch.from_numpy(g[f"{prefix}/w{i}"]))
        lyr.bias.data.copy_(torch.from_numpy(g[f"{prefix}/b{i}"]))
    return enc, net


def test_render_matches_reference(golden):
    g = golden("render")
    enc, net = _render_net(g)
    rays = torch.from_numpy(g["rays"])
    S = g["det/t_rand"].shape[1]
    for tag, perturb in (("det", False), ("jit", True)):
        net.zero_grad()
        ret = R.render(rays, net, None, S, 0, perturb, 4096, 0.0, t_rand=torch.from_numpy(g[f"{tag}/t_rand"]))
        np.testing.assert_allclose(ret["pts"].detach().numpy(), g[f"{tag}/pts"], rtol=0, atol=1e-7)
        np.testing.assert_allclose(ret["acc"].detach().numpy(), g[f"{tag}/acc"], rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(ret["tv_loss"].detach().numpy(), g[f"{tag}/tv_loss"], rtol=1e-5)
        loss = ((ret["acc"] - torch.from_numpy(g[f"{tag}/target"])) ** 2).mean()
        np.testing.assert_allclose(loss.item(), g[f"{tag}/loss"], rtol=1e-5)
        loss.backward()
        ge = g[f"{tag}/g_embeddings"]
        np.testing.assert_allclose(enc.embeddings.grad.numpy(), ge, rtol=1e-4, atol=1e-6 * np.abs(ge).max())
        for i, lyr in enumerate(net.layers):
            np.testing.assert_allclose(lyr.weight.grad.numpy(), g[f"{tag}/gw{i}"], rtol=1e-4, atol=1e-7)
    ret = R.render(rays, net, None, S, 0, False, 4096, 0.0, chunk_size=10)
    np.testing.assert_allclose(ret["acc"].detach().numpy(), g["det/acc_chunked"], rtol=1e-5, atol=1e-7)


def test_raw2outputs_and_sample_pdf(golden):
    g = golden("render")
    acc, w = R.raw2outputs(torch.from_numpy(g["r2o/raw"]), torch.from_numpy(g["r2o/z"]), torch.from_numpy(g["r2o/d"]))
    np.testing.assert_allclose(acc.numpy(), g["r2o/acc"], rtol=1e-6)
    np.testing.assert_allclose(w.numpy(), g["r2o/weights"], rtol=1e-6, atol=1e-12)
    s = R.sample_pdf(torch.from_numpy(g["pdf/bins"]), torch.from_numpy(g["pdf/weights"]), 12, det=True)
    np.testing.assert_allclose(s.numpy(), g["pdf/samples_det"], rtol=1e-6)


def test_fine_pass_matches_reference(golden):
    g = golden("render")
    enc, net = _render_net(g)
    net_fine = N.DensityNetworkRef(enc, bound=0.3, num_layers=4, hidden_dim=32, skips=(2,), out_dim=1)
    for i, lyr in enumerate(net_fine.layers):
        lyr.weight.data.copy_(torch.from_numpy(g[f"net_fine/w{i}"]))
        lyr.bias.data.copy_(torch.from_numpy(g[f"net_fine/b{i}"]))
    S = g["det/t_rand"].shape[1]
    with torch.no_grad():
        ret = R.render(torch.from_numpy(g["rays"]), net, net_fine, S, 8, 0.0, 4096, 0.0)
    np.testing.assert_allclose(ret["acc0"].numpy(), g["fine/acc0"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(ret["weights0"].numpy(), g["fine/weights0"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(ret["acc"].numpy(), g["fine/acc"], rtol=1e-4, atol=1e-6)


def test_loss_and_metrics(golden):
    g = golden("loss_metrics")
    x, y = torch.from_numpy(g["mse/x"]), torch.from_numpy(g["mse/y"])
    loss = {"loss": 0.0}
    LM.calc_mse_loss(loss, x[:30], y[:30])
    LM.calc_mse_loss(loss, x[30:], y[30:])
    np.testing.assert_allclose(loss["loss"].numpy(), g["mse/loss"], rtol=1e-6)
    np.testing.assert_allclose(loss["loss_mse"].numpy(), g["mse/loss_mse"], rtol=1e-6)
    np.testing.assert_allclose(LM.chunked_masked_loss(y, x, None, 30).numpy(), g["mse/loss"], rtol=1e-6)
    assert np.array_equal(LM.get_ptycho_mask(torch.from_numpy(g["mask/hr"])).numpy(), g["mask/mask"])
    a, b = torch.from_numpy(g["psnr/a"]), torch.from_numpy(g["psnr/b"])
    np.testing.assert_allclose(LM.get_psnr_3d(a, b), g["psnr/psnr_3d"], rtol=1e-12)
    np.testing.assert_allclose(LM.get_mse(a, b).numpy(), g["psnr/mse"], rtol=1e-6)
    pa, pb = torch.from_numpy(g["psnr/pa"]), torch.from_numpy(g["psnr/pb"])
    np.testing.assert_allclose(LM.get_psnr(pa.to(torch.complex64), pb.to(torch.complex64)).numpy(), g["psnr/psnr_2d"],
                               rtol=1e-5)
