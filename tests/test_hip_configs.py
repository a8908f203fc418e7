"""GPU parity on the BASELINE.json configurations beyond chest_50: the reference's ray-generation goldens (non-square
detectors, detector offsets, tilted parallel beam), the lamino_chip workload (256 x 356 detector, the reference's own
187 angles, 29 degree tilt), the abdomen shard shape (S = 576), the jaw shape (S = 320), and the chest-size
bf16-vs-fp32 reconstruction PSNR.  Everything goes through the C ABI; the oracle is the checker."""
import numpy as np
import pytest
import torch

from _naf_helpers import crossing_rays, golden_geometry, naf_pair, rel_l2

pytestmark = pytest.mark.gpu

ULP = 1.1920929e-07        # one unit in the last place of 1.0f: coordinates here are O(1) metres


def _raygen(data, angles):
    from neuralvolumetricreconstructionformedicalimages_amd.geometry import ConeGeometry, RayGenerator
    return RayGenerator(ConeGeometry(data), angles, torch.device("cuda"))


# ---- G3 / G4: naf_generate_rays against the reference's get_rays / get_rays2 -------------------------------------
@pytest.mark.parametrize("name", ["cone", "cone_off", "lamino"])
def test_generate_rays_matches_reference_golden(golden, name):
    """Full-projection and pixel-list modes on the 8x6 / 5x7 (offset detector) / 7x5 (tilted parallel) goldens that the
    reference's own TIGREDataset.get_rays (tigre.py:402-456) and get_rays2 (:463-528) produced: the detector is not
    square, so a transposed (row, column) convention (tigre.py:427-428) would fail here."""
    g = golden("geometry")
    want = g[f"{name}/rays"]                                           # [N, H, W, 6]
    N, H, W, _ = want.shape
    gen = _raygen(golden_geometry(g, name), g[f"{name}/angles"])
    assert (gen.H, gen.W) == (H, W)
    near, far = g[f"{name}/near_far"]
    for i in range(N):
        got = gen.rays_for_projection(i).cpu().numpy().reshape(H, W, 8)
        # the kernel evaluates R.[u/DSD, v/DSD, 1] in its own summation order: a couple of ulps at most
        np.testing.assert_allclose(got[..., :6], want[i], rtol=3 * ULP, atol=2 * ULP)
        assert np.all(got[..., 6] == np.float32(near)) and np.all(got[..., 7] == np.float32(far))
        if name == "lamino":
            np.testing.assert_allclose(got[..., :6], g["lamino/rays2"][i], rtol=3 * ULP, atol=2 * ULP)
    # pixel-list mode: every pixel of every projection once, shuffled
    perm = torch.randperm(N * H * W, generator=torch.Generator().manual_seed(1))
    got = gen.rays_for_pixels(perm.cuda()).cpu().numpy()
    np.testing.assert_allclose(got[:, :6], want.reshape(-1, 6)[perm.numpy()], rtol=3 * ULP, atol=2 * ULP)
    # the same list through the dataset-style (projection, row, column) view
    from neuralvolumetricreconstructionformedicalimages_amd.dataset import _LazyRays
    lazy = _LazyRays(gen)
    rows, cols = torch.tensor([0, H - 1, 1]), torch.tensor([W - 1, 0, 2])
    np.testing.assert_allclose(lazy[N - 1, rows, cols].cpu().numpy()[:, :6], want[N - 1][rows, cols], rtol=3 * ULP, atol=2 * ULP)


def _lamino_chip(golden):
    g = golden("lamino_chip")
    data = golden_geometry(g)
    data["tilt_angle"] = float(data["tilt_angle"])
    angles = np.deg2rad(g["angles_deg"].astype(np.float64))           # format_data.py:13-14
    return g, data, angles


def test_lamino_chip_rays_match_reference_get_rays2(golden):
    """The real laminography scan: 187 angles of data/angles_real.npy, 256 rows x 356 columns, 29 degree tilt."""
    g, data, angles = _lamino_chip(golden)
    gen = _raygen(data, angles)
    assert (gen.n_projections, gen.H, gen.W) == (187, 256, 356)
    rows, cols = torch.from_numpy(g["rows"]), torch.from_numpy(g["cols"])
    rr, cc = torch.meshgrid(rows, cols, indexing="ij")
    for k, proj in enumerate(g["pick"]):
        pix = int(proj) * gen.pixels_per_projection + rr.reshape(-1) * gen.W + cc.reshape(-1)
        got = gen.rays_for_pixels(pix.cuda()).cpu().numpy().reshape(len(rows), len(cols), 8)
        np.testing.assert_allclose(got[..., :6], g["rays2"][k], rtol=3 * ULP, atol=2 * ULP)
        assert np.all(got[..., 6] == np.float32(g["near_far"][0])) and np.all(got[..., 7] == np.float32(g["near_far"][1]))
        full = gen.rays_for_projection(int(proj)).reshape(gen.H, gen.W, 8)[rows][:, cols].cpu().numpy()
        assert np.array_equal(full, got)                               # both addressing modes, same bits


def test_lamino_chip_workload_fused_forward_backward_vs_oracle(golden):
    """lamino_chip.yaml shapes (S = 192, T = 2^19, fp32 parity mode) on rays of the tilted parallel scan: projection,
    MLP gradients and table gradient (atomic and binned scatter) against the oracle."""
    from neuralvolumetricreconstructionformedicalimages_amd import fused
    from oracle import render_ref as R
    _, data, angles = _lamino_chip(golden)
    gen = _raygen(data, angles)
    n, S = 96, 192
    pix = torch.randint(0, gen.n_projections * gen.pixels_per_projection, (n,), generator=torch.Generator().manual_seed(2))
    rays = gen.rays_for_pixels(pix.cuda())
    net, ref = naf_pair(seed=21, log2T=19, scale=0.3)
    t_rand = torch.rand(n, S, generator=torch.Generator().manual_seed(3))
    target = torch.rand(n, generator=torch.Generator().manual_seed(4)) * 0.3
    acc_ref = R.render(rays.cpu(), ref, None, S, 0, True, 1 << 20, 0.0, t_rand=t_rand)["acc"]
    ((acc_ref - target) ** 2).mean().backward()
    ge = ref.encoder.embeddings.grad.numpy()
    for mode in (1, 2):
        net.zero_grad()
        with fused.scatter_mode(mode):
            acc = fused.fused_render(rays, net, S, True, t_rand=t_rand.cuda())
            ((acc - target.cuda()) ** 2).mean().backward()
        assert rel_l2(acc.detach().cpu().numpy(), acc_ref.detach().numpy()) < 1e-4        # north_star bar
        assert rel_l2(net.encoder.embeddings.grad.cpu().numpy(), ge) < 2e-4, mode
        for a, b in zip(net.layers, ref.layers):
            assert rel_l2(a.weight.grad.cpu().numpy(), b.weight.grad.numpy()) < 2e-4, mode


def test_lamino_training_matches_oracle_psnr(golden):
    """A short training run on a small tilted parallel scan with the lamino_chip detector aspect (44 columns x 32 rows):
    the fused engine and the oracle take the same 30 Adam steps; the reconstructed volumes agree to 0.1 dB."""
    from neuralvolumetricreconstructionformedicalimages_amd.dataset import TIGREDataset, synthetic_scan
    from neuralvolumetricreconstructionformedicalimages_amd.engine import NAFEngine
    from neuralvolumetricreconstructionformedicalimages_amd.utils import get_psnr_3d
    from oracle import render_ref as R
    from oracle.loss_metrics_ref import get_psnr_3d as psnr_ref
    g, _, angles = _lamino_chip(golden)
    geom = dict(DSD=1500.0, DSO=1000.0, nDetector=[44, 32], dDetector=[8.0, 8.0], nVoxel=[32, 32, 32], dVoxel=[8.0] * 3,
                offOrigin=[0, 0, 0], offDetector=[0, 0], accuracy=0.5, mode="parallel", filter=None, tilt_angle=29)
    data = synthetic_scan(geometry=geom, train_angles=angles[::24], n_val=1, device="cuda")
    ds = TIGREDataset(data, n_rays=128, type="train", device="cuda", seed=3)
    assert ds.rays.shape == (8, 32, 44, 8)
    net, ref = naf_pair(seed=22, log2T=13, scale=1e-4)
    S, steps, lr = 48, 30, 5e-3
    engine = NAFEngine(net, S, perturb=True, lr=lr)
    opt = torch.optim.Adam(ref.parameters(), lr=lr, betas=(0.9, 0.999))
    gen = torch.Generator(device="cuda").manual_seed(5)
    for step in range(steps):
        item = ds[step % len(ds)]
        rays, target = item["rays"], item["projs"]
        t_rand = torch.rand(rays.shape[0], S, device="cuda", generator=gen)
        weight = torch.full((rays.shape[0],), 1.0 / rays.shape[0], device="cuda")
        opt.zero_grad()
        acc_ref = R.render(rays.cpu(), ref, None, S, 0, True, 1 << 20, 0.0, t_rand=t_rand.cpu())["acc"]
        ((acc_ref - target.cpu()) ** 2).mean().backward()
        opt.step()
        acc = engine.backward(rays, target, weight, t_rand=t_rand).clone()
        engine.optimizer_step()
        if step == 0:
            assert rel_l2(acc.cpu().numpy(), acc_ref.detach().numpy()) < 1e-4
    with torch.no_grad():
        vol = net(ds.voxels).squeeze(-1)
        vol_ref = ref(ds.voxels.cpu().reshape(-1, 3)).reshape(vol.shape)
    p, p_ref = get_psnr_3d(vol, ds.image), psnr_ref(vol_ref, ds.image.cpu())
    assert abs(p - p_ref) < 0.1, (p, p_ref)


# ---- abdomen (configs[3]): S = 576, T = 2^19, bf16 tables ----------------------------------------------------------
def test_abdomen_shard_shape_bf16_forward_vs_oracle_and_binned_vs_atomic():
    from neuralvolumetricreconstructionformedicalimages_amd import _abi, fused
    from oracle import render_ref as R
    S = 576
    net, ref = naf_pair(seed=23, log2T=19, scale=0.3)
    # bf16 table: the oracle gets the table rounded once, like the shadow copy the kernels gather from
    ref.encoder.embeddings.data.copy_(ref.encoder.embeddings.data.to(torch.bfloat16).float())
    net.encoder.embeddings.data = net.encoder.embeddings.data.to(torch.bfloat16)
    n = 24
    rays = crossing_rays(n, seed=41)
    t_rand = torch.rand(n, S, generator=torch.Generator().manual_seed(6))
    with torch.no_grad():
        want = R.render(rays, ref, None, S, 0, True, 1 << 20, 0.0, t_rand=t_rand)["acc"].numpy()
        got = fused.fused_render(rays.cuda(), net, S, True, t_rand=t_rand.cuda()).float().cpu().numpy()
    assert rel_l2(got, want) < 1e-2                                    # bf16 MFMA operands (fp32 parity mode is 1e-4)
    # gradient scatter at this ray length: binned == atomic
    n = 256                                                            # 147 456 points: above the binned threshold
    rays = crossing_rays(n, seed=42).cuda()
    t_rand = torch.rand(n, S, device="cuda")
    target = torch.rand(n, device="cuda") * 0.3
    grads = {}
    for mode in (1, 2):
        with fused.scatter_mode(mode):
            net.zero_grad()
            acc = fused.fused_render(rays, net, S, True, t_rand=t_rand, mlp_precision=_abi.BF16)
            (1e4 * (acc - target) ** 2).mean().backward()              # scaled: the gradient comes back in the table dtype
            grads[mode] = net.encoder.embeddings.grad.float().clone()
    a, b = grads[1].double(), grads[2].double()
    assert float((a - b).norm() / a.norm()) < 5e-3


def test_abdomen_shape_s576_fp32_gradients_vs_oracle():
    """abdomen_50.yaml renders 576 samples per ray into a T = 2^19 table: the TABLE gradient of the binned scatter (64 rays =
    36 864 points, above the atomic / binned switch) and the MLP gradients in fp32 parity mode against the oracle's autograd."""
    from neuralvolumetricreconstructionformedicalimages_amd import fused
    from oracle import render_ref as R
    S, n = 576, 64
    net, ref = naf_pair(seed=31, log2T=19, scale=0.3)
    rays = crossing_rays(n, seed=51)
    t_rand = torch.rand(n, S, generator=torch.Generator().manual_seed(16))
    target = torch.rand(n, generator=torch.Generator().manual_seed(17)) * 0.3
    acc_ref = R.render(rays, ref, None, S, 0, True, 1 << 20, 0.0, t_rand=t_rand)["acc"]
    ((acc_ref - target) ** 2).mean().backward()
    with fused.scatter_mode(2):                                        # the binned scatter, whatever the size heuristics say
        acc = fused.fused_render(rays.cuda(), net, S, True, t_rand=t_rand.cuda())
        ((acc - target.cuda()) ** 2).mean().backward()
    assert rel_l2(acc.detach().cpu().numpy(), acc_ref.detach().numpy()) < 1e-4
    g, g_ref = net.encoder.embeddings.grad.cpu().numpy(), ref.encoder.embeddings.grad.numpy()
    assert rel_l2(g, g_ref) < 2e-4
    off = net.encoder.offsets.tolist()
    for lv in (0, 3, 8, 12, 15):                                       # dense, first hashed, hashed, wrapped-dense, finest
        assert rel_l2(g[off[lv]:off[lv + 1]], g_ref[off[lv]:off[lv + 1]]) < 5e-4, lv
    for a, b in zip(net.layers, ref.layers):
        assert rel_l2(a.weight.grad.cpu().numpy(), b.weight.grad.numpy()) < 2e-4


def test_foot_shape_t22_fp32_table_gradient_vs_oracle():
    """foot_50.yaml: T = 2^22 (52.8 M rows, 128 .. 512 buckets per level and the 2 048-point tiles of pass 1), S = 320.  Table
    gradient of 64 rays (20 480 points) through the binned scatter in fp32 parity mode against the oracle's autograd."""
    from neuralvolumetricreconstructionformedicalimages_amd import fused
    from oracle import render_ref as R
    S, n = 320, 64
    net, ref = naf_pair(seed=32, log2T=22, scale=0.3)
    rays = crossing_rays(n, seed=52)
    t_rand = torch.rand(n, S, generator=torch.Generator().manual_seed(18))
    target = torch.rand(n, generator=torch.Generator().manual_seed(19)) * 0.3
    acc_ref = R.render(rays, ref, None, S, 0, True, 1 << 20, 0.0, t_rand=t_rand)["acc"]
    ((acc_ref - target) ** 2).mean().backward()
    with fused.scatter_mode(2):
        acc = fused.fused_render(rays.cuda(), net, S, True, t_rand=t_rand.cuda())
        ((acc - target.cuda()) ** 2).mean().backward()
    assert rel_l2(acc.detach().cpu().numpy(), acc_ref.detach().numpy()) < 1e-4
    g, g_ref = net.encoder.embeddings.grad.cpu().numpy(), ref.encoder.embeddings.grad.numpy()
    assert rel_l2(g, g_ref) < 2e-4
    off = net.encoder.offsets.tolist()
    for lv in (2, 3, 4, 11, 12, 15):                                   # regimes of SURVEY App. A-1 at T = 2^22
        assert rel_l2(g[off[lv]:off[lv + 1]], g_ref[off[lv]:off[lv + 1]]) < 5e-4, lv


# ---- jaw (configs[0]): S = 320 ---------------------------------------------------------------------------------------
def test_jaw_shape_s320_fp32_vs_oracle():
    """jaw_50.yaml renders 320 samples per ray: projection and gradients in fp32 parity mode against the oracle."""
    from neuralvolumetricreconstructionformedicalimages_amd import fused
    from oracle import render_ref as R
    S, n = 320, 40
    net, ref = naf_pair(seed=24, log2T=16, scale=0.3)
    rays = crossing_rays(n, seed=43)
    t_rand = torch.rand(n, S, generator=torch.Generator().manual_seed(7))
    target = torch.rand(n, generator=torch.Generator().manual_seed(8)) * 0.3
    acc_ref = R.render(rays, ref, None, S, 0, True, 1 << 20, 0.0, t_rand=t_rand)["acc"]
    ((acc_ref - target) ** 2).mean().backward()
    acc = fused.fused_render(rays.cuda(), net, S, True, t_rand=t_rand.cuda())
    ((acc - target.cuda()) ** 2).mean().backward()
    assert rel_l2(acc.detach().cpu().numpy(), acc_ref.detach().numpy()) < 1e-4
    assert rel_l2(net.encoder.embeddings.grad.cpu().numpy(), ref.encoder.embeddings.grad.numpy()) < 2e-4
    for a, b in zip(net.layers, ref.layers):
        assert rel_l2(a.weight.grad.cpu().numpy(), b.weight.grad.numpy()) < 2e-4


# ---- chest (configs[1]) at full size: bf16 mode against the fp32 parity mode ---------------------------------------
def test_chest_size_bf16_psnr_within_a_tenth_of_a_db_of_fp32_mode():
    """256^3 phantom, 50 projections of 512 x 512, T = 2^19, S = 192: 200 Adam steps of 16 384 rays in bf16 mode (bf16
    shadow table, bf16 MFMA operands, bf16 scatter records) and in the fp32 parity mode, identical pixels and jitter.
    The reconstructed-volume PSNR must agree within the north star's 0.1 dB at the BASELINE size, not only at 16^3."""
    from neuralvolumetricreconstructionformedicalimages_amd import phantom
    from neuralvolumetricreconstructionformedicalimages_amd.encoder import HashEncoder
    from neuralvolumetricreconstructionformedicalimages_amd.engine import NAFEngine
    from neuralvolumetricreconstructionformedicalimages_amd.geometry import ConeGeometry, RayGenerator, get_voxels
    from neuralvolumetricreconstructionformedicalimages_amd.network import DensityNetwork
    from neuralvolumetricreconstructionformedicalimages_amd.utils import get_psnr_3d
    dev = torch.device("cuda")
    n_voxel, n_rays, steps = 256, 16384, 200
    geo = ConeGeometry(phantom.scan_geometry(n_voxel, "cone"))
    gen = RayGenerator(geo, np.linspace(0, np.pi, 51)[:-1], dev)
    table = phantom.ellipsoid_table(seed=0, extent=float(geo.sVoxel[0]) / 2)
    image = phantom.volume(geo, table, device=dev)
    voxels = torch.tensor(get_voxels(geo), dtype=torch.float32, device=dev)
    pix_gen = torch.Generator(device=dev).manual_seed(9)
    pixels = torch.randint(0, gen.n_projections * gen.pixels_per_projection, (steps, n_rays), device=dev, generator=pix_gen)
    rays = torch.empty(n_rays, 8, device=dev)
    weight = torch.full((n_rays,), 1.0 / n_rays, device=dev)
    psnr, loss = {}, {}
    for dt in (torch.float32, torch.bfloat16):
        torch.manual_seed(0)
        net = DensityNetwork(HashEncoder(3, 16, 2, 16, 19), bound=0.3, num_layers=4, hidden_dim=32, skips=[2], out_dim=1,
                             last_activation="sigmoid").to(dev)
        engine = NAFEngine(net, 192, perturb=True, lr=1e-3, table_dtype=dt, seed=0)
        for step in range(steps):
            gen.rays_for_pixels(pixels[step], out=rays)
            target = phantom.line_integrals(rays, table)
            engine.train_step(rays, target, weight, ray_base=step * n_rays)
        with torch.no_grad():
            vol = torch.cat([net(voxels[i:i + 32].reshape(-1, 3)).reshape(-1, n_voxel, n_voxel) for i in range(0, n_voxel, 32)])
        psnr[dt], loss[dt] = float(get_psnr_3d(vol, image)), float(engine.loss.item())
    assert psnr[torch.float32] > 18.0                                  # the run actually reconstructs something
    assert abs(psnr[torch.bfloat16] - psnr[torch.float32]) < 0.1, psnr
    assert abs(loss[torch.bfloat16] - loss[torch.float32]) < 0.05 * loss[torch.float32], loss


def test_chest_yaml_step_bf16_tracks_the_fp32_mode_for_5000_steps():
    """The reference's OWN step (config/chest_50.yaml: 1 024 rays of one projection, loss = sum of 200-ray chunk means, lr 1e-3) for
    5 000 steps from scratch on the 256^3 phantom, in bf16 mode and in the fp32 parity mode, with identical pixel draws and jitter:
    the reconstructed-volume PSNR must agree within the north star's 0.1 dB at the end (and both runs must have reconstructed
    something).  bench.py's `full_schedule` record carries the same comparison to the end of the 75 000-step schedule."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    dev = torch.device("cuda")
    scan = bench.ChestScan(dev, 1234)
    n_rays, steps = bench.CHEST["yaml_rays"], 5000
    rays = torch.empty(n_rays, 8, device=dev)
    weight, loss_name = bench.step_weights(n_rays, dev)
    assert loss_name.startswith("chunk_sum")
    psnr = {}
    for prec in ("fp32", "bf16"):
        engine = bench.make_chest_engine(dev, prec, bench.CHEST["lr"], seed=0)
        for step in range(steps):
            target, _ = scan.sampler.draw(step, n_rays, rays)
            engine.train_step(rays, target, weight, ray_base=step * n_rays)
        psnr[prec] = scan.volume_psnr(engine.net)
        del engine
    assert psnr["fp32"] > 26.0, psnr
    assert abs(psnr["bf16"] - psnr["fp32"]) < 0.1, psnr


# ---- coordinates outside [0,1] (the reference stays in bounds for ANY input: index % hashmap_size, hashencoder.cu:74) --
@pytest.mark.parametrize("layout", ["blc", "lbc"])
def test_out_of_range_coordinates_follow_the_reference_modulo(layout):
    """x = 1.5, x = 100 and negative / NaN coordinates on dense levels: the stand-alone operator must gather and scatter
    exactly where `index % hashmap_size` puts them (bit-exact against the C oracle), never outside the level."""
    import ctypes
    from neuralvolumetricreconstructionformedicalimages_amd import _abi
    from neuralvolumetricreconstructionformedicalimages_amd.encoder import level_offsets
    from oracle import c_oracle
    L, C, H, log2T = 6, 2, 16, 19                                      # levels 0..2 dense without modulo for x in [0,1]
    offs = level_offsets(3, L, H, log2T)
    rng = np.random.default_rng(3)
    emb = rng.standard_normal((int(offs[-1]), C)).astype(np.float32)
    x = rng.random((64, 3)).astype(np.float32)
    x[0] = [1.5, 0.2, 0.3]
    x[1] = [100.0, 100.0, 100.0]
    x[2] = [-3.0, 0.5, 2.0]
    x[3] = [0.5, 1.0000001, 0.999]
    x[4] = [1e9, 0.1, 0.1]
    x[5] = [np.nan, 0.5, 0.5]
    want, _ = c_oracle.hash_encode_forward(x, emb, offs, H)            # [L, B, C]
    B = x.shape[0]
    xd, ed, od = torch.from_numpy(x).cuda(), torch.from_numpy(emb).cuda(), torch.from_numpy(offs).cuda()
    out = torch.empty((B, L * C) if layout == "blc" else (L, B, C), device="cuda")
    lay = _abi.LAYOUT_BLC if layout == "blc" else _abi.LAYOUT_LBC
    _abi.check(_abi.lib().naf_hash_encode_forward(_abi.ptr(xd), _abi.ptr(ed), _abi.ptr(od), _abi.ptr(out), B, 3, C, L, H, 0, None,
                                                  _abi.F32, lay, None), "hash_encode_forward")
    got = out.cpu().numpy()
    got = got.reshape(B, L, C).transpose(1, 0, 2) if layout == "blc" else got
    finite = ~np.isnan(want)
    assert np.array_equal(got[finite], want[finite]) and np.array_equal(np.isnan(got), np.isnan(want))
    # backward: a sentinel row after the table must stay untouched, the gradient equals the oracle's
    grad = rng.standard_normal((B, L * C)).astype(np.float32)
    grad[5] = 0.0                                                      # NaN weights would poison rows in both implementations
    ge_want, _ = c_oracle.hash_encode_backward(grad, x, emb, offs, H)
    gd = torch.from_numpy(grad).cuda()
    ge = torch.zeros(int(offs[-1]) + 64, C, device="cuda")
    _abi.check(_abi.lib().naf_hash_encode_backward(_abi.ptr(gd), _abi.ptr(xd), _abi.ptr(ed), _abi.ptr(od), _abi.ptr(ge), B, 3, C, L, H,
                                                   0, None, None, _abi.F32, _abi.LAYOUT_BLC, None), "hash_encode_backward")
    torch.cuda.synchronize()
    ge = ge.cpu().numpy()
    assert not ge[int(offs[-1]):].any()
    ok = ~np.isnan(ge_want)
    np.testing.assert_allclose(ge[:int(offs[-1])][ok], ge_want[ok], rtol=0, atol=2e-5 * np.abs(ge_want[ok]).max())
    del ctypes


def test_input_gradient_modes_match_the_oracle():
    """calc_grad_inputs = EXACT (level scale included) and = REFERENCE (hashencoder.cu:153-197 as written) against the C
    oracle, bit for bit; and HashEncoder(reference_compat=...) routes autograd through the chosen mode."""
    from neuralvolumetricreconstructionformedicalimages_amd import _abi
    from neuralvolumetricreconstructionformedicalimages_amd.encoder import HashEncoder, level_offsets
    from oracle import c_oracle
    L, C, H, log2T = 5, 2, 4, 10
    offs = level_offsets(3, L, H, log2T)
    rng = np.random.default_rng(5)
    emb = rng.standard_normal((int(offs[-1]), C)).astype(np.float32)
    x = rng.random((200, 3)).astype(np.float32)
    B = x.shape[0]
    xd, ed, od = torch.from_numpy(x).cuda(), torch.from_numpy(emb).cuda(), torch.from_numpy(offs).cuda()
    jac = {}
    for mode in (_abi.GRAD_INPUTS_EXACT, _abi.GRAD_INPUTS_REFERENCE):
        _, want = c_oracle.hash_encode_forward(x, emb, offs, H, calc_grad_inputs=mode)
        out = torch.empty(B, L * C, device="cuda")
        dy_dx = torch.empty(B, L, 3, C, device="cuda")
        _abi.check(_abi.lib().naf_hash_encode_forward(_abi.ptr(xd), _abi.ptr(ed), _abi.ptr(od), _abi.ptr(out), B, 3, C, L, H, mode,
                                                      _abi.ptr(dy_dx), _abi.F32, _abi.LAYOUT_BLC, None), "hash_encode_forward")
        jac[mode] = dy_dx.cpu().numpy()
        assert np.array_equal(jac[mode], want), mode
    assert not np.allclose(jac[1], jac[2])
    # module level: the exact mode is the true derivative of the encoder output
    enc = HashEncoder(3, L, C, H, log2T).cuda()
    enc.embeddings.data.copy_(ed)
    p = (torch.from_numpy(x).cuda() * 0.58 - 0.29).requires_grad_(True)      # +- eps stays inside the +-0.3 range check
    y = enc(p, 0.3)
    w = torch.randn_like(y)
    (y * w).sum().backward()
    eps = 2e-4                                                        # finest cell of this grid: 0.6 / 64 = 9.4e-3
    for d in range(3):
        step = torch.zeros(3, device="cuda")
        step[d] = eps
        with torch.no_grad():
            fd = ((enc(p + step, 0.3) - enc(p - step, 0.3)) * w).sum(-1) / (2 * eps)
        ok = (fd - p.grad[:, d]).abs() < 2e-2 * fd.abs().max()
        assert ok.float().mean() > 0.9                                 # points next to a cell wall cross it within eps
    enc_ref = HashEncoder(3, L, C, H, log2T, reference_compat=True).cuda()
    enc_ref.embeddings.data.copy_(ed)
    p2 = p.detach().clone().requires_grad_(True)
    (enc_ref(p2, 0.3) * w).sum().backward()
    assert not torch.allclose(p2.grad, p.grad)


def test_non_finite_gradients_stay_visible_in_the_binned_scatter():
    """A NaN target poisons d loss / d acc of one ray.  The reference's atomics would carry the NaN into the rows that ray
    touches; the fixed-point reducer cannot represent it, so it poisons its row sums instead of adding garbage."""
    from neuralvolumetricreconstructionformedicalimages_amd import _abi, fused
    net, _ = naf_pair(seed=25, log2T=14, scale=0.1, oracle=False)
    n, S = 512, 64
    rays = crossing_rays(n, seed=44).cuda()
    target = torch.rand(n, device="cuda") * 0.3
    target[7] = float("nan")
    for mode in (1, 2):
        with fused.scatter_mode(mode):
            net.zero_grad()
            acc = fused.fused_render(rays, net, S, True, seed=1, mlp_precision=_abi.BF16)
            ((acc - target) ** 2).mean().backward()
        assert bool(torch.isnan(net.encoder.embeddings.grad).any()), mode


# ---- foot (configs[4]): T = 2^22 fp16 table, S = 320, at a batch that fills the chip ------------------------------------
def test_foot_shape_4096_rays_t22_fp16_binned_equals_atomic_and_steps():
    """foot_50 shapes at 4 096 rays (1.31 M points): 512 scatter buckets per level, 8-record runs packed 8 lanes each in the
    reducer, wrapped-dense fine levels as pair records.  The binned table gradient equals the reference-style atomic one,
    nothing overflows, and an Adam step moves the fp32 master and its fp16 shadow together."""
    from neuralvolumetricreconstructionformedicalimages_amd.engine import NAFEngine
    from neuralvolumetricreconstructionformedicalimages_amd import phantom
    net, _ = naf_pair(seed=31, log2T=22, scale=1e-2, oracle=False)
    n, S = 4096, 320
    # rays of a cone-beam scan (SURVEY 8d geometry): every sample lies inside the +-0.3 box, as in a real scan -- samples
    # clamped to the box boundary would all be unpaired (cell index 2^k - 1) and fill their tiles' blocks
    gen = _raygen(phantom.scan_geometry(128, "cone"), np.linspace(0, np.pi, 9)[:-1])
    pix = torch.randint(0, gen.n_projections * gen.pixels_per_projection, (n,), generator=torch.Generator().manual_seed(51))
    rays = gen.rays_for_pixels(pix.cuda())
    target = torch.rand(n, device="cuda") * 0.3
    weight = torch.full((n,), 1.0 / n, device="cuda")
    engine = NAFEngine(net, S, perturb=True, lr=1e-3, table_dtype=torch.float16, seed=3)
    grads = {}
    for mode in (1, 2):
        engine.scatter_mode = mode
        engine.grad_flat.zero_()
        engine.backward(rays, target, weight)
        grads[mode] = engine.emb_g.clone().double()
        if mode == 2:
            assert engine.scatter_overflow(n) <= 1e-4 * n * S * 16 * 4       # a tile's block fills up only when most of its pairs are unpaired
    a, b = grads[1], grads[2]
    assert float(a.norm()) > 0 and float((a - b).norm() / a.norm()) < 3e-3      # bf16 records round each contribution once more
    before = engine.emb.clone()
    engine.optimizer_step()
    torch.cuda.synchronize()
    moved = (engine.emb - before).abs()
    assert 0 < float(moved.max()) <= 1.001e-3 and float(engine.emb_g.abs().max()) == 0.0
    assert torch.equal(engine.emb_lp, engine.emb.half())
