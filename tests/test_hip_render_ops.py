"""GPU parity of the stand-alone ray-march operators and Adam (through the C ABI) against the oracle / torch."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _abi():
    from neuralvolumetricreconstructionformedicalimages_amd import _abi
    return _abi


def test_sample_rays_matches_golden(golden):
    A = _abi()
    g = golden("render")
    rays = torch.from_numpy(g["rays"]).cuda()
    n, S = g["det/t_rand"].shape
    for tag, perturb in (("det", 0), ("jit", 1)):
        tr = torch.from_numpy(g[f"{tag}/t_rand"]).cuda()
        z = torch.empty(n, S, device="cuda")
        pts = torch.empty(n, S, 3, device="cuda")
        A.check(A.lib().naf_sample_rays(A.ptr(rays), A.ptr(tr), A.ptr(z), A.ptr(pts), n, S, perturb, 0.3, 0, 0, A.stream_ptr()))
        torch.cuda.synchronize()
        # reference pts are fp32 torch ops; ours follow the same op order: bit-exact up to 1 ulp of |o|+|d z|
        np.testing.assert_allclose(pts.cpu().numpy(), g[f"{tag}/pts"], rtol=0, atol=1.2e-7)


def test_sample_rays_chest_shape_vs_oracle():
    from oracle import render_ref as R
    A = _abi()
    torch.manual_seed(0)
    n, S = 257, 192
    rays = torch.cat([torch.randn(n, 3) * 0.1 + torch.tensor([1.0, 0, 0]), torch.randn(n, 3), torch.full((n, 1), 0.814),
                      torch.full((n, 1), 1.186)], -1)
    tr = torch.rand(n, S)
    z_ref = R.sample_depths(rays[:, 6:7], rays[:, 7:], S, True, tr)
    p_ref = R.points_on_rays(rays, z_ref, 0.3)
    z = torch.empty(n, S, device="cuda")
    pts = torch.empty(n, S, 3, device="cuda")
    rd, td = rays.cuda(), tr.cuda()
    A.check(A.lib().naf_sample_rays(A.ptr(rd), A.ptr(td), A.ptr(z), A.ptr(pts), n, S, 1, 0.3, 0, 0, A.stream_ptr()))
    torch.cuda.synchronize()
    assert torch.equal(z.cpu(), z_ref)                      # same fp32 op order -> bit-exact
    assert torch.equal(pts.cpu(), p_ref)


def test_device_jitter_is_uniform_and_reproducible():
    A = _abi()
    n, S = 4096, 192
    rays = torch.zeros(n, 8, device="cuda")
    rays[:, 3] = 1.0
    rays[:, 6], rays[:, 7] = 0.0, 1.0
    z1 = torch.empty(n, S, device="cuda")
    z2 = torch.empty(n, S, device="cuda")
    pts = torch.empty(n, S, 3, device="cuda")
    A.check(A.lib().naf_sample_rays(A.ptr(rays), None, A.ptr(z1), A.ptr(pts), n, S, 1, 10.0, 7, 0, A.stream_ptr()))
    A.check(A.lib().naf_sample_rays(A.ptr(rays), None, A.ptr(z2), A.ptr(pts), n, S, 1, 10.0, 7, 0, A.stream_ptr()))
    assert torch.equal(z1, z2)
    # strata: sample s lies in [mid(s-1,s), mid(s,s+1)]; normalised position inside the stratum is U[0,1)
    t = torch.linspace(0, 1, S, device="cuda")
    mids = 0.5 * (t[1:] + t[:-1])
    lower = torch.cat([t[:1], mids])
    upper = torch.cat([mids, t[-1:]])
    u = ((z1 - lower) / (upper - lower))[:, 1:-1]
    assert u.min().item() >= 0 and u.max().item() <= 1          # recomputed in fp32: the open end can round to 1.0
    assert abs(u.mean().item() - 0.5) < 2e-3 and abs(u.var().item() - 1 / 12) < 2e-3
    # a different ray base gives a different stream, the same global ray index gives the same numbers
    A.check(A.lib().naf_sample_rays(A.ptr(rays[100:]), None, A.ptr(z2), A.ptr(pts), n - 100, S, 1, 10.0, 7, 100, A.stream_ptr()))
    assert torch.equal(z2[: n - 100], z1[100:])


def test_integrate_forward_backward(golden):
    from oracle import render_ref as R
    A = _abi()
    g = golden("render")
    raw, z, d = (torch.from_numpy(g[k]) for k in ("r2o/raw", "r2o/z", "r2o/d"))
    n, S = z.shape
    rays = torch.cat([torch.zeros(n, 3), d, torch.zeros(n, 2)], -1).cuda()
    acc = torch.empty(n, device="cuda")
    sig = raw[..., 0].contiguous().cuda()
    zd = z.cuda()
    A.check(A.lib().naf_integrate_forward(A.ptr(sig), A.ptr(zd), A.ptr(rays), A.ptr(acc), n, S, A.stream_ptr()))
    np.testing.assert_allclose(acc.cpu().numpy(), g["r2o/acc"], rtol=2e-6)
    ga = torch.randn(n)
    sig_ref = raw.clone().requires_grad_(True)
    R.raw2outputs(sig_ref, z, d)[0].backward(ga)
    gs = torch.empty(n, S, device="cuda")
    gad = ga.cuda()
    A.check(A.lib().naf_integrate_backward(A.ptr(gad), A.ptr(zd), A.ptr(rays), A.ptr(gs), n, S, A.stream_ptr()))
    np.testing.assert_allclose(gs.cpu().numpy(), sig_ref.grad[..., 0].numpy(), rtol=2e-6, atol=1e-12)


@pytest.mark.parametrize("lp", [None, torch.float16, torch.bfloat16])
def test_adam_matches_torch(lp):
    A = _abi()
    torch.manual_seed(1)
    n = 10007                                   # not a multiple of 4: exercises the tail
    p0 = torch.randn(n)
    ref = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([ref], lr=1e-3, betas=(0.9, 0.999))
    p = p0.clone().cuda()
    m = torch.zeros(n, device="cuda")
    v = torch.zeros(n, device="cuda")
    shadow = torch.empty(n, device="cuda", dtype=lp) if lp is not None else None
    for step in range(1, 6):
        g = torch.randn(n) * (10.0 ** (step - 3))
        ref.grad = g.clone()
        opt.step()
        gd = g.cuda()
        A.check(A.lib().naf_adam_step(A.ptr(p), A.ptr(m), A.ptr(v), A.ptr(gd), A.ptr(shadow), A.dtype_code(lp) if lp else 0, n,
                                      1e-3, 0.9, 0.999, 1e-8, step, 1.0, 1, A.stream_ptr()))
        torch.cuda.synchronize()
        assert gd.abs().max().item() == 0.0     # zero_grad fused
        np.testing.assert_allclose(p.cpu().numpy(), ref.detach().numpy(), rtol=2e-6, atol=2e-7)
    if lp is not None:
        assert torch.equal(shadow.cpu(), p.cpu().to(lp))


def test_draw_scan_rays_distinct_valid_uniform_and_shardable():
    """naf_draw_scan_rays (tigre.py:354-372 on the device): distinct pixels, all from the valid lists, their measured
    values and rays; a new seed gives a new draw; slices of one draw (data-parallel shards) tile it; roughly uniform."""
    from neuralvolumetricreconstructionformedicalimages_amd import phantom
    from neuralvolumetricreconstructionformedicalimages_amd.geometry import ConeGeometry, RayGenerator
    geo = ConeGeometry(phantom.scan_geometry(16, "cone"))                    # 32 x 32 detector
    gen = RayGenerator(geo, np.linspace(0, np.pi, 5)[:-1], torch.device("cuda"))
    hw = gen.pixels_per_projection
    g = torch.Generator().manual_seed(0)
    projs = torch.rand(4 * hw, generator=g)
    projs[torch.rand(4 * hw, generator=g) < 0.4] = 0.0                       # 40 % of the pixels saw nothing
    projs = projs.cuda()
    valid = [(torch.nonzero(projs[i * hw:(i + 1) * hw] > 0).reshape(-1) + i * hw).contiguous() for i in range(4)]
    per = 300
    pix, tgt, rays = gen.draw(valid[1:3], per, seed=7, projections=projs)
    assert pix.shape == (600,) and tgt.shape == (600,) and rays.shape == (600, 8)
    for j, lst in enumerate(valid[1:3]):
        seg = pix[j * per:(j + 1) * per]
        assert len(torch.unique(seg)) == per                                 # replace=False
        assert bool(torch.isin(seg, lst).all())                              # only pixels that measured something
    assert torch.equal(tgt, projs[pix]) and bool((tgt > 0).all())
    assert torch.equal(rays, gen.rays_for_pixels(pix))                       # same bits as the pixel-list ray generator
    pix2, _, _ = gen.draw(valid[1:3], per, seed=8, projections=projs)
    assert not torch.equal(pix, pix2)
    # shards: two ranks with the same seed take [0, 250) and [250, 600) of the same draw
    a, ta, ra = gen.draw(valid[1:3], per, seed=7, projections=projs, first=0, count=250)
    b, tb, rb = gen.draw(valid[1:3], per, seed=7, projections=projs, first=250, count=350)
    assert torch.equal(torch.cat([a, b]), pix) and torch.equal(torch.cat([ta, tb]), tgt) and torch.equal(torch.cat([ra, rb]), rays)
    # drawing the whole list is a permutation of it
    n1 = valid[0].numel()
    full, _, _ = gen.draw(valid[:1], n1, seed=3)
    assert torch.equal(torch.sort(full).values, valid[0])
    # uniformity: over many seeds every valid pixel of a list is drawn about equally often
    counts = torch.zeros(4 * hw, device="cuda")
    trials, m = 400, 64
    for s_ in range(trials):
        p_, _, _ = gen.draw(valid[3:4], m, seed=1000 + s_)
        counts[p_] += 1
    c = counts[valid[3]]
    expect = trials * m / valid[3].numel()
    assert float(counts.sum()) == trials * m and float((c - expect).abs().max()) < 6 * expect ** 0.5 + 3
    with pytest.raises(ValueError, match="larger sample than population"):
        gen.draw(valid[:1], n1 + 1, seed=1)
    # a list entry outside the scan (the reference's fancy index would raise): nothing is read through it, its ray and value are NaN
    bad = valid[0].clone()
    bad[5] = 4 * hw + 123
    pb, tb2, rb2 = gen.draw([bad], n1, seed=3, projections=projs)
    hit = pb == 4 * hw + 123
    assert int(hit.sum()) == 1 and bool(torch.isnan(tb2[hit]).all()) and bool(torch.isnan(rb2[hit][:, :6]).all())
    assert torch.equal(tb2[~hit], projs[pb[~hit]]) and torch.equal(rb2[~hit], gen.rays_for_pixels(pb[~hit]))


@pytest.mark.parametrize("table", ["bf16", "fp32"])
def test_a_step_can_carry_the_pixel_draw_of_the_next_one(table):
    """naf_render_train_adam_draw: step k takes the draw of step k + 1 along (spare workgroups of the scatter's first launch in bf16
    mode on the binned scatter; a launch of its own behind the step in fp32 mode and for small batches).  Rays, measured values and
    the trained parameters are bit-identical to drawing with naf_draw_scan_rays in front of every step."""
    from neuralvolumetricreconstructionformedicalimages_amd import phantom
    from neuralvolumetricreconstructionformedicalimages_amd.encoder import HashEncoder
    from neuralvolumetricreconstructionformedicalimages_amd.engine import NAFEngine
    from neuralvolumetricreconstructionformedicalimages_amd.geometry import ConeGeometry, RayGenerator
    from neuralvolumetricreconstructionformedicalimages_amd.network import DensityNetwork
    dev = torch.device("cuda")
    geo = ConeGeometry(phantom.scan_geometry(16, "cone"))                    # 32 x 32 detector
    gen = RayGenerator(geo, np.linspace(0, np.pi, 7)[:-1], dev)
    hw = gen.pixels_per_projection
    g = torch.Generator().manual_seed(3)
    projs = torch.rand(6 * hw, generator=g) * 0.2
    projs[torch.rand(6 * hw, generator=g) < 0.3] = 0.0
    projs = projs.cuda()
    valid = [(torch.nonzero(projs[i * hw:(i + 1) * hw] > 0).reshape(-1) + i * hw).contiguous() for i in range(6)]
    S = 192

    def engine():
        torch.manual_seed(0)
        net = DensityNetwork(HashEncoder(3, 16, 2, 16, 15), bound=0.3, num_layers=4, hidden_dim=32, skips=[2], out_dim=1,
                             last_activation="sigmoid").to(dev)
        return NAFEngine(net, S, perturb=True, lr=1e-2, table_dtype=torch.bfloat16 if table == "bf16" else torch.float32, seed=5)

    def same(x, y, exact):
        return torch.equal(x, y) if exact else bool(torch.allclose(x, y, rtol=1e-4, atol=1e-6))

    for n in (64, 16):                                                       # 12 288 points: the binned scatter; 3 072: the atomic one
        exact = n == 64                                                      # (float atomics: not reproducible from run to run)
        weight = torch.full((n,), 1.0 / n, device=dev)
        a, b, c = engine(), engine(), engine()                         # c: the control -- the same route as a, twice
        rays_a, tgt_a = torch.empty(n, 8, device=dev), torch.empty(n, device=dev)
        rays_b = [torch.empty(n, 8, device=dev) for _ in range(2)]
        tgt_b = [torch.empty(n, device=dev) for _ in range(2)]
        steps = 5
        gen.draw([valid[0]], n, seed=100, projections=projs, rays_out=rays_b[0], target_out=tgt_b[0], want_pixels=False)
        for k in range(steps):
            gen.draw([valid[k % 6]], n, seed=100 + k, projections=projs, rays_out=rays_a, target_out=tgt_a, want_pixels=False)
            a.train_step(rays_a, tgt_a, weight, ray_base=k * n)
            c.train_step(rays_a, tgt_a, weight, ray_base=k * n)
            assert same(a.emb, c.emb, exact) and same(a.mlp, c.mlp, exact), f"the step itself is not reproducible (step {k}, {n} rays)"
            cur = k & 1
            assert torch.equal(rays_b[cur], rays_a) and torch.equal(tgt_b[cur], tgt_a)
            plan = gen.plan_draw([valid[(k + 1) % 6]], n, 100 + k + 1, projs, rays_b[cur ^ 1], tgt_b[cur ^ 1])
            b.train_step(rays_b[cur], tgt_b[cur], weight, ray_base=k * n, next_draw=plan)
            assert same(a.emb, b.emb, exact) and same(a.mlp, b.mlp, exact), f"step {k}, {n} rays"
        torch.cuda.synchronize()
        assert same(a.emb, b.emb, exact) and same(a.mlp, b.mlp, exact) and same(a.loss, b.loss, exact)
        assert same(a.emb_m, b.emb_m, exact) and same(a.emb_v, b.emb_v, exact)
    # the next step's buffers must not be the ones this step reads
    plan = gen.plan_draw([valid[0]], n, 1, projs, rays_a, tgt_a)
    with pytest.raises(RuntimeError, match="must not be the buffers"):
        a.train_step(rays_a, tgt_a, weight, ray_base=0, next_draw=plan)
