"""Edge cases of the hot path on the GPU, through the same entry points as the parity tests: empty and one-element
inputs, the smallest legal sample count, rays that miss the volume (every sample clamped, render.py:103-105), non-finite
rays, batches that straddle the atomic / binned scatter switch.  The reference has no tests of its own; these are the cases
its code paths define (np.random.choice / fancy indexing on empty lists, `torch.clamp`, zero-length chunks in
render.py:52-60)."""
import numpy as np
import pytest
import torch

from _naf_helpers import crossing_rays, naf_pair, rel_l2

pytestmark = pytest.mark.gpu


def _mods():
    from neuralvolumetricreconstructionformedicalimages_amd import _abi, fused
    return _abi, fused


def test_empty_inputs_are_no_ops_everywhere():
    """Zero rays / zero points: every entry point returns empty outputs, launches nothing and leaves gradients at zero."""
    _abi, fused = _mods()
    from neuralvolumetricreconstructionformedicalimages_amd import encoder, phantom
    from neuralvolumetricreconstructionformedicalimages_amd.geometry import ConeGeometry, RayGenerator
    net, _ = naf_pair(seed=3, oracle=False)
    rays = crossing_rays(4).cuda()
    acc = fused.fused_render(rays[:0], net, 32, True)
    assert acc.shape == (0,)
    net.zero_grad()
    (acc.sum() + 0.0 * net.encoder.embeddings.sum()).backward()
    assert float(net.encoder.embeddings.grad.abs().max()) == 0.0
    with torch.no_grad():
        assert fused.field_query(net, torch.zeros(0, 3, device="cuda")).shape[0] == 0
        a, sg, tau = fused.render_samples(rays[:0], net, 32, False)
        assert a.shape == (0,) and sg.shape == (0, 32) and tau.shape == (0, 32)
    enc = encoder.HashEncoder(3, 4, 2, 4, 8).cuda()
    out = enc(torch.zeros(0, 3, device="cuda"), 0.3)
    assert out.shape == (0, 8)
    gen = RayGenerator(ConeGeometry(phantom.scan_geometry(16, "cone")), np.linspace(0, np.pi, 3)[:-1], torch.device("cuda"))
    assert gen.rays_for_pixels(torch.zeros(0, dtype=torch.int64, device="cuda")).shape == (0, 8)


@pytest.mark.parametrize("mode", [1, 2])            # atomic / binned scatter
def test_one_ray_two_samples_vs_oracle(mode):
    """The smallest legal call: one ray, S = 2 (render.py:88-100 needs two samples for the mid-points)."""
    from oracle import render_ref as R
    _abi, fused = _mods()
    net, ref = naf_pair(seed=5)
    rays = crossing_rays(1, seed=9)
    t_rand = torch.tensor([[0.25, 0.75]])
    acc_ref = R.render(rays, ref, None, 2, 0, True, 1 << 20, 0.0, t_rand=t_rand)["acc"]
    (acc_ref ** 2).sum().backward()
    with fused.scatter_mode(mode):
        acc = fused.fused_render(rays.cuda(), net, 2, True, t_rand=t_rand.cuda())
        (acc ** 2).sum().backward()
    assert rel_l2(acc.detach().cpu().numpy(), acc_ref.detach().numpy()) < 1e-5
    assert rel_l2(net.encoder.embeddings.grad.cpu().numpy(), ref.encoder.embeddings.grad.numpy()) < 2e-4


def test_rays_that_miss_the_volume_are_clamped_like_the_reference():
    """All samples outside the +-bound cube: the reference clamps them to the faces (render.py:103-105) and still evaluates the
    network there -- the result is finite, equals the oracle, and the table gradient lands on boundary cells only."""
    from oracle import render_ref as R
    _abi, fused = _mods()
    net, ref = naf_pair(seed=6)
    n, S = 12, 24
    o = torch.tensor([[2.0, 2.0, 2.0]]).repeat(n, 1)
    d = torch.nn.functional.normalize(torch.rand(n, 3, generator=torch.Generator().manual_seed(2)) + 0.2, dim=-1)   # pointing away
    rays = torch.cat([o, d, torch.full((n, 1), 0.1), torch.full((n, 1), 3.0)], -1)
    acc_ref = R.render(rays, ref, None, S, 0, False, 1 << 20, 0.0)["acc"]
    (acc_ref ** 2).sum().backward()
    acc = fused.fused_render(rays.cuda(), net, S, False)
    (acc ** 2).sum().backward()
    assert bool(torch.isfinite(acc).all())
    assert rel_l2(acc.detach().cpu().numpy(), acc_ref.detach().numpy()) < 1e-5
    assert rel_l2(net.encoder.embeddings.grad.cpu().numpy(), ref.encoder.embeddings.grad.numpy()) < 2e-4
    # every sample sits in the (+,+,+) corner cell of each level: 8 rows per level at most
    touched = (net.encoder.embeddings.grad.abs().sum(-1) > 0).sum().item()
    assert 0 < touched <= 8 * 16


@pytest.mark.parametrize("prec", ["f32", "bf16"])
def test_non_finite_rays_poison_their_own_results_only(prec):
    """A NaN / Inf ray yields a NaN projection (the reference's NaN check, render.py:141-144, would print); the other rays of
    the batch are untouched and no access leaves the table (positions are clamped before the cell index is taken)."""
    _abi, fused = _mods()
    net, _ = naf_pair(seed=7, oracle=False)
    p = _abi.F32 if prec == "f32" else _abi.BF16
    rays = crossing_rays(70, seed=4).cuda()
    with torch.no_grad():
        good = fused.fused_render(rays, net, 48, False, mlp_precision=p).clone()
        bad = rays.clone()
        bad[3, 0] = float("nan")
        bad[17, 4] = float("inf")
        bad[40, 6] = float("nan")
        out = fused.fused_render(bad, net, 48, False, mlp_precision=p)
    ok = torch.ones(70, dtype=torch.bool, device="cuda")
    ok[[3, 17, 40]] = False
    assert torch.equal(out[ok], good[ok])
    assert not bool(torch.isfinite(out[~ok]).any())


def test_batches_on_both_sides_of_the_scatter_switch_agree():
    """AUTO picks atomics below 2^13 points per call and the binned scatter from there on: the same rays rendered as one
    batch (binned) and as chunks (atomic) accumulate the same table gradient."""
    _abi, fused = _mods()
    net, _ = naf_pair(seed=8, oracle=False)
    S, n = 64, 160                                               # 10 240 points in one call, 2 560 per chunk of 40
    rays = crossing_rays(n, seed=6).cuda()
    t_rand = torch.rand(n, S, device="cuda")
    target = torch.rand(n, device="cuda") * 0.2
    net.zero_grad()
    acc = fused.fused_render(rays, net, S, True, t_rand=t_rand)
    ((acc - target) ** 2).sum().backward()
    whole = net.encoder.embeddings.grad.clone()
    net.zero_grad()
    for i in range(0, n, 40):
        a = fused.fused_render(rays[i:i + 40], net, S, True, t_rand=t_rand[i:i + 40])
        ((a - target[i:i + 40]) ** 2).sum().backward()
    parts = net.encoder.embeddings.grad
    assert rel_l2(parts.cpu().numpy(), whole.cpu().numpy()) < 2e-6


@pytest.mark.parametrize("n_streams", [2, 3])
def test_engine_pipelined_over_streams_equals_the_single_stream_step(n_streams):
    """NAFEngine(n_streams > 1) pipelines chunks of a ray batch over HIP streams, each lane with its own workspace and gradient
    buffers that are folded at the end: same projection, loss and gradients as the single-stream step (chunking only
    changes the fixed-point scale of each chunk's scatter and the order of a few fp32 sums)."""
    from neuralvolumetricreconstructionformedicalimages_amd.engine import NAFEngine
    S, n = 48, 500                                               # 500 rays in chunks of 96: 6 chunks, the last one ragged
    rays = crossing_rays(n, seed=12).cuda()
    t_rand = torch.rand(n, S, device="cuda")
    target = torch.rand(n, device="cuda") * 0.2
    weight = torch.full((n,), 1.0 / n, device="cuda")
    out = {}
    for tag, kw in (("one", {}), ("many", {"n_streams": n_streams, "chunk_rays": 96})):
        net, _ = naf_pair(seed=21, oracle=False)
        eng = NAFEngine(net, S, perturb=True, lr=1e-3, **kw)
        acc = eng.backward(rays, target, weight, t_rand=t_rand).clone()
        torch.cuda.synchronize()
        out[tag] = (acc.cpu().numpy(), eng.emb_g.clone().cpu().numpy(), eng.mlp_g.clone().cpu().numpy(), float(eng.loss.item()))
        eng.optimizer_step()                                     # and the step after it runs on clean buffers
        acc2 = eng.backward(rays, target, weight, t_rand=t_rand)
        assert bool(torch.isfinite(acc2).all())
    assert rel_l2(out["many"][0], out["one"][0]) < 1e-6
    assert rel_l2(out["many"][1], out["one"][1]) < 1e-5
    assert rel_l2(out["many"][2], out["one"][2]) < 1e-5
    assert abs(out["many"][3] - out["one"][3]) < 1e-6 * abs(out["one"][3])


@pytest.mark.parametrize("table_dtype,log2T,n,S", [(torch.float32, 14, 300, 48), (torch.bfloat16, 14, 300, 48), (torch.float16, 16, 700, 64),
                                                   (torch.bfloat16, 19, 2048, 192), (torch.float32, 12, 20, 16), (torch.bfloat16, 20, 256, 64)])
def test_table_adam_in_the_reducer_equals_the_separate_passes(table_dtype, log2T, n, S):
    """naf_render_train_adam (the gradient reducer finishes every table row with its Adam update) against naf_render_train +
    naf_adam_step: parameters, both moments, the 16-bit shadow table and the loss agree BIT FOR BIT over several steps, the
    gradient table stays all zero; a batch below 2^13 points (atomic scatter: the two passes run one after the other inside
    the call, fp32 atomics make it reproducible to rounding only) included."""
    from neuralvolumetricreconstructionformedicalimages_amd.engine import NAFEngine
    rays = crossing_rays(n, seed=31).cuda()
    target = torch.rand(n, device="cuda") * 0.2
    weight = torch.full((n,), 1.0 / n, device="cuda")
    out = {}
    for fuse in (False, True):
        net, _ = naf_pair(seed=33, log2T=log2T, oracle=False)
        eng = NAFEngine(net, S, perturb=True, lr=3e-3, table_dtype=table_dtype, fuse_table_adam=fuse)
        losses = []
        for step in range(4):
            losses.append(eng.train_step(rays, target, weight, ray_base=step * n).clone())
        torch.cuda.synchronize()
        assert float(eng.emb_g.abs().max()) == 0.0 and eng.step_count == 4
        out[fuse] = (eng.emb.clone(), eng.emb_m.clone(), eng.emb_v.clone(), None if eng.emb_lp is None else eng.emb_lp.clone(),
                     eng.mlp.clone(), torch.stack(losses))
    exact = n * S >= 8192                          # below that the scatter uses fp32 atomics, whose order is not reproducible
    for a, b in zip(out[True], out[False]):
        if a is None or b is None:
            assert a is None and b is None
        elif exact:
            assert torch.equal(a, b)
        else:
            assert float((a.float() - b.float()).abs().max()) <= 1e-6 * max(float(b.float().abs().max()), 1e-3)
    assert float((out[True][0] - naf_pair(seed=33, log2T=log2T, oracle=False)[0].encoder.embeddings.data).abs().max()) > 0     # it trained


@pytest.mark.parametrize("table_dtype", [torch.float32, torch.bfloat16])
def test_adam_tail_folds_in_records_that_overflowed_their_blocks(table_dtype):
    """The `spilled` branch of the reducer's Adam tail: with NAF_CFG_TEST_TINY_BLOCKS a pass-1 block holds a quarter of its tile's
    records, the rest reach the gradient table through counted global atomics, and the tail must add them to its own sums and
    clear them.  Against the separate route (reducer adds its sums to the table, naf_adam_step consumes and clears it) on the
    same tiny blocks -- equal up to the order of the fp32 atomics -- and against the ordinary step without overflow."""
    from neuralvolumetricreconstructionformedicalimages_amd import _abi
    from neuralvolumetricreconstructionformedicalimages_amd.engine import NAFEngine
    n, S = 300, 48
    rays = crossing_rays(n, seed=31).cuda()
    target = torch.rand(n, device="cuda") * 0.2
    weight = torch.full((n,), 1.0 / n, device="cuda")
    out = {}
    for tag, fuse, flags in (("tail", True, _abi.CFG_TEST_TINY_BLOCKS), ("separate", False, _abi.CFG_TEST_TINY_BLOCKS), ("plain", True, 0)):
        net, _ = naf_pair(seed=33, log2T=14, oracle=False)
        eng = NAFEngine(net, S, perturb=True, lr=3e-3, table_dtype=table_dtype, fuse_table_adam=fuse, cfg_flags=flags, scatter_mode=2)
        for step in range(3):
            eng.train_step(rays, target, weight, ray_base=step * n)
        torch.cuda.synchronize()
        spilled = eng.scatter_overflow(n)
        assert (spilled > 0) == (flags != 0), (tag, spilled)             # the tiny blocks really overflowed, the ordinary ones did not
        assert float(eng.emb_g.abs().max()) == 0.0                       # whatever the atomics left in the table was consumed
        out[tag] = (eng.emb.clone(), eng.emb_m.clone(), eng.emb_v.clone())
    for other in ("separate", "plain"):
        for a, b in zip(out["tail"], out[other]):
            assert float((a - b).abs().max()) <= 2e-5 * max(float(b.abs().max()), 1e-3), other


@pytest.mark.parametrize("prec", ["f32", "bf16"])
def test_padding_of_the_last_tile_emits_no_records(prec):
    """300 rays x 48 samples = 14 400 points end 64 points into a pass-1 tile, and every ray's last sample is clamped onto a face
    of the volume, where the x-neighbour pairs are unpaired from level 2 on: the padding threads (clones of that last point
    with a zero gradient) used to emit 8 records each, fill the tile's block and push real records out to atomics.  They emit
    nothing: no overflow, and the table gradient is bit-reproducible from run to run."""
    from neuralvolumetricreconstructionformedicalimages_amd.engine import NAFEngine
    n, S = 300, 48
    rays = crossing_rays(n, seed=31).cuda()
    target = torch.rand(n, device="cuda") * 0.2
    weight = torch.full((n,), 1.0 / n, device="cuda")
    grads = []
    for rep in range(3):
        net, _ = naf_pair(seed=33, log2T=14, oracle=False)
        eng = NAFEngine(net, S, perturb=True, lr=1e-3, table_dtype=torch.float32 if prec == "f32" else torch.bfloat16)
        eng.backward(rays, target, weight)
        torch.cuda.synchronize()
        assert eng.scatter_overflow(n) == 0 and sum(eng.scatter_overflow_levels(n)) == 0
        grads.append(eng.emb_g.clone())
    assert torch.equal(grads[0], grads[1]) and torch.equal(grads[0], grads[2]) and float(grads[0].abs().max()) > 0
