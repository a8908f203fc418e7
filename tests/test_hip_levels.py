"""Level-parallel training entry points (naf_levels_encode / _field_step / _scatter, include/naf_hip.h) driven from ONE process:
N virtual ranks, each owning L / N levels, with the two all-to-alls done by slicing.  The result must be the single-GPU step
(naf_render_train_adam through NAFEngine) on the concatenated batch -- for N = 2 (the reducer applies Adam to the owned levels),
N = 8 (two levels per rank: split reducer launches, gradient written out, separate Adam pass on the owned rows) and a batch small
enough for the atomic scatter.  The multi-process form of the same step is covered in test_hip_dist.py."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _make(seed=0, log2T=14):
    from neuralvolumetricreconstructionformedicalimages_amd.encoder import HashEncoder
    from neuralvolumetricreconstructionformedicalimages_amd.network import DensityNetwork
    torch.manual_seed(seed)
    enc = HashEncoder(3, 16, 2, 16, log2T)
    enc.embeddings.data.uniform_(-0.1, 0.1)
    net = DensityNetwork(enc, bound=0.3, num_layers=4, hidden_dim=32, skips=[2], out_dim=1, last_activation="sigmoid")
    return net.cuda()


def _batch(n, S, seed=11):
    g = torch.Generator().manual_seed(seed)
    ang = torch.rand(n, generator=g) * 6.283
    o = torch.stack([torch.cos(ang), torch.sin(ang), torch.zeros(n)], -1)
    d = (torch.rand(n, 3, generator=g) - 0.5) * 0.4 - o
    rays = torch.cat([o, d, torch.full((n, 1), 0.6), torch.full((n, 1), 1.4)], -1)
    return rays.cuda(), (torch.rand(n, generator=g) * 0.3).cuda(), (torch.rand(n, generator=g) > 0.2).cuda()


def _levels_step(eng, N, rays, target, weight, pad=0, halves=1):
    """One optimisation step of `eng` (a single-process NAFEngine used as a bag of buffers) the level-parallel way.  `pad`: elements
    of NaN between the gradient blocks of two ranks (block_stride_bytes larger than a block).  `halves` = 2: the split form -- every
    rank's rays in two halves that travel separately (one half's all-to-all can then sit behind the other half's kernels); the batch is
    read as [half 0 of rank 0 .. N-1 | half 1 of rank 0 .. N-1], each half is encoded and rendered on its own, and ONE scatter call takes
    the 2 N returned blocks as those of 2 N ranks -- no entry point knows about halves."""
    from neuralvolumetricreconstructionformedicalimages_amd import _abi, fused
    lib, sp = _abi.lib(), _abi.stream_ptr()
    enc = eng.net.encoder
    L, C, S = enc.num_levels, enc.level_dim, eng.n_samples
    n_all = rays.shape[0]
    n_half = n_all // halves                                      # rays of one half, all ranks
    n, per = n_half // N, L // N
    run = n * S * C                                               # elements of one (rank, half, level)
    fdt = torch.float32 if int(eng.mlp_precision) == _abi.F32 else torch.bfloat16
    esz = 4 if fdt == torch.float32 else 2
    cfg_all = eng._cfg(0)
    ws = fused.workspace(cfg_all, n_all * S, eng.device)
    eng.loss.zero_()
    part = torch.zeros(1, device=eng.device)
    acc = torch.empty(n_all, device=eng.device)
    grads = []                                                    # [half * N + rank] -> [L, run]
    for h in range(halves):
        base = h * n_half
        rays_h = rays[base:base + n_half].contiguous()
        cfg_h = eng._cfg(base)
        feats = []
        for k in range(N):                                        # every owner encodes its levels for all points of the half
            out = torch.full((N, per, run), float("nan"), dtype=fdt, device=eng.device)      # one block per destination rank
            _abi.check(lib.naf_levels_encode(_abi.ptr(rays_h), None, _abi.ptr(eng.table), _abi.ptr(eng.offsets), _abi.ptr(out), n_half, N,
                                             ctypes.byref(cfg_h), k * per, (k + 1) * per, sp), "levels_encode")
            feats.append(out)
        for r in range(N):                                        # "all-to-all": rank r receives every owner's levels of its points
            feat = torch.cat([f[r] for f in feats], 0).contiguous()                  # [L, run]
            dfeat = torch.full((L, run), float("nan"), dtype=fdt, device=eng.device)
            cfg = eng._cfg(base + r * n)
            sl = slice(base + r * n, base + (r + 1) * n)
            _abi.check(lib.naf_levels_field_step(_abi.ptr(rays[sl].contiguous()), None, _abi.ptr(target[sl].contiguous()), _abi.ptr(weight[sl].contiguous()),
                                                 _abi.ptr(feat), _abi.ptr(eng.mlp), _abi.ptr(acc[sl]), _abi.ptr(dfeat), _abi.ptr(eng.mlp_g),
                                                 _abi.ptr(part), n, ctypes.byref(cfg), _abi.ptr(ws), None, sp), "levels_field_step")
            eng.loss.add_(part)                                   # (the all-reduce of the real step)
            grads.append(dfeat)
    eng.step_count += 1
    st = _abi.TableAdam()
    st.param, st.exp_avg, st.exp_avg_sq = eng.emb.data_ptr(), eng.emb_m.data_ptr(), eng.emb_v.data_ptr()
    st.param_lp = None if eng.emb_lp is None else eng.emb_lp.data_ptr()
    st.lp_dtype = 0 if eng.emb_lp is None else _abi.dtype_code(eng.table_dtype)
    b1, b2 = eng.betas
    st.n, st.lr, st.beta1, st.beta2, st.eps, st.step, st.grad_scale = eng.emb.numel(), eng.lr, b1, b2, eng.eps, eng.step_count, 1.0
    offs = eng.offsets.tolist()
    fused_tail = []
    V = halves * N                                                # blocks the owner receives
    for k in range(N):                                            # "all-to-all" back: owner k receives its levels' gradients from every rank (and half)
        blocks = torch.full((V, per * run + pad), float("nan"), dtype=fdt, device=eng.device)
        blocks[:, :per * run] = torch.stack([g[k * per:(k + 1) * per].reshape(-1) for g in grads], 0)      # [V, per * run (+ pad)]
        applied = ctypes.c_int(-1)
        _abi.check(lib.naf_levels_scatter(_abi.ptr(rays), None, _abi.ptr(blocks), (per * run + pad) * esz, V, _abi.ptr(eng.offsets), _abi.ptr(eng.emb_g),
                                          n_all, ctypes.byref(cfg_all), k * per, (k + 1) * per, _abi.ptr(ws), ctypes.byref(st),
                                          ctypes.byref(applied), sp), "levels_scatter")
        fused_tail.append(applied.value)
        if not applied.value:
            eng._adam_rows(offs[k * per] * C, offs[(k + 1) * per] * C)
    eng._adam(eng.mlp, eng.mlp_m, eng.mlp_v, eng.mlp_g, None, 0, "adam_step(mlp)")
    return acc, fused_tail


@pytest.mark.parametrize("N,table,n_rays,S,buckets,expect_fused", [
    (2, "fp32", 256, 64, 0, True), (2, "bf16", 256, 64, 0, True),
    (8, "bf16", 512, 64, 0, False), (8, "fp32", 256, 48, 0, False),      # two levels per rank, 64 row buckets: split reducer launches, separate Adam pass
    (8, "bf16", 512, 64, 2, True), (8, "fp32", 256, 48, 2, True),        # ... 256 row buckets (NAF_CFG_MIN_BUCKETS, what the engine asks for): the Adam tail applies
    (4, "bf16", 512, 64, 1, True),
    (4, "bf16", 16, 32, 0, False)])                                      # 512 points: the atomic scatter
def test_level_parallel_entry_points_equal_the_single_gpu_step(N, table, n_rays, S, buckets, expect_fused):
    from neuralvolumetricreconstructionformedicalimages_amd import _abi
    from neuralvolumetricreconstructionformedicalimages_amd.engine import NAFEngine
    dtype = {"fp32": torch.float32, "bf16": torch.bfloat16}[table]
    ref = NAFEngine(_make(), S, perturb=True, lr=1e-2, table_dtype=dtype, seed=5)
    lev = NAFEngine(_make(), S, perturb=True, lr=1e-2, table_dtype=dtype, seed=5)
    lev._levels_flags = buckets << _abi.CFG_MIN_BUCKETS_SHIFT
    rays, target, mask = _batch(n_rays, S)
    weight = mask.float() / mask.float().sum()
    for step in range(2):
        ref.train_step(rays, target, weight)
        acc, fused_tail = _levels_step(lev, N, rays, target, weight)
        assert all(bool(v) == expect_fused for v in fused_tail), fused_tail
        torch.cuda.synchronize()
        # the forward pass sees the same table: the line integrals agree to the rounding of one bf16 / fp32 MLP evaluation order
        np.testing.assert_allclose(acc.cpu().numpy(), ref.acc[:n_rays].cpu().numpy(), rtol=2e-6 if table == "fp32" else 2e-2, atol=1e-6)
        np.testing.assert_allclose(float(lev.loss.item()), float(ref.loss.item()), rtol=1e-4 if table == "fp32" else 1e-2)
        assert float(lev.emb_g.abs().max()) == 0.0                                   # consumed / left clear for the next step
    a, b = lev.emb.cpu().numpy(), ref.emb.cpu().numpy()
    # Adam turns a gradient that is zero up to rounding into a step of +-lr: a handful of such rows may differ by O(lr)
    assert np.mean(np.abs(a - b) > 2e-3) < 1e-3
    np.testing.assert_allclose(lev.mlp.cpu().numpy(), ref.mlp.cpu().numpy(), rtol=0, atol=2e-4 if table == "fp32" else 2e-3)
    m1, m2 = lev.emb_m.cpu().numpy(), ref.emb_m.cpu().numpy()
    assert np.mean(np.abs(m1 - m2) > (1e-5 if table == "fp32" else 2e-2) * float(np.abs(m2).max()) + 1e-9) < 1e-2
    if lev.emb_lp is not None:
        assert torch.equal(lev.emb_lp, lev.emb.to(lev.table_dtype))                 # the shadow follows the master on every owned row


@pytest.mark.parametrize("N,table,n_rays,S,buckets", [(2, "bf16", 256, 64, 0), (4, "bf16", 512, 64, 1), (8, "bf16", 512, 64, 2), (8, "fp32", 256, 48, 2),
                                                       (4, "fp32", 200, 50, 0)])
def test_level_parallel_step_in_two_halves_equals_the_single_gpu_step(N, table, n_rays, S, buckets):
    """The split form: each rank's rays travel as two halves -- two encodes, two MLP passes, ONE scatter over 2 N
    blocks.  Counter-based jitter, so the sample depths depend on where a ray sits in the batch: the reference is the single-GPU step
    on the batch in the order the halves imply, which is the order given."""
    from neuralvolumetricreconstructionformedicalimages_amd import _abi
    from neuralvolumetricreconstructionformedicalimages_amd.engine import NAFEngine
    dtype = {"fp32": torch.float32, "bf16": torch.bfloat16}[table]
    ref = NAFEngine(_make(), S, perturb=True, lr=1e-2, table_dtype=dtype, seed=5)
    lev = NAFEngine(_make(), S, perturb=True, lr=1e-2, table_dtype=dtype, seed=5)
    lev._levels_flags = buckets << _abi.CFG_MIN_BUCKETS_SHIFT
    rays, target, mask = _batch(n_rays, S)
    weight = mask.float() / mask.float().sum()
    for step in range(2):
        ref.train_step(rays, target, weight)
        acc, _ = _levels_step(lev, N, rays, target, weight, halves=2)
        torch.cuda.synchronize()
        np.testing.assert_allclose(acc.cpu().numpy(), ref.acc[:n_rays].cpu().numpy(), rtol=2e-6 if table == "fp32" else 2e-2, atol=1e-6)
        np.testing.assert_allclose(float(lev.loss.item()), float(ref.loss.item()), rtol=1e-4 if table == "fp32" else 1e-2)
        assert float(lev.emb_g.abs().max()) == 0.0
    a, b = lev.emb.cpu().numpy(), ref.emb.cpu().numpy()
    assert np.mean(np.abs(a - b) > 2e-3) < 1e-3
    np.testing.assert_allclose(lev.mlp.cpu().numpy(), ref.mlp.cpu().numpy(), rtol=0, atol=2e-4 if table == "fp32" else 2e-3)
    m1, m2 = lev.emb_m.cpu().numpy(), ref.emb_m.cpu().numpy()
    assert np.mean(np.abs(m1 - m2) > (1e-5 if table == "fp32" else 2e-2) * float(np.abs(m2).max()) + 1e-9) < 1e-2
    if lev.emb_lp is not None:
        assert torch.equal(lev.emb_lp, lev.emb.to(lev.table_dtype))


@pytest.mark.parametrize("N,n_rays,S,buckets,pad,log2T", [
    (2, 256, 64, 0, 0, 14), (4, 512, 64, 1, 6, 14), (8, 512, 64, 2, 2, 14), (16, 400, 48, 2, 10, 14),
    (4, 200, 50, 0, 2, 14),                                               # 10 000 points: the last tile is ragged, ranks end inside tiles
    (2, 8192, 192, 0, 0, 19), (8, 8192, 192, 2, 4, 19)])                  # 1.5 M points: the all-levels-per-workgroup variants of pass 1
def test_scatter_reads_the_gradient_blocks_in_place_bit_for_bit(N, n_rays, S, buckets, pad, log2T):
    """Two bf16 channels: pass 1 of the scatter finds a point's gradient inside its source rank's block and takes the step's maximum
    |gradient| itself (scatter_v2.h, GradBlocks); NAF_CFG_LEVELS_GATHER_PASS keeps the separate re-ordering pass of rounds 3-4.
    Same records, same fixed-point scale: table, moments and shadow are equal bit for bit after two steps.  (Every case here has
    unsplit reducer launches: split ones add their partial sums with fp32 atomics in no fixed order, whichever way the gradients were read;
    test_level_parallel_entry_points_equal_the_single_gpu_step covers them through the in-place route against the single-GPU step.)"""
    from neuralvolumetricreconstructionformedicalimages_amd import _abi
    from neuralvolumetricreconstructionformedicalimages_amd.engine import NAFEngine
    a = NAFEngine(_make(log2T=log2T), S, perturb=True, lr=1e-2, table_dtype=torch.bfloat16, seed=5)
    b = NAFEngine(_make(log2T=log2T), S, perturb=True, lr=1e-2, table_dtype=torch.bfloat16, seed=5)
    a._levels_flags = buckets << _abi.CFG_MIN_BUCKETS_SHIFT
    b._levels_flags = (buckets << _abi.CFG_MIN_BUCKETS_SHIFT) | _abi.CFG_LEVELS_GATHER_PASS
    rays, target, mask = _batch(n_rays, S)
    weight = mask.float() / mask.float().sum()
    start = a.emb.clone()
    for step in range(2):
        _, tail_a = _levels_step(a, N, rays, target, weight, pad=pad)
        _, tail_b = _levels_step(b, N, rays, target, weight, pad=0)
        assert tail_a == tail_b
    torch.cuda.synchronize()
    assert torch.isfinite(a.emb).all()
    assert float((a.emb - start).abs().max()) > 0.0                      # the table moved
    for name in ("emb", "emb_m", "emb_v", "emb_lp", "mlp"):
        assert torch.equal(getattr(a, name), getattr(b, name)), name


def test_a_non_finite_gradient_poisons_the_in_place_scatter_as_it_does_the_gather_pass():
    """The maximum travels as a bit pattern (NaN > Inf > finite), so one Inf among the gradient blocks reaches the fixed-point scale."""
    from neuralvolumetricreconstructionformedicalimages_amd import _abi, fused
    from neuralvolumetricreconstructionformedicalimages_amd.engine import NAFEngine
    S, n_rays, N = 64, 256, 4
    out = []
    for flag in (0, _abi.CFG_LEVELS_GATHER_PASS):
        eng = NAFEngine(_make(), S, perturb=True, lr=1e-2, table_dtype=torch.bfloat16, seed=5)
        eng._levels_flags = flag
        rays, _, _ = _batch(n_rays, S)
        enc = eng.net.encoder
        L, C = enc.num_levels, enc.level_dim
        per, run = L // N, n_rays // N * S * C
        cfg_all = eng._cfg(0)
        ws = fused.workspace(cfg_all, n_rays * S, eng.device)
        g = torch.Generator(device="cuda").manual_seed(3)
        blocks = (torch.randn(N, per * run, device="cuda", generator=g) * 1e-3).to(torch.bfloat16)
        blocks[2, run + 77] = float("inf")
        _abi.check(_abi.lib().naf_levels_scatter(_abi.ptr(rays), None, _abi.ptr(blocks), per * run * 2, N, _abi.ptr(eng.offsets), _abi.ptr(eng.emb_g),
                                                 n_rays, ctypes.byref(cfg_all), per, 2 * per, _abi.ptr(ws), None, None, _abi.stream_ptr()), "levels_scatter")
        torch.cuda.synchronize()
        out.append(eng.emb_g.clone())
    offs = eng.offsets.tolist()
    owned = out[0].reshape(-1)[offs[per] * C:offs[2 * per] * C]
    assert not torch.isfinite(owned).all()
    assert torch.equal(torch.isfinite(out[0]), torch.isfinite(out[1]))
    assert torch.equal(torch.nan_to_num(out[0], nan=7.0, posinf=8.0, neginf=9.0), torch.nan_to_num(out[1], nan=7.0, posinf=8.0, neginf=9.0))


def test_level_parallel_first_step_gradient_matches_the_plain_backward_fp32():
    """fp32, one step, no optimiser: the table gradient the owners write out (adam = NULL) against naf_render_train's."""
    from neuralvolumetricreconstructionformedicalimages_amd import _abi, fused
    from neuralvolumetricreconstructionformedicalimages_amd.engine import NAFEngine
    S, n_rays, N = 64, 256, 4
    ref = NAFEngine(_make(), S, perturb=True, lr=1e-2, seed=5)
    lev = NAFEngine(_make(), S, perturb=True, lr=1e-2, seed=5)
    rays, target, mask = _batch(n_rays, S)
    weight = mask.float() / mask.float().sum()
    ref.backward(rays, target, weight)
    lib, sp = _abi.lib(), _abi.stream_ptr()
    enc = lev.net.encoder
    L, C = enc.num_levels, enc.level_dim
    n, per = n_rays // N, L // N
    run = n * S * C
    cfg_all = lev._cfg(0)
    ws = fused.workspace(cfg_all, n_rays * S, lev.device)
    feats = []
    for k in range(N):
        out = torch.empty(N, per, run, device=lev.device)
        _abi.check(lib.naf_levels_encode(_abi.ptr(rays), None, _abi.ptr(lev.table), _abi.ptr(lev.offsets), _abi.ptr(out), n_rays, N,
                                         ctypes.byref(cfg_all), k * per, (k + 1) * per, sp), "levels_encode")
        feats.append(out)
    # the features are the ones the single-GPU step computes, bit for bit
    full = torch.cat([f.transpose(0, 1) for f in feats], 0).reshape(L, n_rays * S * C)      # [level][rank][points] = [level][all points]
    ws_ref = fused.workspace(ref._cfg(0), n_rays * S, ref.device)
    want = ws_ref[:full.numel() * 4].view(torch.float32).reshape(L, -1)
    assert torch.equal(full, want)
    lev.loss.zero_()
    part = torch.zeros(1, device=lev.device)
    acc = torch.empty(n_rays, device=lev.device)
    grads = []
    for r in range(N):
        feat = torch.cat([f[r] for f in feats], 0).contiguous()
        dfeat = torch.empty(L, run, device=lev.device)
        sl = slice(r * n, (r + 1) * n)
        _abi.check(lib.naf_levels_field_step(_abi.ptr(rays[sl].contiguous()), None, _abi.ptr(target[sl].contiguous()), _abi.ptr(weight[sl].contiguous()),
                                             _abi.ptr(feat), _abi.ptr(lev.mlp), _abi.ptr(acc[sl]), _abi.ptr(dfeat), _abi.ptr(lev.mlp_g),
                                             _abi.ptr(part), n, ctypes.byref(lev._cfg(r * n)), _abi.ptr(ws), None, sp), "levels_field_step")
        lev.loss.add_(part)
        grads.append(dfeat)
    for k in range(N):
        blocks = torch.stack([g[k * per:(k + 1) * per].reshape(-1) for g in grads], 0).contiguous()
        _abi.check(lib.naf_levels_scatter(_abi.ptr(rays), None, _abi.ptr(blocks), per * run * 4, N, _abi.ptr(lev.offsets), _abi.ptr(lev.emb_g),
                                          n_rays, ctypes.byref(cfg_all), k * per, (k + 1) * per, _abi.ptr(ws), None, None, sp), "levels_scatter")
    torch.cuda.synchronize()
    g, w = lev.emb_g.cpu().numpy().ravel(), ref.emb_g.cpu().numpy().ravel()
    assert np.linalg.norm(g - w) <= 1e-5 * np.linalg.norm(w)
    np.testing.assert_allclose(lev.mlp_g.cpu().numpy(), ref.mlp_g.cpu().numpy(), rtol=1e-4, atol=1e-7 * float(ref.mlp_g.abs().max()) + 1e-12)
    np.testing.assert_allclose(float(lev.loss.item()), float(ref.loss.item()), rtol=1e-5)


def test_level_parallel_entry_points_reject_bad_arguments():
    from neuralvolumetricreconstructionformedicalimages_amd import _abi, fused
    from neuralvolumetricreconstructionformedicalimages_amd.engine import NAFEngine
    eng = NAFEngine(_make(), 32, perturb=True, lr=1e-2)
    rays, target, mask = _batch(16, 32)
    cfg = eng._cfg(0)
    ws = fused.workspace(cfg, 16 * 32, eng.device)
    lib, sp = _abi.lib(), _abi.stream_ptr()
    out = torch.empty(16, 16 * 32 * 2, device="cuda")
    assert lib.naf_levels_encode(_abi.ptr(rays), None, _abi.ptr(eng.table), _abi.ptr(eng.offsets), _abi.ptr(out), 16, 2, ctypes.byref(cfg), 4, 4, sp) != 0
    assert lib.naf_levels_encode(_abi.ptr(rays), None, _abi.ptr(eng.table), _abi.ptr(eng.offsets), _abi.ptr(out), 16, 2, ctypes.byref(cfg), 8, 17, sp) != 0
    assert lib.naf_levels_encode(_abi.ptr(rays), None, _abi.ptr(eng.table), _abi.ptr(eng.offsets), _abi.ptr(out), 16, 3, ctypes.byref(cfg), 0, 8, sp) != 0
    # 16 rays cannot come from 3 ranks with equal shares; a block stride shorter than a block
    assert lib.naf_levels_scatter(_abi.ptr(rays), None, _abi.ptr(out), 4096, 3, _abi.ptr(eng.offsets), _abi.ptr(eng.emb_g), 16, ctypes.byref(cfg), 0, 8,
                                  _abi.ptr(ws), None, None, sp) != 0
    assert lib.naf_levels_scatter(_abi.ptr(rays), None, _abi.ptr(out), 64, 2, _abi.ptr(eng.offsets), _abi.ptr(eng.emb_g), 16, ctypes.byref(cfg), 0, 8,
                                  _abi.ptr(ws), None, None, sp) != 0
    assert b"levels" in _abi.lib().naf_last_error()
