"""GPU parity of the HIP hash-grid encoder (through the C ABI) against the CPU oracle.

Bars: forward fp32 is BIT-EXACT against oracle/hash_ref.c (same op order, explicit fma); backward uses fp32
atomics whose order is not deterministic -> rel. tolerance 2e-5 of the largest gradient; 16-bit tables within
one storage ulp of the fp32 result computed from the same (rounded) table."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _mods():
    from neuralvolumetricreconstructionformedicalimages_amd import _abi, encoder
    from oracle import c_oracle, hashgrid_ref
    return _abi, encoder, c_oracle, hashgrid_ref


def _run_forward(_abi, x, emb, offs, H, layout, calc=False):
    B, D = x.shape
    L, C = offs.numel() - 1, emb.shape[1]
    shape = (B, L * C) if layout == _abi.LAYOUT_BLC else (L, B, C)
    out = torch.empty(shape, device="cuda", dtype=emb.dtype)
    jac = torch.empty(B, L, D, C, device="cuda", dtype=emb.dtype) if calc else None
    _abi.check(_abi.lib().naf_hash_encode_forward(_abi.ptr(x), _abi.ptr(emb), _abi.ptr(offs), _abi.ptr(out), B, D, C, L, H,
                                                  int(calc), _abi.ptr(jac), _abi.dtype_code(emb.dtype), layout,
                                                  _abi.stream_ptr()))
    torch.cuda.synchronize()
    return out, jac


@pytest.mark.parametrize("log2T,L,H,B", [(19, 16, 16, 5000), (12, 8, 4, 3001), (22, 16, 16, 2000)])
def test_forward_fp32_bit_exact(log2T, L, H, B):
    _abi, encoder, c_oracle, hr = _mods()
    offs = hr.level_offsets(L, H, log2T, 3)
    rng = np.random.default_rng(log2T)
    x = rng.random((B, 3), dtype=np.float32)
    x[0], x[1], x[2] = 0.0, 1.0, [0.0, 1.0, 0.5]           # cube corners / faces
    emb = rng.uniform(-1, 1, (int(offs[-1]), 2)).astype(np.float32)
    ref, _ = c_oracle.hash_encode_forward(x, emb, offs, H)
    xd, ed, od = torch.from_numpy(x).cuda(), torch.from_numpy(emb).cuda(), torch.from_numpy(offs).cuda()
    out_lbc, _ = _run_forward(_abi, xd, ed, od, H, _abi.LAYOUT_LBC)
    assert np.array_equal(out_lbc.cpu().numpy(), ref)
    out_blc, _ = _run_forward(_abi, xd, ed, od, H, _abi.LAYOUT_BLC)
    assert np.array_equal(out_blc.cpu().numpy(), ref.transpose(1, 0, 2).reshape(B, -1))


@pytest.mark.parametrize("D,C", [(2, 1), (2, 2), (2, 4), (2, 8), (3, 1), (3, 4), (3, 8)])
def test_forward_all_D_C(D, C):
    _abi, encoder, c_oracle, hr = _mods()
    offs = hr.level_offsets(6, 4, 9, D)
    rng = np.random.default_rng(D * 10 + C)
    x = rng.random((777, D), dtype=np.float32)
    emb = rng.uniform(-1, 1, (int(offs[-1]), C)).astype(np.float32)
    ref, jref = c_oracle.hash_encode_forward(x, emb, offs, 4, calc_grad_inputs=True)
    out, jac = _run_forward(_abi, torch.from_numpy(x).cuda(), torch.from_numpy(emb).cuda(), torch.from_numpy(offs).cuda(),
                            4, _abi.LAYOUT_LBC, calc=True)
    assert np.array_equal(out.cpu().numpy(), ref)
    assert np.array_equal(jac.cpu().numpy(), jref)


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
def test_forward_16bit_tables(dt):
    _abi, encoder, c_oracle, hr = _mods()
    offs = hr.level_offsets(16, 16, 19, 3)
    rng = np.random.default_rng(5)
    x = rng.random((4000, 3), dtype=np.float32)
    emb = torch.from_numpy(rng.uniform(-1, 1, (int(offs[-1]), 2)).astype(np.float32)).to(dt)
    ref, _ = c_oracle.hash_encode_forward(x, emb.float().numpy(), offs, 16)          # fp32 math on the rounded table
    out, _ = _run_forward(_abi, torch.from_numpy(x).cuda(), emb.cuda(), torch.from_numpy(offs).cuda(), 16, _abi.LAYOUT_LBC)
    expect = torch.from_numpy(ref).to(dt)                                               # one rounding at the store
    assert torch.equal(out.cpu(), expect)


@pytest.mark.parametrize("layout", ["blc", "lbc"])
def test_backward_fp32(layout):
    _abi, encoder, c_oracle, hr = _mods()
    offs = hr.level_offsets(16, 16, 19, 3)
    rng = np.random.default_rng(9)
    B = 6000
    x = rng.random((B, 3), dtype=np.float32)
    g = rng.standard_normal((B, 32)).astype(np.float32)
    emb = np.zeros((int(offs[-1]), 2), dtype=np.float32)
    ref, _ = c_oracle.hash_encode_backward(g, x, emb, offs, 16)
    gd = torch.from_numpy(g).cuda()
    lay = _abi.LAYOUT_BLC
    if layout == "lbc":
        gd = gd.view(B, 16, 2).permute(1, 0, 2).contiguous()
        lay = _abi.LAYOUT_LBC
    ge = torch.zeros(emb.shape, device="cuda")
    xd, od = torch.from_numpy(x).cuda(), torch.from_numpy(offs).cuda()
    _abi.check(_abi.lib().naf_hash_encode_backward(_abi.ptr(gd), _abi.ptr(xd), None, _abi.ptr(od), _abi.ptr(ge), B, 3, 2, 16, 16,
                                                   0, None, None, _abi.F32, lay, _abi.stream_ptr()))
    torch.cuda.synchronize()
    np.testing.assert_allclose(ge.cpu().numpy(), ref, rtol=0, atol=2e-5 * np.abs(ref).max())


@pytest.mark.parametrize("layout,dtype,C,log2T", [("blc", "fp32", 2, 19), ("lbc", "fp32", 2, 19), ("blc", "bf16", 2, 19), ("lbc", "fp16", 4, 16),
                                                  ("blc", "fp32", 4, 21)])
def test_backward_with_workspace_equals_the_atomic_route(layout, dtype, C, log2T):
    """naf_hash_encode_backward_ws (the binned two-pass scatter of the training path behind the operator's signature: caller's
    workspace, `+=` into grad_embeddings) against naf_hash_encode_backward (the reference scheme, one atomic per corner and channel) on
    the same inputs: 1e-5 of the largest sum for fp32 gradients (both accumulate the same fp32 products; only the order differs),
    3e-3 where the records carry 16-bit gradients.  Coordinates include points outside [0, 1] and samples of rays (runs of equal
    cells on the coarse levels); the caller's content of grad_embeddings is kept; a NULL workspace falls back to the atomic route."""
    _abi, encoder, c_oracle, hr = _mods()
    L, H = 16, 16
    offs = torch.from_numpy(hr.level_offsets(L, H, log2T, 3)).cuda()
    g0 = torch.Generator().manual_seed(11)
    n_rays, S = 96, 192                                            # 18 432 points: above the 2^13 floor of the binned scatter
    o = torch.rand(n_rays, 1, 3, generator=g0)
    d = torch.rand(n_rays, 1, 3, generator=g0) - 0.5
    # some samples leave the unit cube -- on its high side only: there the reference's weights stay in [0, 1] (a NEGATIVE coordinate
    # makes them -1e5 per dimension on the fine levels, and sums of such terms differ between two runs of the SAME atomic kernel)
    x = (o + d * torch.linspace(0, 1.2, S).view(1, S, 1)).clamp(min=0.0).reshape(-1, 3).contiguous()
    inside = ((x >= 0) & (x <= 1)).all(dim=1)
    assert 0.02 < float((~inside).float().mean()) < 0.5
    B = x.shape[0]
    tdt = {"fp32": torch.float32, "bf16": torch.bfloat16, "fp16": torch.float16}[dtype]
    g = torch.randn(B, L * C, generator=g0).to(tdt)
    lay = _abi.LAYOUT_BLC
    gd = g.cuda()
    if layout == "lbc":
        gd = gd.view(B, L, C).permute(1, 0, 2).contiguous()
        lay = _abi.LAYOUT_LBC
    xd = x.cuda()
    dtc = _abi.dtype_code(tdt)
    need = int(_abi.lib().naf_hash_encode_workspace_bytes(B, 3, C, L, log2T, dtc))
    assert need > 0
    assert int(_abi.lib().naf_hash_encode_workspace_bytes(4096, 3, C, L, log2T, dtc)) == 0      # small batches: the atomic route
    assert int(_abi.lib().naf_hash_encode_workspace_bytes(B, 2, C, L, log2T, dtc)) == 0         # D = 2 likewise
    ws = torch.empty(need + 256, dtype=torch.uint8, device="cuda")
    n_rows = int(offs[-1])
    base = torch.rand(n_rows, C, device="cuda")
    out = {}
    for route in ("atomic", "ws", "ws_null"):
        ge = base.clone()
        if route == "atomic":
            _abi.check(_abi.lib().naf_hash_encode_backward(_abi.ptr(gd), _abi.ptr(xd), None, _abi.ptr(offs), _abi.ptr(ge), B, 3, C, L, H,
                                                           0, None, None, dtc, lay, _abi.stream_ptr()))
        else:
            w = ws if route == "ws" else None
            _abi.check(_abi.lib().naf_hash_encode_backward_ws(_abi.ptr(gd), _abi.ptr(xd), None, _abi.ptr(offs), _abi.ptr(ge), B, 3, C, L, H,
                                                              0, None, None, dtc, lay, log2T, _abi.ptr(w), 0 if w is None else w.numel(),
                                                              _abi.stream_ptr()))
        torch.cuda.synchronize()
        out[route] = (ge - base).double().cpu().numpy()
    # Points outside [0, 1] take the atomic route inside the binned scatter too (its fixed-point sums are scaled for weights in [0, 1]);
    # the comparison is relative per element, with an absolute floor taken from the in-range points alone.
    ge_in = torch.zeros(n_rows, C, device="cuda")
    keep = inside.cuda()
    g_in = (gd if layout == "blc" else gd.permute(1, 0, 2).reshape(B, L * C)).clone()
    g_in[~keep] = 0
    g_in = g_in if layout == "blc" else g_in.view(B, L, C).permute(1, 0, 2).contiguous()
    x_in = xd.clone()
    x_in[~keep] = 0.5
    _abi.check(_abi.lib().naf_hash_encode_backward(_abi.ptr(g_in), _abi.ptr(x_in), None, _abi.ptr(offs), _abi.ptr(ge_in), B, 3, C, L, H,
                                                   0, None, None, dtc, lay, _abi.stream_ptr()))
    torch.cuda.synchronize()
    scale = float(ge_in.abs().max())
    tol = 1e-5 if dtype == "fp32" else 3e-3
    np.testing.assert_allclose(out["ws"], out["atomic"], rtol=max(tol, 2e-5), atol=tol * scale)
    np.testing.assert_allclose(out["ws_null"], out["atomic"], rtol=2e-5, atol=2e-5 * scale)


def test_backward_accumulates_and_input_grad():
    _abi, encoder, c_oracle, hr = _mods()
    offs = hr.level_offsets(5, 4, 10, 3)
    rng = np.random.default_rng(2)
    B = 500
    x = rng.random((B, 3), dtype=np.float32)
    emb = rng.uniform(-1, 1, (int(offs[-1]), 2)).astype(np.float32)
    g = rng.standard_normal((B, 10)).astype(np.float32)
    _, jref = c_oracle.hash_encode_forward(x, emb, offs, 4, calc_grad_inputs=True)
    ge_ref, gi_ref = c_oracle.hash_encode_backward(g, x, emb, offs, 4, dy_dx=jref)
    xd, ed, od, gd = (torch.from_numpy(a).cuda() for a in (x, emb, offs, g))
    _, jac = _run_forward(_abi, xd, ed, od, 4, _abi.LAYOUT_BLC, calc=True)
    ge = torch.ones(emb.shape, device="cuda")               # (+=) contract: starts from the caller's content
    gi = torch.zeros(B, 3, device="cuda")
    _abi.check(_abi.lib().naf_hash_encode_backward(_abi.ptr(gd), _abi.ptr(xd), _abi.ptr(ed), _abi.ptr(od), _abi.ptr(ge), B, 3, 2, 5, 4,
                                                   1, _abi.ptr(jac), _abi.ptr(gi), _abi.F32, _abi.LAYOUT_BLC, _abi.stream_ptr()))
    torch.cuda.synchronize()
    np.testing.assert_allclose(ge.cpu().numpy() - 1.0, ge_ref, rtol=0, atol=1e-4)
    np.testing.assert_allclose(gi.cpu().numpy(), gi_ref, rtol=1e-4, atol=1e-5)


def test_error_contract():
    _abi, encoder, c_oracle, hr = _mods()
    x = torch.zeros(4, 3, device="cuda")
    emb = torch.zeros(100, 3, device="cuda")
    offs = torch.tensor([0, 100], dtype=torch.int32, device="cuda")
    out = torch.zeros(4, 3, device="cuda")
    rc = _abi.lib().naf_hash_encode_forward(_abi.ptr(x), _abi.ptr(emb), _abi.ptr(offs), _abi.ptr(out), 4, 3, 3, 1, 16, 0, None, 0, 0, None)
    assert rc == -2 and b"C must be 1, 2, 4, or 8" in _abi.lib().naf_last_error()
    with pytest.raises(RuntimeError, match="C must be 1, 2, 4, or 8"):
        _abi.check(rc)
    assert _abi.lib().naf_hash_encode_forward(None, _abi.ptr(emb), _abi.ptr(offs), _abi.ptr(out), 4, 3, 2, 1, 16, 0, None, 0, 0, None) == -1
    # B == 0 is a no-op like an empty launch
    assert _abi.lib().naf_hash_encode_forward(_abi.ptr(x), _abi.ptr(emb), _abi.ptr(offs), _abi.ptr(out), 0, 3, 2, 1, 16, 0, None, 0, 0, None) == 0


def test_module_matches_oracle_module_and_raises_out_of_range():
    _abi, encoder, c_oracle, hr = _mods()
    torch.manual_seed(0)
    enc = encoder.HashEncoder(3, 16, 2, 16, 19).cuda()
    ref = hr.HashEncoderRef(3, 16, 2, 16, 19)
    ref.embeddings.data.copy_(enc.embeddings.data.cpu())
    pts = (torch.rand(3000, 3) - 0.5) * 0.6
    y = enc(pts.cuda(), 0.3)
    y_ref = ref(pts, 0.3)
    np.testing.assert_allclose(y.detach().cpu().numpy(), y_ref.detach().numpy(), rtol=1e-5, atol=1e-9)
    g = torch.randn_like(y_ref)
    y.backward(g.cuda())
    y_ref.backward(g)
    gr = ref.embeddings.grad.numpy()
    np.testing.assert_allclose(enc.embeddings.grad.cpu().numpy(), gr, rtol=0, atol=2e-5 * np.abs(gr).max())
    with pytest.raises(ValueError, match="not in"):
        enc(torch.tensor([[0.0, 0.0, 0.31]], device="cuda"), 0.3)


def test_full_size_properties():
    """BASELINE-size launch (B = 2^21 points, T = 2^19) checked through size-independent properties:
    constant table -> constant output; linearity in the table; <g, F(E)> == <F^T g, E>."""
    _abi, encoder, c_oracle, hr = _mods()
    offs = hr.level_offsets(16, 16, 19, 3)
    od = torch.from_numpy(offs).cuda()
    B = 1 << 21
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.rand(B, 3, device="cuda", generator=g)
    rows = int(offs[-1])
    const = torch.full((rows, 2), 0.375, device="cuda")
    out, _ = _run_forward(_abi, x, const, od, 16, _abi.LAYOUT_LBC)
    assert (out - 0.375).abs().max().item() < 3e-7
    e1 = torch.rand(rows, 2, device="cuda", generator=g) - 0.5
    e2 = torch.rand(rows, 2, device="cuda", generator=g) - 0.5
    f1, _ = _run_forward(_abi, x, e1, od, 16, _abi.LAYOUT_LBC)
    f2, _ = _run_forward(_abi, x, e2, od, 16, _abi.LAYOUT_LBC)
    f12, _ = _run_forward(_abi, x, (e1 + 2 * e2).contiguous(), od, 16, _abi.LAYOUT_LBC)
    assert (f12 - (f1 + 2 * f2)).abs().max().item() < 1e-5
    gr = torch.randn(16, B, 2, device="cuda", generator=g)
    ge = torch.zeros(rows, 2, device="cuda")
    _abi.check(_abi.lib().naf_hash_encode_backward(_abi.ptr(gr), _abi.ptr(x), None, _abi.ptr(od), _abi.ptr(ge), B, 3, 2, 16, 16,
                                                   0, None, None, _abi.F32, _abi.LAYOUT_LBC, _abi.stream_ptr()))
    lhs = (f1.double() * gr.double()).sum().item()
    rhs = (ge.double() * e1.double()).sum().item()
    assert abs(lhs - rhs) <= 1e-4 * max(abs(lhs), 1.0)
