"""Analytic pins for the hash-encoder oracle (the reference holds no vectors for it)."""
import numpy as np
import torch

from oracle import c_oracle, hashgrid_ref as hr


def _setup(log2T=12, L=8, H=4, C=2, seed=0):
    offs = hr.level_offsets(L, H, log2T, 3)
    rng = np.random.default_rng(seed)
    emb = rng.uniform(-1, 1, (int(offs[-1]), C)).astype(np.float32)
    return offs, emb, rng


def test_c_and_numpy_forward_bit_identical():
    offs, emb, rng = _setup()
    x = rng.random((513, 3), dtype=np.float32)
    x[0] = 0.0
    x[1] = 1.0
    x[2] = [0.0, 1.0, 0.5]
    out_c, _ = c_oracle.hash_encode_forward(x, emb, offs, 4)
    out_n = hr.hash_encode_forward(x, emb, offs, 4)
    assert np.array_equal(out_c.transpose(1, 0, 2).reshape(513, -1), out_n)


def test_full_size_table_agreement():
    offs = hr.level_offsets(16, 16, 19, 3)
    rng = np.random.default_rng(1)
    emb = rng.uniform(-1e-4, 1e-4, (int(offs[-1]), 2)).astype(np.float32)
    x = rng.random((300, 3), dtype=np.float32)
    out_c, _ = c_oracle.hash_encode_forward(x, emb, offs, 16)
    assert np.array_equal(out_c.transpose(1, 0, 2).reshape(300, -1), hr.hash_encode_forward(x, emb, offs, 16))


def test_constant_table_gives_constant_output():
    offs, emb, rng = _setup()
    emb[:] = 0.375                                    # weights sum to 1 (up to fp32 rounding)
    x = rng.random((200, 3), dtype=np.float32)
    out = hr.hash_encode_forward(x, emb, offs, 4)
    np.testing.assert_allclose(out, 0.375, rtol=3e-7)


def test_trilinear_reproduction_on_dense_level():
    # level 0 (res 4, 125 rows, dense x+5y+25z): a table linear in the grid coords is reproduced exactly
    offs, emb, rng = _setup()
    g = np.arange(5, dtype=np.float32)
    X, Y, Z = np.meshgrid(g, g, g, indexing="ij")
    lin = (0.5 * X + 0.25 * Y - 0.125 * Z + 1.0)
    rows = (X + 5 * Y + 25 * Z).astype(np.int64).ravel()
    emb[rows, 0] = lin.ravel()
    emb[rows, 1] = -lin.ravel()
    x = rng.random((300, 3), dtype=np.float32)
    out = hr.hash_encode_forward(x, emb, offs, 4)
    pos = x * 3.0 + 0.5                                # scale = 2^0*4 - 1
    expect = 0.5 * pos[:, 0] + 0.25 * pos[:, 1] - 0.125 * pos[:, 2] + 1.0
    np.testing.assert_allclose(out[:, 0], expect, rtol=1e-6)
    np.testing.assert_allclose(out[:, 1], -expect, rtol=1e-6)


def test_backward_is_adjoint_of_forward():
    offs, emb, rng = _setup()
    x = rng.random((257, 3), dtype=np.float32)
    g = rng.standard_normal((257, 16)).astype(np.float32)
    out = hr.hash_encode_forward(x, emb, offs, 4).astype(np.float64)
    ge_c, _ = c_oracle.hash_encode_backward(g, x, emb, offs, 4)
    ge_n = hr.hash_encode_backward(g, x, offs, 4, emb.shape[0], 2)
    np.testing.assert_allclose(ge_c, ge_n, rtol=1e-5, atol=1e-6)
    # <g, F(E)> == <F^T g, E> because F is linear in the table
    lhs = float((out * g).sum())
    rhs = float((ge_n.astype(np.float64) * emb).sum())
    np.testing.assert_allclose(lhs, rhs, rtol=1e-5)


def test_torch_encoder_matches_numpy_and_autograd_matches_scatter():
    offs, emb, rng = _setup()
    enc = hr.HashEncoderRef(3, 8, 2, 4, 12)
    enc.embeddings.data.copy_(torch.from_numpy(emb))
    pts = (rng.random((100, 3), dtype=np.float32) - 0.5) * 0.6     # inside [-0.3, 0.3]
    y = enc(torch.from_numpy(pts), 0.3)
    x01 = ((torch.from_numpy(pts) + 0.3) / 0.6).numpy()
    np.testing.assert_allclose(y.detach().numpy(), hr.hash_encode_forward(x01, emb, offs, 4), rtol=1e-5, atol=1e-6)
    g = rng.standard_normal((100, 16)).astype(np.float32)
    y.backward(torch.from_numpy(g))
    np.testing.assert_allclose(enc.embeddings.grad.numpy(), hr.hash_encode_backward(g, x01, offs, 4, emb.shape[0], 2),
                               rtol=1e-4, atol=1e-5)


def test_range_check_raises():
    enc = hr.HashEncoderRef(3, 4, 2, 4, 10)
    try:
        enc(torch.tensor([[0.0, 0.0, 0.31]]), 0.3)
    except ValueError:
        return
    raise AssertionError("expected ValueError (hashgrid.py:122-123)")


def test_dy_dx_exact_mode_is_the_derivative_and_reference_mode_is_the_documented_defect():
    # dy_dx has the reference layout [B,L,D,C].  calc_grad_inputs=1 is the true derivative (level scale included);
    # calc_grad_inputs=2 restates what the reference stores: no scale, `nd > gd` dimension pick.
    offs, emb, rng = _setup(L=3)
    x = rng.random((50, 3)).astype(np.float32) * 0.8 + 0.1
    _, j = c_oracle.hash_encode_forward(x, emb, offs, 4, calc_grad_inputs=1)
    _, jr = c_oracle.hash_encode_forward(x, emb, offs, 4, calc_grad_inputs=2)
    eps = 1e-3
    for d in range(3):
        xp, xm = x.copy(), x.copy()
        xp[:, d] += eps
        xm[:, d] -= eps
        fp = hr.hash_encode_forward(xp, emb, offs, 4).reshape(50, 3, 2)
        fm = hr.hash_encode_forward(xm, emb, offs, 4).reshape(50, 3, 2)
        for lvl in range(1):                       # level 0: cell size 1/3 >> eps, few points cross a cell
            scale = 2.0 ** lvl * 4 - 1
            fd = (fp[:, lvl] - fm[:, lvl]) / (2 * eps)
            ok = np.abs(fd - j[:, lvl, d]) < 5e-2 * scale
            assert ok.mean() > 0.95
    for lvl in range(3):
        scale = np.float32(2.0 ** lvl * 4 - 1)
        # the last dimension picks the right corners in the reference too: only the scale is missing
        np.testing.assert_allclose(jr[:, lvl, 2] * scale, j[:, lvl, 2], rtol=1e-6, atol=1e-7)
        # the other two do not (dimension gd itself is used as an interpolation dimension, one is never set)
        assert not np.allclose(jr[:, lvl, 0] * scale, j[:, lvl, 0], rtol=1e-3, atol=1e-6)
