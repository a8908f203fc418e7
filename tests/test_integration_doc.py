"""INTEGRATION.md section 1 is executable documentation: the ctypes `backend.py` a reference maintainer would drop in is
extracted from the markdown and driven exactly like the reference's hashgrid.py drives its pybind module
(src/encoder/hashencoder/hashgrid.py:30-35, 59-64); results must equal our own drop-in module."""
import os
import re

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _backend_from_doc():
    text = open(os.path.join(REPO, "INTEGRATION.md")).read()
    block = re.search(r"```python\n(# backend\.py.*?)```", text, re.S).group(1)
    from neuralvolumetricreconstructionformedicalimages_amd import build
    os.environ["NAF_HIP_LIB"] = build.LIB_PATH
    scope = {}
    exec(compile(block, "INTEGRATION.md#backend.py", "exec"), scope)
    return scope["_backend"]


def test_documented_ctypes_backend_matches_the_drop_in_module():
    from neuralvolumetricreconstructionformedicalimages_amd.encoder import HashEncoder, hash_encode
    _backend = _backend_from_doc()
    torch.manual_seed(0)
    enc = HashEncoder(3, 16, 2, 16, 15).cuda()
    enc.embeddings.data.uniform_(-0.3, 0.3)
    B, D, C, L, H = 1000, 3, 2, 16, 16
    x = torch.rand(B, D, device="cuda")
    emb, offsets = enc.embeddings.detach(), enc.offsets.cuda()

    # forward exactly as hashgrid.py:28-38: outputs [L, B, C], then permute to [B, L*C]
    outputs = torch.zeros(L, B, C, device="cuda")
    dy_dx = torch.zeros(1, device="cuda")
    _backend.hash_encode_forward(x, emb, offsets, outputs, B, D, C, L, H, False, dy_dx)
    got = outputs.permute(1, 0, 2).reshape(B, L * C)
    want = hash_encode(x, enc.embeddings, offsets, H, False)          # our drop-in for hashgrid.py's hash_encode, same [0,1] inputs
    np.testing.assert_allclose(got.cpu().numpy(), want.detach().cpu().numpy(), rtol=0, atol=2e-7)

    # backward as hashgrid.py:52-66: grad [B, L*C] -> grad_embeddings (zeros_like, accumulated)
    grad = torch.randn(B, L * C, device="cuda")
    grad_embeddings = torch.zeros_like(emb)
    _backend.hash_encode_backward(grad, x, emb, offsets, grad_embeddings, B, D, C, L, H, False, dy_dx, torch.zeros_like(x))
    enc.zero_grad()
    want.backward(grad)
    ref = enc.embeddings.grad
    assert float((grad_embeddings - ref).abs().max()) <= 2e-5 * float(ref.abs().max())

    # a training-sized batch (above 2^13 points): the documented backend lends the library a workspace and gets the binned scatter
    B2 = 20000
    x2 = torch.rand(B2, D, device="cuda")
    grad2 = torch.randn(B2, L * C, device="cuda")
    ge2 = torch.zeros_like(emb)
    _backend.hash_encode_backward(grad2, x2, emb, offsets, ge2, B2, D, C, L, H, False, dy_dx, torch.zeros_like(x2))
    enc.zero_grad()
    hash_encode(x2, enc.embeddings, offsets, H, False).backward(grad2)       # no log2_hashmap_size: the atomic route of the module
    ref2 = enc.embeddings.grad
    assert float((ge2 - ref2).abs().max()) <= 2e-5 * float(ref2.abs().max())
