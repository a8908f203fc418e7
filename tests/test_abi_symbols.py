"""The C-ABI library loads and exports every symbol include/naf_hip.h declares (no compute calls: no GPU needed),
and the ctypes signature table covers exactly the declared entry points."""
import ctypes
import os
import re

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(REPO, "include", "naf_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(naf_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_reference_operator_pair():
    names = _declared()
    # the two entry points of the reference's pybind module (bindings.cpp:5-8)
    assert "naf_hash_encode_forward" in names and "naf_hash_encode_backward" in names
    assert len(names) >= 15


def test_library_exports_every_declared_symbol():
    from neuralvolumetricreconstructionformedicalimages_amd import _abi, build
    import shutil
    if not os.path.exists(build.LIB_PATH):
        if shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"):
            pytest.skip("libnaf_hip.so not built and no hipcc here")
        build.build_library()
    lib = ctypes.CDLL(build.LIB_PATH)
    missing = [n for n in _declared() if not hasattr(lib, n)]
    assert not missing, missing
    assert sorted(_abi.SIGNATURES) == _declared()
    assert _abi.lib().naf_abi_version() == 5
    assert _abi.lib().naf_last_error() is not None


def test_argument_validation_needs_no_gpu():
    """Null pointers / unsupported shapes are rejected before any HIP call."""
    from neuralvolumetricreconstructionformedicalimages_amd import _abi
    lib = _abi.lib()
    assert lib.naf_hash_encode_forward(None, None, None, None, 4, 3, 2, 16, 16, 0, None, 0, 0, None) == -1
    one = ctypes.c_void_p(16)
    assert lib.naf_hash_encode_forward(one, one, one, one, 4, 3, 3, 16, 16, 0, None, 0, 0, None) == -2
    assert b"C must be 1, 2, 4, or 8" in lib.naf_last_error()
    assert lib.naf_hash_encode_forward(one, one, one, one, 4, 5, 2, 16, 16, 0, None, 0, 0, None) == -2
    assert lib.naf_adam_step(one, one, one, one, None, 0, 10, 1e-3, 0.9, 0.999, 1e-8, 0, 1.0, 0, None) == -1
    assert lib.naf_hash_encode_forward(one, one, one, one, 4, 3, 2, 16, 16, 3, one, 0, 0, None) == -1      # calc_grad_inputs in {0,1,2}
    cfg = _abi.RenderCfg(n_samples=192, perturb=1, bound=0.3, L=16, C=2, H=16, table_dtype=2, mlp_precision=2,
                         last_activation=0, seed=0, ray_index_base=0, log2_hashmap_size=19)
    small = lib.naf_render_workspace_bytes(ctypes.byref(cfg), 1000 * 192)
    big = lib.naf_render_workspace_bytes(ctypes.byref(cfg), 16384 * 192)
    assert 0 < small < big
    bad = _abi.RenderCfg(n_samples=192, perturb=1, bound=0.3, L=8, C=2, H=16, table_dtype=0, mlp_precision=0,
                         last_activation=0, seed=0, ray_index_base=0, log2_hashmap_size=19)
    assert lib.naf_render_forward(one, None, one, one, one, one, 8, ctypes.byref(bad), one, None) == -2
    # the scatter mode is part of the cfg (the library keeps no process-wide mode): it sizes the workspace ...
    atomic = _abi.RenderCfg(n_samples=192, perturb=1, bound=0.3, L=16, C=2, H=16, table_dtype=2, mlp_precision=2,
                            last_activation=0, seed=0, ray_index_base=0, log2_hashmap_size=19, scatter_mode=_abi.SCATTER_ATOMIC)
    assert lib.naf_render_workspace_bytes(ctypes.byref(atomic), 16384 * 192) < big
    # ... and an unknown mode is refused
    atomic.scatter_mode = 7
    assert lib.naf_render_forward(one, None, one, one, one, one, 8, ctypes.byref(atomic), one, None) == -1
    with pytest.raises(RuntimeError, match="scatter_mode"):
        _abi.check(lib.naf_render_forward(one, None, one, one, one, one, 8, ctypes.byref(atomic), one, None), "render_forward")
    # an empty batch is a no-op whose per-batch pointers are not examined (zero-length chunks, empty data-parallel shards)
    assert lib.naf_render_forward(None, None, None, None, None, None, 0, ctypes.byref(cfg), None, None) == 0
    assert lib.naf_hash_encode_forward(None, None, None, None, 0, 3, 2, 16, 16, 0, None, 0, 0, None) == 0
    assert lib.naf_adam_step(None, None, None, None, None, 0, 0, 1e-3, 0.9, 0.999, 1e-8, 1, 1.0, 0, None) == 0
    # explicit depths (the fine pass) travel in t_rand: it cannot be absent then
    cfg.flags = _abi.CFG_EXPLICIT_DEPTHS
    assert lib.naf_render_forward(one, None, one, one, one, one, 8, ctypes.byref(cfg), one, None) == -1
    assert b"EXPLICIT_DEPTHS" in lib.naf_last_error()
    # a voxel grid of more than 2^31 points is refused before its size can wrap
    cfg.flags = 0
    dims = (ctypes.c_uint32 * 3)(0xffffffff, 0xffffffff, 0xffffffff)
    ends = (ctypes.c_double * 3)(-0.1, -0.1, -0.1), (ctypes.c_double * 3)(0.1, 0.1, 0.1)
    assert lib.naf_field_forward_grid(ends[0], ends[1], dims, one, one, one, one, ctypes.byref(cfg), one, 1 << 30, None) == -1
    assert b"2^31 points" in lib.naf_last_error()
    # ... and one whose workspace cannot hold the features of even 1024 points is refused, not overrun
    dims = (ctypes.c_uint32 * 3)(64, 64, 64)
    assert lib.naf_field_forward_grid(ends[0], ends[1], dims, one, one, one, one, ctypes.byref(cfg), one, 4096, None) == -1
    assert b"workspace too small" in lib.naf_last_error()
    # forward-only calls ask for the features alone (a 1024^3 query wanted 196 GiB of TRAINING workspace): 64 B per
    # point in bf16 mode, nothing with the fused kernel, and the front end caps what it allocates (the grid is walked in ranges)
    from neuralvolumetricreconstructionformedicalimages_amd import fused
    n = 1 << 30
    assert lib.naf_forward_workspace_bytes(ctypes.byref(cfg), n) == 64 * n + 256
    assert lib.naf_render_workspace_bytes(ctypes.byref(cfg), n) > 2.5 * lib.naf_forward_workspace_bytes(ctypes.byref(cfg), n)    # (172 B per point with the 8-byte scatter records of round 4)
    cfg.flags = _abi.CFG_FORWARD_FUSED
    assert lib.naf_forward_workspace_bytes(ctypes.byref(cfg), n) == 256
    assert fused.FORWARD_WORKSPACE_CAP <= 8 << 30


def test_product_has_no_cpu_fallback():
    """Tensors on the CPU are refused loudly; nothing under the package imports the oracle."""
    import torch
    from neuralvolumetricreconstructionformedicalimages_amd import _abi, encoder
    with pytest.raises(RuntimeError, match="no CPU path"):
        _abi.ptr(torch.zeros(3))
    enc = encoder.HashEncoder(3, 4, 2, 4, 8)
    with pytest.raises(RuntimeError):
        enc(torch.zeros(5, 3), 0.3)
    pkg = os.path.join(REPO, "neuralvolumetricreconstructionformedicalimages_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(root, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f
