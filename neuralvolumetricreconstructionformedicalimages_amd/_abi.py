"""ctypes binding of libnaf_hip.so (C ABI: include/naf_hip.h).

Replaces the pybind11 module `_hash_encoder` of the reference (src/encoder/hashencoder/src/bindings.cpp:5-8,
loaded by backend.py:6-16).  There is deliberately NO fallback: if the library is missing or a call fails the
caller gets a RuntimeError -- the product path never routes through a CPU implementation.
"""
from __future__ import annotations

import ctypes
import os

import torch

from . import build as _build

F32, F16, BF16 = 0, 1, 2
LAYOUT_LBC, LAYOUT_BLC = 0, 1
MLP_PARAMS = 4225
SCATTER_AUTO, SCATTER_ATOMIC, SCATTER_BINNED = 0, 1, 2
CFG_PER_LEVEL_LAUNCHES = 1
CFG_EXPLICIT_DEPTHS = 2
CFG_LEVELS_INTERLEAVED = 4
CFG_FORWARD_FUSED = 8
CFG_FUSED_STORE_FEATURES = 16
CFG_ENCODE_TWO_GATHERS = 32
CFG_ENCODE_WINDOWS = 64
CFG_BACKWARD_ONE_WAVE_PER_SIMD = 128
CFG_ENCODE_LEVEL_MAJOR = 256
CFG_ENCODE_GROUPS_2 = 512
CFG_ENCODE_GROUPS_4 = 1024
CFG_SCATTER_PAIR12 = 2048
CFG_TEST_TINY_BLOCKS = 4096
CFG_LEVELS_GATHER_PASS = 32768
CFG_MIN_BUCKETS_SHIFT = 13          # bits 13-14: at least 64 << value row buckets per level in the binned scatter
GRAD_INPUTS_NONE, GRAD_INPUTS_EXACT, GRAD_INPUTS_REFERENCE = 0, 1, 2

_DTYPE_CODE = {torch.float32: F32, torch.float16: F16, torch.bfloat16: BF16}
_lib = None


class RenderCfg(ctypes.Structure):
    """struct naf_render_cfg (include/naf_hip.h)."""
    _fields_ = [
        ("n_samples", ctypes.c_uint32), ("perturb", ctypes.c_int32), ("bound", ctypes.c_float),
        ("L", ctypes.c_uint32), ("C", ctypes.c_uint32), ("H", ctypes.c_uint32),
        ("table_dtype", ctypes.c_int32), ("mlp_precision", ctypes.c_int32), ("last_activation", ctypes.c_int32),
        ("seed", ctypes.c_uint64), ("ray_index_base", ctypes.c_uint32), ("log2_hashmap_size", ctypes.c_uint32),
        ("scatter_mode", ctypes.c_int32), ("flags", ctypes.c_uint32),
    ]


MAX_GRAD_BUCKETS = 16


class GradBuckets(ctypes.Structure):
    """struct naf_grad_buckets (include/naf_hip.h): level ranges + the hipEvent_t recorded when each is final."""
    _fields_ = [
        ("n_buckets", ctypes.c_uint32), ("level_begin", ctypes.c_uint32 * MAX_GRAD_BUCKETS),
        ("level_end", ctypes.c_uint32 * MAX_GRAD_BUCKETS), ("ready", ctypes.c_void_p * MAX_GRAD_BUCKETS),
        ("mlp_ready", ctypes.c_void_p),
    ]


MAX_DRAW_SEGMENTS = 16


class TableAdam(ctypes.Structure):
    """struct naf_table_adam (include/naf_hip.h): the table's Adam state for naf_render_train_adam."""
    _fields_ = [
        ("param", ctypes.c_void_p), ("exp_avg", ctypes.c_void_p), ("exp_avg_sq", ctypes.c_void_p), ("param_lp", ctypes.c_void_p),
        ("lp_dtype", ctypes.c_int32), ("n", ctypes.c_uint64), ("lr", ctypes.c_float), ("beta1", ctypes.c_float),
        ("beta2", ctypes.c_float), ("eps", ctypes.c_float), ("step", ctypes.c_uint32), ("grad_scale", ctypes.c_float),
        ("mlp_param", ctypes.c_void_p), ("mlp_exp_avg", ctypes.c_void_p), ("mlp_exp_avg_sq", ctypes.c_void_p),
    ]


class ScanDraw(ctypes.Structure):
    """struct naf_scan_draw (include/naf_hip.h): valid-pixel lists of the projections a step draws from."""
    _fields_ = [
        ("n_segments", ctypes.c_uint32), ("rays_per_segment", ctypes.c_uint32),
        ("valid", ctypes.c_void_p * MAX_DRAW_SEGMENTS), ("n_valid", ctypes.c_uint32 * MAX_DRAW_SEGMENTS),
    ]


class NextDraw(ctypes.Structure):
    """struct naf_next_draw (include/naf_hip.h): the arguments of naf_draw_scan_rays for the NEXT step's pixels."""
    _fields_ = [
        ("draw", ScanDraw), ("poses", ctypes.c_void_p), ("projections", ctypes.c_void_p), ("pixels", ctypes.c_void_p),
        ("target", ctypes.c_void_p), ("rays", ctypes.c_void_p),
        ("first_draw", ctypes.c_uint32), ("n_draws", ctypes.c_uint32), ("n_projections", ctypes.c_uint32), ("det_w", ctypes.c_uint32),
        ("det_h", ctypes.c_uint32),
        ("du", ctypes.c_float), ("dv", ctypes.c_float), ("ou", ctypes.c_float), ("ov", ctypes.c_float), ("DSD", ctypes.c_float),
        ("near", ctypes.c_float), ("far", ctypes.c_float), ("parallel", ctypes.c_int32), ("seed", ctypes.c_uint64),
    ]


# name -> (restype, argtypes); mirrors include/naf_hip.h one to one (checked by tests/test_abi_symbols.py)
_vp, _u32, _u64, _i32, _f32 = ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_int, ctypes.c_float
SIGNATURES = {
    "naf_last_error": (ctypes.c_char_p, []),
    "naf_abi_version": (_i32, []),
    "naf_profile_enable": (_i32, [_i32]),
    "naf_profile_collect": (_i32, [ctypes.c_char_p, ctypes.c_size_t]),
    "naf_hash_encode_forward": (_i32, [_vp, _vp, _vp, _vp, _u32, _u32, _u32, _u32, _u32, _i32, _vp, _i32, _i32, _vp]),
    "naf_hash_encode_backward": (_i32, [_vp, _vp, _vp, _vp, _vp, _u32, _u32, _u32, _u32, _u32, _i32, _vp, _vp, _i32, _i32, _vp]),
    "naf_hash_encode_workspace_bytes": (ctypes.c_size_t, [_u32, _u32, _u32, _u32, _u32, _i32]),
    "naf_hash_encode_backward_ws": (_i32, [_vp, _vp, _vp, _vp, _vp, _u32, _u32, _u32, _u32, _u32, _i32, _vp, _vp, _i32, _i32, _u32, _vp,
                                           ctypes.c_size_t, _vp]),
    "naf_sample_rays": (_i32, [_vp, _vp, _vp, _vp, _u32, _u32, _i32, _f32, _u64, _u32, _vp]),
    "naf_fine_depths": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _u32, _u32, _u32, _i32, _i32, _u64, _u32, _vp, _vp]),
    "naf_draw_scan_rays": (_i32, [ctypes.POINTER(ScanDraw), _vp, _vp, _vp, _vp, _vp, _u32, _u32, _u32, _u32, _u32, _f32, _f32, _f32, _f32,
                                  _f32, _f32, _f32, _i32, _u64, _vp]),
    "naf_generate_rays": (_i32, [_vp, _vp, ctypes.c_int64, _vp, _u64, _u32, _u32, _u32, _f32, _f32, _f32, _f32, _f32, _f32, _f32, _i32, _vp]),
    "naf_integrate_forward": (_i32, [_vp, _vp, _vp, _vp, _u32, _u32, _vp]),
    "naf_integrate_backward": (_i32, [_vp, _vp, _vp, _vp, _u32, _u32, _vp]),
    "naf_scatter_overflow_count": (_i32, [ctypes.POINTER(RenderCfg), _u64, _vp, ctypes.POINTER(ctypes.c_uint32)]),
    "naf_scatter_overflow_levels": (_i32, [ctypes.POINTER(RenderCfg), _u64, _vp, ctypes.POINTER(ctypes.c_uint32 * 32)]),
    "naf_render_workspace_bytes": (ctypes.c_size_t, [ctypes.POINTER(RenderCfg), _u64]),
    "naf_forward_workspace_bytes": (ctypes.c_size_t, [ctypes.POINTER(RenderCfg), _u64]),
    "naf_render_forward": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _u32, ctypes.POINTER(RenderCfg), _vp, _vp]),
    "naf_render_forward_samples": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _u32, ctypes.POINTER(RenderCfg), _vp, _vp]),
    "naf_render_backward": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _u32, ctypes.POINTER(RenderCfg), _vp, _i32, _vp]),
    "naf_render_train": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _u32, ctypes.POINTER(RenderCfg), _vp, _vp]),
    "naf_render_train_bucketed": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _u32, ctypes.POINTER(RenderCfg), _vp,
                                         ctypes.POINTER(GradBuckets), _vp]),
    "naf_render_train_adam": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _u32, ctypes.POINTER(RenderCfg), _vp,
                                     ctypes.POINTER(TableAdam), _vp]),
    "naf_render_train_adam_draw": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _u32, ctypes.POINTER(RenderCfg), _vp,
                                          ctypes.POINTER(TableAdam), ctypes.POINTER(NextDraw), _vp]),
    "naf_levels_encode": (_i32, [_vp, _vp, _vp, _vp, _vp, _u32, _u32, ctypes.POINTER(RenderCfg), _u32, _u32, _vp]),
    "naf_levels_field_step": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _u32, ctypes.POINTER(RenderCfg), _vp, _vp, _vp]),
    "naf_levels_scatter": (_i32, [_vp, _vp, _vp, ctypes.c_size_t, _u32, _vp, _vp, _u32, ctypes.POINTER(RenderCfg), _u32, _u32, _vp,
                                  ctypes.POINTER(TableAdam), ctypes.POINTER(ctypes.c_int), _vp]),
    "naf_field_forward": (_i32, [_vp, _vp, _vp, _vp, _vp, _u32, ctypes.POINTER(RenderCfg), _vp, _vp]),
    "naf_field_forward_grid": (_i32, [ctypes.POINTER(ctypes.c_double * 3), ctypes.POINTER(ctypes.c_double * 3),
                                      ctypes.POINTER(ctypes.c_uint32 * 3), _vp, _vp, _vp, _vp, ctypes.POINTER(RenderCfg), _vp,
                                      ctypes.c_size_t, _vp]),
    "naf_adam_step": (_i32, [_vp, _vp, _vp, _vp, _vp, _i32, _u64, _f32, _f32, _f32, _f32, _u32, _f32, _i32, _vp]),
    "naf_normalize_inputs": (_i32, [_vp, _u64, _f32, _vp, _vp, _vp]),
}


def library_path():
    return _build.LIB_PATH


def lib():
    """Load (once) the prebuilt in-tree library; raise loudly if it is absent or incomplete."""
    global _lib
    if _lib is None:
        path = library_path()
        if not os.path.exists(path):
            # a fresh checkout has sources only (*.so is git-ignored): compile them now if the ROCm toolchain is here;
            # there is no other way to run -- no CPU fallback exists for the NAF hot path
            try:
                _build.build_library()
            except Exception as e:
                raise RuntimeError(
                    f"{path} not found and could not be built ({e}).  Build it with "
                    "`python -c 'import __graft_entry__ as g; g.build()'` (hipcc --offload-arch=gfx950).  "
                    "There is no CPU fallback for the NAF hot path.") from e
        handle = ctypes.CDLL(path)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)          # AttributeError if the .so lacks a declared symbol
            fn.restype, fn.argtypes = res, args
        _lib = handle
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = lib().naf_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"libnaf_hip {what} failed ({rc}): {msg}")


def dtype_code(dt):
    try:
        return _DTYPE_CODE[dt]
    except KeyError:
        raise RuntimeError(f"libnaf_hip: unsupported dtype {dt}") from None


def ptr(t):
    """Device pointer of a tensor that must already live on the GPU and be contiguous (hashencoder.cu:17-18)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("libnaf_hip: tensor must be a CUDA/HIP tensor (no CPU path)")
    if not t.is_contiguous():
        raise RuntimeError("libnaf_hip: tensor must be contiguous")
    return ctypes.c_void_p(t.data_ptr())


def stream_ptr():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def profile_enable(on=True):
    check(lib().naf_profile_enable(int(on)), "profile_enable")


def profile_collect():
    """-> {kernel: (launches, total_ms)} for everything launched since profile_enable(True)."""
    buf = ctypes.create_string_buffer(1 << 16)
    check(lib().naf_profile_collect(buf, len(buf)), "profile_collect")
    out = {}
    for line in buf.value.decode().splitlines():
        name, count, ms = line.rsplit(" ", 2)
        out[name] = (int(count), float(ms))
    return out
