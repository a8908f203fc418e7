"""oracle/c_oracle.py -- TEST INFRASTRUCTURE ONLY: ctypes loader for oracle/_build/libnaf_oracle.so."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libnaf_oracle.so")
_lib = None


def build(force=False):
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, "hash_ref.c")):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
        _lib.naf_oracle_grid_index.restype = ctypes.c_uint32
    return _lib


def _p(a, t):
    return a.ctypes.data_as(ctypes.POINTER(t))


def grid_index(pos_grid, hashmap_size, resolution, C=1, ch=0):
    pg = np.ascontiguousarray(pos_grid, dtype=np.uint32)
    return int(lib().naf_oracle_grid_index(ctypes.c_uint32(pg.shape[0]), ctypes.c_uint32(C), ctypes.c_uint32(ch),
                                           ctypes.c_uint32(hashmap_size), ctypes.c_uint32(resolution),
                                           _p(pg, ctypes.c_uint32)))


def hash_encode_forward(x01, embeddings, offsets, H, calc_grad_inputs=False):
    """-> outputs [L,B,C] (level-major, as the reference kernel writes it), dy_dx [B,L,D,C] or None."""
    x = np.ascontiguousarray(x01, dtype=np.float32)
    emb = np.ascontiguousarray(embeddings, dtype=np.float32)
    offs = np.ascontiguousarray(offsets, dtype=np.int32)
    B, D = x.shape
    C, L = emb.shape[1], offs.shape[0] - 1
    out = np.zeros((L, B, C), dtype=np.float32)
    dy_dx = np.zeros((B, L, D, C), dtype=np.float32) if calc_grad_inputs else np.zeros(1, dtype=np.float32)
    u = ctypes.c_uint32
    lib().naf_oracle_hash_encode_forward(_p(x, ctypes.c_float), _p(emb, ctypes.c_float), _p(offs, ctypes.c_int32),
                                         _p(out, ctypes.c_float), u(B), u(D), u(C), u(L), u(H),
                                         ctypes.c_int(int(calc_grad_inputs)), _p(dy_dx, ctypes.c_float))
    return out, (dy_dx if calc_grad_inputs else None)


def hash_encode_backward(grad, x01, embeddings, offsets, H, dy_dx=None):
    """grad [B, L*C] -> (grad_embeddings [rows, C], grad_inputs [B,D] or None)."""
    g = np.ascontiguousarray(grad, dtype=np.float32)
    x = np.ascontiguousarray(x01, dtype=np.float32)
    emb = np.ascontiguousarray(embeddings, dtype=np.float32)
    offs = np.ascontiguousarray(offsets, dtype=np.int32)
    B, D = x.shape
    C, L = emb.shape[1], offs.shape[0] - 1
    ge = np.zeros_like(emb)
    calc = dy_dx is not None
    gi = np.zeros((B, D), dtype=np.float32) if calc else np.zeros(1, dtype=np.float32)
    jj = np.ascontiguousarray(dy_dx, dtype=np.float32) if calc else np.zeros(1, dtype=np.float32)
    u = ctypes.c_uint32
    lib().naf_oracle_hash_encode_backward(_p(g, ctypes.c_float), _p(x, ctypes.c_float), _p(emb, ctypes.c_float),
                                          _p(offs, ctypes.c_int32), _p(ge, ctypes.c_float), u(B), u(D), u(C), u(L),
                                          u(H), ctypes.c_int(int(calc)), _p(jj, ctypes.c_float), _p(gi, ctypes.c_float))
    return ge, (gi if calc else None)


def corners(x, level, H, hashmap_size, C=1):
    x = np.ascontiguousarray(x, dtype=np.float32)
    D = x.shape[0]
    idx = np.zeros(1 << D, dtype=np.uint32)
    w = np.zeros(1 << D, dtype=np.float32)
    u = ctypes.c_uint32
    lib().naf_oracle_corners(_p(x, ctypes.c_float), u(D), u(C), u(level), u(H), u(hashmap_size),
                             _p(idx, ctypes.c_uint32), _p(w, ctypes.c_float))
    return idx, w
