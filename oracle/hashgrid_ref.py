"""oracle/hashgrid_ref.py -- TEST INFRASTRUCTURE ONLY.

Vectorised numpy / torch restatement of the reference's multi-resolution hash-grid encoder, written
independently of oracle/hash_ref.c so the two can be checked against each other.

Follows (reference paths relative to /root/reference):
  * level table / offsets / init ........ src/encoder/hashencoder/hashgrid.py:92-113
  * grid index (dense / wrapped / hash) .. src/encoder/hashencoder/src/hashencoder.cu:36-74
  * forward trilinear gather ............. src/encoder/hashencoder/src/hashencoder.cu:99-149
  * backward scatter ..................... src/encoder/hashencoder/src/hashencoder.cu:210-271
  * range check + [-size,size]->[0,1] .... src/encoder/hashencoder/hashgrid.py:118-137

Parity status: "parity unpinned by the reference" -- it has no tests/golden vectors for the encoder and its
CUDA source does not build here.  Pinned instead by hand-derived integer KATs
(tests/test_index_kat.py) and by agreement with the scalar C restatement.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import numpy as np
import torch

PRIMES = (np.uint32(1), np.uint32(19349663), np.uint32(83492791))


def level_offsets(num_levels=16, base_resolution=16, log2_hashmap_size=19, input_dim=3):
    """hashgrid.py:92-102 -> int32[L+1] row offsets; T_l = min(2^log2T, (H*2^l + 1)^D)."""
    max_params = 2 ** log2_hashmap_size
    offs, off = [], 0
    for lvl in range(num_levels):
        res = base_resolution * 2 ** lvl
        offs.append(off)
        off += min(max_params, (res + 1) ** input_dim)
    offs.append(off)
    return np.asarray(offs, dtype=np.int32)


def level_scale_res(level, base_resolution):
    """hashencoder.cu:99-100 (fp32 scale, uint32 resolution)."""
    scale = np.float32(np.exp2(np.float32(level)) * np.float32(base_resolution) - np.float32(1.0))
    resolution = np.uint32(int(np.ceil(scale)) + 1)
    return scale, resolution


def grid_index(pos_grid, hashmap_size, resolution):
    """hashencoder.cu:55-74 on uint32 arrays.  pos_grid: uint32 [..., D] -> row index uint32 [...]."""
    pos_grid = np.asarray(pos_grid, dtype=np.uint32)
    D = pos_grid.shape[-1]
    T = np.uint32(hashmap_size)
    with np.errstate(over="ignore"):
        stride = np.uint32(1)
        index = np.zeros(pos_grid.shape[:-1], dtype=np.uint32)
        d = 0
        while d < D and stride <= T:
            index = index + pos_grid[..., d] * stride          # uint32 wrap
            stride = np.uint32((int(stride) * (int(resolution) + 1)) & 0xFFFFFFFF)
            d += 1
        if stride > T:
            index = np.zeros(pos_grid.shape[:-1], dtype=np.uint32)
            for dd in range(D):
                index = index ^ (pos_grid[..., dd] * PRIMES[dd])
    return index % T


def _fma32(a, b, c):
    """fp32 fma emulated through fp64 (products of two fp32 are exact in fp64)."""
    return (np.asarray(a, np.float64) * np.asarray(b, np.float64) + np.asarray(c, np.float64)).astype(np.float32)


def corners(x01, level, offsets, base_resolution):
    """Per-point corner rows and weights for one level.

    x01: float32 [B, D] in [0,1].  Returns (rows int64 [B, 2^D] absolute table rows, w float32 [B, 2^D])."""
    x01 = np.ascontiguousarray(x01, dtype=np.float32)
    B, D = x01.shape
    scale, res = level_scale_res(level, base_resolution)
    T = int(offsets[level + 1]) - int(offsets[level])
    pos = _fma32(x01, scale, np.float32(0.5))
    pg = np.floor(pos).astype(np.uint32)
    frac = (pos - pg.astype(np.float32)).astype(np.float32)
    rows = np.empty((B, 1 << D), dtype=np.int64)
    w = np.empty((B, 1 << D), dtype=np.float32)
    one = np.float32(1.0)
    for corner in range(1 << D):
        wc = np.ones(B, dtype=np.float32)
        pl = np.empty_like(pg)
        for d in range(D):
            if corner & (1 << d):
                wc = (wc * frac[:, d]).astype(np.float32)
                pl[:, d] = pg[:, d] + np.uint32(1)
            else:
                wc = (wc * (one - frac[:, d])).astype(np.float32)
                pl[:, d] = pg[:, d]
        rows[:, corner] = grid_index(pl, T, res).astype(np.int64) + int(offsets[level])
        w[:, corner] = wc
    return rows, w


def hash_encode_forward(x01, embeddings, offsets, base_resolution):
    """-> float32 [B, L*C] (the layout _hash_encode.forward returns after its permute, hashgrid.py:40)."""
    emb = np.asarray(embeddings, dtype=np.float32)
    L = len(offsets) - 1
    B, C = x01.shape[0], emb.shape[1]
    out = np.zeros((B, L, C), dtype=np.float32)
    for lvl in range(L):
        rows, w = corners(x01, lvl, offsets, base_resolution)
        acc = np.zeros((B, C), dtype=np.float32)
        for corner in range(rows.shape[1]):
            acc = _fma32(emb[rows[:, corner]], w[:, corner:corner + 1], acc)
        out[:, lvl] = acc
    return out.reshape(B, L * C)


def hash_encode_backward(grad, x01, offsets, base_resolution, n_rows, C):
    """grad float32 [B, L*C] -> grad_embeddings float64-accumulated, returned as float32 [n_rows, C]."""
    L = len(offsets) - 1
    B = x01.shape[0]
    g = np.asarray(grad, dtype=np.float32).reshape(B, L, C)
    acc = np.zeros((n_rows, C), dtype=np.float64)
    for lvl in range(L):
        rows, w = corners(x01, lvl, offsets, base_resolution)
        for corner in range(rows.shape[1]):
            np.add.at(acc, rows[:, corner], (w[:, corner:corner + 1] * g[:, lvl]).astype(np.float64))
    return acc.astype(np.float32)


# ----------------------------------------------------------------------------------------------
# Pure-PyTorch differentiable encoder: this is the "pure-PyTorch CPU path" timed as cpu_baseline.
# ----------------------------------------------------------------------------------------------
class HashEncoderRef(torch.nn.Module):
    """Drop-in (CPU, autograd) restatement of reference HashEncoder (hashgrid.py:77-137)."""

    def __init__(self, input_dim=3, num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=19):
        super().__init__()
        self.input_dim, self.num_levels, self.level_dim = input_dim, num_levels, level_dim
        self.base_resolution, self.log2_hashmap_size = base_resolution, log2_hashmap_size
        self.output_dim = num_levels * level_dim
        offs = level_offsets(num_levels, base_resolution, log2_hashmap_size, input_dim)
        self.offsets = torch.from_numpy(offs)
        self.embeddings = torch.nn.Parameter(torch.zeros(int(offs[-1]), level_dim))
        self.embeddings.data.uniform_(-1e-4, 1e-4)                   # hashgrid.py:111-113

    @staticmethod
    def _index(pl, T, res):
        """uint32 index arithmetic carried in int64 with explicit masking (torch lacks uint32 math)."""
        M = 0xFFFFFFFF
        D = pl.shape[-1]
        stride, d = 1, 0
        index = torch.zeros(pl.shape[:-1], dtype=torch.int64)
        while d < D and stride <= T:
            index = (index + pl[..., d] * stride) & M
            stride = (stride * (res + 1)) & M
            d += 1
        if stride > T:
            primes = (1, 19349663, 83492791)
            index = torch.zeros(pl.shape[:-1], dtype=torch.int64)
            for dd in range(D):
                index = index ^ ((pl[..., dd] * primes[dd]) & M)
        return index % T

    def forward(self, inputs, size=1):
        lo, hi = inputs.min().item(), inputs.max().item()
        if lo < -size or hi > size:                                     # hashgrid.py:122-123
            raise ValueError(f"HashGrid encoder: inputs range [{lo}, {hi}] not in [{-size}, {size}]!")
        x = (inputs + size) / (2 * size)
        prefix = list(x.shape[:-1])
        x = x.reshape(-1, self.input_dim).float()
        D = self.input_dim
        feats = []
        for lvl in range(self.num_levels):
            scale = float(2.0 ** lvl * self.base_resolution - 1.0)
            res = int(np.ceil(scale)) + 1
            T = int(self.offsets[lvl + 1] - self.offsets[lvl])
            # fp32 fma through fp64, as in the numpy oracle
            pos = (x.double() * scale + 0.5).float()
            pg = torch.floor(pos)
            frac = pos - pg
            pg = pg.long()
            acc = 0
            for corner in range(1 << D):
                w = torch.ones(x.shape[0])
                pl = []
                for d in range(D):
                    if corner & (1 << d):
                        w = w * frac[:, d]
                        pl.append(pg[:, d] + 1)
                    else:
                        w = w * (1 - frac[:, d])
                        pl.append(pg[:, d])
                rows = self._index(torch.stack(pl, -1), T, res) + int(self.offsets[lvl])
                acc = acc + w[:, None] * self.embeddings[rows]
            feats.append(acc)
        out = torch.cat(feats, -1)
        return out.reshape(prefix + [self.output_dim])
