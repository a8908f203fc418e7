"""Integer known-answer tests for the grid index (hashencoder.cu:36-74), hand-derived per regime
(SURVEY.md App. A-1), checked on BOTH oracle restatements.  Bit-exact."""
import numpy as np
import pytest

from oracle import c_oracle, hashgrid_ref as hr

M32 = 0xFFFFFFFF
P1, P2 = 19349663, 83492791


def test_offsets_table_T19():
    offs = hr.level_offsets(16, 16, 19, 3)
    assert offs.dtype == np.int32
    assert list(offs[:5]) == [0, 4913, 40850, 315475, 839763]
    assert offs[12] == 5034067 and offs[15] == 6606931 and offs[16] == 7131219


def test_offsets_table_T22():
    offs = hr.level_offsets(16, 16, 22, 3)
    assert offs[3] == 315475 and offs[4] == 315475 + 129 ** 3 == 2462164
    assert offs[16] == 52793812


def _expected(level, T, xyz):
    x, y, z = xyz
    res = 16 * 2 ** level
    stride, idx, d = 1, 0, 0
    for d in range(3):
        if stride > T:
            break
        idx = (idx + xyz[d] * stride) & M32
        stride = (stride * (res + 1)) & M32
    if stride > T:
        idx = (x ^ ((y * P1) & M32) ^ ((z * P2) & M32)) & M32
    return idx % T


CASES = [
    # (level, T_l, xyz, regime, closed form)
    (0, 4913, (3, 5, 7), "dense", 3 + 17 * 5 + 289 * 7),
    (0, 4913, (16, 16, 16), "dense", 4912),
    (1, 35937, (32, 1, 2), "dense", 32 + 33 + 2 * 1089),
    (2, 274625, (64, 64, 64), "dense", 274624),
    (3, 524288, (3, 5, 7), "hash", (3 ^ (5 * P1 & M32) ^ (7 * P2 & M32)) % 524288),
    (7, 524288, (2047, 1024, 1), "hash", (2047 ^ (1024 * P1 & M32) ^ (P2 & M32)) % 524288),
    (12, 524288, (3, 5, 7), "wrapped-dense", ((3 + 65537 * 5 + 131073 * 7) & M32) % 524288),
    (12, 524288, (65536, 65536, 65536), "wrapped-dense", ((65536 * (1 + 65537 + 131073)) & M32) % 524288),
    (13, 524288, (100000, 3, 9), "wrapped-dense", ((100000 + 131073 * 3 + 262145 * 9) & M32) % 524288),
    (14, 524288, (3, 5, 7), "hash (wrapped stride 524289 > T by one)", (3 ^ (5 * P1 & M32) ^ (7 * P2 & M32)) % 524288),
    (15, 524288, (524288, 524288, 1), "hash", (524288 ^ (524288 * P1 & M32) ^ P2) % 524288),
]


@pytest.mark.parametrize("level,T,xyz,regime,closed", CASES)
def test_grid_index_kat(level, T, xyz, regime, closed):
    res = 16 * 2 ** level
    assert _expected(level, T, xyz) == closed, regime
    assert c_oracle.grid_index(xyz, T, res) == closed, regime
    assert int(hr.grid_index(np.array([xyz], dtype=np.uint32), T, res)[0]) == closed, regime


def test_T22_regimes():
    # level 3 becomes dense, levels 14/15 become wrapped-dense at T=2^22 (App. A-1)
    assert c_oracle.grid_index((3, 5, 7), 129 ** 3, 128) == 3 + 129 * 5 + 129 * 129 * 7
    T = 1 << 22
    assert c_oracle.grid_index((3, 5, 7), T, 262144) == ((3 + 262145 * 5 + 524289 * 7) & M32) % T
    assert c_oracle.grid_index((3, 5, 7), T, 524288) == ((3 + 524289 * 5 + 1048577 * 7) & M32) % T
    assert c_oracle.grid_index((3, 5, 7), T, 1024) == (3 ^ (5 * P1 & M32) ^ (7 * P2 & M32)) % T


def test_two_restatements_agree_on_random_indices():
    rng = np.random.default_rng(0)
    for log2T in (10, 19, 22):
        offs = hr.level_offsets(16, 16, log2T, 3)
        for level in range(16):
            T = int(offs[level + 1] - offs[level])
            res = 16 * 2 ** level
            pg = rng.integers(0, res + 1, size=(64, 3), dtype=np.uint32)
            a = hr.grid_index(pg, T, res)
            b = np.array([c_oracle.grid_index(p, T, res) for p in pg], dtype=np.uint32)
            assert np.array_equal(a, b), (log2T, level)
            assert (a < T).all()


def test_2d_index():
    # D=2: level 0 dense x + 17 y
    assert c_oracle.grid_index((3, 5), 289, 16) == 3 + 17 * 5
    assert int(hr.grid_index(np.array([[3, 5]], dtype=np.uint32), 289, 16)[0]) == 88
