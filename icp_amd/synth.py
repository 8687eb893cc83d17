"""Synthetic workloads of BASELINE.json's configs 4 and 5, generated with a libm-free
counter RNG so that every host produces the same bits.

  u(c)      = splitmix64(c)                      (one 64-bit draw per counter value c)
  unif(c)   = (u(c) >> 11) * 2^-53               in [0, 1)
  gauss(c)  = sum_{k<12} unif(12 c + k) - 6      Irwin-Hall(12), variance 1

Config 4 (one pair, 3D, n = m): target_j[axis] = 20*unif(S + 3j + axis) - 10 with
S = 0x1CB00001; source_j = R target_j + t + 0.01*gauss(S2 + 3j + axis) with S2 = S + 2^40,
R = Rz(3deg) Ry(2deg) Rx(1deg), t = (0.05, -0.03, 0.02).  The point order is kept (ICP does not
use it).

Config 5 (batch): pair p has its own counter base S + p * 2^32: target uniform in [-10,10]^D,
source = R_p target + t_p + 0.01*gauss, rotation angle(s) in U(-5deg, 5deg) and translation in
U(-0.2, 0.2)^D drawn from the pair's own stream.
"""
from __future__ import annotations

import numpy as np

SEED = 0x1CB00001
_M = (1 << 64) - 1


def splitmix64(c: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = (c.astype(np.uint64) + np.uint64(0x9E3779B97F4A7C15))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def unif(c: np.ndarray) -> np.ndarray:
    return (splitmix64(c) >> np.uint64(11)).astype(np.float64) * (2.0 ** -53)


def gauss(c: np.ndarray) -> np.ndarray:
    c = c.astype(np.uint64)
    s = np.zeros(c.shape, np.float64)
    with np.errstate(over="ignore"):
        for k in range(12):
            s += unif(c * np.uint64(12) + np.uint64(k))
    return s - 6.0


def _rot3(rx, ry, rz):
    cx, sx, cy, sy, cz, sz = np.cos(rx), np.sin(rx), np.cos(ry), np.sin(ry), np.cos(rz), np.sin(rz)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def _cloud(base: int, n: int, dim: int) -> np.ndarray:
    with np.errstate(over="ignore"):
        c = np.uint64(base & _M) + np.arange(n * dim, dtype=np.uint64)
    return (20.0 * unif(c) - 10.0).reshape(n, dim)


def _noise(base: int, n: int, dim: int, sigma: float) -> np.ndarray:
    with np.errstate(over="ignore"):
        c = np.uint64(base & _M) + np.arange(n * dim, dtype=np.uint64)
    return (sigma * gauss(c)).reshape(n, dim)


def config4(n: int = 1_000_000, clustered: bool = False):
    """(source, target, R, t) of BASELINE config 4.  clustered=True puts the target on 64 random
    planes (non-uniform cell occupancy)."""
    target = _cloud(SEED, n, 3)
    if clustered:
        k = 64
        normals = _cloud(SEED + (1 << 41), k, 3)
        normals /= np.linalg.norm(normals, axis=1, keepdims=True)
        offs = _cloud(SEED + (1 << 42), k, 1)[:, 0] * 0.5
        which = (np.arange(n) * 2654435761 % k).astype(np.int64)
        nn_ = normals[which]
        d = np.einsum("ij,ij->i", target, nn_) - offs[which]
        target = target - d[:, None] * nn_
        target = np.ascontiguousarray(target)
    R = _rot3(np.radians(1.0), np.radians(2.0), np.radians(3.0))
    t = np.array([0.05, -0.03, 0.02])
    source = target @ R.T + t + _noise(SEED + (1 << 40), n, 3, 0.01)
    return np.ascontiguousarray(source), target, R, t


def config5_pairs(pair_ids, dim: int, n: int):
    """Concatenated clouds of the given pair ids: (sources, targets, offsets, true transforms)."""
    P = len(pair_ids)
    src = np.empty((P * n, dim))
    dst = np.empty((P * n, dim))
    Ts = np.zeros((P, dim + 1, dim + 1))
    for k, p in enumerate(pair_ids):
        base = SEED + (int(p) << 32) + (dim << 28)
        tgt = _cloud(base, n, dim)
        par = unif(np.uint64((base + (1 << 27)) & _M) + np.arange(8, dtype=np.uint64))
        ang = np.radians(10.0 * par[:3] - 5.0)
        tr = 0.4 * par[3:3 + dim] - 0.2
        if dim == 2:
            c, s = np.cos(ang[0]), np.sin(ang[0])
            R = np.array([[c, -s], [s, c]])
        else:
            R = _rot3(ang[0], ang[1], ang[2])
        s_ = tgt @ R.T + tr + _noise(base + (1 << 26), n, dim, 0.01)
        src[k * n:(k + 1) * n] = s_
        dst[k * n:(k + 1) * n] = tgt
        Ts[k, :dim, :dim] = R
        Ts[k, :dim, dim] = tr
        Ts[k, dim, dim] = 1.0
    off = np.arange(P + 1, dtype=np.int64) * n
    return src, dst, off, Ts
