"""Seeded synthetic batches of the benchmark configs.

RNG: one splitmix64 step of (seed * 2**32 + k), k = i*16 + j for instance i, coordinate j;
r = (z >> 11) * 2**-53.  Seeds: C2 1, C3 2, C4 3, C5 4.  numpy only; returns SoA arrays
x [nx, B], udes [nu, B] ready to upload.
"""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def uniform(seed, i, j):
    """r in [0,1) for instances i (uint64 array) and coordinate j."""
    with np.errstate(over="ignore"):
        z = (np.uint64(seed) << np.uint64(32)) + (i.astype(np.uint64) * np.uint64(16) + np.uint64(j))
        z = z + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def make_batch(cfg, B, first=0):
    i = np.arange(first, first + B, dtype=np.uint64)
    if cfg == 2:    # DoubleIntegrator explicit: x in [-1.2,1.2]^2 (safe box is +-1), uDes in [-1.5,1.5]
        x = np.stack([-1.2 + 2.4 * uniform(1, i, 0), -1.2 + 2.4 * uniform(1, i, 1)])
        u = (-1.5 + 3.0 * uniform(1, i, 2))[None, :]
    elif cfg == 3:  # InvertedPendulum implicit
        x = np.stack([-1.5 + 3.0 * uniform(2, i, 0), -1.5 + 3.0 * uniform(2, i, 1)])
        u = (-1.5 + 3.0 * uniform(2, i, 2))[None, :]
    elif cfg == 4:  # segway TB: x_j = 0.05 * xBound_j * (2r - 1)
        xb = [3.0, 3.0, np.pi / 6, np.pi]
        x = np.stack([0.05 * xb[j] * (2.0 * uniform(3, i, j) - 1.0) for j in range(4)])
        u = (-5.0 + 10.0 * uniform(3, i, 4))[None, :]
    elif cfg == 5:  # robust pendulum
        x = np.stack([-3.0 + 6.0 * uniform(4, i, 0), -3.0 + 6.0 * uniform(4, i, 1)])
        u = (-1.5 + 3.0 * uniform(4, i, 2))[None, :]
    else:
        raise ValueError(f"unknown config {cfg}")
    return np.ascontiguousarray(x), np.ascontiguousarray(u)
