"""Host-side check of the lemma the two-pass critical-sample search of the trajectory kernels rests on
(asif_amd/csrc/k_implicit.hip, k_tb.hip): with samples ordered by (value, index) and blocks of M
consecutive samples ordered by (block minimum, block index), every one of the K first samples lies in one of the
K first blocks -- also with ties, plateaus, ragged last blocks and fewer than K blocks."""
import numpy as np


def _topk_samples(h, K):
    order = sorted(range(len(h)), key=lambda s: (h[s], s))
    return order[:K]


def _two_pass(h, K, M):
    n = len(h)
    nblk = (n + M - 1) // M
    bmin = [min(h[b * M:(b + 1) * M]) for b in range(nblk)]
    blocks = sorted(range(nblk), key=lambda b: (bmin[b], b))[:K]
    cand = [s for b in sorted(blocks) for s in range(b * M, min((b + 1) * M, n))]
    # streaming selection with strict '<' over samples arriving in increasing index = (value, index) order
    return sorted(cand, key=lambda s: (h[s], s))[:K]


def test_k_smallest_samples_lie_in_k_smallest_blocks():
    rng = np.random.default_rng(7)
    for trial in range(400):
        n = int(rng.integers(1, 400))
        K = int(rng.choice([1, 4, 10]))
        M = int(rng.choice([1, 4, 8, 16, 32]))
        kind = trial % 4
        if kind == 0:
            h = rng.normal(size=n)
        elif kind == 1:   # few distinct values: ties everywhere
            h = rng.integers(0, 3, size=n).astype(float)
        elif kind == 2:   # transient then a plateau of identical values (a trajectory that converged)
            h = np.concatenate([np.sort(rng.normal(size=n // 3))[::-1], np.full(n - n // 3, -5.0)])
        else:             # V shape: falling margin, then rising
            t = np.arange(n) - n * rng.random()
            h = np.abs(t) + 0.0
        h = list(h)
        want = _topk_samples(h, K)
        got = _two_pass(h, K, M)
        assert got == want, (trial, n, K, M)
