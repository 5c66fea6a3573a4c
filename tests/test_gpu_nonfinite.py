"""Non-finite states (NaN, +-inf, overflow) through every filter class: the return code equals the oracle's on every
poisoned instance, nothing hangs, and the finite instances of the same launch are bit-for-bit what they are without
the poison.  ADVICE r2: the one-variable solve of the default explicit path read NaN rows as met and returned rc 1 with
uAct = clip(uDes); the reference (OSQP on NaN data) runs to max_iter -> filter() returns -1, uAct untouched
(src/asif.cpp:199-209, src/qpwrapper_osqp.cpp:225-238)."""
import numpy as np
import pytest

import gpu_util

pytestmark = pytest.mark.gpu


def _poison(x):
    B = x.shape[1]
    idx = []
    k = 3
    for v in (np.nan, np.inf, -np.inf, 1.7e308):
        for j in range(x.shape[0]):
            x[j, k] = v
            idx.append(k)
            k += 11
    x[:, k] = np.nan
    idx.append(k)
    assert k < B
    return np.array(idx)


@pytest.mark.parametrize("cfg", [2, 3, 4, 5, 9, 11, 12])
@pytest.mark.parametrize("polish", [2, 1])
def test_rc_matches_oracle_and_neighbours_untouched(hip, oracle, cfg, polish):
    B = 512 if cfg in (3,) else 1024
    x, udes = gpu_util.workloads.make_batch(cfg, B)
    s = hip.default_solver(polish=polish)
    clean = gpu_util.run_filter(cfg, B, solver=s, x=x.copy(), udes=udes, uact_init=7.0, relax_init=-7.0)
    bad = _poison(x)
    out = gpu_util.run_filter(cfg, B, solver=s, x=x, udes=udes, uact_init=7.0, relax_init=-7.0)
    ua, rl, rc = gpu_util.oracle_filter(oracle, cfg, x, udes, uact_init=7.0, relax_init=-7.0)
    assert np.array_equal(out["rc"], rc), (out["rc"][bad], rc[bad])
    assert np.all(rc[bad] != 1) and np.all(rc[bad] != 2)
    if cfg in (2, 5, 11):  # classes that leave uAct / relax untouched on failure
        assert np.all(out["uact"][:, bad] == 7.0) and np.all(out["relax"][:, bad] == -7.0)
    keep = np.ones(B, bool)
    keep[bad] = False
    assert np.array_equal(out["rc"][keep], clean["rc"][keep])
    for k in ("uact", "relax"):
        if cfg in (3, 4, 9, 12):
            # trajectory kernels: a wave that holds a NaN lane re-runs its blocks on the checking step (trig evaluated
            # per step instead of carried along the block): its other lanes move by rounding, not more
            assert np.abs(out[k][..., keep] - clean[k][..., keep]).max() <= 1e-10, k
        else:
            assert np.array_equal(out[k][..., keep], clean[k][..., keep]), k


def test_udes_nan_fails_too(hip, oracle):
    B = 256
    x, udes = gpu_util.workloads.make_batch(2, B)
    udes[0, 9] = np.nan
    out = gpu_util.run_filter(2, B, x=x, udes=udes, uact_init=7.0)
    ua, rl, rc = gpu_util.oracle_filter(oracle, 2, x, udes, uact_init=7.0)
    assert np.array_equal(out["rc"], rc) and rc[9] == -1 and out["uact"][0, 9] == 7.0


def test_rollout_counts_the_failure(hip):
    import torch
    B, T = 256, 5
    x, udes = gpu_util.workloads.make_batch(2, B)
    x *= 0.5
    x[0, 7] = np.nan
    dev = torch.device("cuda:0")
    flt = hip.Filter(*hip.CONFIGS[2][:2])
    tx, tu = torch.from_numpy(x).to(dev), torch.from_numpy(udes).to(dev)
    uact = torch.full((1, B), 7.0, dtype=torch.float64, device=dev)
    relax = torch.zeros((1, B), dtype=torch.float64, device=dev)
    nfail = torch.zeros(B, dtype=torch.int32, device=dev)
    flt.rollout(T, 0.01, tx, tu, uact, relax, nfail)
    torch.cuda.synchronize()
    assert int(nfail[7]) == T and float(uact[0, 7]) == 7.0
    flt.close()
