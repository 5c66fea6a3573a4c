"""C5 (BASELINE.json configs[4]): robust inverted pendulum (affine-arithmetic rows) on the GPU.

The assembled 12 x 18 rows must equal the oracle's (whose affine forms are pinned bit-exactly against
the reference's libaffa) to 1e-12; the only transcendental is one sin().  u* <= 1e-6 against the
exact optimum; rc identical.  The golden interval Lie derivatives generated from the reference's
libaffa (tests/golden/affa_ip_robust_lie.json) are checked directly against the device rows.
"""
import json
import os

import numpy as np
import pytest

import gpu_util

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_rows_match_oracle(hip, oracle):
    B = 2048
    out = gpu_util.run_assemble(5, B)
    model, variant = oracle.CONFIGS[5]
    o = oracle.default_options(model, variant)
    A, b, code, _ = oracle.assemble_batch(model, variant, o, np.ascontiguousarray(out["x"].T))
    assert np.all(out["code"] == 1) and np.all(code == 1)
    np.testing.assert_allclose(out["A"].T, A, rtol=1e-12, atol=1e-13)
    assert np.array_equal(out["b"].T, b)


def test_rows_match_reference_libaffa_golden(hip):
    g = json.load(open(os.path.join(GOLD, "affa_ip_robust_lie.json")))
    x = np.ascontiguousarray(np.array([c["x"] for c in g["cases"]]).T)
    out = gpu_util.run_assemble(5, x.shape[1], x=x)
    nc, N = 12, 4
    A = out["A"].reshape(18, nc, -1)  # [col, row, instance]
    for s in range(N):
        col = 2 + 4 * s
        row = 3 * s
        np.testing.assert_allclose(A[1, row], [c["h"][s] for c in g["cases"]], rtol=1e-14, atol=1e-15)
        np.testing.assert_allclose(A[col + 0, row], [c["Lgh_lo"][s] for c in g["cases"]], rtol=1e-13, atol=1e-15)
        np.testing.assert_allclose(A[col + 2, row], [-c["Lgh_hi"][s] for c in g["cases"]], rtol=1e-13, atol=1e-15)
        np.testing.assert_allclose(A[col + 1, row], [c["Lfh_lo"][s] for c in g["cases"]], rtol=1e-12, atol=1e-15)
        np.testing.assert_allclose(A[col + 3, row], [-c["Lfh_hi"][s] for c in g["cases"]], rtol=1e-12, atol=1e-15)


@pytest.mark.parametrize("lanes", [2, 4, 8])
def test_filter_matches_exact_optimum(hip, oracle, lanes):
    B = 8192
    s = hip.default_solver(lanes_per_qp=lanes)
    out = gpu_util.run_filter(5, B, solver=s, uact_init=7.0, relax_init=-7.0)
    ua, rl, rc = gpu_util.oracle_filter(oracle, 5, out["x"], out["udes"], uact_init=7.0, relax_init=-7.0)
    assert np.array_equal(out["rc"], rc), f"rc mismatches {(out['rc'] != rc).sum()}"
    assert np.abs(out["uact"] - ua).max() <= 1e-6
    ok = rc == 1
    assert np.abs(out["relax"][:, ok] - rl[:, ok]).max() <= 1e-6
    assert np.all(out["uact"][0, ~ok] == 7.0)


def test_against_full_qp_admm(hip, oracle):
    """The eliminated 2-variable solve must agree with the oracle's OSQP-style ADMM run on the FULL
    18-variable QP the reference assembles (tight tolerance), not only with the reduced exact solver."""
    B = 512
    out = gpu_util.run_filter(5, B)
    s = oracle.admm_settings(eps_abs=1e-10, eps_rel=1e-10, max_iter=200000)
    ua, rl, rc = gpu_util.oracle_filter(oracle, 5, out["x"], out["udes"], solver=oracle.SOLVER_ADMM, settings=s)
    ok = (rc == 1) & (out["rc"] == 1)
    assert ok.sum() >= B - 4
    assert np.abs(out["uact"] - ua)[:, ok].max() <= 1e-5
