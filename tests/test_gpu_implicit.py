"""C3 (BASELINE.json configs[2]): InvertedPendulum implicit filter on the GPU vs the oracle.

Rows: the device integrates 5000 Euler steps with FMA contraction and ocml sin/cos, the oracle
without contraction and with glibc; rows are compared to rtol 1e-9 / atol 1e-11 (observed ~1e-13).
u*: <= 1e-6 against the exact optimum; rc identical.
"""
import numpy as np
import pytest

import gpu_util

pytestmark = pytest.mark.gpu


def _oracle_rows(oracle, x):
    model, variant = oracle.CONFIGS[3]
    o = oracle.default_options(model, variant)
    return oracle.assemble_batch(model, variant, o, np.ascontiguousarray(x.T))


def test_rows_and_critical_samples(hip, oracle):
    B = 192
    out = gpu_util.run_assemble(3, B)
    A, b, code, _ = _oracle_rows(oracle, out["x"])
    assert np.all(out["code"] == 1) and np.all(code == 1)
    np.testing.assert_allclose(out["A"].T, A, rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(out["b"].T, b, rtol=1e-9, atol=1e-11)
    # critical sample indexes are reported in diag[0:10]; spot-check the SURVEY 8(c) point x=(0.1,0)
    x = np.array([[0.1], [0.0]])
    one = gpu_util.run_assemble(3, 1, x=x)
    crit = one["diag"][:10, 0].astype(int)
    assert list(crit) == list(range(10))  # ties -> lowest index first (reference: 1 0 2 3 ...)
    A1 = one["A"][:, 0].reshape(3, 41).T
    # rows of trajectory sample 1 (rows 4..7 here, rows 0..3 in the reference's tie order) and the backup row
    np.testing.assert_allclose(A1[4], [-0.001, 3.041592653589793, 0.0], rtol=1e-12)
    np.testing.assert_allclose(one["b"][4, 0], 9.9833416646828154e-05, rtol=1e-12)
    np.testing.assert_allclose(A1[6], [0.997, 3.1413924870064398, 0.0], rtol=1e-12)
    np.testing.assert_allclose(one["b"][6, 0], -0.099533916396887676, rtol=1e-12)
    np.testing.assert_allclose(A1[40], [-1.7835876603724658e-05, 0.0, 0.049998208951217787], rtol=1e-9)
    np.testing.assert_allclose(one["b"][40, 0], 1.780616500241058e-06, rtol=1e-9)


@pytest.mark.parametrize("lanes", [4, 8, 16])
def test_filter_matches_exact_optimum(hip, oracle, lanes):
    B = 256
    s = hip.default_solver(lanes_per_qp=lanes)
    out = gpu_util.run_filter(3, B, solver=s, uact_init=7.0, relax_init=-7.0)
    ua, rl, rc = gpu_util.oracle_filter(oracle, 3, out["x"], out["udes"], uact_init=7.0, relax_init=-7.0)
    assert np.array_equal(out["rc"], rc), f"rc mismatches {(out['rc'] != rc).sum()}"
    assert (rc == -1).sum() >= 5 and (rc == 1).sum() >= 200
    assert np.abs(out["uact"] - ua).max() <= 1e-6
    ok = rc == 1
    assert np.abs(out["relax"][:, ok] - rl[:, ok]).max() <= 1e-5
    assert np.all(out["relax"][:, ~ok] == -7.0)  # relax untouched on failure, src/asif_implicit.cpp:348-355


def test_full_size_properties(hip):
    """B = 16 384: returned (u, relax) satisfy the assembled rows and bounds; fallback inputs are
    the saturated backup controller; halves == whole."""
    B = 16384
    out = gpu_util.run_filter(3, B)
    rows = gpu_util.run_assemble(3, B)
    rc = out["rc"]
    assert set(np.unique(rc)) <= {1, -1}
    ok = rc == 1
    u = out["uact"][0]
    A = rows["A"].reshape(3, 41, B)
    lhs = A[0] * u + A[1] * out["relax"][0] + A[2] * out["relax"][1]
    assert (rows["b"] - lhs)[:, ok].max() <= 1e-6
    assert np.all(np.abs(u) <= 1.5)
    assert np.all(out["relax"][0, ok] >= 10.0 - 1e-9) and np.all(out["relax"][1, ok] >= 5.0 - 1e-9)
    x = out["x"]
    ub = np.clip(-3.0 * x[0] + -3.0 * x[1], -1.5, 1.5)
    assert np.allclose(u[~ok], ub[~ok], rtol=0, atol=1e-15)
    h1 = gpu_util.run_filter(3, B // 2, first=0)
    h2 = gpu_util.run_filter(3, B // 2, first=B // 2)
    assert np.array_equal(np.concatenate([h1["uact"], h2["uact"]], axis=1), out["uact"])
    assert np.array_equal(np.concatenate([h1["rc"], h2["rc"]]), rc)


def test_fused_solve_and_its_hand_over_to_stage_two(hip, monkeypatch):
    """InvertedPendulum_Implicit, default solver mode: the rows kernel solves each instance's 3 x 41 QP itself and stages
    nothing.  ASIF_HIP_IM_FUSE forces the hand-over to stage 2 that no seeded instance takes by itself (2: every
    instance, 3: every second one), 0 is the two-launch path: uAct, relax, rc, diagnostics bitwise identical across all."""
    import gpu_util
    outs = []
    for v in (None, "0", "2", "3"):
        if v is None:
            monkeypatch.delenv("ASIF_HIP_IM_FUSE", raising=False)
        else:
            monkeypatch.setenv("ASIF_HIP_IM_FUSE", v)
        outs.append(gpu_util.run_filter(3, 1027, uact_init=7.0, relax_init=-7.0))
    ref = outs[0]
    assert {1, -1} <= set(np.unique(ref["rc"]).tolist())
    for o in outs[1:]:
        assert np.array_equal(o["rc"], ref["rc"]) and np.array_equal(o["uact"], ref["uact"])
        assert np.array_equal(o["relax"], ref["relax"]) and np.array_equal(o["diag"], ref["diag"])
