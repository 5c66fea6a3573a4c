"""asif_hip_options.integrator = 1: ASIFimplicit with the backup trajectory of the reference's USE_ODEINT build
(dopri5 dense output, src/asif_implicit.cpp:427-460) on the device vs the oracle's restatement.
The device evaluates sin/cos with its own kernels and pow() with ocml, the oracle with glibc: the adaptive controller
sees inputs that differ in the last bits and now and then accepts a step the other rejects; the two trajectories are
then two valid answers of the same controlled integration, apart by its tolerance.  Rows are therefore compared at ten
times the integrator tolerance (1e-6 -> 2e-5 relative, 2e-6 absolute; most entries agree to ~1e-12), and the
filtered input -- whose sensitivity to the rows is O(10) -- at 1e-5 with the integrator at 1e-8.  rc identical."""
import numpy as np
import pytest

import gpu_util

pytestmark = pytest.mark.gpu


def _opts(hip, oracle, cfg, tol=1e-6):
    od = hip.default_options(*hip.CONFIGS[cfg][:2])
    oo = oracle.default_options(*oracle.CONFIGS[cfg])
    for o in (od, oo):
        o.integrator = 1
        o.backTrajAbsTol = o.backTrajRelTol = tol
    return od, oo


@pytest.mark.parametrize("cfg", [3, 9])
def test_rows_match_oracle(hip, oracle, cfg):
    B = 192
    od, oo = _opts(hip, oracle, cfg)
    out = gpu_util.run_assemble(cfg, B, options=od)
    model, variant = oracle.CONFIGS[cfg]
    A, b, code, _ = oracle.assemble_batch(model, variant, oo, np.ascontiguousarray(out["x"].T))
    assert np.all(out["code"] == 1)
    np.testing.assert_allclose(out["A"].T, A, rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(out["b"].T, b, rtol=2e-5, atol=2e-6)
    assert (np.abs(out["A"].T - A) <= 1e-9 * (1 + np.abs(A))).mean() > 0.9  # nearly everything to rounding
    # and they are NOT the Euler rows
    eul = gpu_util.run_assemble(cfg, B)
    assert np.abs(eul["A"] - out["A"]).max() > 1e-7


def test_filter_matches_exact_optimum(hip, oracle):
    B = 512
    od, oo = _opts(hip, oracle, 3, tol=1e-8)
    out = gpu_util.run_filter(3, B, options=od, uact_init=7.0, relax_init=-7.0)
    model, variant = oracle.CONFIGS[3]
    ua, rl, rc = oracle.filter_batch(model, variant, oo, np.ascontiguousarray(out["x"].T),
                                     np.ascontiguousarray(out["udes"].T), uact_init=np.full((B, 1), 7.0))
    assert np.array_equal(out["rc"], rc)
    assert (rc == 1).sum() > 300 and (rc == -1).sum() >= 5
    assert np.abs(out["uact"] - ua.T).max() <= 1e-5


def test_tolerance_is_honoured(hip):
    B = 128
    rows = {}
    for tol in (1e-5, 1e-8, 1e-11):
        od = hip.default_options(*hip.CONFIGS[3][:2])
        od.integrator = 1
        od.backTrajAbsTol = od.backTrajRelTol = tol
        rows[tol] = gpu_util.run_assemble(3, B, options=od)
    d1 = np.abs(rows[1e-5]["A"] - rows[1e-11]["A"]).max()
    d2 = np.abs(rows[1e-8]["A"] - rows[1e-11]["A"]).max()
    assert d2 < 1e-3 * d1 and d2 <= 1e-6 and d1 <= 5e-2, (d1, d2)


def test_classes_without_it_refuse(hip):
    o = hip.default_options(hip.MODEL_INVERTED_PENDULUM, hip.IMPLICIT_RB)
    o.integrator = 1
    with pytest.raises(hip.AsifHipError):
        hip.Filter(hip.MODEL_INVERTED_PENDULUM, hip.IMPLICIT_RB, options=o)
    # (ASIFimplicitTB has it since round 3: tests/test_gpu_tb_dopri.py; ASIFimplicitRB's held input makes the rhs
    # time-dependent and stays on forward Euler)
    o = hip.default_options(hip.MODEL_DOUBLE_INTEGRATOR, hip.EXPLICIT)
    o.integrator = 1
    with pytest.raises(hip.AsifHipError):
        hip.Filter(hip.MODEL_DOUBLE_INTEGRATOR, hip.EXPLICIT, options=o)


def test_failed_integration_fails_the_filter(hip, oracle):
    """NaN / overflowing states under the adaptive integrator: the lane stops stepping (no hang), its rows are
    non-finite and the filter returns what the oracle returns (rc -1); the neighbouring lanes are unaffected."""
    B = 128
    od, oo = _opts(hip, oracle, 3)
    x, udes = gpu_util.workloads.make_batch(3, B)
    clean = gpu_util.run_filter(3, B, options=od, x=x.copy(), udes=udes)
    x[0, 5], x[1, 17], x[:, 40] = np.nan, np.nan, 1.7e308
    out = gpu_util.run_filter(3, B, options=od, x=x, udes=udes, uact_init=7.0)
    model, variant = oracle.CONFIGS[3]
    ua, rl, rc = oracle.filter_batch(model, variant, oo, np.ascontiguousarray(x.T), np.ascontiguousarray(udes.T),
                                     uact_init=np.full((B, 1), 7.0))
    assert np.array_equal(out["rc"], rc) and list(rc[[5, 17, 40]]) == [-1, -1, -1]
    keep = np.ones(B, bool); keep[[5, 17, 40]] = False
    assert np.array_equal(out["uact"][:, keep], clean["uact"][:, keep])
