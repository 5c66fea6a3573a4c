"""asif_hip_options.integrator = 1 for ASIFimplicitTB: the backup trajectory of the reference's USE_ODEINT build (dopri5
dense output at t = i backTrajDt, src/asif_implicit_tb.cpp:431-463) on the device vs the oracle's restatement.  As for
ASIFimplicit (tests/test_gpu_implicit_dopri.py) the two adaptive controllers see inputs that differ in the last bits, so
rows are compared at ten times the integrator tolerance, the filtered input at 1e-5 with the integrator at 1e-8; branch
codes identical except where the first sample inside the backup set is a tie at that tolerance (none on these
batches)."""
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


@pytest.mark.parametrize("cfg,B", [(12, 512), (8, 128), (4, 512)])
def test_rows_and_branches_match_oracle(hip, oracle, cfg, B):
    od, oo = _opts(hip, oracle, cfg, tol=1e-8)
    out = gpu_util.run_assemble(cfg, B, options=od)
    model, variant = oracle.CONFIGS[cfg]
    A, b, code, diag = oracle.assemble_batch(model, variant, oo, np.ascontiguousarray(out["x"].T))
    assert np.array_equal(out["code"], code), np.where(out["code"] != code)[0][:10]
    m = code == 1
    assert m.sum() >= 5
    np.testing.assert_allclose(out["A"].T[m], A[m], rtol=2e-6, atol=2e-7)
    np.testing.assert_allclose(out["b"].T[m], b[m], rtol=2e-6, atol=2e-7)
    assert np.array_equal(out["diag"][2][m], diag[m, 2])  # idxHit
    np.testing.assert_allclose(out["diag"][0][m], diag[m, 0], rtol=1e-12)  # TTS_ = idxHit * backTrajDt
    eul = gpu_util.run_assemble(cfg, B)  # and they are NOT the Euler rows
    both = m & (eul["code"] == 1)
    assert np.abs(eul["A"].T[both] - out["A"].T[both]).max() > 1e-7


@pytest.mark.parametrize("cfg,B", [(12, 2048), (4, 2048)])
def test_filter_matches_exact_optimum(hip, oracle, cfg, B):
    od, oo = _opts(hip, oracle, cfg, tol=1e-8)
    out = gpu_util.run_filter(cfg, B, options=od, uact_init=7.0, relax_init=-7.0)
    model, variant = oracle.CONFIGS[cfg]
    ua, rl, rc = oracle.filter_batch(model, variant, oo, np.ascontiguousarray(out["x"].T),
                                     np.ascontiguousarray(out["udes"].T), uact_init=np.full((B, 1), 7.0), nthreads=8)
    assert np.array_equal(out["rc"], rc), np.where(out["rc"] != rc)[0][:10]
    assert {1, 2, -3} <= set(rc.tolist())
    assert np.abs(out["uact"] - ua.T).max() <= 1e-5
