"""Second time-to-backup-set model: the inverted pendulum of examples/InvertedPendulum_ImplicitTB.cpp
(half-space backup set, velocity-tracking backup controller, 11 551-sample backup trajectory -- the long-horizon
stress case) on the GPU vs the oracle.  Same bar as C4: branch and return codes identical, rows after up to 11 550
Euler steps at rtol 1e-7, u* <= 1e-6 against the exact optimum.
"""
import numpy as np
import pytest

import gpu_util

pytestmark = pytest.mark.gpu
CFG = 8


def test_rows_codes_and_diagnostics(hip, oracle):
    B = 512
    out = gpu_util.run_assemble(CFG, B)
    assert (out["dims"].nx, out["dims"].nv, out["dims"].nc, out["dims"].npBT) == (2, 2, 18, 11551)
    model, variant = oracle.CONFIGS[CFG]
    o = oracle.default_options(model, variant)
    A, b, code, diag = oracle.assemble_batch(model, variant, o, np.ascontiguousarray(out["x"].T))
    assert np.array_equal(out["code"], code)
    assert {1, 2, -3} <= set(np.unique(code))
    m = code == 1
    np.testing.assert_allclose(out["A"].T[m], A[m], rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(out["b"].T[m], b[m], rtol=1e-7, atol=1e-9)
    t = code == 2
    assert np.all(out["A"].T[t] == 0.0) and np.all(out["b"].T[t] == -1e20)
    np.testing.assert_allclose(out["diag"][0][m], diag[m, 0], rtol=1e-12)  # TTS_
    np.testing.assert_allclose(out["diag"][1][m], diag[m, 1], rtol=1e-8)   # BTorthoBS_
    assert np.array_equal(out["diag"][2][m], diag[m, 2])                    # idxHit


def test_filter_matches_exact_optimum(hip, oracle):
    B = 2048
    out = gpu_util.run_filter(CFG, B, uact_init=77.0, relax_init=-7.0)
    ua, rl, rc = gpu_util.oracle_filter(oracle, CFG, out["x"], out["udes"], uact_init=77.0, relax_init=-7.0)
    assert np.array_equal(out["rc"], rc), f"rc mismatches {(out['rc'] != rc).sum()}"
    assert (rc == 2).sum() > 30 and (rc == -3).sum() > 100 and (rc == 1).sum() > 1000
    assert np.abs(out["uact"] - ua).max() <= 1e-6
    ok = (rc == 1) | (rc == 2)
    assert np.abs(out["relax"][:, ok] - rl[:, ok]).max() <= 1e-6
    # failures fall back to the saturated backup controller u = 10 (pi/10 - omega)
    fb = rc < 0
    assert np.allclose(out["uact"][0][fb], np.clip(10.0 * (np.pi / 10.0 - out["x"][1][fb]), -1.5, 1.5), atol=1e-12)
