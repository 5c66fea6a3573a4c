"""Second backup-trajectory model: the double integrator of examples/DoubleIntegrator_implicit.cpp
(npBTSS = 4 critical samples -> 17 x 3 QP, 201-sample trajectory, the shipped mPpPt with its +1 entry)
on the GPU vs the oracle.  Same bar as C3: rows at rtol 1e-9, u* <= 1e-6 against the exact optimum, rc identical.
"""
import numpy as np
import pytest

import gpu_util

pytestmark = pytest.mark.gpu
CFG = 9


def test_rows_match_oracle(hip, oracle):
    B = 2048
    out = gpu_util.run_assemble(CFG, B)
    assert (out["dims"].nv, out["dims"].nc, out["dims"].npBT, out["dims"].npBTSS) == (3, 17, 201, 4)
    model, variant = oracle.CONFIGS[CFG]
    o = oracle.default_options(model, variant)
    A, b, code, _ = oracle.assemble_batch(model, variant, o, np.ascontiguousarray(out["x"].T))
    assert np.all(out["code"] == 1) and np.all(code == 1)
    np.testing.assert_allclose(out["A"].T, A, rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(out["b"].T, b, rtol=1e-9, atol=1e-11)


@pytest.mark.parametrize("lanes", [0, 2, 8])
def test_filter_matches_exact_optimum(hip, oracle, lanes):
    B = 8192
    s = hip.default_solver(lanes_per_qp=lanes)
    out = gpu_util.run_filter(CFG, B, solver=s, uact_init=7.0, relax_init=-7.0)
    ua, rl, rc = gpu_util.oracle_filter(oracle, CFG, out["x"], out["udes"], uact_init=7.0, relax_init=-7.0)
    assert np.array_equal(out["rc"], rc), f"rc mismatches {(out['rc'] != rc).sum()}"
    assert (rc == -1).sum() > 2000 and (rc == 1).sum() > 2000
    assert np.abs(out["uact"] - ua).max() <= 1e-6
    ok = rc == 1
    assert np.abs(out["relax"][:, ok] - rl[:, ok]).max() <= 1e-5
    assert np.all(out["relax"][:, ~ok] == -7.0)
    # failures fall back to the saturated backup controller u = K x (src/asif_implicit.cpp:348-355)
    fb = ~ok
    uk = np.clip(-10.0 * out["x"][0] - 20.0 * out["x"][1], -1.0, 1.0)
    assert np.allclose(out["uact"][0][fb], uk[fb], atol=1e-12)


def test_fused_solve_and_its_hand_over_to_stage_two(hip, monkeypatch):
    """DoubleIntegrator_implicit, default solver mode: the rows kernel solves each instance's 3 x 17 QP itself and stages
    nothing.  ASIF_HIP_IM_FUSE forces the hand-over to stage 2 that no seeded instance takes by itself (2: every
    instance, 3: every second one), 0 is the two-launch path: uAct, relax, rc, diagnostics bitwise identical across all."""
    import gpu_util
    outs = []
    for v in (None, "0", "2", "3"):
        if v is None:
            monkeypatch.delenv("ASIF_HIP_IM_FUSE", raising=False)
        else:
            monkeypatch.setenv("ASIF_HIP_IM_FUSE", v)
        outs.append(gpu_util.run_filter(9, 5003, uact_init=7.0, relax_init=-7.0))
    ref = outs[0]
    assert {1, -1} <= set(np.unique(ref["rc"]).tolist())
    for o in outs[1:]:
        assert np.array_equal(o["rc"], ref["rc"]) and np.array_equal(o["uact"], ref["uact"])
        assert np.array_equal(o["relax"], ref["relax"]) and np.array_equal(o["diag"], ref["diag"])
