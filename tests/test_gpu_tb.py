"""C4 (BASELINE.json configs[3]): segway time-to-backup-set filter on the GPU vs the oracle.

Branch codes (2 inside the backup set / 1 rows assembled / -3 backup set never reached) and the final
return codes must be identical.  Rows pass through 300 Euler steps of trig-heavy dynamics with FMA
contraction and ocml vs glibc transcendentals: compared at rtol 1e-7 (observed ~1e-11); u* <= 1e-6
against the exact optimum.
"""
import numpy as np
import pytest

import gpu_util

pytestmark = pytest.mark.gpu


def test_rows_codes_and_diagnostics(hip, oracle):
    B = 2048
    out = gpu_util.run_assemble(4, B)
    model, variant = oracle.CONFIGS[4]
    o = oracle.default_options(model, variant)
    A, b, code, diag = oracle.assemble_batch(model, variant, o, np.ascontiguousarray(out["x"].T))
    assert np.array_equal(out["code"], code)
    assert {1, 2, -3} <= set(np.unique(code))
    m = code == 1
    np.testing.assert_allclose(out["A"].T[m], A[m], rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(out["b"].T[m], b[m], rtol=1e-7, atol=1e-9)
    t = code == 2  # trivial rows: A = 0, b = -inf (1e20)
    assert np.all(out["A"].T[t] == 0.0) and np.all(out["b"].T[t] == -1e20)
    np.testing.assert_allclose(out["diag"][0][m], diag[m, 0], rtol=1e-12)  # TTS_
    np.testing.assert_allclose(out["diag"][1][m], diag[m, 1], rtol=1e-8)   # BTorthoBS_
    assert np.array_equal(out["diag"][2][m], diag[m, 2])                    # idxHit
    assert np.all(out["diag"][1][t] == 1.0) and np.all(out["diag"][1][code == -3] == 0.0)


@pytest.mark.parametrize("lanes", [2, 4])
def test_filter_matches_exact_optimum(hip, oracle, lanes):
    B = 4096
    s = hip.default_solver(lanes_per_qp=lanes)
    out = gpu_util.run_filter(4, B, solver=s, uact_init=77.0, relax_init=-7.0)
    ua, rl, rc = gpu_util.oracle_filter(oracle, 4, out["x"], out["udes"], uact_init=77.0, relax_init=-7.0)
    assert np.array_equal(out["rc"], rc), f"rc mismatches {(out['rc'] != rc).sum()}"
    assert (rc == 2).sum() > 500 and (rc == -3).sum() > 1000 and (rc == 1).sum() >= 10
    assert np.abs(out["uact"] - ua).max() <= 1e-6
    ok = (rc == 1) | (rc == 2)
    assert np.abs(out["relax"][:, ok] - rl[:, ok]).max() <= 1e-6
    assert np.all(out["relax"][:, ~ok] == -7.0)


def test_full_size_properties(hip):
    """One GPU's share of C4 (32 768 instances): fallback = saturated backup controller, solved
    instances satisfy their rows, halves == whole."""
    B = 32768
    out = gpu_util.run_filter(4, B)
    rows = gpu_util.run_assemble(4, B)
    rc = out["rc"]
    assert set(np.unique(rc)) <= {1, 2, -1, -2, -3}
    x = out["x"]
    K = np.array([44.7214, 44.6528, 150.1612, 37.6492])
    xt = x + np.array([0.0, 0.0, -0.1383244254, 0.0])[:, None]
    ub = np.clip(K @ xt, -20.0, 20.0)
    fb = rc < 0
    assert np.allclose(out["uact"][0][fb], ub[fb], rtol=0, atol=1e-12)
    ok = (rc == 1) | (rc == 2)
    A = rows["A"].reshape(2, 18, B)
    lhs = A[0] * out["uact"][0] + A[1] * out["relax"][0]
    assert (rows["b"] - lhs)[:, ok].max() <= 1e-6
    inside = rc == 2
    assert np.allclose(out["uact"][0][inside], np.clip(out["udes"][0][inside], -20, 20), atol=1e-9)
    h1 = gpu_util.run_filter(4, B // 2, first=0)
    h2 = gpu_util.run_filter(4, B // 2, first=B // 2)
    assert np.array_equal(np.concatenate([h1["uact"], h2["uact"]], axis=1), out["uact"])
    assert np.array_equal(np.concatenate([h1["rc"], h2["rc"]]), rc)


def test_two_role_pass_is_bitwise_the_fused_pass(hip):
    """Segway, pass 1 dealt to two waves (x on one SIMD, gradients + sensitivity one block behind on another)
    against the fused pass (ASIF_HIP_TB_SPLIT=0, read once per process: each side in its own process): rows, diagnostics,
    uAct, relax, rc bitwise identical on a ragged batch that mixes all three branches -- a batch above the switch-over and
    its shards below it must agree, which is what the multi-GPU path is tested on."""
    import hashlib
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, json, hashlib; sys.path.insert(0, %r); sys.path.insert(0, %r + '/tests'); import gpu_util\n"
            "r = gpu_util.run_assemble(4, 4099); f = gpu_util.run_filter(4, 4099, uact_init=7.0, relax_init=-7.0)\n"
            "print(json.dumps({k: hashlib.sha256(v.tobytes()).hexdigest() for k, v in "
            "(('A', r['A']), ('b', r['b']), ('code', r['code']), ('diag', r['diag']), ('uact', f['uact']), ('relax', f['relax']), ('rc', f['rc']))}))\n"
            % (root, root))
    res = []
    for v in ("1", "0"):
        o = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, ASIF_HIP_TB_SPLIT=v), capture_output=True,
                           text=True, timeout=300)
        assert o.returncode == 0, o.stderr[-1500:]
        res.append(json.loads(o.stdout.strip().split("\n")[-1]))
    assert res[0] == res[1]


@pytest.mark.parametrize("B", [1, 63, 65, 1000])
def test_small_and_ragged_batches_on_the_two_role_pass(hip, oracle, B):
    out = gpu_util.run_filter(4, B, uact_init=7.0, relax_init=-7.0)
    ua, rl, rc = gpu_util.oracle_filter(oracle, 4, out["x"], out["udes"], uact_init=7.0, relax_init=-7.0)
    assert np.array_equal(out["rc"], rc)
    assert np.abs(out["uact"] - ua).max() <= 1e-6


@pytest.mark.parametrize("cfg,B", [(4, 4099), (12, 3001)])
def test_fused_solve_and_its_hand_over_to_stage_two(hip, monkeypatch, cfg, B):
    """ASIFimplicitTB, default solver mode: the rows kernel solves each instance's 2 x 18 QP itself and stages nothing
    (DESIGN 4.2).  An instance its stage leaves undecided is handed to stage 2 with its rows and a mark -- no seeded
    instance is, so ASIF_HIP_TB_FUSE forces it: 2 marks every instance that has a QP, 3 every second one (waves with
    marked and unmarked lanes side by side), 0 is the two-launch path of before.  uAct, relax, rc and the iteration
    diagnostics bitwise identical across all four, untouched slots untouched."""
    outs = []
    for v in (None, "0", "2", "3"):
        if v is None:
            monkeypatch.delenv("ASIF_HIP_TB_FUSE", raising=False)
        else:
            monkeypatch.setenv("ASIF_HIP_TB_FUSE", v)
        outs.append(gpu_util.run_filter(cfg, B, uact_init=7.0, relax_init=-7.0))
    ref = outs[0]
    assert {1, 2, -3} <= set(np.unique(ref["rc"]).tolist())
    for o in outs[1:]:
        assert np.array_equal(o["rc"], ref["rc"]) and np.array_equal(o["uact"], ref["uact"])
        assert np.array_equal(o["relax"], ref["relax"]) and np.array_equal(o["diag"], ref["diag"])
