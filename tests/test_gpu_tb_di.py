"""Third time-to-backup-set model: the double integrator of examples/DoubleIntegrator_implicit_tb.cpp (disc-shaped
backup set of radius 0.01 -- the only shipped TB example whose backup set has a Hessian besides the segway's -- and a
2 101-sample trajectory; 7 001 samples after the example's updateOptions at half time) on the GPU vs the oracle.
Same bar as C4 / C8: branch and return codes identical, rows at rtol 1e-7, u* <= 1e-6 against the exact optimum.
"""
import numpy as np
import pytest
import torch

import gpu_util

pytestmark = pytest.mark.gpu
CFG = 12


def test_rows_codes_and_diagnostics(hip, oracle):
    B = 1024
    out = gpu_util.run_assemble(CFG, B)
    assert (out["dims"].nx, out["dims"].nv, out["dims"].nc, out["dims"].npBT) == (2, 2, 18, 2101)
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
    B = 4096
    out = gpu_util.run_filter(CFG, B, uact_init=77.0, relax_init=-7.0)
    ua, rl, rc = gpu_util.oracle_filter(oracle, CFG, out["x"], out["udes"], uact_init=77.0, relax_init=-7.0)
    assert np.array_equal(out["rc"], rc), f"rc mismatches {(out['rc'] != rc).sum()}"
    assert (rc == 2).sum() > 100 and (rc == -3).sum() > 1000 and (rc == 1).sum() > 1000
    assert np.abs(out["uact"] - ua).max() <= 1e-6
    ok = (rc == 1) | (rc == 2)
    assert np.abs(out["relax"][:, ok] - rl[:, ok]).max() <= 1e-6
    # failures fall back to the saturated backup controller u = K x
    fb = rc < 0
    assert np.allclose(out["uact"][0][fb], np.clip(-10.0 * out["x"][0][fb] - 20.0 * out["x"][1][fb], -1.0, 1.0), atol=1e-12)


def test_after_the_examples_update_options(hip, oracle):
    """examples/DoubleIntegrator_implicit_tb.cpp:134-139: backTrajHorizon 2 -> 7 through updateOptions(), which sizes the
    trajectory WITHOUT (1 + backTrajExtend) (src/asif_implicit_tb.cpp:377 vs :177): 7 001 samples, not 7 351; with the
    longer horizon states ten times further out reach the disc."""
    B = 1024
    model, variant, _ = hip.CONFIGS[CFG]
    flt = hip.Filter(model, variant)
    o = flt.options
    o.backTrajHorizon = 7.0
    flt.update_options(o)
    assert flt.dims.npBT == 7001
    x, udes = gpu_util.workloads.make_batch(CFG, B)
    x = np.ascontiguousarray(x * 8.0)  # +-0.32
    dev = torch.device("cuda:0")
    tx, tu = torch.from_numpy(x).to(dev), torch.from_numpy(udes).to(dev)
    uact = torch.full((1, B), 3.0, dtype=torch.float64, device=dev)
    relax = torch.zeros((1, B), dtype=torch.float64, device=dev)
    rc = torch.zeros(B, dtype=torch.int32, device=dev)
    diag = torch.zeros((flt.dims.ndiag, B), dtype=torch.float64, device=dev)
    flt.filter(tx, tu, uact, relax, rc, diag)
    torch.cuda.synchronize()
    flt.close()
    om, ov = oracle.CONFIGS[CFG]
    oo = oracle.default_options(om, ov)
    oo.backTrajHorizon = 7.0
    oo.backTrajExtend = 0.0  # the oracle sizes as initialize() does: 7 (1 + 0) / 0.001 + 1 = 7 001
    assert oracle.dims(om, ov, oo).npBT == 7001
    ua, rl, orc = oracle.filter_batch(om, ov, oo, np.ascontiguousarray(x.T), np.ascontiguousarray(udes.T),
                                      oracle.SOLVER_EXACT, uact_init=np.full((B, 1), 3.0))
    assert np.array_equal(rc.cpu().numpy(), orc), f"rc mismatches {(rc.cpu().numpy() != orc).sum()}"
    assert (orc == 1).sum() > 300 and (orc == -3).sum() > 50
    assert np.abs(uact.cpu().numpy().T - ua).max() <= 1e-6


def test_ragged_batches_equal_their_prefix_of_a_larger_one(hip):
    big = gpu_util.run_filter(CFG, 1000, uact_init=5.0)
    for B in (1, 63, 65):
        out = gpu_util.run_filter(CFG, B, uact_init=5.0)
        assert np.array_equal(out["rc"], big["rc"][:B])
        assert np.array_equal(out["uact"], big["uact"][:, :B])
