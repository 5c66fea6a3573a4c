"""class ASIFimplicitRB (src/asif_implicit_robust.cpp) on the GPU vs the oracle: the backup input
held over backContDt, interval safety margins under x_unc (bit-exact, libaffa arithmetic), the learned
residual on the first row, the n_debug sample selection, and the reduction to ASIFimplicit (bitwise).
Same bar as C3: rows at rtol 1e-9, u* <= 1e-6 against the exact optimum, rc identical."""
import numpy as np
import pytest
import torch

import gpu_util
from asif_amd import workloads

pytestmark = pytest.mark.gpu
CFG = 10
RB = 5


def _both_options(hip, oracle, model_dev, model_or, x_unc=workloads.RB_X_UNC, learning=True, **kw):
    od = hip.default_options(model_dev, RB)
    oo = oracle.default_options(model_or, oracle.VAR_IMPLICIT_RB)
    for o in (od, oo):
        for i, v in enumerate(x_unc):
            o.x_unc[i] = v
        for k, v in kw.items():
            setattr(o, k, v)
    w = None
    if learning:
        w = workloads.make_learning()
        od.use_learning = 1
        oo.set_learning(oracle.Learning.from_dict(w))
    return od, oo, w


def test_rows_match_oracle_with_hold_uncertainty_and_learning(hip, oracle):
    B = 1024
    od, oo, w = _both_options(hip, oracle, 1, oracle.MODEL_IP)
    out = gpu_util.run_assemble(CFG, B, options=od, learning=w)
    d = out["dims"]
    assert (d.nv, d.nc, d.npBT, d.npBTSS, d.ndiag) == (3, 41, 5001, 10, 10 + 2 + 1 + 1 + 1)
    xT = np.ascontiguousarray(out["x"].T)
    A, b, code, _ = oracle.assemble_batch(oracle.MODEL_IP, oracle.VAR_IMPLICIT_RB, oo, xT)
    assert np.all(out["code"] == 1) and np.all(code == 1)
    np.testing.assert_allclose(out["A"].T, A, rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(out["b"].T, b, rtol=1e-9, atol=1e-11)
    # diagnostics = the class's public members: critical samples, Dh_index_, Lfh_diff, Lgh_diff
    for i in range(0, B, 97):
        oracle.assemble(oracle.MODEL_IP, oracle.VAR_IMPLICIT_RB, oo, xT[i])
        dh, lf, lg = oracle.rb_last_learning()
        assert np.array_equal(out["diag"][:10, i].astype(int), oracle.last_crit_idx())
        np.testing.assert_allclose(out["diag"][10:12, i], dh[:2], rtol=1e-9, atol=1e-12)
        assert abs(out["diag"][12, i] - lf) <= 1e-11 and abs(out["diag"][13, i] - lg[0]) <= 1e-11
    # and they are not the plain implicit class's rows
    plain = gpu_util.run_assemble(3, B, x=out["x"])
    dA = (out["A"] - plain["A"]).reshape(3, 41, B)
    assert np.abs(dA[1, :40]).min() > 0.0 and np.abs(dA[0, 0]).min() > 0.0


def test_interval_margins_are_bit_exact(hip, oracle):
    """A horizon of npBTSS - 1 steps keeps every sample, and sample 0 is the input state itself: its margins
    must equal libaffa's evaluation of the interval safety set (oracle/or_affine.c, pinned on
    tests/golden/affa_box_safety_interval.json) to the last bit."""
    B = 2048
    for model_dev, model_or, cfg_plain in ((1, oracle.MODEL_IP, 3), (7, 5, 9)):
        od, oo, _ = _both_options(hip, oracle, model_dev, model_or, x_unc=(0.03, 0.007), learning=False)
        K = 10 if model_dev == 1 else 4
        for o in (od, oo):
            o.backTrajHorizon = (K - 1) * o.backTrajDt
        x, _ = workloads.make_batch(cfg_plain, B)
        out = gpu_util.run_assemble(CFG, B, options=od, x=x, model=model_dev, variant=RB)
        d = out["dims"]
        assert d.npBT == K and d.npBTSS == K
        A = out["A"].reshape(d.nv, d.nc, B)
        idx = out["diag"][:K].astype(int)
        assert np.array_equal(np.sort(idx, axis=0), np.tile(np.arange(K)[:, None], (1, B)))
        for i in range(B):
            k = int(np.where(idx[:, i] == 0)[0][0])
            lo = oracle.rb_safety_lo(model_or, oo, x[:, i])
            assert A[1, 4 * k:4 * k + 4, i].tolist() == lo.tolist()
        Ao, bo, _, _ = oracle.assemble_batch(model_or, oracle.VAR_IMPLICIT_RB, oo, np.ascontiguousarray(x.T))
        np.testing.assert_allclose(out["A"].T, Ao, rtol=1e-12, atol=1e-13)
        np.testing.assert_allclose(out["b"].T, bo, rtol=1e-12, atol=1e-13)


def test_reduces_to_implicit_bitwise(hip, oracle):
    B = 4096
    od, _, _ = _both_options(hip, oracle, 1, oracle.MODEL_IP, x_unc=(0.0, 0.0), learning=False)
    od.backContDt = od.backTrajDt
    a = gpu_util.run_filter(CFG, B, options=od, uact_init=3.0, relax_init=-3.0)
    p = gpu_util.run_filter(3, B, x=a["x"], udes=a["udes"], uact_init=3.0, relax_init=-3.0)
    assert np.array_equal(a["rc"], p["rc"]) and np.array_equal(a["uact"], p["uact"])
    assert np.array_equal(a["relax"], p["relax"])


@pytest.mark.parametrize("lanes", [0, 4])
def test_filter_matches_exact_optimum(hip, oracle, lanes):
    B = 4096
    od, oo, w = _both_options(hip, oracle, 1, oracle.MODEL_IP)
    s = hip.default_solver(lanes_per_qp=lanes)
    out = gpu_util.run_filter(CFG, B, solver=s, options=od, learning=w, uact_init=7.0, relax_init=-7.0)
    ua, rl, rc = oracle.filter_batch(oracle.MODEL_IP, oracle.VAR_IMPLICIT_RB, oo, np.ascontiguousarray(out["x"].T),
                                     np.ascontiguousarray(out["udes"].T), uact_init=np.full((B, 1), 7.0), nthreads=8)
    assert np.array_equal(out["rc"], rc), f"rc mismatches {(out['rc'] != rc).sum()}"
    ok = rc == 1
    assert ok.sum() > 2000 and (~ok).sum() > 20
    assert np.abs(out["uact"][0] - ua[:, 0]).max() <= 1e-6
    assert np.abs(out["relax"][:, ok] - rl[ok].T).max() <= 1e-5
    assert np.all(out["relax"][:, ~ok] == -7.0)
    uk = np.clip(-3.0 * out["x"][0] - 3.0 * out["x"][1], -1.5, 1.5)  # src/asif_implicit_robust.cpp:427-433
    assert np.allclose(out["uact"][0][~ok], uk[~ok], atol=1e-12)
    # the learned residual and the uncertainty change the answer: it is not the implicit class's
    p = gpu_util.run_filter(3, B, x=out["x"], udes=out["udes"])
    assert np.abs(p["uact"] - out["uact"]).max() > 1e-3


def test_n_debug_and_di_model(hip, oracle):
    B = 1024
    # n_debug picks the sample whose Dh feeds the networks; out of range falls back to the most critical one
    od, oo, w = _both_options(hip, oracle, 1, oracle.MODEL_IP, n_debug=300)
    out = gpu_util.run_assemble(CFG, B, options=od, learning=w)
    A, b, _, _ = oracle.assemble_batch(oracle.MODEL_IP, oracle.VAR_IMPLICIT_RB, oo, np.ascontiguousarray(out["x"].T))
    np.testing.assert_allclose(out["A"].T, A, rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(out["b"].T, b, rtol=1e-9, atol=1e-11)
    od2, _, _ = _both_options(hip, oracle, 1, oracle.MODEL_IP, n_debug=99999)
    od3, _, _ = _both_options(hip, oracle, 1, oracle.MODEL_IP)
    a2 = gpu_util.run_assemble(CFG, B, options=od2, learning=w)
    a3 = gpu_util.run_assemble(CFG, B, options=od3, learning=w)
    assert np.array_equal(a2["A"], a3["A"]) and np.array_equal(a2["b"], a3["b"])
    assert not np.array_equal(out["A"], a3["A"])
    # the double integrator of examples/DoubleIntegrator_implicit.cpp under the same class
    od, oo, w = _both_options(hip, oracle, 7, 5, x_unc=(0.01, 0.02), backContDt=0.05)
    x, u = workloads.make_batch(9, 4096)
    outf = gpu_util.run_filter(CFG, 4096, options=od, learning=w, x=x, udes=u, model=7, variant=RB, uact_init=7.0)
    ua, rl, rc = oracle.filter_batch(5, oracle.VAR_IMPLICIT_RB, oo, np.ascontiguousarray(x.T),
                                     np.ascontiguousarray(u.T), uact_init=np.full((4096, 1), 7.0), nthreads=8)
    assert np.array_equal(outf["rc"], rc)
    assert (rc == 1).sum() > 500 and (rc == -1).sum() > 500
    assert np.abs(outf["uact"][0] - ua[:, 0]).max() <= 1e-6


def test_learning_without_weights_fails_loudly(hip, oracle):
    od, _, _ = _both_options(hip, oracle, 1, oracle.MODEL_IP)  # use_learning = 1, no set_learning
    with pytest.raises(Exception):
        gpu_util.run_assemble(CFG, 64, options=od)
    # and malformed networks are refused
    flt = hip.Filter(1, RB, options=od)
    w = workloads.make_learning(hidden=(48, 16))
    with pytest.raises(Exception):
        flt.set_learning(w)
    w = workloads.make_learning()
    w["d_drift_in"] = 3
    with pytest.raises(Exception):
        flt.set_learning(w)
    flt.close()


def test_rejected_weights_leave_the_handle_as_it_was(hip, oracle):
    """asif_hip_set_learning validates and uploads before it swaps: a refused call keeps the previous networks."""
    od, _, _ = _both_options(hip, oracle, 1, oracle.MODEL_IP)
    good = workloads.make_learning()
    before = gpu_util.run_assemble(CFG, 256, options=od, learning=good)
    flt = hip.Filter(1, RB, options=od)
    flt.set_learning(good)
    with pytest.raises(Exception):
        flt.set_learning(workloads.make_learning(hidden=(48, 16)))
    d = flt.dims
    dev = torch.device("cuda:0")
    x = torch.from_numpy(before["x"]).to(dev)
    A = torch.zeros(d.nc * d.nv, 256, dtype=torch.float64, device=dev)
    b = torch.zeros(d.nc, 256, dtype=torch.float64, device=dev)
    code = torch.zeros(256, dtype=torch.int32, device=dev)
    flt.assemble(x, A, b, code)
    torch.cuda.synchronize()
    assert np.array_equal(A.cpu().numpy(), before["A"]) and np.array_equal(b.cpu().numpy(), before["b"])
    flt.close()


def test_plain_implicit_class_with_its_learned_residual(hip, oracle):
    """use_learning / n_debug / learning_data_ also exist in ASIFimplicit (include/asif_implicit.h:23,33,125;
    src/asif_implicit.cpp:585-588): variant IMPLICIT with the option set adds the same residual, nothing else."""
    B = 2048
    w = workloads.make_learning()
    od = hip.default_options(1, 1)
    od.use_learning = 1
    oo = oracle.default_options(oracle.MODEL_IP, oracle.VAR_IMPLICIT)
    oo.set_learning(oracle.Learning.from_dict(w))
    out = gpu_util.run_assemble(3, B, options=od, learning=w)
    A, b, _, _ = oracle.assemble_batch(oracle.MODEL_IP, oracle.VAR_IMPLICIT, oo, np.ascontiguousarray(out["x"].T))
    np.testing.assert_allclose(out["A"].T, A, rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(out["b"].T, b, rtol=1e-9, atol=1e-11)
    plain = gpu_util.run_assemble(3, B, x=out["x"])
    dA = (out["A"] - plain["A"]).reshape(3, 41, B)
    db = out["b"] - plain["b"]
    assert np.abs(dA[0, 0]).min() > 0 and np.abs(db[0]).min() > 0
    assert np.all(dA[0, 1:] == 0) and np.all(dA[1:] == 0) and np.all(db[1:] == 0)  # bitwise elsewhere
    f = gpu_util.run_filter(3, B, options=od, learning=w, x=out["x"], udes=workloads.make_batch(3, B)[1], uact_init=7.0)
    ua, rl, rc = oracle.filter_batch(oracle.MODEL_IP, oracle.VAR_IMPLICIT, oo, np.ascontiguousarray(f["x"].T),
                                     np.ascontiguousarray(f["udes"].T), uact_init=np.full((B, 1), 7.0), nthreads=8)
    assert np.array_equal(f["rc"], rc) and np.abs(f["uact"][0] - ua[:, 0]).max() <= 1e-6


def test_update_options_equals_fresh_handle(hip, oracle):
    """updateOptions(options) (src/asif_implicit_robust.cpp:436-477) on a live handle: same rows as a handle created
    with those options; the uploaded networks survive the update."""
    import torch
    B = 512
    w = workloads.make_learning()
    od, _, _ = _both_options(hip, oracle, 1, oracle.MODEL_IP)
    flt = hip.Filter(1, RB, options=od)
    flt.set_learning(w)
    od2, _, _ = _both_options(hip, oracle, 1, oracle.MODEL_IP, x_unc=(0.05, 0.0), backContDt=0.004, relaxCost=20.0)
    flt.update_options(od2)
    x, _ = workloads.make_batch(CFG, B)
    dev = torch.device("cuda:0")
    d = flt.dims
    A = torch.zeros((d.nc * d.nv, B), dtype=torch.float64, device=dev)
    b = torch.zeros((d.nc, B), dtype=torch.float64, device=dev)
    code = torch.zeros(B, dtype=torch.int32, device=dev)
    flt.assemble(torch.from_numpy(x).to(dev), A, b, code)
    torch.cuda.synchronize()
    fresh = gpu_util.run_assemble(CFG, B, options=od2, learning=w, x=x)
    assert np.array_equal(A.cpu().numpy(), fresh["A"]) and np.array_equal(b.cpu().numpy(), fresh["b"])
    flt.close()
