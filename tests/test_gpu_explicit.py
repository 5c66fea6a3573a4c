"""C2 (BASELINE.json configs[1]): DoubleIntegrator explicit CBF filter on the GPU vs the oracle.

Tolerance: the north star asks |u* - u_ref| <= 1e-5 with u_ref = exact optimum of the QP the
reference assembles; the tests hold the HIP path to 1e-6.  Return codes must match
exactly; rows (A, b) to 1e-12 (the only differences are FMA contractions).
"""
import numpy as np
import pytest

import gpu_util

pytestmark = pytest.mark.gpu

U_TOL = 1e-6


def test_rows_match_oracle(hip, oracle):
    B = 4096
    out = gpu_util.run_assemble(2, B)
    model, variant = oracle.CONFIGS[2]
    o = oracle.default_options(model, variant)
    A, b, code, _ = oracle.assemble_batch(model, variant, o, np.ascontiguousarray(out["x"].T))
    assert np.array_equal(out["code"], code)
    np.testing.assert_allclose(out["A"].T, A, rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(out["b"].T, b, rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("lanes", [1, 2, 4])
def test_filter_matches_exact_optimum(hip, oracle, lanes):
    B = 8192
    s = hip.default_solver(lanes_per_qp=lanes)
    out = gpu_util.run_filter(2, B, solver=s, uact_init=7.0, relax_init=-7.0)
    ua, rl, rc = gpu_util.oracle_filter(oracle, 2, out["x"], out["udes"], uact_init=7.0, relax_init=-7.0)
    assert np.array_equal(out["rc"], rc), f"rc mismatches: {(out['rc'] != rc).sum()}"
    assert (rc == -1).sum() > 100 and (rc == 1).sum() > 1000  # both branches exercised
    err = np.abs(out["uact"] - ua).max()
    assert err <= U_TOL, err
    assert np.abs(out["relax"] - rl).max() <= U_TOL
    # untouched slots on QP failure (src/asif.cpp:208-209)
    assert np.all(out["uact"][0, rc == -1] == 7.0) and np.all(out["relax"][0, rc == -1] == -7.0)


def test_full_size_properties(hip):
    """B = 65 536 (the benchmark size): feasibility of the returned u against the assembled rows,
    input bounds, determinism, and shard equivalence (two halves == whole, bitwise)."""
    B = 65536
    out = gpu_util.run_filter(2, B)
    rows = gpu_util.run_assemble(2, B)
    ok = out["rc"] == 1
    assert set(np.unique(out["rc"])) <= {1, -1}
    u = out["uact"][0]
    d = out["relax"][0]
    assert np.all(u[ok] >= -1.0) and np.all(u[ok] <= 1.0)
    assert np.allclose(d[ok], 5.0, atol=1e-7)  # pinned relaxation variable
    A = rows["A"].reshape(2, 4, B)
    lhs = A[0] * u + A[1] * d
    viol = (rows["b"] - lhs)[:, ok].max()
    assert viol <= 1e-6, viol
    again = gpu_util.run_filter(2, B)
    assert np.array_equal(again["uact"], out["uact"]) and np.array_equal(again["rc"], out["rc"])
    h1 = gpu_util.run_filter(2, B // 2, first=0)
    h2 = gpu_util.run_filter(2, B // 2, first=B // 2)
    assert np.array_equal(np.concatenate([h1["uact"], h2["uact"]], axis=1), out["uact"])
    assert np.array_equal(np.concatenate([h1["rc"], h2["rc"]]), out["rc"])


def test_edge_cases(hip, oracle):
    # empty batch is a no-op
    import torch
    flt = hip.Filter(hip.MODEL_DOUBLE_INTEGRATOR, hip.EXPLICIT)
    e = torch.zeros((2, 0), dtype=torch.float64, device="cuda")
    flt.filter(e, e[:1], e[:1], e[:1], torch.zeros(0, dtype=torch.int32, device="cuda"))
    # ragged batch (not a multiple of the wave size) and special states: origin, corners, v = 0 switch
    x = np.array([[0.0, 1.0, -1.0, 0.999, -0.999, 0.5, 0.5, 0.0, 1.2],
                  [0.0, 1.0, -1.0, 0.0, 0.0, 0.0, -0.0, 1e-300, -1.2]])
    ud = np.array([[1.0, 1.5, -1.5, 1.0, -1.0, 0.0, 0.3, -0.2, 1.0]])
    out = gpu_util.run_filter(2, x.shape[1], x=x, udes=ud, uact_init=9.0)
    ua, rl, rc = gpu_util.oracle_filter(oracle, 2, x, ud, uact_init=9.0)
    assert np.array_equal(out["rc"], rc)
    assert np.abs(out["uact"] - ua).max() <= U_TOL


def test_presolve_closed_form_equals_admm_path(hip, oracle):
    """asif_hip_solver.presolve = 1: the pinned relaxation variable is eliminated and the one-variable QP is a
    clip.  Same optimum (to rounding) and the same return codes as the exact solver and as the ADMM path."""
    B = 65536
    pre = gpu_util.run_filter(2, B, solver=hip.default_solver(presolve=1), uact_init=7.0, relax_init=-7.0)
    adm = gpu_util.run_filter(2, B, uact_init=7.0, relax_init=-7.0)
    ua, rl, rc = gpu_util.oracle_filter(oracle, 2, pre["x"], pre["udes"], uact_init=7.0, relax_init=-7.0)
    assert np.array_equal(pre["rc"], rc) and np.array_equal(adm["rc"], rc)
    assert np.abs(pre["uact"] - ua).max() <= 1e-12
    assert np.abs(pre["uact"] - adm["uact"]).max() <= 1e-12
    ok = rc == 1
    assert np.all(pre["relax"][0, ok] == 5.0) and np.all(pre["relax"][0, ~ok] == -7.0)
    assert np.all(pre["uact"][0, ~ok] == 7.0)


def test_whole_line_store_kernel_is_bitwise_the_masked_store_kernel(hip, oracle):
    """From 6 M instances the explicit filter stores every lane of uAct / relax (a failed lane writes back the value it
    read from its slot) instead of storing under the mask of the successful lanes: partial-line stores cost HBM a
    read-modify-write (profiles/r03/stream_pattern_microbench.txt).  An odd batch just above the switch-over against the
    same instances in two launches below it (masked stores): uAct, relax, rc bitwise equal, untouched slots untouched,
    and equal to the oracle on a sample."""
    import torch
    from asif_amd import workloads
    B = 6291456 + 1
    x, udes = workloads.make_batch(2, B)
    dev = torch.device("cuda:0")
    flt = hip.Filter(*hip.CONFIGS[2][:2])

    def run(lo, hi):
        n = hi - lo
        tx = torch.from_numpy(np.ascontiguousarray(x[:, lo:hi])).to(dev)
        tu = torch.from_numpy(np.ascontiguousarray(udes[:, lo:hi])).to(dev)
        uact = torch.full((1, n), 7.0, dtype=torch.float64, device=dev)
        relax = torch.full((1, n), -7.0, dtype=torch.float64, device=dev)
        rc = torch.zeros(n, dtype=torch.int32, device=dev)
        flt.filter(tx, tu, uact, relax, rc)
        torch.cuda.synchronize()
        return uact.cpu().numpy(), relax.cpu().numpy(), rc.cpu().numpy()

    whole = run(0, B)
    h1, h2 = run(0, B // 2), run(B // 2, B)
    flt.close()
    for k in range(3):
        assert np.array_equal(whole[k], np.concatenate([h1[k], h2[k]], axis=-1))
    ua, rl, rc = whole
    fail = rc == -1
    assert fail.sum() > B // 10
    assert np.all(ua[0, fail] == 7.0) and np.all(rl[0, fail] == -7.0)
    for sl in (slice(0, 8192), slice(B - 4097, B)):
        uo, ro, rco = gpu_util.oracle_filter(oracle, 2, x[:, sl], udes[:, sl], uact_init=7.0, relax_init=-7.0)
        assert np.array_equal(rc[sl], rco)
        assert np.abs(ua[:, sl] - uo).max() <= U_TOL
