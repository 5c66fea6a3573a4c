"""The oracle's two QP solvers cross-checked: exact enumeration vs OSQP-style ADMM at tight
tolerance, KKT conditions of the exact answer in numpy, infeasibility, return codes of filter()."""
import numpy as np


def _config_qps(oracle, cfg, B):
    model, variant = oracle.CONFIGS[cfg]
    o = oracle.default_options(model, variant)
    d = oracle.dims(model, variant, o)
    x, u = oracle.make_batch(cfg, B)
    A, b, code, _ = oracle.assemble_batch(model, variant, o, x)
    keep = (code == 1) | (code == 2)
    A, b, u = A[keep], b[keep], u[keep]
    n = len(A)
    Hd, c, lb, ub = (np.zeros((n, d.nv)) for _ in range(4))
    be = None
    for i in range(n):
        Hd[i], c[i], lb[i], ub[i], be = oracle.qp_static(model, variant, o, u[i])
    return d, Hd, c, A, b, lb, ub, be


def test_exact_solution_satisfies_kkt(oracle):
    for cfg, B in ((2, 512), (3, 48), (4, 2048)):
        d, Hd, c, A, b, lb, ub, be = _config_qps(oracle, cfg, B)
        sol, st, _ = oracle.qp_solve_batch(d.nv, d.nc, Hd, c, A, b, lb, ub, be, oracle.SOLVER_EXACT)
        ok = st == 1
        assert ok.sum() > 10
        for i in np.where(ok)[0][:200]:
            Am = A[i].reshape(d.nv, d.nc).T
            x = sol[i]
            assert np.all(Am @ x - b[i] >= -1e-9 * (1 + np.abs(b[i])))
            assert np.all(x >= lb[i] - 1e-9) and np.all(x <= ub[i] + 1e-9)
            # stationarity: 2Hx + c = A' mu + nu with mu >= 0 on active rows (NNLS via least squares on the active set)
            grad = 2 * Hd[i] * x + c[i]
            act = np.abs(Am @ x - b[i]) <= 1e-8 * (1 + np.abs(b[i]))
            cols = [Am[r] for r in np.where(act)[0]]
            for j in range(d.nv):
                e = np.zeros(d.nv)
                e[j] = 1
                if abs(x[j] - lb[i][j]) <= 1e-9:
                    cols.append(e)
                if abs(x[j] - ub[i][j]) <= 1e-9:
                    cols.append(-e)
            if cols:
                G = np.array(cols).T
                mu, res, *_ = np.linalg.lstsq(G, grad, rcond=None)
                assert np.abs(G @ mu - grad).max() <= 1e-7 * (1 + np.abs(grad).max())
            else:
                assert np.abs(grad).max() <= 1e-9


def test_admm_tight_matches_exact_all_configs(oracle):
    for cfg, B in ((2, 2048), (3, 48), (4, 256), (5, 256)):
        model, variant = oracle.CONFIGS[cfg]
        o = oracle.default_options(model, variant)
        x, u = oracle.make_batch(cfg, B)
        ua, rl, rc = oracle.filter_batch(model, variant, o, x, u, oracle.SOLVER_EXACT)
        s = oracle.admm_settings(eps_abs=1e-9, eps_rel=1e-9, max_iter=50000)
        ua2, rl2, rc2 = oracle.filter_batch(model, variant, o, x, u, oracle.SOLVER_ADMM, s)
        assert np.array_equal(rc, rc2), cfg
        ok = (rc == 1) | (rc == 2)
        assert np.nanmax(np.abs(ua[ok] - ua2[ok])) <= 2e-6, cfg


def test_reduced_kkt_equals_full_kkt(oracle):
    d, Hd, c, A, b, lb, ub, be = _config_qps(oracle, 2, 512)
    s0 = oracle.admm_settings(eps_abs=1e-8, eps_rel=1e-8)
    s1 = oracle.admm_settings(eps_abs=1e-8, eps_rel=1e-8, reduced_kkt=1)
    a0, st0, it0 = oracle.qp_solve_batch(d.nv, d.nc, Hd, c, A, b, lb, ub, be, oracle.SOLVER_ADMM, s0)
    a1, st1, it1 = oracle.qp_solve_batch(d.nv, d.nc, Hd, c, A, b, lb, ub, be, oracle.SOLVER_ADMM, s1)
    assert np.array_equal(st0, st1) and np.array_equal(it0, it1)
    ok = st0 == 1
    assert np.abs(a0[ok] - a1[ok]).max() <= 1e-9


def test_device_algorithm_emulation_is_exact(oracle):
    """The oracle's ADMM with the device's settings (power-of-two Ruiz x4, reduced KKT, active-set finish
    with Farkas certificates) must decide every instance like the exact solver -- this is the CPU
    emulation the HIP kernel was developed against."""
    for cfg, B in ((2, 8192), (3, 256), (4, 2048)):
        d, Hd, c, A, b, lb, ub, be = _config_qps(oracle, cfg, B)
        ex, stex, _ = oracle.qp_solve_batch(d.nv, d.nc, Hd, c, A, b, lb, ub, be, oracle.SOLVER_EXACT)
        s = oracle.admm_settings(max_iter=4000, polish=1, check_termination=5, adaptive_rho_interval=5,
                                 eps_abs=1e-8, eps_rel=1e-8, reduced_kkt=1, scaling_pow2=1, scaling=4)
        sol, st, it = oracle.qp_solve_batch(d.nv, d.nc, Hd, c, A, b, lb, ub, be, oracle.SOLVER_ADMM, s)
        assert np.array_equal(st == 1, stex == 1), cfg
        ok = st == 1
        assert np.abs(sol[ok] - ex[ok]).max() <= 1e-9, cfg
        assert it.max() <= 50, cfg
        # the device's defaults: finish first (polish 2), checks every 2 iterations, 2 Ruiz passes
        s = oracle.admm_settings(max_iter=4000, polish=2, check_termination=2, adaptive_rho_interval=2,
                                 eps_abs=1e-8, eps_rel=1e-8, reduced_kkt=1, scaling_pow2=1, scaling=2)
        sol, st, it = oracle.qp_solve_batch(d.nv, d.nc, Hd, c, A, b, lb, ub, be, oracle.SOLVER_ADMM, s)
        assert np.array_equal(st == 1, stex == 1), cfg
        ok = st == 1
        assert np.abs(sol[ok] - ex[ok]).max() <= 1e-9, cfg
        assert (it == 0).mean() > 0.9, cfg  # decided before the first iteration


def test_infeasible_and_edge_qps(oracle):
    # x >= 1 and x <= 0 through a row: infeasible
    Hd = np.array([[1.0, 1.0]]); c = np.zeros((1, 2)); lb = np.array([[-5.0, -5.0]]); ub = np.array([[5.0, 0.0]])
    A = np.array([[0.0, 0.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0]])  # 4x2 col-major: row 1 = [0, 1]
    b = np.array([[-1e20, 1.0, -1e20, -1e20]])
    sol, st, _ = oracle.qp_solve_batch(2, 4, Hd, c, A, b, lb, ub, None, oracle.SOLVER_EXACT)
    assert st[0] == -3
    sol, st, _ = oracle.qp_solve_batch(2, 4, Hd, c, A, b, lb, ub, None, oracle.SOLVER_ADMM)
    assert st[0] in (-3, 3)
    # unconstrained optimum inside the box: x = -c / (2H)
    c = np.array([[-2.0, 4.0]]); ub = np.array([[5.0, 5.0]]); b = np.full((1, 4), -1e20)
    sol, st, _ = oracle.qp_solve_batch(2, 4, Hd, c, A, b, lb, ub, None, oracle.SOLVER_EXACT)
    assert st[0] == 1 and np.allclose(sol[0], [1.0, -2.0])


def test_filter_return_codes_and_untouched_outputs(oracle):
    model, variant = oracle.CONFIGS[2]
    o = oracle.default_options(model, variant)
    x = np.array([[0.0, 0.0], [1.2, 1.2]])  # safe / far outside (delta is pinned: infeasible)
    u = np.array([[0.3], [0.3]])
    ua, rl, rc = oracle.filter_batch(model, variant, o, x, u, oracle.SOLVER_EXACT, uact_init=np.full((2, 1), 9.0))
    assert list(rc) == [1, -1]
    assert abs(ua[0, 0] - 0.3) < 1e-12 and rl[0, 0] == 5.0
    assert ua[1, 0] == 9.0 and np.isnan(rl[1, 0])  # src/asif.cpp:208-209: untouched
