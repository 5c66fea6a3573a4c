"""QPWrapperHost's Newton stage (asif_amd/host/qp_alm_host.cpp behind tests/host_alm_driver.cpp): the wave kernels'
method -- proximal method of multipliers + semismooth Newton -- on the calling thread, for the problems the reference's
robust and realizable classes hand their solver for one agent.  Same bar as the kernels (tests/test_gpu_qp_lds.py):
status identical to the oracle's exact solver on EVERY instance, |u - u_ref| <= 1e-6.  CPU only."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from test_oracle_qp import _config_qps

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
U_TOL = 1e-6


@pytest.fixture(scope="module")
def alm(tmp_path_factory):
    if os.environ.get("ASIF_SAN_DIR"):  # tests/test_sanitizers.py: the build of `make -C tests san`
        so = os.path.join(os.environ["ASIF_SAN_DIR"], "libalm_host_san.so")
    else:
        so = str(tmp_path_factory.mktemp("alm") / "libalm_host.so")
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", os.path.join(ROOT, "tests", "host_alm_driver.cpp"),
                               os.path.join(ROOT, "asif_amd", "host", "qp_alm_host.cpp"), "-o", so])
    lib = C.CDLL(so)

    def solve(nv, nc, H, c, A, b, lb, ub, be=None, diag=True, eps=1e-8, max_newton=400, warm=None, warm_in=False):
        # warm = (x [B, nv], y [B, nc + nv]) float64 C-contiguous arrays, written by the call, read first with warm_in
        B = c.shape[0]
        arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in (H, c, A, b, lb, ub)]
        sol = np.zeros((B, nv))
        st = np.zeros(B, dtype=np.int32)
        nw = np.zeros(B, dtype=np.int32)
        P = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
        bep = np.ascontiguousarray(be, dtype=np.uint8).ctypes.data_as(C.POINTER(C.c_uint8)) if be is not None else None
        wx, wy = (P(warm[0]), P(warm[1])) if warm is not None else (None, None)
        r = lib.alm_host_solve_batch(nv, nc, C.c_int64(B), int(diag), *[P(a) for a in arrs], bep, C.c_double(eps), max_newton,
                                     P(sol), st.ctypes.data_as(C.POINTER(C.c_int32)), nw.ctypes.data_as(C.POINTER(C.c_int32)),
                                     wx, wy, int(warm_in))
        assert r == 0
        return sol, st, nw
    return solve


def test_robust_full_18x12_every_instance(alm, oracle):
    B = 2048
    d, Hd, c, A, b, lb, ub, be = _config_qps(oracle, 5, B)
    assert (d.nv, d.nc) == (18, 12) and be.sum() == 8
    model, variant = oracle.CONFIGS[5]
    o = oracle.default_options(model, variant)
    x, u = oracle.make_batch(5, B)
    ua, rl, rc = oracle.filter_batch(model, variant, o, x, u, oracle.SOLVER_EXACT)
    sol, st, nw = alm(d.nv, d.nc, Hd, c, A, b, lb, ub, be)
    assert np.all(rc == 1) and np.array_equal(st, rc)
    assert np.abs(sol[:, 0].clip(o.lb[0], o.ub[0]) - ua[:, 0]).max() <= U_TOL
    assert np.abs(sol[:, 1] - rl[:, 0]).max() <= U_TOL
    assert sol[:, 2:].min() >= -1e-7  # multipliers stay in their cone
    assert nw.max() <= 40


def test_robust_data_22x15_every_status(alm, oracle):
    hp = oracle.load_halfplanes()
    z = oracle.RobustData(hp)
    B = 2048
    x, u = oracle.make_batch_robust_data(hp, B)
    ua, rl, rc = z.filter(x, u)
    A, b, code, sel = z.assemble(x)
    Hd, c, lb, ub = (np.zeros((B, z.nv)) for _ in range(4))
    for i in range(B):
        Hd[i], c[i], lb[i], ub[i], be = z.qp_static(u[i])
    sol, st, nw = alm(z.nv, z.nc, Hd, c, A, b, lb, ub, be)
    assert (rc == -1).sum() > 100
    assert np.array_equal(st == 1, rc == 1), f"{((st == 1) != (rc == 1)).sum()} status mismatches"
    assert np.all(st[rc != 1] == -3)  # primal infeasible, the raw OSQP-style code
    ok = rc == 1
    assert np.abs(sol[ok, 0].clip(-20, 20) - ua[ok, 0]).max() <= U_TOL
    assert np.abs(sol[ok, 1] - rl[ok, 0]).max() <= U_TOL


@pytest.mark.parametrize("kernel,shape,B", [("100Hz", (38, 29), 384), ("10Hz_50pt", (62, 47), 192), ("10Hz", (86, 65), 128)])
def test_realizable_full_problem(alm, oracle, kernel, shape, B):
    """What ASIFrealizable::filter hands to QPsolver_ (src/asif_realizable.cpp:300-340): the full lifted problem."""
    k = oracle.load_kernel(kernel)
    z = oracle.Realizable(k)
    assert (z.nv, z.nc) == shape
    x, u = oracle.make_batch_realizable(k, B)
    ua, rl, rc = z.filter(x, u)
    A, b, code, info = z.assemble(x)
    keep = code == 1
    x, u, ua, rc, A, b = x[keep], u[keep], ua[keep], rc[keep], A[keep], b[keep]
    n = len(x)
    Hd, c, lb, ub = (np.zeros((n, z.nv)) for _ in range(4))
    for i in range(n):
        Hd[i], c[i], lb[i], ub[i], be = z.qp_static(u[i])
    sol, st, nw = alm(z.nv, z.nc, Hd, c, A, b, lb, ub, be)
    assert np.array_equal(st == 1, rc == 1), f"{((st == 1) != (rc == 1)).sum()} of {n} status mismatches"
    ok = rc == 1
    assert ok.sum() > B // 4
    assert np.abs(sol[ok, 0].clip(-20, 20) - ua[ok, 0]).max() <= U_TOL


def test_filter_shapes_match_the_exact_solver(alm, oracle):
    """The small classes' QPs too (the active-set stage owns them in QPWrapperHost; the method must not depend on that)."""
    for cfg, B in ((2, 2048), (4, 1024), (3, 128)):
        d, Hd, c, A, b, lb, ub, be = _config_qps(oracle, cfg, B)
        ex, stex, _ = oracle.qp_solve_batch(d.nv, d.nc, Hd, c, A, b, lb, ub, be, oracle.SOLVER_EXACT)
        sol, st, nw = alm(d.nv, d.nc, Hd, c, A, b, lb, ub, be)
        assert np.array_equal(st == 1, stex == 1), cfg
        assert np.all(st[stex != 1] == -3), cfg
        ok = st == 1
        assert np.abs(sol[ok, 0] - ex[ok, 0]).max() <= U_TOL, cfg


@pytest.mark.parametrize("nv,nc", [(6, 9), (20, 30), (70, 70)])
def test_full_cost_matrix_kkt(alm, nv, nc):
    """diagonalCost = false (src/qpwrapper_osqp.cpp:276-309): optimality checked directly -- feasibility, and
    stationarity 2Hx + c = A'mu + nu with mu >= 0 on the active rows (non-negative least squares).  Only the upper
    triangle of H may be read: the lower one is filled with garbage."""
    from scipy.optimize import nnls
    from test_gpu_qp_lds import _random_dense
    rng = np.random.default_rng(nv * 100 + nc)
    B = 48
    H, c, A, b, lb, ub = _random_dense(rng, B, nv, nc)
    Hcm = H.transpose(0, 2, 1).copy()  # column-major per instance: Hcm[b, j, i] = H[i, j]
    jj, ii = np.meshgrid(np.arange(nv), np.arange(nv), indexing="ij")  # entry (i, j) lives at [j, i]; below the diagonal: i > j
    Hcm[:, jj[ii > jj], ii[ii > jj]] = 1e6
    sol, st, nw = alm(nv, nc, Hcm.reshape(B, nv * nv), c, A.reshape(B, nv * nc), b, lb, ub, diag=False)
    assert np.all(st == 1)
    for i in range(B):
        x = sol[i]
        Am = A[i].T
        assert (Am @ x - b[i]).min() >= -1e-8 and (x - lb[i]).min() >= -1e-8 and (ub[i] - x).min() >= -1e-8
        grad = 2 * H[i] @ x + c[i]
        cols = [Am[r] for r in np.where(Am @ x - b[i] <= 1e-7)[0]]
        cols += [np.eye(nv)[j] for j in np.where(x - lb[i] <= 1e-7)[0]]
        cols += [-np.eye(nv)[j] for j in np.where(ub[i] - x <= 1e-7)[0]]
        if cols:
            mu, res = nnls(np.array(cols).T, grad)
            assert res <= 1e-6 * (1 + np.abs(grad).max()), (i, res)
        else:
            assert np.abs(grad).max() <= 1e-7


def test_data_outside_the_domain_and_budget(alm, oracle):
    """Non-finite data: -2 (OSQP's max_iter value, as the device path) with a zero solution, neighbours untouched;
    a Newton budget of 1 on a problem that needs more: -2 as well."""
    d, Hd, c, A, b, lb, ub, be = _config_qps(oracle, 5, 8)
    A2, c2, ub2 = A.copy(), c.copy(), ub.copy()
    A2[1, 3] = np.nan
    c2[3, 0] = np.inf
    ub2[5, 2] = np.nan
    sol, st, nw = alm(d.nv, d.nc, Hd, c2, A2, b, lb, ub2, be)
    ref, stref, _ = alm(d.nv, d.nc, Hd, c, A, b, lb, ub, be)
    bad = np.array([1, 3, 5])
    good = np.setdiff1d(np.arange(8), bad)
    assert np.all(st[bad] == -2) and np.all(sol[bad] == 0.0)
    assert np.array_equal(st[good], stref[good]) and np.array_equal(sol[good], ref[good])
    sol1, st1, nw1 = alm(d.nv, d.nc, Hd, c, A, b, lb, ub, be, max_newton=1)
    assert np.all(nw1 <= 1) and np.any(st1 == -2)


def test_warm_start_realizable_two_control_steps(alm, oracle):
    """OSQP's warm_start = 1 between two solve() calls of one workspace (the reference's wrapper leaves it on,
    src/qpwrapper_osqp.cpp:68-69): the 38 x 29 problem of ASIFrealizable at a state and at the state a plant step later,
    the second solve started from the first one's iterate and multipliers.  Same verdicts, |u - u_ref| <= 1e-6, less
    than half the Newton steps of a cold solve."""
    k = oracle.load_kernel("100Hz")
    z = oracle.Realizable(k)
    B = 256
    x, u = oracle.make_batch_realizable(k, B)
    rng = np.random.default_rng(38)
    x2 = x + 0.002 * rng.normal(size=x.shape)
    keep = (z.assemble(x)[2] == 1) & (z.assemble(x2)[2] == 1)
    x, x2, u = x[keep], x2[keep], u[keep]
    n = len(x)

    def qps(x):
        A, b, code, info = z.assemble(x)
        Hd, c, lb, ub = (np.zeros((n, z.nv)) for _ in range(4))
        for i in range(n):
            Hd[i], c[i], lb[i], ub[i], be = z.qp_static(u[i])
        H = np.zeros((n, z.nv * z.nv))
        H[:, :: z.nv + 1] = Hd
        return H, c, A, b, lb, ub, be

    ua, rl, rc = z.filter(x2, u)
    warm = (np.full((n, z.nv), np.nan), np.full((n, z.nv + z.nc), np.nan))
    s1, st1, n1 = alm(z.nv, z.nc, *qps(x), diag=False, warm=warm)
    assert np.all(np.isfinite(warm[0])) and np.all(np.isfinite(warm[1]))
    assert np.array_equal(warm[0][st1 == 1], s1[st1 == 1]) and np.all(warm[0][st1 != 1] == 0.0)
    q2 = qps(x2)
    sw, stw, nw = alm(z.nv, z.nc, *q2, diag=False, warm=warm, warm_in=True)
    sc, stc, nc_ = alm(z.nv, z.nc, *q2, diag=False)
    assert np.array_equal(stw == 1, rc == 1) and np.array_equal(stc == 1, rc == 1)
    ok = rc == 1
    assert ok.sum() > 100
    assert np.abs(sw[ok, 0].clip(-20, 20) - ua[ok, 0]).max() <= U_TOL
    assert nw[ok].sum() < 0.5 * nc_[ok].sum()


def test_warm_start_robust_18x12(alm, oracle):
    """The lifted robust problem needs four Newton steps from a cold start; a warm one is no shorter on a randomly moved
    state (it is in the closed loops, where the active set stays) -- what must hold is the answer."""
    B = 1024
    d, Hd, c, A, b, lb, ub, be = _config_qps(oracle, 5, B)
    model, variant = oracle.CONFIGS[5]
    o = oracle.default_options(model, variant)
    x, u = oracle.make_batch(5, B)
    rng = np.random.default_rng(5)
    x2 = x + 0.005 * rng.normal(size=x.shape)
    A2, b2, code, _ = oracle.assemble_batch(model, variant, o, x2)
    ua, rl, rc = oracle.filter_batch(model, variant, o, x2, u, oracle.SOLVER_EXACT)
    H = np.zeros((B, d.nv * d.nv))
    H[:, :: d.nv + 1] = Hd
    warm = (np.zeros((B, d.nv)), np.zeros((B, d.nv + d.nc)))
    s1, st1, n1 = alm(d.nv, d.nc, H, c, A, b, lb, ub, be, diag=False, warm=warm)
    s0, st0, n0 = alm(d.nv, d.nc, H, c, A, b, lb, ub, be, diag=False)
    assert np.array_equal(s1, s0) and np.array_equal(n1, n0)  # writing the block changes nothing
    sw, stw, nw = alm(d.nv, d.nc, H, c, A2, b2, lb, ub, be, diag=False, warm=warm, warm_in=True)
    assert np.array_equal(stw, rc)
    assert np.abs(sw[:, 0].clip(o.lb[0], o.ub[0]) - ua[:, 0]).max() <= U_TOL
    assert np.abs(sw[:, 1] - rl[:, 0]).max() <= U_TOL
    # the unchanged problem: two multiplier updates, next to no Newton steps
    su, stu, nu = alm(d.nv, d.nc, H, c, A2, b2, lb, ub, be, diag=False, warm=warm, warm_in=True)
    assert np.array_equal(stu, rc) and nu.mean() <= 3 and np.abs(su[:, :2] - sw[:, :2]).max() <= U_TOL
