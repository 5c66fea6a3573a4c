"""The product's in-register dual active-set solver (asif_amd/csrc/gi_small.hpp) compiled for the HOST with one
lane per QP (tests/host_gi_driver.cpp) against the oracle's exact enumeration: same verdict on every instance of
the configs' QPs and of randomised stress families (nearly parallel rows, pinned variables, equality rows, zero
rows, degenerate vertices, crossed bounds), optimum to 1e-9.  CPU only; the device runs the same source."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from test_oracle_qp import _config_qps

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def gi(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("gi") / "libgi_host.so")
    if os.environ.get("ASIF_SAN_DIR"):  # tests/test_sanitizers.py: the build of `make -C tests san`
        so = os.path.join(os.environ["ASIF_SAN_DIR"], "libgi_host_san.so")
    else:
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
                               "-I" + os.path.join(ROOT, "asif_amd", "csrc"),
                               os.path.join(ROOT, "tests", "host_gi_driver.cpp"), "-o", so])
    lib = C.CDLL(so)

    def solve(nv, nc, Hd, c, A, b, lb, ub, be, max_steps=28):
        B = c.shape[0]
        arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in (Hd, c, A, b, lb, ub)]
        sol = np.zeros((B, nv))
        st = np.zeros(B, dtype=np.int32)
        steps = np.zeros(B, dtype=np.int32)
        bep = np.ascontiguousarray(be, dtype=np.uint8).ctypes.data_as(C.POINTER(C.c_uint8)) if be is not None else None
        r = lib.gi_host_solve_batch(nv, nc, C.c_int64(B), *[a.ctypes.data_as(C.POINTER(C.c_double)) for a in arrs],
                                    bep, max_steps, sol.ctypes.data_as(C.POINTER(C.c_double)),
                                    st.ctypes.data_as(C.POINTER(C.c_int32)), steps.ctypes.data_as(C.POINTER(C.c_int32)))
        assert r == 0
        return sol, st, steps
    return solve


@pytest.mark.parametrize("cfg,B", [(2, 16384), (3, 1024), (4, 8192), (9, 2048)])
def test_config_qps_decided_like_the_exact_solver(oracle, gi, cfg, B):
    d, Hd, c, A, b, lb, ub, be = _config_qps(oracle, cfg, B)
    ex, stex, _ = oracle.qp_solve_batch(d.nv, d.nc, Hd, c, A, b, lb, ub, be, oracle.SOLVER_EXACT)
    sol, st, steps = gi(d.nv, d.nc, Hd, c, A, b, lb, ub, be)
    assert (st == 0).sum() == 0, "undecided instances on a seeded workload"
    assert np.array_equal(st == 1, stex == 1)
    assert np.array_equal(st == 2, stex != 1)
    ok = st == 1
    assert np.abs(sol[ok] - ex[ok]).max() <= 1e-12
    assert steps.max() <= 4


def _family(rng, nv, nc, B, kind):
    Hd = np.exp(rng.uniform(np.log(0.5), np.log(100), (B, nv)))
    c = rng.normal(0, 3, (B, nv))
    A = rng.normal(0, 1, (B, nv, nc))
    b = rng.normal(-0.5, 1, (B, nc))
    lb = -rng.uniform(0.5, 3, (B, nv))
    ub = rng.uniform(0.5, 3, (B, nv))
    be = np.zeros(nc, dtype=np.uint8)
    if kind == "parallel":
        eps = 10.0 ** rng.uniform(-7, -1, (B, 1, 1))
        A[:, :, 1:2] = A[:, :, 0:1] * rng.uniform(0.5, 2, (B, 1, 1)) + eps * rng.normal(0, 1, (B, nv, 1))
        A[:, :, 3:4] = -A[:, :, 2:3] + eps * rng.normal(0, 1, (B, nv, 1))
    elif kind == "pinned":
        ub[:, -1] = lb[:, -1] = rng.uniform(1, 6, B)
    elif kind == "eqrow":
        be[0] = 1
    elif kind == "zero":
        A[:, :, 0] = 0.0
        b[:, 0] = rng.choice([-1e20, -1.0, 1.0], B)
        A[:, 0, 1] = 0.0
    elif kind == "degenerate":
        p = rng.normal(0, 1, (B, nv))
        b = np.einsum("bjr,bj->br", A, p)
    elif kind == "crossed":
        lb[:, 0], ub[:, 0] = ub[:, 0].copy(), lb[:, 0].copy()
    return Hd, c, A.reshape(B, nv * nc), b, lb, ub, be


@pytest.mark.parametrize("nv,nc", [(2, 4), (2, 18), (3, 17), (3, 41)])
@pytest.mark.parametrize("kind", ["plain", "parallel", "pinned", "eqrow", "zero", "degenerate", "crossed"])
def test_stress_families(oracle, gi, nv, nc, kind):
    import zlib
    rng = np.random.default_rng(zlib.crc32(f"{nv}x{nc}:{kind}".encode()))  # str hashes change from run to run
    B = 4000
    Hd, c, A, b, lb, ub, be = _family(rng, nv, nc, B, kind)
    ex, stex, _ = oracle.qp_solve_batch(nv, nc, Hd, c, A, b, lb, ub, be, oracle.SOLVER_EXACT)
    sol, st, steps = gi(nv, nc, Hd, c, A, b, lb, ub, be)
    und = st == 0
    # "undecided" hands the problem to the ADMM iterations; it must stay the exception and never be a wrong verdict
    assert und.mean() <= 0.03, und.mean()
    diff = np.where(((st == 1) != (stex == 1)) & ~und)[0]
    # Rows built to pass through ONE point leave problems whose feasible set is that point within rounding: the
    # enumeration (1e-12 relative, long double) and this solver (1e-12 on the selection, 1e-9 at a vertex its working
    # set pins) may then disagree about "feasible".  Such a verdict is accepted only with a point that meets every row.
    for k in diff:
        assert kind == "degenerate" and st[k] == 1, (kind, k, st[k], stex[k])
        Am = A[k].reshape(nv, nc).T
        assert (b[k] - Am @ sol[k]).max() <= 1e-9 * (1 + np.abs(b[k]).max())
        assert np.all(sol[k] >= lb[k] - 1e-9) and np.all(sol[k] <= ub[k] + 1e-9)
    assert len(diff) <= 2
    ok = (st == 1) & (stex == 1)
    if ok.any():
        assert (np.abs(sol[ok] - ex[ok]) / (1 + np.abs(ex[ok]))).max() <= 1e-9


def test_cost_without_curvature_is_left_to_the_iterations(gi):
    Hd = np.array([[1.0, 0.0]])
    z = np.zeros((1, 2))
    sol, st, steps = gi(2, 4, Hd, z, np.zeros((1, 8)), np.full((1, 4), -1e20), z - 1, z + 1, None)
    assert st[0] == 0 and steps[0] == 0


@pytest.mark.parametrize("nv,nc", [(1, 8), (2, 4), (3, 17)])
@pytest.mark.parametrize("poison", [np.nan, np.inf, -np.inf])
def test_nonfinite_data_is_a_failure_never_a_solution(oracle, gi, nv, nc, poison):
    """A NaN / inf state reaches the solver as NaN / inf rows.  The reference's OSQP never converges on such data and
    returns max_iter (filter(): rc -1, uAct untouched); comparisons written the positive way read a NaN row as met and
    would hand back clip(uDes) as "safe" (ADVICE r2).  Verdict 3 = failed on every poisoned instance, whatever entry
    holds the poison; the oracle's stand-in says max_iter (-2) for the same problems; clean instances are unaffected."""
    rng = np.random.default_rng(7)
    B = 600
    Hd, c, A, b, lb, ub, be = _family(rng, nv, nc, B, "plain")
    if nv >= 2:
        ub[:, -1] = lb[:, -1] = 5.0  # the explicit class's pinned relaxation variable: elimination path
    clean = np.arange(B) % 3 == 0
    for k in np.where(~clean)[0]:
        where = rng.integers(0, 4 if poison != poison else 3)
        if where == 0:
            A[k, rng.integers(0, nv * nc)] = poison
        elif where == 1:
            b[k, rng.integers(0, nc)] = poison
        elif where == 2:
            c[k, rng.integers(0, nv)] = poison
        else:
            (lb if rng.integers(0, 2) else ub)[k, rng.integers(0, nv)] = np.nan
    sol, st, _ = gi(nv, nc, Hd, c, A, b, lb, ub, be)
    assert np.all(st[~clean] == 3)
    assert np.all(st[clean] != 3)
    _, stex, _ = oracle.qp_solve_batch(nv, nc, Hd, c, A, b, lb, ub, be, oracle.SOLVER_EXACT)
    assert np.all(stex[~clean] == -2)
    assert np.array_equal(st[clean] == 1, stex[clean] == 1)


def test_all_nan_rows_of_the_explicit_filter(gi):
    """ADVICE r2's reproduction: every row NaN (the state was NaN), delta pinned -> was verdict 1 with x = clip(uDes)."""
    Hd = np.array([[1.0, 10.0]])
    c = np.array([[-2.0 * 0.3, -2.0 * 10.0 * 5.0]])
    A = np.full((1, 8), np.nan)
    b = np.full((1, 4), np.nan)
    sol, st, _ = gi(2, 4, Hd, c, A, b, np.array([[-1.0, 5.0]]), np.array([[1.0, 5.0]]), None)
    assert st[0] == 3
    A = np.zeros((1, 8)); A[0, 2] = np.nan  # a single NaN entry
    sol, st, _ = gi(2, 4, Hd, c, A, np.full((1, 4), -1.0), np.array([[-1.0, 5.0]]), np.array([[1.0, 5.0]]), None)
    assert st[0] == 3


def test_shipped_half_plane_problems_are_all_decided(oracle, gi):
    """ASIFrobust on the shipped 100 half-planes (examples/DoubleIntegrator_Robust.cpp, 5 kept per call): after the
    exact elimination of the multipliers the QP has rows [lo/hi(Lgh), h] (u, delta) >= -lo(Lfh) whose Lgh entries
    are affine-arithmetic noise of 1e-9 ... 1e-14 next to h ~ -0.09 -- pairs of rows, and rows against the bound on
    delta, meet at angles of 1e-8 ... 1e-12.  Round 2's stage declined 76 of 8 192 such instances (the normal-equations
    matrix of two such normals is singular to rounding) and their waves fell through to ADMM + finish; with the square
    system solved on the normals themselves and the codimension-one direction in closed form it decides every one,
    verdict equal to the exact enumeration's."""
    hp = oracle.load_halfplanes()
    z = oracle.RobustData(hp)
    B = 8192
    x, u = oracle.make_batch_robust_data(hp, B)
    A, b, code, sel = z.assemble(x)
    M, nc, nv = z.nc // 3, z.nc, z.nv
    A3 = A.reshape(B, nv, nc)
    nr = 2 * M
    A2, b2 = np.zeros((B, 2, nr)), np.zeros((B, nr))
    for s in range(M):  # the two plain rows per safety function (oracle/or_filter.c: robust_exact)
        iRow, iCol = 3 * s, 2 + 4 * s
        for p, col, sign in ((0, iCol, 1.0), (1, iCol + 2, -1.0)):
            A2[:, 0, 2 * s + p] = sign * A3[:, col, iRow]
            A2[:, 1, 2 * s + p] = A3[:, 1, iRow]
            b2[:, 2 * s + p] = -A3[:, iCol + 1, iRow]
    Hd, c, lb, ub = (np.zeros((B, 2)) for _ in range(4))
    for i in range(B):
        H_, c_, lb_, ub_, _ = z.qp_static(u[i])
        Hd[i], c[i], lb[i], ub[i] = H_[:2], c_[:2], lb_[:2], ub_[:2]
    Af = np.ascontiguousarray(A2.reshape(B, 2 * nr))
    ex, stex, _ = oracle.qp_solve_batch(2, nr, Hd, c, Af, b2, lb, ub, None, oracle.SOLVER_EXACT)
    sol, st, steps = gi(2, nr, Hd, c, Af, b2, lb, ub, None)
    assert (st == 0).sum() == 0
    assert np.array_equal(st == 1, stex == 1) and np.array_equal(st == 2, stex != 1)
    assert (stex != 1).sum() > 1000
    ok = st == 1
    assert np.abs(sol[ok] - ex[ok]).max() <= 1e-11


def test_conflict_threshold_is_relative_1e_9(oracle, gi):
    """Where "infeasible" starts (the one soak disagreement): a row that the bounds contradict by eps.
    The dual active-set stage calls a conflict at a vertex infeasible beyond kActiveTol = 1e-9 relative to the row's
    own terms (gi_small.hpp); the oracle's exact enumeration counts a row violated beyond 1e-12.  Above both thresholds
    the verdicts agree; between them -- problems infeasible by 1e-12 ... 1e-9, which OSQP at its default tolerance
    calls solved -- the stage returns the optimum of the problem with that row met to rounding."""
    nv, nc = 2, 4
    eps = np.array([1e-3, 1e-6, 1e-8, 3e-10, 1e-11, 0.0, -1e-6])
    B = len(eps)
    Hd = np.tile([1.0, 50.0], (B, 1))
    c = np.tile([-2.0 * 0.3, -500.0], (B, 1))
    A = np.zeros((B, nv, nc))
    A[:, 0, 0] = 1.0                      # row 0: u >= 1 + eps against u <= 1
    b = np.full((B, nc), -1e20)
    b[:, 0] = 1.0 + eps
    lb = np.tile([-1.0, 5.0], (B, 1))
    ub = np.tile([1.0, 5.0], (B, 1))       # the explicit class's pinned relaxation variable
    sol, st, _ = gi(nv, nc, Hd, c, A.reshape(B, -1), b, lb, ub, None)
    ex, stex, _ = oracle.qp_solve_batch(nv, nc, Hd, c, A.reshape(B, -1), b, lb, ub, None, oracle.SOLVER_EXACT)
    assert list(st) == [2, 2, 2, 1, 1, 1, 1]          # infeasible down to 1e-8, solved from 3e-10 on
    assert list(stex == 1) == [False, False, False, False, False, True, True]  # the oracle: infeasible down to 1e-11
    agree = (eps > 2e-9) | (eps < 1e-12)
    assert np.array_equal((st == 1)[agree], (stex == 1)[agree])
    assert np.abs(sol[3:6, 0] - 1.0).max() <= 1e-9 and abs(sol[6, 0] - (1.0 - 1e-6)) <= 1e-12
