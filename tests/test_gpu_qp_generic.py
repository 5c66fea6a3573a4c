"""asif_hip_qp_solve_batch: the QPWrapperAbstract path for pre-assembled problems.

Shapes of the filter classes run on the in-register kernels (dual active-set stage, exact); every other shape
runs one wavefront per QP with the factor in LDS (qp_lds.hpp; tests/test_gpu_qp_lds.py holds its parity tests).
solver.polish = 0 selects the plain OSQP-style ADMM wave kernel (admm_wave.hpp), whose accuracy is set by eps.
Tolerances: 1e-6; status identical on every instance.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _solve(hip, Hd, c, A, b, lb, ub, be=None, **solver_kw):
    """AoS numpy in ([B,nv], [B,nc*nv] col-major, ...) -> SoA on the GPU and back."""
    B, nv = c.shape
    nc = b.shape[1]
    dev = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a.T)).to(dev)
    sol = torch.zeros((nv, B), dtype=torch.float64, device=dev)
    status = torch.zeros(B, dtype=torch.int32, device=dev)
    iters = torch.zeros(B, dtype=torch.int32, device=dev)
    hip.qp_solve_batch(t(Hd), t(c), t(A), t(b), t(lb), t(ub), sol, status, iters, be=be,
                       solver=hip.default_solver(**solver_kw))
    torch.cuda.synchronize()
    return sol.cpu().numpy().T, status.cpu().numpy(), iters.cpu().numpy()


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


@pytest.mark.parametrize("cfg,B", [(2, 4096), (3, 256), (4, 4096)])
def test_filter_shapes_in_register_kernel(hip, oracle, cfg, B):
    d, Hd, c, A, b, lb, ub, be = _config_qps(oracle, cfg, B)
    ex, stex, _ = oracle.qp_solve_batch(d.nv, d.nc, Hd, c, A, b, lb, ub, be, oracle.SOLVER_EXACT)
    sol, st, it = _solve(hip, Hd, c, A, b, lb, ub, be)
    assert np.array_equal(st == 1, stex == 1)
    assert np.all(st[stex != 1] == -3)  # primal infeasible, the raw OSQP-style status
    ok = st == 1
    assert np.abs(sol[ok] - ex[ok]).max() <= 1e-6


def test_same_problems_on_the_plain_admm_wave_kernel(hip, oracle):
    d, Hd, c, A, b, lb, ub, be = _config_qps(oracle, 2, 2048)
    ex, stex, _ = oracle.qp_solve_batch(d.nv, d.nc, Hd, c, A, b, lb, ub, be, oracle.SOLVER_EXACT)
    sol, st, it = _solve(hip, Hd, c, A, b, lb, ub, be, lanes_per_qp=64, polish=0)
    feas = stex == 1
    assert np.all(st[feas] == 1)
    assert np.abs(sol[feas] - ex[feas]).max() <= 1e-5
    # infeasible ones: certificate (-3) or, for the barely infeasible, iteration limit (-2); never "solved"
    assert np.all(np.isin(st[~feas], (-3, -2)))
    assert (st[~feas] == -3).mean() > 0.9


def test_robust_full_18x12_problem(hip, oracle):
    """The QP the reference hands to OSQP for C5, multipliers included (nv = 18, nc = 12, 8 equality rows,
    H zero on the multipliers): wave-per-QP LDS kernel vs the exact (u, delta), every instance."""
    d, Hd, c, A, b, lb, ub, be = _config_qps(oracle, 5, 512)
    assert (d.nv, d.nc) == (18, 12) and be.sum() == 8
    model, variant = oracle.CONFIGS[5]
    o = oracle.default_options(model, variant)
    x, u = oracle.make_batch(5, 512)
    ua, rl, rc = oracle.filter_batch(model, variant, o, x, u, oracle.SOLVER_EXACT)
    sol, st, it = _solve(hip, Hd, c, A, b, lb, ub, be)
    assert np.all(rc == 1)
    assert np.all(st == 1)
    assert np.abs(sol[:, 0] - ua[:, 0]).max() <= 1e-6
    assert np.abs(sol[:, 1] - rl[:, 0]).max() <= 1e-6
    assert sol[:, 2:].min() >= -1e-7  # multipliers stay in their cone


def test_random_medium_qps_against_oracle_admm(hip, oracle):
    rng = np.random.default_rng(7)
    B, nv, nc = 256, 5, 7
    Hd = rng.uniform(0.5, 3.0, (B, nv))
    c = rng.normal(0, 2, (B, nv))
    A = rng.normal(0, 1, (B, nc * nv))
    x0 = rng.normal(0, 1, (B, nv))
    Am = A.reshape(B, nv, nc).transpose(0, 2, 1)
    b = np.einsum("brv,bv->br", Am, x0) - rng.uniform(0, 1, (B, nc))  # x0 strictly feasible ...
    be = np.zeros(nc, dtype=np.uint8)
    be[0] = 1
    b[:, 0] = np.einsum("bv,bv->b", Am[:, 0], x0)                      # ... and on the equality row
    lb = x0 - rng.uniform(0.1, 2, (B, nv))
    ub = x0 + rng.uniform(0.1, 2, (B, nv))
    b[:16, 1] += 100.0  # make the first 16 infeasible: row 1 cannot be met inside the box
    s = oracle.admm_settings(eps_abs=1e-10, eps_rel=1e-10, max_iter=100000)
    ref, stref, _ = oracle.qp_solve_batch(nv, nc, Hd, c, A, b, lb, ub, be, oracle.SOLVER_ADMM, s)
    sol, st, it = _solve(hip, Hd, c, A, b, lb, ub, be)
    assert np.all(stref[:16] != 1) and np.all(st[:16] == -3)
    assert np.all(st[16:] == 1)
    ok = (stref == 1) & (st == 1)
    assert np.abs(sol[ok] - ref[ok]).max() <= 1e-6
    # feasibility of what came back
    res = np.einsum("brv,bv->br", Am, sol) - b
    assert res[ok][:, 1:].min() >= -1e-5 and np.abs(res[ok][:, 0]).max() <= 1e-5


def test_unsupported_shape_is_an_error(hip):
    dev = torch.device("cuda:0")
    nv, nc, B = 140, 4, 8
    z = lambda r: torch.zeros((r, B), dtype=torch.float64, device=dev)
    with pytest.raises(hip.AsifHipError):
        hip.qp_solve_batch(z(nv) + 1, z(nv), z(nc * nv), z(nc), z(nv) - 1, z(nv) + 1, z(nv),
                           torch.zeros(B, dtype=torch.int32, device=dev))


@pytest.mark.parametrize("nv,nc,pins,eqrow,bmean", [(2, 4, (1,), False, -1.0), (2, 4, (0,), False, -1.0),
                                                     (3, 41, (1, 2), False, -2.5), (3, 41, (2,), False, -2.0),
                                                     (2, 18, (1,), True, -3.0), (3, 41, (0, 1), True, -4.0)])
def test_pinned_variables_are_eliminated(hip, oracle, nv, nc, pins, eqrow, bmean):
    """Variables pinned by their bounds in every QP of the batch: the dual active-set stage eliminates them (wave-
    uniformly) and, when one variable is left, solves the problem as its one-variable fixed point (gi_small.hpp).
    Random problems, some infeasible, rows with a zero coefficient on the free variable, optionally an equality row;
    verdict and optimum against the exact enumeration."""
    rng = np.random.default_rng(100 * nv + nc + 7 * len(pins) + (1 if eqrow else 0))
    B = 4096
    Hd = np.exp(rng.uniform(np.log(0.5), np.log(50), (B, nv)))
    c = rng.normal(0, 3, (B, nv))
    A = rng.normal(0, 1, (B, nv, nc))
    A[rng.random((B, nv, nc)) < 0.1] = 0.0
    b = rng.normal(bmean, 1, (B, nc))  # bmean tuned per shape for a mix of feasible and infeasible problems
    lb = -rng.uniform(0.5, 3, (B, nv))
    ub = rng.uniform(0.5, 3, (B, nv))
    for j in pins:
        lb[:, j] = ub[:, j] = rng.uniform(-1, 1, B)
    be = np.zeros(nc, dtype=np.uint8)
    if eqrow:
        be[1] = 1
        free = [j for j in range(nv) if j not in pins]
        A[:, free[0], 1] = rng.choice([-1.0, 1.0], B) * rng.uniform(0.5, 2, B)  # keep the equality row solvable
    Af = np.ascontiguousarray(A.reshape(B, nv * nc))
    ex, stex, _ = oracle.qp_solve_batch(nv, nc, Hd, c, Af, b, lb, ub, be, oracle.SOLVER_EXACT)
    sol, st, it = _solve(hip, Hd, c, Af, b, lb, ub, be)
    assert 0.02 < (stex == 1).mean() < 0.98, "the family should mix feasible and infeasible problems"
    assert np.array_equal(st == 1, stex == 1), f"{((st == 1) != (stex == 1)).sum()} verdicts differ"
    ok = st == 1
    assert np.abs(sol[ok] - ex[ok]).max() <= 1e-6
    for j in pins:
        assert np.array_equal(sol[ok][:, j], lb[ok][:, j])
