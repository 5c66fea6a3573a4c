"""The wave-per-QP LDS kernel (asif_amd/csrc/qp_lds.hpp) behind asif_hip_qp_solve_batch(_dense): the QPs the
reference's own classes hand to their solver, multipliers and all -- ASIFrobust 18 x 12 (C5) and 22 x 15 (shipped
half-planes), ASIFrealizable 38 x 29 / 62 x 47 / 86 x 65 -- plus full cost matrices (diagonalCost = false,
src/qpwrapper_osqp.cpp:276-309), against the oracle's exact (u, delta).  Bar: status identical on EVERY instance,
|u - u_ref| <= 1e-6 (north star 1e-5)."""
import numpy as np
import pytest
import torch

from test_gpu_qp_generic import _config_qps, _solve

pytestmark = pytest.mark.gpu

U_TOL = 1e-6


def _solve_dense(hip, H, c, A, b, lb, ub, be=None, **solver_kw):
    B, nv = c.shape
    dev = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a.T)).to(dev)
    sol = torch.zeros((nv, B), dtype=torch.float64, device=dev)
    status = torch.zeros(B, dtype=torch.int32, device=dev)
    iters = torch.zeros(B, dtype=torch.int32, device=dev)
    hip.qp_solve_batch_dense(t(H), t(c), t(A), t(b), t(lb), t(ub), sol, status, iters, be=be,
                             solver=hip.default_solver(**solver_kw))
    torch.cuda.synchronize()
    return sol.cpu().numpy().T, status.cpu().numpy(), iters.cpu().numpy()


def test_robust_full_18x12_every_instance(hip, oracle):
    B = 8192
    d, Hd, c, A, b, lb, ub, be = _config_qps(oracle, 5, B)
    assert (d.nv, d.nc) == (18, 12) and be.sum() == 8
    model, variant = oracle.CONFIGS[5]
    o = oracle.default_options(model, variant)
    x, u = oracle.make_batch(5, B)
    ua, rl, rc = oracle.filter_batch(model, variant, o, x, u, oracle.SOLVER_EXACT)
    sol, st, it = _solve(hip, Hd, c, A, b, lb, ub, be)
    assert np.all(rc == 1) and np.array_equal(st, rc)
    assert np.abs(sol[:, 0].clip(o.lb[0], o.ub[0]) - ua[:, 0]).max() <= U_TOL
    assert np.abs(sol[:, 1] - rl[:, 0]).max() <= U_TOL
    assert sol[:, 2:].min() >= -1e-7  # multipliers stay in their cone
    assert it.max() < 400


def test_two_waves_per_simd_instantiation_gives_the_same_bits(hip, oracle):
    """From 16 384 problems on the half-wave kernel is launched with its register allocation held to two waves per
    SIMD (k_qp.hip: kInvTwoWavesMin) -- other code, with spills: same solution, status and Newton count, bit for
    bit, as the one-wave build gives the same problems in a batch of 2 048.  On the 22 x 15 problems of the shipped
    half-planes (the 18 x 12 instantiation fits two waves as it is and has no second build)."""
    hp = oracle.load_halfplanes()
    z = oracle.RobustData(hp)
    B = 2048
    x, u = oracle.make_batch_robust_data(hp, B)
    A, b, code, sel = z.assemble(x)
    Hd, c, lb, ub = (np.zeros((B, z.nv)) for _ in range(4))
    for i in range(B):
        Hd[i], c[i], lb[i], ub[i], be = z.qp_static(u[i])
    sol1, st1, it1 = _solve(hip, Hd, c, A, b, lb, ub, be)
    rep = lambda a: np.tile(a, (8, 1))
    sol8, st8, it8 = _solve(hip, rep(Hd), rep(c), rep(A), rep(b), rep(lb), rep(ub), be)
    assert len(st8) == 16384 and (st1 == 1).sum() > 1000 and (st1 != 1).sum() > 100
    for k in range(8):
        blk = slice(B * k, B * (k + 1))
        assert np.array_equal(st8[blk], st1) and np.array_equal(it8[blk], it1)
        assert np.array_equal(sol8[blk], sol1)


def test_large_batch_of_18x12_equals_small_batches(hip, oracle):
    d, Hd, c, A, b, lb, ub, be = _config_qps(oracle, 5, 4096)
    sol1, st1, it1 = _solve(hip, Hd, c, A, b, lb, ub, be)
    rep = lambda a: np.tile(a, (4, 1))
    sol4, st4, it4 = _solve(hip, rep(Hd), rep(c), rep(A), rep(b), rep(lb), rep(ub), be)
    assert len(st4) == 16384 and np.all(st1 == 1)
    for k in range(4):
        blk = slice(4096 * k, 4096 * (k + 1))
        assert np.array_equal(st4[blk], st1) and np.array_equal(it4[blk], it1)
        assert np.array_equal(sol4[blk], sol1)


@pytest.mark.parametrize("nv,nc", [(12, 20), (19, 30), (23, 12), (23, 28), (30, 14), (31, 32), (36, 50), (50, 60), (64, 64)])
def test_every_padded_instantiation_of_the_inverse_kernel(hip, oracle, nv, nc):
    """One random family per compiled size of qp_inv.hpp -- <20,32>, <24,16>, <24,32>, <32,16>, <32,32> on half a wave,
    <40,64>, <64,64> on a whole one (18 x 12, 22 x 15, 38 x 29 and 62 x 47 above cover the others): a strictly feasible
    point by construction, one equality row, the first eight problems made infeasible; against the oracle's ADMM run
    to 1e-10."""
    rng = np.random.default_rng(1000 * nv + nc)
    B = 48
    Hd = rng.uniform(0.5, 3.0, (B, nv))
    c = rng.normal(0, 2, (B, nv))
    A = rng.normal(0, 1, (B, nc * nv))
    x0 = rng.normal(0, 1, (B, nv))
    Am = A.reshape(B, nv, nc).transpose(0, 2, 1)
    b = np.einsum("brv,bv->br", Am, x0) - rng.uniform(0, 1, (B, nc))
    be = np.zeros(nc, dtype=np.uint8)
    be[0] = 1
    b[:, 0] = np.einsum("bv,bv->b", Am[:, 0], x0)
    lb = x0 - rng.uniform(0.1, 2, (B, nv))
    ub = x0 + rng.uniform(0.1, 2, (B, nv))
    b[:8, 1] += 1000.0  # row 1 cannot be met inside the box
    s = oracle.admm_settings(eps_abs=1e-10, eps_rel=1e-10, max_iter=200000)
    ref, stref, _ = oracle.qp_solve_batch(nv, nc, Hd, c, A, b, lb, ub, be, oracle.SOLVER_ADMM, s)
    sol, st, it = _solve(hip, Hd, c, A, b, lb, ub, be)
    assert np.all(stref[:8] != 1) and np.all(st[:8] == -3)
    assert np.all(st[8:] == 1) and np.all(stref[8:] == 1)
    assert np.abs(sol[8:] - ref[8:]).max() <= U_TOL
    res = np.einsum("brv,bv->br", Am, sol) - b
    assert res[8:, 1:].min() >= -1e-6 and np.abs(res[8:, 0]).max() <= 1e-6


def test_robust_data_22x15_every_status(hip, oracle):
    hp = oracle.load_halfplanes()
    z = oracle.RobustData(hp)
    B = 2048
    x, u = oracle.make_batch_robust_data(hp, B)
    ua, rl, rc = z.filter(x, u)
    A, b, code, sel = z.assemble(x)
    Hd, c, lb, ub = (np.zeros((B, z.nv)) for _ in range(4))
    for i in range(B):
        Hd[i], c[i], lb[i], ub[i], be = z.qp_static(u[i])
    sol, st, it = _solve(hip, Hd, c, A, b, lb, ub, be)
    assert (rc == -1).sum() > 100
    assert np.array_equal(st == 1, rc == 1), f"{((st == 1) != (rc == 1)).sum()} status mismatches"
    assert np.all(st[rc != 1] == -3)  # primal infeasible, the raw OSQP-style code
    ok = rc == 1
    assert np.abs(sol[ok, 0].clip(-20, 20) - ua[ok, 0]).max() <= U_TOL
    assert np.abs(sol[ok, 1] - rl[ok, 0]).max() <= U_TOL


@pytest.mark.parametrize("kernel,shape", [("100Hz", (38, 29)), ("10Hz_50pt", (62, 47)), ("10Hz", (86, 65))])
def test_realizable_full_problem(hip, oracle, kernel, shape):
    """What ASIFrealizable::filter hands to QPsolver_ (src/asif_realizable.cpp:300-340): the full lifted problem."""
    k = oracle.load_kernel(kernel)
    z = oracle.Realizable(k)
    assert (z.nv, z.nc) == shape
    B = 768
    x, u = oracle.make_batch_realizable(k, B)
    ua, rl, rc = z.filter(x, u)
    A, b, code, info = z.assemble(x)
    keep = code == 1
    x, u, ua, rc, A, b = x[keep], u[keep], ua[keep], rc[keep], A[keep], b[keep]
    n = len(x)
    Hd, c, lb, ub = (np.zeros((n, z.nv)) for _ in range(4))
    for i in range(n):
        Hd[i], c[i], lb[i], ub[i], be = z.qp_static(u[i])
    sol, st, it = _solve(hip, Hd, c, A, b, lb, ub, be)
    assert np.array_equal(st == 1, rc == 1), f"{((st == 1) != (rc == 1)).sum()} of {n} status mismatches"
    ok = rc == 1
    assert ok.sum() > 100
    assert np.abs(sol[ok, 0].clip(-20, 20) - ua[ok, 0]).max() <= U_TOL


def test_filter_shapes_on_the_lds_kernel(hip, oracle):
    for cfg, B in ((2, 4096), (4, 2048), (3, 128)):
        d, Hd, c, A, b, lb, ub, be = _config_qps(oracle, cfg, B)
        ex, stex, _ = oracle.qp_solve_batch(d.nv, d.nc, Hd, c, A, b, lb, ub, be, oracle.SOLVER_EXACT)
        sol, st, it = _solve(hip, Hd, c, A, b, lb, ub, be, lanes_per_qp=64)
        assert np.array_equal(st == 1, stex == 1), cfg
        assert np.all(st[stex != 1] == -3), cfg
        ok = st == 1
        assert np.abs(sol[ok, 0] - ex[ok, 0]).max() <= U_TOL, cfg


def _random_dense(rng, B, nv, nc):
    M = rng.normal(0, 1, (B, nv, nv))
    H = np.einsum("bij,bkj->bik", M, M) / nv + 0.2 * np.eye(nv)
    c = rng.normal(0, 2, (B, nv))
    A = rng.normal(0, 1, (B, nv, nc))
    x0 = rng.normal(0, 1, (B, nv))
    b = np.einsum("bjr,bj->br", A, x0) - rng.uniform(0, 1, (B, nc))
    lb = x0 - rng.uniform(0.1, 2, (B, nv))
    ub = x0 + rng.uniform(0.1, 2, (B, nv))
    return H, c, A, b, lb, ub


@pytest.mark.parametrize("nv,nc", [(6, 9), (20, 30), (70, 70)])
def test_full_cost_matrix_kkt(hip, nv, nc):
    """diagonalCost = false: optimality checked directly -- feasibility, and stationarity 2Hx + c = A'mu + nu with
    mu >= 0 on the active rows (non-negative least squares)."""
    from scipy.optimize import nnls
    rng = np.random.default_rng(nv * 100 + nc)
    B = 64
    H, c, A, b, lb, ub = _random_dense(rng, B, nv, nc)
    Hcm = H.transpose(0, 2, 1).reshape(B, nv * nv)  # column-major per instance (H symmetric anyway)
    sol, st, it = _solve_dense(hip, Hcm, c, A.reshape(B, nv * nc), b, lb, ub)
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


def test_dense_entry_reads_the_upper_triangle_and_equals_the_diagonal_entry(hip, oracle):
    d, Hd, c, A, b, lb, ub, be = _config_qps(oracle, 5, 256)
    B, nv = c.shape
    H = np.zeros((B, nv, nv))
    H[:, np.arange(nv), np.arange(nv)] = Hd
    H += np.tril(np.full((nv, nv), 123.0), -1)  # garbage below the diagonal must not be read (col-major: i > j)
    Hcm = H.transpose(0, 2, 1).reshape(B, nv * nv)
    s1, st1, _ = _solve(hip, Hd, c, A, b, lb, ub, be)
    s2, st2, _ = _solve_dense(hip, Hcm, c, A, b, lb, ub, be)
    assert np.array_equal(st1, st2)
    assert np.abs(s1[:, :2] - s2[:, :2]).max() <= 1e-9


def test_shape_limits(hip):
    dev = torch.device("cuda:0")
    B = 4
    z = lambda r: torch.zeros((r, B), dtype=torch.float64, device=dev)
    st = torch.zeros(B, dtype=torch.int32, device=dev)
    with pytest.raises(hip.AsifHipError):  # beyond 128 variables
        hip.qp_solve_batch(z(130) + 1, z(130), z(130 * 4), z(4), z(130) - 1, z(130) + 1, z(130), st)
    with pytest.raises(hip.AsifHipError):  # 128 x 100: 235 KB of LDS
        hip.qp_solve_batch(z(128) + 1, z(128), z(128 * 100), z(100), z(128) - 1, z(128) + 1, z(128), st)
    # 100 x 60 with a diagonal cost fits in 131 KB (box-constrained least squares: the answer is the clipped centre)
    nv, nc = 100, 60
    c = torch.full((nv, B), -4.0, dtype=torch.float64, device=dev)
    sol = z(nv)
    hip.qp_solve_batch(z(nv) + 1, c, z(nc * nv), z(nc) - 1e20, z(nv) - 1, z(nv) + 1, sol, st)
    torch.cuda.synchronize()
    assert np.all(st.cpu().numpy() == 1) and np.abs(sol.cpu().numpy() - 1.0).max() <= 1e-9


def test_plain_admm_wave_kernel_on_the_lifted_robust_problem(hip, oracle):
    """polish = 0, lanes_per_qp = 64: the north star's literal layout (OSQP-style ADMM, one wavefront per QP, iterates
    and factor in LDS) on the shape it exists for, the lifted 18 x 12 robust QP.  Round 2 left ~0.5 % of the instances
    at max_iter with a solution 6e-3 off and status -2 where the oracle says 1; what the iterations leave undecided now
    goes through the exact LDS method in a second launch.  Status equal on every instance; |u - u_ref| <= 1e-5 with the
    iterations' tolerances at 1e-9 (without a finish step their accuracy is what eps buys: 1.2e-5 at the default 1e-8)."""
    B = 2048
    d, Hd, c, A, b, lb, ub, be = _config_qps(oracle, 5, B)
    model, variant = oracle.CONFIGS[5]
    o = oracle.default_options(model, variant)
    x, u = oracle.make_batch(5, B)
    ua, rl, rc = oracle.filter_batch(model, variant, o, x, u, oracle.SOLVER_EXACT)
    sol, st, it = _solve(hip, Hd, c, A, b, lb, ub, be, lanes_per_qp=64, polish=0, eps_abs=1e-9, eps_rel=1e-9)
    assert np.array_equal(st, rc), f"{(st != rc).sum()} status mismatches"
    assert np.abs(sol[:, 0].clip(o.lb[0], o.ub[0]) - ua[:, 0]).max() <= 1e-5
    assert np.abs(sol[:, 1] - rl[:, 0]).max() <= 1e-5
    assert it.max() >= 1000 and np.median(it) <= 400  # the iterations did run; a few instances used the whole budget


def test_plain_admm_wave_kernel_on_shipped_half_planes_22x15(hip, oracle):
    """Same mode on the 22 x 15 problems of DoubleIntegrator_Robust (a quarter infeasible): feasible <-> 1 on every
    instance; an infeasible one ends with the certificate (-3) from either pass, never 'solved'."""
    hp = oracle.load_halfplanes()
    z = oracle.RobustData(hp)
    B = 2048
    x, u = oracle.make_batch_robust_data(hp, B)
    ua, rl, rc = z.filter(x, u)
    A, b, code, sel = z.assemble(x)
    Hd, c, lb, ub = (np.zeros((B, z.nv)) for _ in range(4))
    for i in range(B):
        Hd[i], c[i], lb[i], ub[i], be = z.qp_static(u[i])
    sol, st, it = _solve(hip, Hd, c, A, b, lb, ub, be, lanes_per_qp=64, polish=0, eps_abs=1e-9, eps_rel=1e-9)
    assert np.array_equal(st == 1, rc == 1), f"{((st == 1) != (rc == 1)).sum()} status mismatches"
    assert np.all(st[rc != 1] == -3)
    ok = rc == 1
    assert (np.abs(sol[ok, 0] - ua[ok, 0]) / np.maximum(1.0, np.abs(ua[ok, 0]))).max() <= 1e-5  # |u| up to 20 here


@pytest.mark.parametrize("B", [1, 2, 777])
def test_two_problems_per_wave_do_not_see_each_other(hip, oracle, B):
    """qp_inv.hpp pairs problems 2k and 2k+1 in one wavefront (an odd batch leaves the last half-wave idle): the answer
    of a problem must not depend on its partner -- the first B of the 18 x 12 problems solved alone equal, bit for bit,
    the same problems inside a batch of 2 048, with a different partner for every odd index when B is odd."""
    d, Hd, c, A, b, lb, ub, be = _config_qps(oracle, 5, 2048)
    full = _solve(hip, Hd, c, A, b, lb, ub, be)
    part = _solve(hip, Hd[:B], c[:B], A[:B], b[:B], lb[:B], ub[:B], be)
    for x, y in zip(full, part):
        assert np.array_equal(x[:B], y)
    # and shifted by one: every problem changes its partner and its half of the wave
    sh = _solve(hip, Hd[1:B + 1], c[1:B + 1], A[1:B + 1], b[1:B + 1], lb[1:B + 1], ub[1:B + 1], be)
    for x, y in zip(full, sh):
        assert np.array_equal(x[1:B + 1], y)


def test_exact_size_18x12_instantiation_gives_the_padded_one_s_bits(oracle, tmp_path):
    """18 x 12 runs on qp_inv_kernel<18, 12> instead of the padded <20, 16>, 22 x 15 on <22, 16> instead of <24, 16>
    (k_qp.hip), 38 x 29 on the whole-wave <38, 30, 64> instead of <40, 32, 64>: the padding only ever added zeros
    to the same summation chains, so solution, status and Newton count are the same bits.  The developer switch
    ASIF_HIP_QP_INV_EXACT=0 is read once per process: two child processes."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, numpy as np, torch\n"
        f"sys.path.insert(0, {root!r}); sys.path.insert(0, {os.path.join(root, 'tests')!r})\n"
        "import oracle_lib as O; O.build()\n"
        "from asif_amd import capi\n"
        "from test_gpu_qp_generic import _config_qps, _solve\n"
        "d, Hd, c, A, b, lb, ub, be = _config_qps(O, 5, 4096)\n"
        "sol, st, it = _solve(capi, Hd, c, A, b, lb, ub, be)\n"
        "hp = O.load_halfplanes(); z = O.RobustData(hp); B = 1024\n"
        "x, u = O.make_batch_robust_data(hp, B); A, b, code, sel = z.assemble(x)\n"
        "Hd, c, lb, ub = (np.zeros((B, z.nv)) for _ in range(4))\n"
        "for i in range(B): Hd[i], c[i], lb[i], ub[i], be = z.qp_static(u[i])\n"
        "sol2, st2, it2 = _solve(capi, Hd, c, A, b, lb, ub, be)\n"
        "k = O.load_kernel('100Hz'); r = O.Realizable(k); x, u = O.make_batch_realizable(k, 256)\n"
        "A, b, code, info = r.assemble(x); keep = code == 1; x, u, A, b = x[keep], u[keep], A[keep], b[keep]\n"
        "Hd, c, lb, ub = (np.zeros((len(x), r.nv)) for _ in range(4))\n"
        "for i in range(len(x)): Hd[i], c[i], lb[i], ub[i], be = r.qp_static(u[i])\n"
        "sol3, st3, it3 = _solve(capi, Hd, c, A, b, lb, ub, be)\n"
        "from test_gpu_qp_lds import _robust_qps_with_n_halfplanes\n"
        "extra = {}\n"
        "for N in (1, 2, 3, 6, 7, 8):\n"
        "    d, q, _ = _robust_qps_with_n_halfplanes(O, N, 256)\n"
        "    s_, st_, it_ = _solve(capi, *q)\n"
        "    extra[f'solN{N}'], extra[f'stN{N}'], extra[f'itN{N}'] = s_, st_, it_\n"
        "np.savez(sys.argv[1], sol=sol, st=st, it=it, sol2=sol2, st2=st2, it2=it2, sol3=sol3, st3=st3, it3=it3, **extra)\n")
    outs = []
    for exact in ("1", "0"):
        f = str(tmp_path / f"exact{exact}.npz")
        env = dict(os.environ, ASIF_HIP_QP_INV_EXACT=exact)
        r = subprocess.run([sys.executable, "-c", code, f], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(np.load(f))
    a, b = outs
    assert np.all(a["st"] == 1) and np.array_equal(a["st"], b["st"]) and np.array_equal(a["it"], b["it"])
    assert np.array_equal(a["sol"], b["sol"])
    # 22 x 15 on <22, 16> against <24, 16>, infeasible problems included
    assert (a["st2"] != 1).sum() > 50 and np.array_equal(a["st2"], b["st2"]) and np.array_equal(a["it2"], b["it2"])
    assert np.array_equal(a["sol2"], b["sol2"])
    # 38 x 29 on the whole-wave <38, 30, 64> against <40, 32, 64>
    assert len(a["st3"]) > 100 and np.array_equal(a["st3"], b["st3"]) and np.array_equal(a["it3"], b["it3"])
    assert np.array_equal(a["sol3"], b["sol3"])
    # ASIFrobust's other sizes: 6 x 3, 10 x 6, 14 x 9, 26 x 18, 30 x 21 (half-wave), 34 x 24 (whole-wave)
    for N in (1, 2, 3, 6, 7, 8):
        for k in ("sol", "st", "it"):
            assert np.array_equal(a[f"{k}N{N}"], b[f"{k}N{N}"]), (N, k)


def _robust_qps_with_n_halfplanes(oracle, N, B):
    """The lifted (2 + 4 N) x 3 N problem of ASIFrobust for a safe set of N half-planes 1 - a_k . x >= 0, a_k spread
    round the circle at the radius of BASELINE's C5 box (1 / pi), on C5's seeded states."""
    model, variant = oracle.CONFIGS[5]
    o = oracle.default_options(model, variant)
    o.nHalfPlanes = N
    for k in range(N):
        th = 2 * np.pi * (k + 0.25) / N
        o.halfPlanes[2 * k] = np.cos(th) / np.pi
        o.halfPlanes[2 * k + 1] = np.sin(th) / np.pi
    d = oracle.dims(model, variant, o)
    assert (d.nv, d.nc) == (2 + 4 * N, 3 * N)
    x, u = oracle.make_batch(5, B)
    A, b, code, _ = oracle.assemble_batch(model, variant, o, x)
    assert np.all(code == 1)
    Hd, c, lb, ub = (np.zeros((B, d.nv)) for _ in range(4))
    for i in range(B):
        Hd[i], c[i], lb[i], ub[i], be = oracle.qp_static(model, variant, o, u[i])
    ua, rl, rc = oracle.filter_batch(model, variant, o, x, u, oracle.SOLVER_EXACT)
    return d, (Hd, c, A, b, lb, ub, be), (o, ua, rl, rc)


@pytest.mark.parametrize("N", [1, 2, 3, 5, 6, 7, 8])
def test_robust_lifted_problem_for_every_number_of_safety_functions(hip, oracle, N):
    """ASIFrobust hands its solver a (2 + 4 N) x 3 N problem (src/asif_robust.cpp:21-22): 6 x 3 ... 34 x 24 for the
    N = 1 ... 8 the library carries -- each on a kernel of its own size (k_qp.hip; N = 4 is the 18 x 12 above).  Status
    equal to the oracle's exact filter on every instance, |u - u_ref| <= 1e-6; the relaxation, which reaches the
    hundreds on the states far outside these sets, to 1e-5 of its size (the closed-loop tests' bar)."""
    B = 1024
    d, q, (o, ua, rl, rc) = _robust_qps_with_n_halfplanes(oracle, N, B)
    sol, st, it = _solve(hip, *q)
    assert np.array_equal(st == 1, rc == 1), f"{((st == 1) != (rc == 1)).sum()} status mismatches"
    ok = rc == 1
    assert ok.sum() > B // 2
    assert np.abs(sol[ok, 0].clip(o.lb[0], o.ub[0]) - ua[ok, 0]).max() <= U_TOL
    assert (np.abs(sol[ok, 1] - rl[ok, 0]) / np.maximum(1.0, np.abs(rl[ok, 0]))).max() <= 1e-5
    print(f"N={N}: max|delta| {np.abs(rl[ok, 0]).max():.1f}, max|delta - delta_ref| {np.abs(sol[ok, 1] - rl[ok, 0]).max():.1e}")
