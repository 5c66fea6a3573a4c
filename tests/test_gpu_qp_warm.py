"""asif_hip_qp_solve_batch_warm: the start OSQP's warm_start = 1 gives the second and later solve() calls of one
workspace (the reference's wrapper leaves it on, src/qpwrapper_osqp.cpp:68-69,217-245) on the wave-level kernels
(qp_inv.hpp half-wave and whole-wave variants, qp_lds.hpp).  The optimum does not depend on the start: verdicts equal
to the cold solve's and the oracle's on every instance, |u - u_ref| <= 1e-6 (a warm solve ends when three multiplier
updates in a row meet the termination test: qp_inv.hpp), and far fewer Newton steps where a cold solve needs many (the
realizable filter's 38 x 29 / 86 x 65: 9-11 -> 3-5); the robust filter's 18 x 12 takes four from a cold start and about
as many from a randomly moved neighbour -- its gain is in the closed loops (tests/test_gpu_host_cpp.py runs them warm,
profiles/r04/warm_start_loop_time.txt)."""
import numpy as np
import pytest
import torch

from test_gpu_qp_generic import _config_qps

pytestmark = pytest.mark.gpu

U_TOL = 1e-6


class _Workspace:
    """Device buffers of B solver workspaces of one shape: sol / status / iters and the warm block."""

    def __init__(self, hip, B, nv, nc):
        self.hip, self.B, self.nv, self.nc = hip, B, nv, nc
        dev = torch.device("cuda:0")
        self.sol = torch.zeros((nv, B), dtype=torch.float64, device=dev)
        self.status = torch.zeros(B, dtype=torch.int32, device=dev)
        self.iters = torch.zeros(B, dtype=torch.int32, device=dev)
        self.wx = torch.full((nv, B), float("nan"), dtype=torch.float64, device=dev)  # garbage until written
        self.wy = torch.full((nc + nv, B), float("nan"), dtype=torch.float64, device=dev)

    def solve(self, Hd, c, A, b, lb, ub, be, warm, H=None, **solver_kw):
        dev = torch.device("cuda:0")
        t = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a.T)).to(dev)
        self.hip.qp_solve_batch_warm(t(Hd), t(H), t(c), t(A), t(b), t(lb), t(ub), self.sol, self.status, self.wx,
                                     self.wy, warm, iters=self.iters, be=be,
                                     solver=self.hip.default_solver(**solver_kw))
        torch.cuda.synchronize()
        return self.sol.cpu().numpy().T.copy(), self.status.cpu().numpy().copy(), self.iters.cpu().numpy().copy()


def _robust_qps_at(oracle, x, u):
    model, variant = oracle.CONFIGS[5]
    o = oracle.default_options(model, variant)
    d = oracle.dims(model, variant, o)
    A, b, code, _ = oracle.assemble_batch(model, variant, o, x)
    assert np.all(code == 1)
    n = len(x)
    Hd, c, lb, ub = (np.zeros((n, d.nv)) for _ in range(4))
    be = None
    for i in range(n):
        Hd[i], c[i], lb[i], ub[i], be = oracle.qp_static(model, variant, o, u[i])
    ua, rl, rc = oracle.filter_batch(model, variant, o, x, u, oracle.SOLVER_EXACT)
    return (Hd, c, A, b, lb, ub, be), (o, ua, rl, rc)


def test_lifted_robust_problem_two_control_steps(hip, oracle):
    """18 x 12 (ASIFrobust, C5) on the half-wave kernel: step k cold, step k + 1 -- the state moved as a plant step
    moves it, the desired input with it -- warm from step k and, beside it, cold."""
    B = 4096
    x, u = oracle.make_batch(5, B)
    rng = np.random.default_rng(5)
    x2 = x + 0.01 * rng.normal(size=x.shape)
    u2 = u + 0.02 * rng.normal(size=u.shape)
    q1, _ = _robust_qps_at(oracle, x, u)
    q2, (o, ua, rl, rc) = _robust_qps_at(oracle, x2, u2)
    ws = _Workspace(hip, B, 18, 12)
    s1, st1, it1 = ws.solve(*q1, warm=False)
    assert np.all(st1 == 1)
    wx = ws.wx.cpu().numpy().T
    assert np.array_equal(wx, s1)  # the warm block's x is the solution handed back
    wy = ws.wy.cpu().numpy().T
    assert np.all(np.isfinite(wy))
    # y is OSQP's dual of l <= [A; I] x <= u: stationarity 2 H x + c + [A; I]' y = 0 at the solution
    Hd, c, A, b = q1[0], q1[1], q1[2], q1[3]
    Am = A.reshape(B, 18, 12).transpose(0, 2, 1)
    grad = 2 * Hd * s1 + c + np.einsum("brv,br->bv", Am, wy[:, :12]) + wy[:, 12:]
    assert np.abs(grad).max() <= 1e-6
    assert wy[:, :12][:, ~np.asarray(q1[6], dtype=bool)].max() <= 1e-9  # A x >= b: multipliers of the lower side only
    sw, stw, itw = ws.solve(*q2, warm=True)
    cold = _Workspace(hip, B, 18, 12)
    sc, stc, itc = cold.solve(*q2, warm=False)
    assert np.array_equal(stw, rc) and np.array_equal(stc, rc)
    for s in (sw, sc):
        assert np.abs(s[:, 0].clip(o.lb[0], o.ub[0]) - ua[:, 0]).max() <= U_TOL
        assert np.abs(s[:, 1] - rl[:, 0]).max() <= U_TOL
    assert np.abs(sw[:, :2] - sc[:, :2]).max() <= U_TOL  # (each is within the solver's 1e-7 of the exact optimum)
    print(f"18x12 Newton steps per problem: cold {itc.mean():.2f} (max {itc.max()}), warm {itw.mean():.2f} (max {itw.max()})")
    assert itw.mean() < 2 * itc.mean()  # (no gain promised on this shape from a randomly moved state; no blow-up either)


def test_unchanged_problem_takes_next_to_nothing(hip, oracle):
    B = 2048
    x, u = oracle.make_batch(5, B)
    q, _ = _robust_qps_at(oracle, x, u)
    ws = _Workspace(hip, B, 18, 12)
    s1, st1, it1 = ws.solve(*q, warm=False)
    s2, st2, it2 = ws.solve(*q, warm=True)
    assert np.array_equal(st1, st2) and np.abs(s1[:, :2] - s2[:, :2]).max() <= U_TOL
    assert it2.max() <= 6 and it2.mean() <= 3.5 and it2.mean() < it1.mean() and it1.mean() >= 4


def test_warm_flag_off_is_the_cold_entry_bit_for_bit(hip, oracle):
    """warm_in = 0 reads nothing of the (NaN-filled) block: the same bits as asif_hip_qp_solve_batch."""
    from test_gpu_qp_generic import _solve
    d, Hd, c, A, b, lb, ub, be = _config_qps(oracle, 5, 1024)
    ws = _Workspace(hip, len(c), d.nv, d.nc)
    s1, st1, it1 = ws.solve(Hd, c, A, b, lb, ub, be, warm=False)
    s0, st0, it0 = _solve(hip, Hd, c, A, b, lb, ub, be)
    assert np.array_equal(s1, s0) and np.array_equal(st1, st0) and np.array_equal(it1, it0)


def test_a_block_that_was_never_written_is_a_cold_start(hip, oracle):
    d, Hd, c, A, b, lb, ub, be = _config_qps(oracle, 5, 512)
    ws = _Workspace(hip, len(c), d.nv, d.nc)  # NaN everywhere
    s1, st1, it1 = ws.solve(Hd, c, A, b, lb, ub, be, warm=True)
    ref = _Workspace(hip, len(c), d.nv, d.nc)
    s0, st0, it0 = ref.solve(Hd, c, A, b, lb, ub, be, warm=False)
    # the same path from the same zeros; the warm call only asks for two more multiplier updates at the end
    assert np.array_equal(st1, st0) and np.abs(s1[:, :2] - s0[:, :2]).max() <= U_TOL and np.all(it1 >= it0)
    assert (it1 - it0).max() <= 6


def test_infeasible_problem_leaves_a_cold_start_behind(hip, oracle):
    """DoubleIntegrator_Robust on the shipped half-planes (22 x 15): a quarter of the seeded problems are infeasible.
    Their warm block is zeros afterwards; a second, warm call gives every verdict again."""
    hp = oracle.load_halfplanes()
    z = oracle.RobustData(hp)
    B = 1024
    x, u = oracle.make_batch_robust_data(hp, B)
    ua, rl, rc = z.filter(x, u)
    A, b, code, sel = z.assemble(x)
    Hd, c, lb, ub = (np.zeros((B, z.nv)) for _ in range(4))
    for i in range(B):
        Hd[i], c[i], lb[i], ub[i], be = z.qp_static(u[i])
    ws = _Workspace(hip, B, z.nv, z.nc)
    s1, st1, it1 = ws.solve(Hd, c, A, b, lb, ub, be, warm=False)
    assert np.array_equal(st1 == 1, rc == 1) and (rc != 1).sum() > 50
    wx, wy = ws.wx.cpu().numpy().T, ws.wy.cpu().numpy().T
    assert np.all(wx[rc != 1] == 0.0) and np.all(wy[rc != 1] == 0.0)
    s2, st2, it2 = ws.solve(Hd, c, A, b, lb, ub, be, warm=True)
    assert np.array_equal(st2, st1)
    ok = rc == 1
    assert np.abs(s2[ok, 0].clip(-20, 20) - ua[ok, 0]).max() <= U_TOL
    assert it2[ok].mean() < 0.7 * it1[ok].mean()
    assert np.array_equal(it2[~ok], it1[~ok])  # cold again: the same path to the same certificate


@pytest.mark.parametrize("kernel,shape", [("100Hz", (38, 29)), ("10Hz", (86, 65))])
def test_realizable_lifted_problem_warm(hip, oracle, kernel, shape):
    """Whole-wave variant of qp_inv.hpp (38 x 29) and the factorising LDS kernel (86 x 65)."""
    k = oracle.load_kernel(kernel)
    z = oracle.Realizable(k)
    assert (z.nv, z.nc) == shape
    B = 384
    x, u = oracle.make_batch_realizable(k, B)
    rng = np.random.default_rng(38)

    def qps(x, u):
        ua, rl, rc = z.filter(x, u)
        A, b, code, info = z.assemble(x)
        Hd, c, lb, ub = (np.zeros((len(x), z.nv)) for _ in range(4))
        be = None
        for i in range(len(x)):
            Hd[i], c[i], lb[i], ub[i], be = z.qp_static(u[i])
        return (Hd, c, A, b, lb, ub, be), ua, rc, code

    _, _, _, code = qps(x, u)
    x2 = x + 0.002 * rng.normal(size=x.shape)
    _, _, _, code2 = qps(x2, u)
    keep = (code == 1) & (code2 == 1)
    x, x2, u = x[keep], x2[keep], u[keep]
    q1, _, _, _ = qps(x, u)
    q2, ua, rc, _ = qps(x2, u)
    n = len(x)
    assert n > 100
    ws = _Workspace(hip, n, *shape)
    ws.solve(*q1, warm=False)
    sw, stw, itw = ws.solve(*q2, warm=True)
    cold = _Workspace(hip, n, *shape)
    sc, stc, itc = cold.solve(*q2, warm=False)
    assert np.array_equal(stw == 1, rc == 1) and np.array_equal(stc == 1, rc == 1)
    ok = rc == 1
    assert np.abs(sw[ok, 0].clip(-20, 20) - ua[ok, 0]).max() <= U_TOL
    print(f"{shape}: Newton steps cold {itc[ok].mean():.2f}, warm {itw[ok].mean():.2f}")
    if shape == (38, 29):
        assert itw[ok].sum() < 0.9 * itc[ok].sum()
    # (86 x 65 on the factorising kernel: 9.5 cold, 14 warm on this batch -- the start is honoured, not a gain there)


def test_full_cost_matrix_warm(hip):
    from test_gpu_qp_lds import _random_dense
    rng = np.random.default_rng(77)
    B, nv, nc = 64, 20, 30
    H, c, A, b, lb, ub = _random_dense(rng, B, nv, nc)
    Hcm = H.transpose(0, 2, 1).reshape(B, nv * nv)
    Af = A.reshape(B, nv * nc)
    ws = _Workspace(hip, B, nv, nc)
    s1, st1, it1 = ws.solve(None, c, Af, b, lb, ub, None, warm=False, H=Hcm)
    c2 = c + 0.01 * rng.normal(size=c.shape)
    sw, stw, itw = ws.solve(None, c2, Af, b, lb, ub, None, warm=True, H=Hcm)
    cold = _Workspace(hip, B, nv, nc)
    sc, stc, itc = cold.solve(None, c2, Af, b, lb, ub, None, warm=False, H=Hcm)
    assert np.all(st1 == 1) and np.all(stw == 1) and np.all(stc == 1)
    assert np.abs(sw - sc).max() <= U_TOL  # strictly convex: the whole solution is unique
    print(f"full H 20x30: Newton steps cold {itc.mean():.2f}, warm {itw.mean():.2f}")
    assert itw.sum() <= 1.5 * itc.sum()


def test_small_shapes_and_plain_admm_leave_the_block_alone(hip, oracle):
    d, Hd, c, A, b, lb, ub, be = _config_qps(oracle, 2, 256)  # 2 x 4: in registers
    ws = _Workspace(hip, len(c), d.nv, d.nc)
    s, st, it = ws.solve(Hd, c, A, b, lb, ub, be, warm=True)
    assert np.all(np.isnan(ws.wx.cpu().numpy())) and np.all(np.isnan(ws.wy.cpu().numpy()))
    ex, stex, _ = oracle.qp_solve_batch(d.nv, d.nc, Hd, c, A, b, lb, ub, be, oracle.SOLVER_EXACT)
    assert np.array_equal(st == 1, stex == 1)
    d, Hd, c, A, b, lb, ub, be = _config_qps(oracle, 5, 256)  # 18 x 12 under polish = 0: plain ADMM, two launches
    ws = _Workspace(hip, len(c), d.nv, d.nc)
    s, st, it = ws.solve(Hd, c, A, b, lb, ub, be, warm=True, lanes_per_qp=64, polish=0)
    assert np.all(np.isnan(ws.wx.cpu().numpy())) and np.all(st == 1)


def test_argument_errors(hip):
    dev = torch.device("cuda:0")
    z = lambda r: torch.zeros((r, 4), dtype=torch.float64, device=dev)
    st = torch.zeros(4, dtype=torch.int32, device=dev)
    with pytest.raises(hip.AsifHipError):  # both forms of the cost
        hip.qp_solve_batch_warm(z(5) + 1, z(25), z(5), z(10), z(2), z(5) - 1, z(5) + 1, z(5), st, z(5), z(7), False)
    with pytest.raises(hip.AsifHipError):  # no warm block
        hip.qp_solve_batch_warm(z(5) + 1, None, z(5), z(10), z(2), z(5) - 1, z(5) + 1, z(5), st, None, None, False)


def test_ragged_batch_with_a_leading_dimension(hip, oracle):
    """B = 37 (an odd batch: the last wave's second half has no problem) inside arrays of leading dimension 48: the same
    bits as the dense call, the 11 columns beyond the batch untouched in every output."""
    d, Hd, c, A, b, lb, ub, be = _config_qps(oracle, 5, 37)
    B, ld = 37, 48
    dev = torch.device("cuda:0")

    def wide(a):  # [B, k] -> view [k, B] of a [k, ld] tensor whose other columns hold a marker
        w = torch.full((a.shape[1], ld), 7.25, dtype=torch.float64, device=dev)
        w[:, :B] = torch.from_numpy(np.ascontiguousarray(a.T)).to(dev)
        return w
    ins = [wide(a) for a in (Hd, c, A, b, lb, ub)]
    outs = {k: torch.full((r, ld), -3.5, dtype=torch.float64, device=dev) for k, r in (("sol", 18), ("wx", 18), ("wy", 30))}
    st = torch.full((ld,), -77, dtype=torch.int32, device=dev)
    it = torch.full((ld,), -77, dtype=torch.int32, device=dev)
    v = lambda t: t[:, :B]
    hip.qp_solve_batch_warm(v(ins[0]), None, *[v(t) for t in ins[1:]], v(outs["sol"]), st[:B], v(outs["wx"]), v(outs["wy"]),
                            False, iters=it[:B], be=be)
    torch.cuda.synchronize()
    ws = _Workspace(hip, B, 18, 12)
    s0, st0, it0 = ws.solve(Hd, c, A, b, lb, ub, be, warm=False)
    assert np.array_equal(outs["sol"][:, :B].cpu().numpy().T, s0) and np.array_equal(st[:B].cpu().numpy(), st0)
    assert np.array_equal(outs["wx"][:, :B].cpu().numpy(), ws.wx.cpu().numpy())
    assert np.array_equal(outs["wy"][:, :B].cpu().numpy(), ws.wy.cpu().numpy())
    for t in outs.values():
        assert np.all(t[:, B:].cpu().numpy() == -3.5)
    assert np.all(st[B:].cpu().numpy() == -77) and np.all(it[B:].cpu().numpy() == -77)
    # and a warm second call over the same strided block
    hip.qp_solve_batch_warm(v(ins[0]), None, *[v(t) for t in ins[1:]], v(outs["sol"]), st[:B], v(outs["wx"]), v(outs["wy"]),
                            True, iters=it[:B], be=be)
    torch.cuda.synchronize()
    assert np.array_equal(st[:B].cpu().numpy(), st0) and np.abs(outs["sol"][:2, :B].cpu().numpy().T - s0[:, :2]).max() <= U_TOL
    for t in outs.values():
        assert np.all(t[:, B:].cpu().numpy() == -3.5)
