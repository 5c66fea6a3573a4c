"""C1 (BASELINE.json configs[0]): the reference's DoubleIntegrator closed loop, single agent, on this
build's C++ mirror of the reference API (ASIF::ASIF + ASIF::QPWrapperHip, asif_amd/host/).  The program
is the drop-in test: same constructor, initialize(), filter() and plant integration as
examples/DoubleIntegrator.cpp:63-116, first 2500 steps.  Compared step by step with the oracle's closed
loop (exact QP optimum): |uAct - u_ref| <= 1e-6 (north star 1e-5), identical rc, state within 1e-9.
It also pushes 256 copies of the state through filterBatch() every 100 steps (must equal filter())."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "asif_amd", "host")


def test_double_integrator_closed_loop(hip, oracle):
    check_double_integrator_closed_loop(oracle, "hip")


# The closed-loop checks below take the solver the program is run with: "hip" (QPWrapperHip, the tests of this file) or
# "host" (QPWrapperHost, `--solver host`: no device anywhere in the program; tests/test_host_solver_cpp.py runs them in
# the GPU-less container).
def check_double_integrator_closed_loop(oracle, solver):
    exe = os.path.join(HOST, "double_integrator")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", HOST, "-s"])
    steps = 2500
    extra = ["--batch", "256"] if solver == "hip" else ["--solver", "host"]
    out = subprocess.run([exe, "--steps", str(steps)] + extra, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    rows = np.array([[float(v) for v in line.split(",")] for line in out.stdout.strip().split("\n")[1:]])
    assert rows.shape == (steps, 7)
    model, variant = oracle.CONFIGS[2]
    o = oracle.default_options(model, variant)
    # Per step, on the state the program was actually in: its input must be the exact optimum.
    # (A free-running second closed loop is not comparable to 1e-6: as the agent comes to rest at the
    # wall, Lgh = -v -> 0 and du/dx grows without bound, so 1e-16 differences are amplified.)
    xprev = np.vstack([[0.0, 0.0], rows[:-1, 1:3]])
    ua, rl, rc = oracle.filter_batch(model, variant, o, xprev, np.ones((steps, 1)), oracle.SOLVER_EXACT)
    assert np.array_equal(rows[:, 6].astype(int), rc) and np.all(rc == 1)
    assert np.abs(rows[:, 4] - ua[:, 0]).max() <= 1e-6
    assert np.abs(rows[:, 5] - 5.0).max() <= 1e-9  # pinned relaxation variable
    # plant integration of the example (explicit Euler, dt = 1 ms) reproduced from its own inputs
    xn = xprev + 0.001 * np.stack([xprev[:, 1], rows[:, 4]], axis=1)
    assert np.abs(xn - rows[:, 1:3]).max() <= 1e-15
    # and the free-running oracle loop agrees while the loop gain is still moderate
    x = np.zeros(2)
    for k in range(800):
        u1, _, _ = oracle.filter_batch(model, variant, o, x[None, :], np.array([[1.0]]), oracle.SOLVER_EXACT)
        x = x + 0.001 * np.array([x[1], u1[0, 0]])
        assert np.abs(rows[k, 1:3] - x).max() <= 1e-9
    # the filter must have intervened: uDes = 1 drives the agent towards the x = 1 wall
    assert rows[:, 4].min() < 0.0 and rows[:, 1].max() < 1.0 + 1e-6


def _run_backup(kind, n, *extra, solver="hip"):
    exe = os.path.join(HOST, "backup_filters")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", HOST, "-s"])
    out = subprocess.run([exe, kind, str(n)] + [str(e) for e in extra] + (["--solver", "host"] if solver == "host" else []),
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    return np.array([[float(v) for v in line.split(",")] for line in out.stdout.strip().split("\n")[1:]])


def test_implicit_class_single_agent_and_batch(hip, oracle):
    """ASIF::ASIFimplicit with host std::function callbacks (trajectory + rows on the host, QP on the GPU)
    and filterBatch() (all on the GPU) against the oracle's exact answer on the C3 states."""
    n = 24
    rows = _run_backup("implicit", n)
    x, u = oracle.make_batch(3, n)
    model, variant = oracle.CONFIGS[3]
    ua, rl, rc = oracle.filter_batch(model, variant, oracle.default_options(model, variant), x, u, oracle.SOLVER_EXACT)
    assert np.array_equal(rows[:, 4].astype(int), rc) and np.array_equal(rows[:, 6].astype(int), rc)
    assert np.abs(rows[:, 1] - ua[:, 0]).max() <= 1e-6      # single agent
    assert np.abs(rows[:, 5] - ua[:, 0]).max() <= 1e-6      # batch
    ok = rc == 1
    assert np.abs(rows[ok, 2] - rl[ok, 0]).max() <= 1e-5 and np.abs(rows[ok, 3] - rl[ok, 1]).max() <= 1e-5


def test_implicit_class_published_lie_derivatives(hip, oracle):
    """The three vectors ASIFimplicit publishes for the learning pipeline (include/asif_implicit.h:122-124) after a
    single-agent filter(): Lfh_out_ = -b and Lgh_out_[.][0] = the input column of the oracle's rows; Dh_out_ with the
    reference's index arithmetic (src/asif_implicit.cpp:556-561: entry [i][j] is element nx i + j of the COLUMN-major
    npTC x nx array) -- Dh is recovered from the rows through f = (x1, sin x0), g = (0, 1)."""
    n = 12
    out = _run_backup("implicit-out", n)
    assert out.shape == (n, 41 + 41 + 82)
    x, _ = oracle.make_batch(3, n)
    model, variant = oracle.CONFIGS[3]
    A, b, code, _ = oracle.assemble_batch(model, variant, oracle.default_options(model, variant), x)
    for i in range(n):
        Lfh, Lgh, Dho = out[i, :41], out[i, 41:82], out[i, 82:].reshape(41, 2)
        np.testing.assert_allclose(Lfh, -b[i], rtol=1e-12, atol=1e-13)
        np.testing.assert_allclose(Lgh, A[i][:41], rtol=1e-12, atol=1e-13)
        f0, f1 = x[i, 1], np.sin(x[i, 0])
        Dh = np.concatenate([(Lfh - Lgh * f1) / f0, Lgh])  # column-major 41 x 2: d/dx0 then d/dx1
        np.testing.assert_allclose(Dho.reshape(-1), Dh, rtol=1e-9, atol=1e-9 * np.abs(Dh).max() / abs(f0))


def test_tb_class_single_agent_and_batch(hip, oracle):
    n = 300
    rows = _run_backup("tb", n)
    x, u = oracle.make_batch(4, n)
    model, variant = oracle.CONFIGS[4]
    ua, rl, rc = oracle.filter_batch(model, variant, oracle.default_options(model, variant), x, u, oracle.SOLVER_EXACT)
    assert np.array_equal(rows[:, 4].astype(int), rc) and np.array_equal(rows[:, 6].astype(int), rc)
    assert {1, 2, -3} <= set(rc.tolist())
    assert np.abs(rows[:, 1] - ua[:, 0]).max() <= 1e-6
    assert np.abs(rows[:, 5] - ua[:, 0]).max() <= 1e-6


def test_segway_tb_closed_loop(hip, oracle):
    check_segway_tb_closed_loop(oracle, "hip")


def check_segway_tb_closed_loop(oracle, solver):
    """BASELINE config 4's example loop (examples/segway_implicit_tb.cpp:236-275) through the C++ class, literally: from
    rest, uDes = 0, plant Euler at 1 ms, updateOptions(backTrajHorizon = 6) at half time (1 600 of the example's 10 990
    steps).  Rest is not an equilibrium of the model (the pitch's is 0.138 rad): the agent drifts out of the backup set
    within a fraction of a second.  Every step's input must be the exact optimum of the QP the reference assembles on
    the state the program was in (or the saturated backup controller where the reference applies it), with the options
    in force: 316 samples before the update, 601 after (no 1 + backTrajExtend, src/asif_implicit_tb.cpp:377)."""
    steps = 1600
    rows = _run_backup("tb-loop", steps, solver=solver)
    assert rows.shape == (steps, 10)
    upd = rows[:, 9].astype(int)
    first = int(np.argmax(upd))
    assert 799 <= first <= 802 and np.all(upd[first:] == 1)
    model, variant = oracle.CONFIGS[4]
    o1 = oracle.default_options(model, variant)
    o2 = oracle.default_options(model, variant)
    o2.backTrajHorizon = 6.0
    o2.backTrajExtend = 0.0
    assert oracle.dims(model, variant, o1).npBT == 316 and oracle.dims(model, variant, o2).npBT == 601
    x = np.ascontiguousarray(rows[:, 1:5])
    ud = np.zeros((steps, 1))
    ua = np.empty(steps)
    rc = np.empty(steps, dtype=int)
    for o, sl in ((o1, slice(0, first)), (o2, slice(first, steps))):
        a, _, r = oracle.filter_batch(model, variant, o, x[sl], ud[sl], oracle.SOLVER_EXACT, nthreads=8)
        ua[sl], rc[sl] = a[:, 0], r
    assert np.array_equal(rows[:, 8].astype(int), rc), np.where(rows[:, 8].astype(int) != rc)[0][:10]
    assert np.abs(rows[:, 5] - ua).max() <= 1e-6
    assert 2 in rc and (rc != 2).sum() > 100  # starts inside the backup set, leaves it


@pytest.mark.parametrize("kind,cfg,steps,run", [("implicit-loop", 3, 300, 5), ("dii-loop", 9, 1200, 0), ("tbip-loop", 8, 160, 1)])
def test_remaining_example_loops_step_by_step(hip, oracle, kind, cfg, steps, run):
    check_example_loop_step_by_step(oracle, kind, cfg, steps, run, "hip")


def check_example_loop_step_by_step(oracle, kind, cfg, steps, run, solver):
    """The main() loops of examples/InvertedPendulum_Implicit.cpp (run 5 of its ten start states),
    examples/DoubleIntegrator_implicit.cpp (fused-gradient constructor, updateOptions(backTrajHorizon = 5) at half time)
    and examples/InvertedPendulum_ImplicitTB.cpp (its second start state) through the C++ classes: every step's input is
    the exact optimum of the QP the reference assembles on the state the program was in, rc identical, and the plant
    step is the example's."""
    rows = _run_backup(kind, steps, run, solver=solver)
    assert rows.shape == (steps, 8)
    model, variant = oracle.CONFIGS[cfg]
    x = np.ascontiguousarray(rows[:, 1:3])
    ud = np.full((steps, 1), 1.0 if cfg == 9 else 0.0)
    upd = rows[:, 7].astype(int)
    first = int(np.argmax(upd)) if upd.any() else steps
    o1 = oracle.default_options(model, variant)
    o2 = oracle.default_options(model, variant)
    o2.backTrajHorizon = 5.0
    ua = np.empty(steps)
    rc = np.empty(steps, dtype=int)
    for o, sl in ((o1, slice(0, first)), (o2, slice(first, steps))):
        if sl.start >= sl.stop:
            continue
        a, _, r = oracle.filter_batch(model, variant, o, x[sl], ud[sl], oracle.SOLVER_EXACT, nthreads=8)
        ua[sl], rc[sl] = a[:, 0], r
    if cfg == 9:
        assert 599 <= first <= 602 and oracle.dims(model, variant, o2).npBT == 501
    assert np.array_equal(rows[:, 6].astype(int), rc), np.where(rows[:, 6].astype(int) != rc)[0][:10]
    assert np.abs(rows[:, 3] - ua).max() <= 1e-6
    f1 = np.sin(x[:-1, 0]) if cfg in (3, 8) else 0.0 * x[:-1, 0]
    xn = x[:-1] + 0.001 * np.stack([x[:-1, 1], f1 + rows[:-1, 3]], axis=1)
    assert np.abs(xn - x[1:]).max() <= 1e-15


def test_tb_class_double_integrator_single_agent_and_batch(hip, oracle):
    """examples/DoubleIntegrator_implicit_tb.cpp's model through the class's fused-gradient constructor
    (dynamicsWithGradient, include/asif_implicit_tb.h:74-84) and through filterBatch()."""
    n = 300
    rows = _run_backup("tbdi", n)
    x, u = oracle.make_batch(12, n)
    model, variant = oracle.CONFIGS[12]
    ua, rl, rc = oracle.filter_batch(model, variant, oracle.default_options(model, variant), x, u, oracle.SOLVER_EXACT)
    assert np.array_equal(rows[:, 4].astype(int), rc) and np.array_equal(rows[:, 6].astype(int), rc)
    assert {1, 2, -3} <= set(rc.tolist())
    assert np.abs(rows[:, 1] - ua[:, 0]).max() <= 1e-6
    assert np.abs(rows[:, 5] - ua[:, 0]).max() <= 1e-6


def test_double_integrator_tb_closed_loop(hip, oracle):
    check_double_integrator_tb_closed_loop(oracle, "hip")


def check_double_integrator_tb_closed_loop(oracle, solver, steps=5000):
    """The example's own loop (examples/DoubleIntegrator_implicit_tb.cpp:105-160): 5 000 control steps from (0.1, 0.1)
    with uDes = 0.9, updateOptions(backTrajHorizon = 7) once t > 2.5 s.  Per step, on the state the program was in, the
    input must be the exact optimum of the QP the reference assembles with the options in force (after the update the
    trajectory has 7 001 samples: the (1 + backTrajExtend) factor is dropped, src/asif_implicit_tb.cpp:377)."""
    rows = _run_backup("tbdi-loop", steps, solver=solver)
    batch_row = None
    if solver == "hip":  # the program's last line: filterBatch() on the final state
        assert rows.shape == (steps + 1, 8)
        batch_row, rows = rows[-1], rows[:-1]
    assert rows.shape == (steps, 8)
    upd = rows[:, 7].astype(int)
    first = int(np.argmax(upd))
    assert steps // 2 - 1 <= first <= steps // 2 + 2 and np.all(upd[first:] == 1) and np.all(upd[:first] == 0)
    model, variant = oracle.CONFIGS[12]
    o1 = oracle.default_options(model, variant)
    o2 = oracle.default_options(model, variant)
    o2.backTrajHorizon = 7.0
    o2.backTrajExtend = 0.0
    assert oracle.dims(model, variant, o2).npBT == 7001
    ud = np.full((steps, 1), 0.9)
    x = np.ascontiguousarray(rows[:, 1:3])
    ua = np.empty(steps)
    rc = np.empty(steps, dtype=int)
    for o, sl in ((o1, slice(0, first)), (o2, slice(first, steps))):
        a, _, r = oracle.filter_batch(model, variant, o, x[sl], ud[sl], oracle.SOLVER_EXACT, nthreads=8)
        ua[sl], rc[sl] = a[:, 0], r
    assert np.array_equal(rows[:, 6].astype(int), rc), np.where(rows[:, 6].astype(int) != rc)[0][:10]
    assert np.abs(rows[:, 3] - ua).max() <= 1e-6
    assert {1, -3} <= set(rc.tolist())  # out of reach of the small backup set at first, filtering later
    # the plant step of the example, reproduced from its own inputs
    xn = x[:-1] + 0.001 * np.stack([x[:-1, 1], rows[:-1, 3]], axis=1)
    assert np.abs(xn - x[1:]).max() <= 1e-15
    if batch_row is not None:  # filterBatch() on the state after the last step, options as updated
        a, _, r = oracle.filter_batch(model, variant, o2, batch_row[None, 1:3], np.array([[0.9]]), oracle.SOLVER_EXACT)
        assert int(batch_row[6]) == r[0] and abs(batch_row[3] - a[0, 0]) <= 1e-6


def test_robust_class_single_agent_and_batch(hip, oracle):
    check_robust_class(oracle, "hip")


def _solver_args(solver):
    return ["--solver", "host"] if solver == "host" else []


def check_robust_class(oracle, solver):
    """ASIF::ASIFrobust: host affine arithmetic (asif_affine.h) must reproduce the oracle's rows bit for bit
    (the oracle is pinned against the reference's libaffa); single-agent filter() solves the FULL 18-variable
    QP on the wave-per-QP kernel (plain ADMM, 1e-5), filterBatch() the eliminated one (1e-6)."""
    exe = os.path.join(HOST, "robust_pendulum")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", HOST, "-s"])
    n = 48
    out = subprocess.run([exe] + _solver_args(solver) + [str(n)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = out.stdout.strip().split("\n")[1:]
    res = np.array([[float(v) for v in l.split(",")] for l in lines if not l.startswith("A,")])
    rowsA = np.array([[float(v) for v in l.split(",")[2:]] for l in lines if l.startswith("A,")])
    x, u = oracle.make_batch(5, n)
    model, variant = oracle.CONFIGS[5]
    o = oracle.default_options(model, variant)
    A, b, code, _ = oracle.assemble_batch(model, variant, o, x)
    assert np.array_equal(rowsA, A)
    ua, rl, rc = oracle.filter_batch(model, variant, o, x, u, oracle.SOLVER_EXACT)
    assert np.array_equal(res[:, 3].astype(int), rc)
    assert np.abs(res[:, 1] - ua[:, 0]).max() <= 1e-5
    assert np.abs(res[:, 2] - rl[:, 0]).max() <= 1e-5
    if solver != "host":  # (no filterBatch without a device)
        assert np.array_equal(res[:, 5].astype(int), rc)
        assert np.abs(res[:, 4] - ua[:, 0]).max() <= 1e-6


@pytest.mark.parametrize("p,ud,steps", [(0.8, 0.0, 280), (1.0, 1.5, 155), (1.2, -1.5, 210)])
def test_robust_pendulum_closed_loop(hip, oracle, p, ud, steps):
    check_robust_pendulum_closed_loop(oracle, p, ud, steps, "hip")


@pytest.mark.parametrize("p,ud,steps", [(0.8, 0.0, 280), (1.0, 1.5, 155)])
def test_robust_pendulum_closed_loop_warm_started(hip, oracle, p, ud, steps, monkeypatch):
    """The same loops with QPWrapperHip::warmStart on (ASIF_HIP_QP_WARM=1): every solve() after the first starts from
    the previous control step's iterate and multipliers, as the reference's OSQP workspace does.  Same bar."""
    monkeypatch.setenv("ASIF_HIP_QP_WARM", "1")
    check_robust_pendulum_closed_loop(oracle, p, ud, steps, "hip")


def check_robust_pendulum_closed_loop(oracle, p, ud, steps, solver):
    """The main() loop of examples/InvertedPendulum_Robust.cpp:134-175 (ROBUST flavour: steps of 10 ms from (0.5, 0), plant
    gain p in {pMin, 1, pMax}) through ASIF::ASIFrobust: each step's affine-arithmetic rows on the host and its
    18-variable QP on the GPU, return code / input / relaxation against the oracle on the state the program was in.
    With the example's uDes = 0 the pendulum falls towards x0 = pi and the box's position rows (Lgh = 0) can only be
    relaxed, more at every step, until the state leaves the box at step 282; with uDes = +-1.5 the velocity reaches its
    half-plane and the filter takes the input back."""
    exe = os.path.join(HOST, "robust_pendulum")
    out = subprocess.run([exe] + _solver_args(solver) + ["--loop", str(steps), repr(p), repr(ud)], capture_output=True, text=True,
                         timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    rows = np.array([[float(v) for v in l.split(",")] for l in out.stdout.strip().split("\n")[1:]])
    assert rows.shape == (steps, 6)
    x = np.ascontiguousarray(rows[:, 1:3])
    model, variant = oracle.CONFIGS[5]
    o = oracle.default_options(model, variant)
    ua, rl, rc = oracle.filter_batch(model, variant, o, x, np.full((steps, 1), ud), oracle.SOLVER_EXACT)
    assert np.array_equal(rows[:, 5].astype(int), rc) and np.all(rc == 1)
    assert np.abs(rows[:, 3] - ua[:, 0]).max() <= 1e-5
    assert (np.abs(rows[:, 4] - rl[:, 0]) / np.maximum(1.0, np.abs(rl[:, 0]))).max() <= 1e-5
    assert np.abs(x).max() < np.pi
    if ud == 0.0:
        assert rows[-1, 4] > 20.0 and np.abs(rows[:, 3]).max() <= 1e-5  # relaxed, never actuated
    else:
        assert np.abs(rows[:, 3] - ud).max() > 0.3  # the filter acts through the velocity rows


def test_realizable_class_single_agent_and_batch(hip, oracle, tmp_path):
    check_realizable_class(oracle, tmp_path, "hip")


def _write_kernel(k, kfile):
    with open(kfile, "w") as f:
        nF, nA = k["facetVertices"].shape[0], k["maxActiveConstraints"]
        f.write(f"{k['vertices'].shape[0]} {nF} {k['maxCriticalFacets']} {nA}\n")
        for v in k["vertices"]:
            f.write(f"{float(v[0])!r} {float(v[1])!r}\n")
        for i in range(nF):
            f.write(" ".join([str(int(t)) for t in k["facetVertices"][i]] + [repr(float(t)) for t in k["facetNormals"][i]] +
                             [str(int(t)) for t in k["facetActive"][i]]) + "\n")


def check_realizable_class(oracle, tmp_path, solver):
    """ASIF::ASIFrealizable: facet search through facetSolver_ (2 x 5 QPs on the GPU), host affine arithmetic and
    the full 29 x 38 rows must reproduce the oracle's rows bit for bit; single-agent filter() (the lifted 38 x 29
    problem through QPWrapperHip, as src/asif_realizable.cpp:300-340 hands it to its solver) and filterBatch()
    against the oracle's exact optimum; rc 1 / -2 identical."""
    exe = os.path.join(HOST, "realizable_di")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", HOST, "-s"])
    k = oracle.load_kernel("100Hz")
    kfile = tmp_path / "kernel.txt"
    _write_kernel(k, kfile)
    n = 96
    x, u = oracle.make_batch_realizable(k, n)
    stdin = "".join(f"{float(x[i, 0])!r} {float(x[i, 1])!r} {float(u[i, 0])!r}\n" for i in range(n))
    out = subprocess.run([exe] + _solver_args(solver) + [str(kfile)], input=stdin, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = out.stdout.strip().split("\n")[1:]
    res = np.array([[float(v) for v in l.split(",")] for l in lines if l[0] not in "Ab"])
    rowsA = np.array([[float(v) for v in l.split(",")[2:]] for l in lines if l.startswith("A,")])
    rowsb = np.array([[float(v) for v in l.split(",")[2:]] for l in lines if l.startswith("b,")])
    z = oracle.Realizable(k)
    A, b, code, info = z.assemble(x)
    assert np.array_equal(res[:, 5].astype(int), info[:, 0])  # critical-facet counts found through the facet QPs
    assert np.array_equal(rowsA, A) and np.array_equal(rowsb, b)
    ua, rl, rc = z.filter(x, u)
    assert np.array_equal(res[:, 4].astype(int), rc)
    assert {1, -2} <= set(rc.tolist()) and (info[:, 0] > 0).sum() > 10
    ok = rc == 1
    # single agent: QPsolver_ gets the full 38 x 29 problem as the reference's does (wave-per-QP LDS kernel)
    assert np.abs(res[ok, 1] - ua[ok, 0]).max() <= 1e-6
    assert np.abs(res[ok, 3] - rl[ok, 1]).max() <= 1e-6
    # relax[0] = solutionFull[nu] is a multiplier with zero cost: any value from its smallest feasible one up is optimal
    assert (res[ok, 2] - rl[ok, 0]).min() >= -1e-6
    if solver != "host":
        assert np.array_equal(res[:, 8].astype(int), rc)
        assert np.abs(res[ok, 6] - ua[ok, 0]).max() <= 1e-6      # batch
        assert np.abs(res[ok, 7] - rl[ok, 1]).max() <= 1e-6


def test_realizable_sampled_closed_loop(hip, oracle, tmp_path):
    check_realizable_sampled_closed_loop(oracle, tmp_path, "hip")


def check_realizable_sampled_closed_loop(oracle, tmp_path, solver, steps=3000):
    """The main() loop of examples/DoubleIntegrator_RealizableSampled.cpp:96-190 (the 100 Hz kernel: plant at 1 kHz, the
    filter on every tenth step, uDes = 20, the example's moving input bounds) through ASIF::ASIFrealizable: 300 filter
    calls, each solving its facet QPs and the lifted 38 x 29 problem on the GPU; return code and filtered input against
    the oracle on the state the program was in."""
    exe = os.path.join(HOST, "realizable_di")
    k = oracle.load_kernel("100Hz")
    kfile = tmp_path / "kernel.txt"
    _write_kernel(k, kfile)
    out = subprocess.run([exe] + _solver_args(solver) + [str(kfile), "--loop", str(steps)], capture_output=True, text=True,
                         timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    rows = np.array([[float(v) for v in l.split(",")] for l in out.stdout.strip().split("\n")[1:]])
    assert rows.shape == (steps // 10, 8)
    x = np.ascontiguousarray(rows[:, 1:3])
    z = oracle.Realizable(k)
    ua, rl, rc = z.filter(x, np.full((len(x), 1), 20.0))
    assert np.array_equal(rows[:, 6].astype(int), rc), np.where(rows[:, 6].astype(int) != rc)[0][:10]
    ok = rc == 1
    assert ok.sum() > len(x) // 2
    assert np.abs(rows[ok, 3] - ua[ok, 0]).max() <= 1e-5  # (inputs up to 20; the north star's bound)
    assert rows[:, 3].min() < 19.0  # the filter does intervene on the way to the kernel's boundary


def test_robust_class_on_shipped_data(hip, oracle, tmp_path):
    check_robust_class_on_shipped_data(oracle, tmp_path, "hip")


def check_robust_class_on_shipped_data(oracle, tmp_path, solver):
    """ASIF::ASIFrobust as examples/DoubleIntegrator_Robust.cpp builds it (npSSmax = 5 of the 100 shipped half-planes):
    rows bit-identical to the oracle's; single-agent filter() solves the full 22 x 15 QP on the wave-per-QP LDS
    kernel (every return code equal, 1e-6), filterBatch() the eliminated one."""
    exe = os.path.join(HOST, "di_robust")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", HOST, "-s"])
    hp = oracle.load_halfplanes()
    hfile = tmp_path / "hp.txt"
    with open(hfile, "w") as f:
        f.write(f"{hp.shape[0]}\n")
        for a in hp:
            f.write(f"{float(a[0])!r} {float(a[1])!r}\n")
    n = 64
    x, u = oracle.make_batch_robust_data(hp, n)
    stdin = "".join(f"{float(x[i, 0])!r} {float(x[i, 1])!r} {float(u[i, 0])!r}\n" for i in range(n))
    out = subprocess.run([exe] + _solver_args(solver) + [str(hfile)], input=stdin, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = out.stdout.strip().split("\n")[1:]
    res = np.array([[float(v) for v in l.split(",")] for l in lines if not l.startswith("A,")])
    rowsA = np.array([[float(v) for v in l.split(",")[2:]] for l in lines if l.startswith("A,")])
    z = oracle.RobustData(hp)
    A, b, code, sel = z.assemble(x)
    assert np.array_equal(rowsA, A)
    ua, rl, rc = z.filter(x, u)
    ok = rc == 1
    if solver != "host":
        assert np.array_equal(res[:, 6].astype(int), rc)                 # batch: every code
        assert np.abs(res[ok, 4] - ua[ok, 0]).max() <= 1e-6 and np.abs(res[ok, 5] - rl[ok, 0]).max() <= 1e-6
    assert np.array_equal(res[:, 3].astype(int), rc)                     # single agent: every code
    assert np.abs(res[ok, 1] - ua[ok, 0]).max() <= 1e-6


def test_double_integrator_robust_closed_loop(hip, oracle, tmp_path):
    check_double_integrator_robust_closed_loop(oracle, tmp_path, "hip")


def check_double_integrator_robust_closed_loop(oracle, tmp_path, solver):
    """The main() loop of examples/DoubleIntegrator_Robust.cpp:88-131 (600 steps of 10 ms from rest, uDes = 20, the shipped
    100 half-planes with npSSmax = 5) through ASIF::ASIFrobust: every step's 22 x 15 QP is solved on the GPU and must give
    the oracle's return code and, where solved, its input on the state the program was in."""
    exe = os.path.join(HOST, "di_robust")
    hp = oracle.load_halfplanes()
    hfile = tmp_path / "hp.txt"
    with open(hfile, "w") as f:
        f.write(f"{hp.shape[0]}\n")
        for a in hp:
            f.write(f"{float(a[0])!r} {float(a[1])!r}\n")
    steps = 600
    out = subprocess.run([exe] + _solver_args(solver) + [str(hfile), "--loop", str(steps)], capture_output=True, text=True,
                         timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    rows = np.array([[float(v) for v in l.split(",")] for l in out.stdout.strip().split("\n")[1:]])
    assert rows.shape == (steps, 6)
    x = np.ascontiguousarray(rows[:, 1:3])
    z = oracle.RobustData(hp)
    ua, rl, rc = z.filter(x, np.full((steps, 1), 20.0))
    assert np.array_equal(rows[:, 5].astype(int), rc), np.where(rows[:, 5].astype(int) != rc)[0][:10]
    ok = rc == 1
    assert ok.sum() > steps // 2
    # (the lifted problem's inputs reach 20: 1e-5, the north star's bound, is 5e-7 of that; measured 1.6e-6)
    assert np.abs(rows[ok, 3] - ua[ok, 0]).max() <= 1e-5 and np.abs(rows[ok, 4] - rl[ok, 0]).max() <= 1e-5
    # the agent is driven towards the boundary of the safe set and the filter takes the input away from uDes
    assert rows[:, 3].min() < 19.0 and rows[0, 3] > 19.9


@pytest.mark.parametrize("plain", [False, True])
def test_implicit_rb_class_single_agent_and_batch(hip, oracle, plain):
    check_implicit_rb_class(oracle, plain, "hip")


def check_implicit_rb_class(oracle, plain, solver):
    """ASIF::ASIFimplicitRB (held backup input, interval margins from the user's safetySet_int on host AAF operands,
    learned residual) and ASIF::ASIFimplicit with use_learning: single-agent filter() with host callbacks and
    filterBatch() on the GPU against the oracle's exact answer; the class's public diagnostics Dh_index_ /
    learning_data_.Lfh_diff / Lgh_diff against the oracle's."""
    from asif_amd import workloads
    exe = os.path.join(HOST, "implicit_rb")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", HOST, "-s"])
    n = 24
    out = subprocess.run([exe, str(n)] + (["plain"] if plain else []) + (["--solver", "host"] if solver == "host" else []),
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    rows = np.array([[float(v) for v in line.split(",")] for line in out.stdout.strip().split("\n")[1:]])
    model = oracle.MODEL_IP
    variant = oracle.VAR_IMPLICIT if plain else oracle.VAR_IMPLICIT_RB
    o = oracle.default_options(model, variant)
    if not plain:
        o.x_unc[0], o.x_unc[1] = workloads.RB_X_UNC
    L = oracle.Learning.from_dict(workloads.make_learning())
    o.set_learning(L)
    x, u = oracle.make_batch(3 if plain else 10, n)
    ua, rl, rc = oracle.filter_batch(model, variant, o, x, u, oracle.SOLVER_EXACT)
    assert np.array_equal(rows[:, 4].astype(int), rc)
    assert np.abs(rows[:, 1] - ua[:, 0]).max() <= 1e-6      # single agent
    if solver == "hip":                                      # batch (the host-solver run has no device)
        assert np.array_equal(rows[:, 6].astype(int), rc) and np.abs(rows[:, 5] - ua[:, 0]).max() <= 1e-6
    ok = rc == 1
    assert np.abs(rows[ok, 2] - rl[ok, 0]).max() <= 1e-5 and np.abs(rows[ok, 3] - rl[ok, 1]).max() <= 1e-5
    for i in range(n):
        oracle.assemble(model, variant, o, x[i])
        dh, lf, lg = oracle.rb_last_learning()
        assert abs(rows[i, 7] - dh[0]) <= 1e-12 and abs(rows[i, 8] - lf) <= 1e-13 and abs(rows[i, 9] - lg[0]) <= 1e-13


def test_qpwrapper_hip_contract(hip, tmp_path):
    """ASIF::QPWrapperHip: full cost matrix, set-up vs solver verdicts kept apart (an infeasible first solve is not
    a set-up failure; a shape beyond the kernels is), the reference's largest shape accepted."""
    exe = str(tmp_path / "qpw")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I" + os.path.join(HOST, "include"),
                           "-I" + os.path.join(ROOT, "include"), "-I/opt/rocm/include", "-D__HIP_PLATFORM_AMD__",
                           os.path.join(ROOT, "tests", "host_qpwrapper_driver.cpp"),
                           os.path.join(HOST, "libasif_host.a"), "-L" + os.path.join(ROOT, "asif_amd"), "-lasif_hip",
                           "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + os.path.join(ROOT, "asif_amd"),
                           "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    r = {l.split()[0]: [float(v) for v in l.split()[1:]] for l in out.stdout.strip().split("\n")}
    # min 2x0^2 + 2x0x1 + 3x1^2 - 2x0 - 6x1 on [0, 0.9]^2 with x0 + x1 >= 1: on that row the cost is 3x1^2 - 6x1,
    # so x1 sits at its bound 0.9 and x0 = 0.1
    assert r["dense"][:2] == [0, 1] and abs(r["dense"][2] - 0.1) <= 1e-8 and abs(r["dense"][3] - 0.9) <= 1e-8
    assert r["infeasible_init"][0] == 0 and r["infeasible_init"][1] == -3 and r["infeasible_init"][2] == 0
    assert r["infeasible_init"][3] == 1 and abs(r["infeasible_init"][4] - 1.0) <= 1e-8 and abs(r["infeasible_init"][5]) <= 1e-8
    assert r["big"][:2] == [0, 1] and abs(r["big"][2] - 0.25) <= 1e-8 and abs(r["big"][3] - 0.25) <= 1e-8
    assert r["toobig"][0] == -3 and r["toobig"][1] == -10 and r["toobig"][2] == -3  # ASIF_HIP_EUNSUPPORTED / STATUS_UNSOLVED
    # one agent's control step at the plug-in (update, solve, read back): solved, and of the order of a launch +
    # synchronisation -- no staging copies (the value itself is printed for the record, not asserted tightly)
    assert r["latency_us"][0] == 1 and r["latency_us"][1] < 500.0
    print("QPWrapperHip latency per solve [us]:", r["latency_us"][1])


def test_explicit_class_two_input_model_and_reduced_row_budget(hip, oracle):
    """ASIF::ASIF bound to a model other than the double integrator (two inputs) and with npSSmax = 2 of 4 rows:
    filterBatch() (fused kernel of the bound model) equals filter() (host rows + QPWrapperHip) before and after
    updateOptions() -- round 2 rebuilt the device options from the double integrator's defaults whatever was bound
    and never passed npSSmax at bind time (EINVAL).  Both against the oracle's exact optimum as well."""
    exe = os.path.join(HOST, "explicit_variants")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", HOST, "-s"])
    n = 96
    out = subprocess.run([exe, str(n)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l.split(",") for l in out.stdout.strip().split("\n")[1:]]
    seen = set()
    for name, nu in (("planar2", 2), ("di_keep2", 1)):
        for phase in (0, 1):
            rows = np.array([[float(v) for v in l[2:]] for l in lines if l[0] == name and int(l[1]) == phase])
            assert rows.shape == (n, 3 + 2 * nu + 2), (name, phase, rows.shape)
            rc1, rcb = rows[:, 1].astype(int), rows[:, 2].astype(int)
            u1, ub = rows[:, 3:3 + nu], rows[:, 3 + nu:3 + 2 * nu]
            assert np.array_equal(rc1, rcb), (name, phase)
            assert set(rc1.tolist()) <= {1, -1} and (rc1 == 1).sum() >= n // 4
            ok = rc1 == 1
            assert np.abs(u1[ok] - ub[ok]).max() <= 1e-6, (name, phase)
            assert np.all(ub[~ok] == 7.0) and np.all(u1[~ok] == 7.0)  # untouched on failure, both paths
            assert np.abs(rows[ok, -2] - rows[ok, -1]).max() <= 1e-9  # pinned relaxation variable, same value
            assert np.abs(rows[ok, -1] - (5.0 if phase == 0 else 2.0)).max() <= 1e-9
            seen.add((name, phase))
    assert len(seen) == 4
