"""The oracle's restatement of the reference's USE_ODEINT build (dopri5 with dense output, src/asif_implicit.cpp:427-460;
oracle/or_assembly.c dopri5_*) checked against an independent integration: classical RK4 at a tenth of the sample
spacing in numpy, own closed loop of examples/InvertedPendulum_Implicit.cpp.  Boost.odeint is absent, so odeint's own
numbers cannot be pinned; what CAN be checked is that the restated tableau, continuous extension and controller
integrate the same ODE to the requested accuracy."""
import numpy as np

LB, UB, R_SAT = -1.5, 1.5, 0.1


def _sat(u):
    rng, mid = UB - LB, 0.5 * (UB + LB)
    uc = 2.0 * (u - mid) / rng
    bev = R_SAT * np.tan(np.pi / 8)
    start, stop, yc = 1 - np.cos(np.pi / 4) * bev, 1 + bev, 1 - R_SAT
    if abs(uc) <= start:
        return u
    if abs(uc) >= stop:
        return UB if uc > 0 else LB
    s = np.sqrt(R_SAT ** 2 - (abs(uc) - stop) ** 2)
    return np.sign(uc) * 0.5 * (s + yc) * rng + mid


def _f(x):
    return np.array([x[1], np.sin(x[0]) + _sat(-3.0 * x[0] - 3.0 * x[1])])


def _rk4(x0, T, dt):
    x = np.array(x0, dtype=float)
    for _ in range(int(round(T / dt))):
        k1 = _f(x); k2 = _f(x + 0.5 * dt * k1); k3 = _f(x + 0.5 * dt * k2); k4 = _f(x + dt * k3)
        x = x + dt / 6 * (k1 + 2 * k2 + 2 * k3 + k4)
    return x


def test_dopri5_trajectory_end_matches_rk4(oracle):
    model, variant = oracle.MODEL_IP, oracle.VAR_IMPLICIT
    x = np.array([[0.1, 0.0], [0.7, -0.4], [-1.2, 0.9], [1.4, 1.3]])
    P = np.array([[1.25, 0.25], [0.25, 0.25]])
    ends = [_rk4(xi, 5.0, 1e-3) for xi in x]
    hb_ref = np.array([0.05 - e @ P @ e for e in ends])
    errs = {}
    for tol in (1e-6, 1e-9):
        o = oracle.default_options(model, variant)
        o.integrator = 1
        o.backTrajAbsTol = o.backTrajRelTol = tol
        A, b, code, _ = oracle.assemble_batch(model, variant, o, x)
        hb = np.array([A[i].reshape(3, 41).T[40, 2] for i in range(len(x))])  # backup-set value at the trajectory end
        errs[tol] = np.abs(hb - hb_ref).max()
    assert errs[1e-6] <= 2e-6 and errs[1e-9] <= 5e-9, errs
    # forward Euler at the example's dt = 1e-3 is O(dt) away from both
    o = oracle.default_options(model, variant)
    A, b, code, _ = oracle.assemble_batch(model, variant, o, x)
    hb_e = np.array([A[i].reshape(3, 41).T[40, 2] for i in range(len(x))])
    assert 1e-7 < np.abs(hb_e - hb_ref).max() < 1e-3


def test_dopri5_rows_converge_with_the_tolerance(oracle):
    model, variant = oracle.MODEL_IP, oracle.VAR_IMPLICIT
    x, _ = oracle.make_batch(3, 24)
    rows = {}
    for tol in (1e-5, 1e-7, 1e-10):
        o = oracle.default_options(model, variant)
        o.integrator = 1
        o.backTrajAbsTol = o.backTrajRelTol = tol
        rows[tol] = oracle.assemble_batch(model, variant, o, x)[:2]
    d1 = max(np.abs(rows[1e-5][k] - rows[1e-10][k]).max() for k in (0, 1))
    d2 = max(np.abs(rows[1e-7][k] - rows[1e-10][k]).max() for k in (0, 1))
    assert d2 < d1 and d2 <= 1e-5 and d1 <= 1e-3, (d1, d2)


def test_dopri5_failure_ends_and_fails_the_filter(oracle):
    """odeint throws when a step cannot be made (the reference has no handler); a NaN error estimate is dropped by the
    controller's max() and would be ACCEPTED.  The restatement marks the trajectory failed instead: later samples are
    NaN, the rows non-finite, and filter() fails like a failed solve (rc -1, backup controller) -- and the call ends."""
    model, variant = oracle.MODEL_IP, oracle.VAR_IMPLICIT
    o = oracle.default_options(model, variant)
    o.integrator = 1
    x = np.array([[np.nan, 0.0], [0.2, np.nan], [1.7e308, 1.7e308], [0.1, 0.0]])
    u = np.zeros((4, 1))
    A, b, code, _ = oracle.assemble_batch(model, variant, o, x)
    assert not np.isfinite(b[:3]).all(axis=1).any() and np.isfinite(b[3]).all() and np.isfinite(A[3]).all()
    ua, rl, rc = oracle.filter_batch(model, variant, o, x, u, uact_init=np.full((4, 1), 7.0))
    assert list(rc) == [-1, -1, -1, 1]


def test_tb_with_dopri5_hits_at_the_first_sample_after_the_continuous_hitting_time(oracle):
    """ASIFimplicitTB in the USE_ODEINT build (src/asif_implicit_tb.cpp:431-463): the samples are dense output at
    t = i backTrajDt, the hit is the first SAMPLE inside the backup set -- i.e. idxHit = ceil(tau / dt) for the continuous
    hitting time tau, which tests/test_oracle_tb_rows_fd.py computes independently (RK4 + bisection, the double
    integrator's disc); forward Euler at dt = 1e-3 can be a sample or two off."""
    from test_oracle_tb_rows_fd import _di_hit
    model, variant = oracle.CONFIGS[12]
    o = oracle.default_options(model, variant)
    o.integrator = 1
    o.backTrajAbsTol = o.backTrajRelTol = 1e-10
    x, _ = oracle.make_batch(12, 200)
    A, b, code, diag = oracle.assemble_batch(model, variant, o, x)
    idx = [k for k in np.where(code == 1)[0] if 0.05 < diag[k, 0] < 2.0][:20]
    assert len(idx) >= 15
    exact = 0
    for k in idx:
        tau = _di_hit(x[k])
        want = int(np.ceil(tau / 1e-3 - 1e-6))
        assert abs(int(diag[k, 2]) - want) <= 1, (diag[k, 2], tau)  # (a hitting time within 1e-9 of a sample may round either way)
        exact += int(diag[k, 2]) == want
        assert abs(diag[k, 0] - diag[k, 2] * 1e-3) <= 1e-12  # TTS_ = idxHit * backTrajDt: stamped, not accumulated (:451)
    assert exact >= len(idx) - 1
