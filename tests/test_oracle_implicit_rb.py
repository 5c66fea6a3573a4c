"""Oracle restatement of ASIFimplicitRB (src/asif_implicit_robust.cpp, include/asif_learning_utils.h):
interval safety margins pinned on the reference's libaffa (golden vectors), the zero-order hold of the
backup input and the learned residual checked against an independent numpy restatement, and the
reductions to ASIFimplicit the class admits."""
import json
import math
import os

import numpy as np

from asif_amd import workloads

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _rb_options(oracle, learning=True, x_unc=workloads.RB_X_UNC):
    m, v = oracle.CONFIGS[10]
    o = oracle.default_options(m, v)
    o.x_unc[0], o.x_unc[1] = x_unc
    L = None
    if learning:
        L = oracle.Learning.from_dict(workloads.make_learning())
        o.set_learning(L)
    return m, v, o, L


def test_interval_safety_lower_ends_equal_libaffa_golden(oracle):
    with open(os.path.join(GOLD, "affa_box_safety_interval.json")) as f:
        cases = json.load(f)["cases"]
    assert len(cases) >= 100
    for c in cases:
        model = oracle.MODEL_IP if c["model"] == "pendulum" else 5
        o = oracle.default_options(model, oracle.VAR_IMPLICIT_RB)
        o.x_unc[0], o.x_unc[1] = c["x_unc"]
        h = oracle.rb_safety_lo(model, o, np.array(c["x"]))
        assert h.tolist() == c["h_lo"], c  # bit-exact


def test_reduces_to_implicit_without_hold_uncertainty_and_learning(oracle):
    m, v, o, _ = _rb_options(oracle, learning=False, x_unc=(0.0, 0.0))
    x, u = oracle.make_batch(10, 48)
    o.backContDt = o.backTrajDt  # the hold is refreshed at every step
    A0, b0, _, _ = oracle.assemble_batch(m, oracle.VAR_IMPLICIT, o, x)
    A1, b1, _, _ = oracle.assemble_batch(m, v, o, x)
    assert np.array_equal(A0, A1) and np.array_equal(b0, b1)
    ua0, rl0, rc0 = oracle.filter_batch(m, oracle.VAR_IMPLICIT, o, x, u)
    ua1, rl1, rc1 = oracle.filter_batch(m, v, o, x, u)
    assert np.array_equal(rc0, rc1) and np.array_equal(ua0, ua1) and np.array_equal(rl0, rl1, equal_nan=True)
    # and the DI implicit model too
    o5 = oracle.default_options(5, v)
    o5.backContDt = o5.backTrajDt
    x5, _ = oracle.make_batch(9, 64)
    A0, b0, _, _ = oracle.assemble_batch(5, oracle.VAR_IMPLICIT, o5, x5)
    A1, b1, _, _ = oracle.assemble_batch(5, v, o5, x5)
    assert np.array_equal(A0, A1) and np.array_equal(b0, b1)


def _pendulum_rb_numpy(o, x0):
    """Independent restatement of the RB trajectory for the pendulum: forward Euler with the backup input
    sampled when t >= t_last + backContDt - 1e-4 (src/asif_implicit_robust.cpp:891-903), t = i*dt stamped on
    the rhs of sample i-1 (:567).  Returns the states [npBT, 2] and sensitivities [npBT, 2, 2]."""
    dt, T = o.backTrajDt, o.backTrajHorizon
    n = int(round(T / dt) + 1)
    K = np.array([-3.0, -3.0])
    lb, ub, r = o.lb[0], o.ub[0], o.satSharpness
    bevelL = r * math.tan(math.pi / 8)
    bstart, bstop = 1 - math.cos(math.pi / 4) * bevelL, 1 + bevelL

    def sat(u):
        rng, mid = ub - lb, (ub + lb) / 2
        uc = 2 * (u - mid) / rng
        if uc >= bstop:
            return ub, 0.0
        if uc <= -bstop:
            return lb, 0.0
        if -bstart <= uc <= bstart:
            return u, 1.0
        if uc > bstart:
            s = math.sqrt(r * r - (uc - bstop) ** 2)
            return 0.5 * (s + 1 - r) * rng + mid, (bstop - uc) / s
        s = math.sqrt(r * r - (uc + bstop) ** 2)
        return 0.5 * (-s - (1 - r)) * rng + mid, (bstop + uc) / s

    xs = np.zeros((n, 2))
    Qs = np.zeros((n, 2, 2))
    xs[0], Qs[0] = x0, np.eye(2)
    t_last, u_hold, samples = 0.0, 0.0, []
    for i in range(1, n):
        x, Q = xs[i - 1], Qs[i - 1]
        t = i * dt
        if t <= dt:
            t_last = -1.0
        if t >= t_last + o.backContDt - 0.0001:
            u_hold, t_last = float(K @ x), t
            samples.append(i)
        us, dus = sat(u_hold)
        f = np.array([x[1], math.sin(x[0]) + us])
        Df = np.array([[0.0, 1.0], [math.cos(x[0]), 0.0]]) + np.outer([0.0, 1.0], dus * K)
        xs[i] = x + dt * f
        Qs[i] = Q + dt * (Df @ Q)
    return xs, Qs, samples


def test_zero_order_hold_trajectory_matches_numpy_restatement(oracle):
    m, v, o, _ = _rb_options(oracle, learning=False, x_unc=(0.0, 0.0))
    d = oracle.dims(m, v, o)
    for x0 in (np.array([0.4, -0.3]), np.array([-1.2, 1.1])):
        xs, Qs, samples = _pendulum_rb_numpy(o, x0)
        assert samples[:4] == [1, 11, 21, 31] and len(samples) == 500  # 10 Euler steps per hold
        A, b, code = oracle.assemble(m, v, o, x0)
        assert code == 1
        A = A.reshape(d.nv, d.nc)
        crit = oracle.last_crit_idx()
        # safe rows: [Dh_i Q_k g | h_i(x_k)], b = -Dh_i Q_k f at the critical samples the oracle picked
        Dh = np.array([[-1.0, 0.0], [1.0, 0.0], [0.0, 1.0], [0.0, -1.0]])
        f0, g0 = np.array([x0[1], math.sin(x0[0])]), np.array([0.0, 1.0])
        hmin = np.minimum(np.minimum(math.pi - xs[:, 0], xs[:, 0] + math.pi),
                          np.minimum(xs[:, 1] + math.pi, math.pi - xs[:, 1]))
        assert abs(hmin[crit[0]] - hmin.min()) <= 1e-12
        for k, s in enumerate(crit):
            h = np.array([math.pi - xs[s, 0], xs[s, 0] + math.pi, xs[s, 1] + math.pi, math.pi - xs[s, 1]])
            DhQ = Dh @ Qs[s]
            assert np.abs(A[1, 4 * k:4 * k + 4] - h).max() <= 1e-11
            assert np.abs(A[0, 4 * k:4 * k + 4] - DhQ @ g0).max() <= 1e-9
            assert np.abs(b[4 * k:4 * k + 4] + DhQ @ f0).max() <= 1e-9
        # backup row
        P = np.array([[1.25, 0.25], [0.25, 0.25]])
        xT, QT = xs[-1], Qs[-1]
        assert abs(A[2, 40] - (0.05 - xT @ P @ xT)) <= 1e-11
        assert abs(A[0, 40] - (-2 * P @ xT) @ QT @ g0) <= 1e-9
        # the hold matters: the plain implicit rows differ
        A0, _, _ = oracle.assemble(m, oracle.VAR_IMPLICIT, o, x0)
        assert np.abs(A0.reshape(d.nv, d.nc) - A).max() > 1e-6


def _mlp(w, net, vin):
    W1 = w[f"w_1_{net}"].reshape(-1, w[f"d_{net}_hidden"]).T  # column-major [hidden x in]
    W2 = w[f"w_2_{net}"].reshape(-1, w[f"d_{net}_hidden_2"]).T
    W3 = w[f"w_3_{net}"].reshape(-1, w[f"d_{net}_out"]).T
    a1 = np.maximum(0.0, W1 @ vin + w[f"b_1_{net}"])
    a2 = np.maximum(0.0, W2 @ a1 + w[f"b_2_{net}"])
    return W3 @ a2 + w[f"b_3_{net}"]


def test_learned_residual_and_uncertainty_touch_only_what_the_reference_touches(oracle):
    m, v, o, L = _rb_options(oracle)
    w = workloads.make_learning()
    d = oracle.dims(m, v, o)
    x, _ = oracle.make_batch(10, 24)
    o_plain = oracle.default_options(m, v)  # hold only
    for xi in x:
        A, b, code = oracle.assemble(m, v, o, xi)
        dh, lf, lg = oracle.rb_last_learning()
        A0, b0, _ = oracle.assemble(m, v, o_plain, xi)
        A, A0 = A.reshape(d.nv, d.nc), A0.reshape(d.nv, d.nc)
        vin = np.concatenate([xi, dh[:2]])
        assert abs(_mlp(w, "drift", vin)[0] - lf) <= 1e-13 and abs(_mlp(w, "act", vin)[0] - lg[0]) <= 1e-13
        # include/asif_learning_utils.h:149-154: Lfh[0] and Lgh[0] only
        assert abs((A[0, 0] - A0[0, 0]) - lg[0]) <= 1e-13 and abs((b0[0] - b[0]) - lf) <= 1e-13
        assert np.array_equal(A[0, 1:], A0[0, 1:]) and np.array_equal(b[1:], b0[1:])
        # x_unc: the safe rows' margins drop by the radius of the coordinate they bound; backup row untouched
        dh_col = A0[1, :40] - A[1, :40]
        want = np.tile([o.x_unc[0], o.x_unc[0], o.x_unc[1], o.x_unc[1]], 10)
        assert np.abs(dh_col - want).max() <= 1e-12
        assert np.array_equal(A[2], A0[2]) and A[1, 40] == 0.0
        # Dh_index_ = first column of Dh_SS * Q at the most critical sample = rows 0/1 of column 0: -Q00, +Q00
        assert dh[0] == -dh[1]


def test_n_debug_selects_the_sample_feeding_the_network(oracle):
    m, v, o, L = _rb_options(oracle)
    xi = np.array([0.7, -0.9])
    A, b, _ = oracle.assemble(m, v, o, xi)
    dh_crit = oracle.rb_last_learning()[0].copy()
    o.n_debug = 250
    A1, b1, _ = oracle.assemble(m, v, o, xi)
    dh_dbg = oracle.rb_last_learning()[0].copy()
    assert not np.array_equal(dh_crit, dh_dbg) and A1[0] != A[0]
    xs, Qs, _ = _pendulum_rb_numpy(o, xi)
    assert abs(dh_dbg[0] + Qs[250][0, 0]) <= 1e-9
    for bad in (-7, 5000, 123456):  # src/asif_implicit_robust.cpp:298-303: outside (-1, npBT-1) -> -1
        o.n_debug = bad
        A2, b2, _ = oracle.assemble(m, v, o, xi)
        assert np.array_equal(A2, A) and np.array_equal(b2, b)


def test_device_algorithm_emulation_decides_rb_like_exact(oracle):
    m, v, o, L = _rb_options(oracle)
    x, u = oracle.make_batch(10, 96)
    ua, rl, rc = oracle.filter_batch(m, v, o, x, u, oracle.SOLVER_EXACT)
    s = oracle.admm_settings(max_iter=4000, polish=2, check_termination=2, adaptive_rho_interval=2,
                             eps_abs=1e-8, eps_rel=1e-8, reduced_kkt=1, scaling_pow2=1, scaling=2)
    ua2, rl2, rc2 = oracle.filter_batch(m, v, o, x, u, oracle.SOLVER_ADMM, s)
    assert np.array_equal(rc, rc2)
    assert set(np.unique(rc)) <= {1, -1} and (rc == 1).sum() > 40
    assert np.nanmax(np.abs(ua - ua2)) <= 1e-9
    bad = rc == -1  # src/asif_implicit_robust.cpp:427-433: saturated backup controller
    if bad.any():
        want = np.clip(x[bad] @ np.array([-3.0, -3.0]), o.lb[0], o.ub[0])
        assert np.abs(ua[bad, 0] - want).max() <= 1e-15


def test_plain_implicit_class_carries_the_same_learned_residual(oracle):
    """include/asif_implicit.h:23,33,125 + src/asif_implicit.cpp:585-588: ASIFimplicit has n_debug / use_learning /
    learning_data_ as well."""
    m = oracle.MODEL_IP
    o = oracle.default_options(m, oracle.VAR_IMPLICIT)
    x, _ = oracle.make_batch(3, 16)
    A0, b0, _, _ = oracle.assemble_batch(m, oracle.VAR_IMPLICIT, o, x)
    w = workloads.make_learning()
    o.set_learning(oracle.Learning.from_dict(w))
    d = oracle.dims(m, oracle.VAR_IMPLICIT, o)
    for i, xi in enumerate(x):
        A, b, _ = oracle.assemble(m, oracle.VAR_IMPLICIT, o, xi)
        dh, lf, lg = oracle.rb_last_learning()
        vin = np.concatenate([xi, dh[:2]])
        assert abs(_mlp(w, "drift", vin)[0] - lf) <= 1e-13 and abs(_mlp(w, "act", vin)[0] - lg[0]) <= 1e-13
        dA = A.reshape(d.nv, d.nc) - A0[i].reshape(d.nv, d.nc)
        assert abs(dA[0, 0] - lg[0]) <= 1e-13 and abs((b0[i][0] - b[0]) - lf) <= 1e-13
        dA[0, 0] = 0.0
        assert np.all(dA == 0.0) and np.array_equal(b[1:], b0[i][1:])
