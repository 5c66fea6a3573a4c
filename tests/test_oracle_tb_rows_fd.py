"""Independent check of the one assembly with no reference-derived pin at all: the time-to-safety row and the
orthogonality row of ASIFimplicitTB (src/asif_implicit_tb.cpp:588-641) as the oracle restates them.

Both rows are gradients THROUGH THE HITTING MAP x0 -> phi_tau(x0)(x0) of the backup trajectory:
    d tau / d x0 = -(grad h_B . Q) / (grad h_B . fCL)            (implicit function theorem at h_B = 0)
so the test recomputes the hitting time and the angle at the hit from scratch -- its own closed loop of
examples/InvertedPendulum_ImplicitTB.cpp (velocity-tracking controller, bevelled saturation, pendulum dynamics)
integrated with RK4 (fourth order at the same step: far below the O(dt) of the oracle's Euler), the crossing located by bisection on a dense output -- and differentiates
them by central differences along g.  The oracle's rows use forward Euler at dt = 1e-3 and evaluate at the first
SAMPLE inside the backup set, so agreement is expected to O(dt): rtol 3e-2.  A wrong sign, a missing 1/cos factor or a
transposed sensitivity would be off by O(1).

The orthogonality row is NOT the true derivative in the reference: it pushes the state through
J = Q - fCL (grad h_B Q) without the 1 / (grad h_B . fCL) of the hitting map and books the derivative of |fCL| under
|grad h_B| (src/asif_implicit_tb.cpp:601-641; SURVEY App. B "preserve").  For a half-space backup set (Hessian 0,
|grad h_B| = 1) that formula collapses to
    row = (1 - grad h_B . fCL) / |fCL|  *  grad h_B' DfCL J ,
which the second test evaluates with a sensitivity Q obtained by differencing this file's own flow at the hit time."""
import numpy as np

LB, UB, R_SAT = -1.5, 1.5, 0.1


def _sat(u, LB=LB, UB=UB):
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


def _fcl(x):
    u = 10.0 * (np.pi / 10.0 - x[1])
    return np.array([x[1], np.sin(x[0]) + _sat(u)])


def _hb(x):
    return x[0] - np.pi / 2 + 0.1


def _hit(x0, dt=1e-3, tmax=12.0):
    """(tau, state at the crossing) of the continuous backup flow."""
    x, t = np.array(x0, dtype=float), 0.0
    assert _hb(x) < 0
    while t < tmax:
        k1 = _fcl(x); k2 = _fcl(x + 0.5 * dt * k1); k3 = _fcl(x + 0.5 * dt * k2); k4 = _fcl(x + dt * k3)
        xn = x + dt / 6 * (k1 + 2 * k2 + 2 * k3 + k4)
        if _hb(xn) >= 0:
            lo, hi = 0.0, dt  # bisection on a cubic-Hermite-free substitute: re-integrate the fraction of the step
            for _ in range(50):
                m = 0.5 * (lo + hi)
                k1 = _fcl(x); k2 = _fcl(x + 0.5 * m * k1); k3 = _fcl(x + 0.5 * m * k2); k4 = _fcl(x + m * k3)
                xm = x + m / 6 * (k1 + 2 * k2 + 2 * k3 + k4)
                if _hb(xm) >= 0:
                    hi = m
                else:
                    lo = m
            return t + hi, xm
        x, t = xn, t + dt
    raise AssertionError("backup set not reached")


def _flow(x0, T, dt=1e-3):
    x = np.array(x0, dtype=float)
    n = int(round(T / dt))
    for _ in range(n):
        k1 = _fcl(x); k2 = _fcl(x + 0.5 * dt * k1); k3 = _fcl(x + 0.5 * dt * k2); k4 = _fcl(x + dt * k3)
        x = x + dt / 6 * (k1 + 2 * k2 + 2 * k3 + k4)
    return x


def _ortho(xh):
    f = _fcl(xh)  # grad h_B = (1, 0)
    return f[0] / np.hypot(f[0], f[1])


def test_tts_and_ortho_rows_are_derivatives_through_the_hitting_map(oracle):
    model, variant = oracle.MODEL_IP_TB, oracle.VAR_TB
    o = oracle.default_options(model, variant)
    d = oracle.dims(model, variant, o)
    assert (d.nv, d.nc) == (2, 18)
    x, _ = oracle.make_batch(8, 96)
    A, b, code, diag = oracle.assemble_batch(model, variant, o, x)
    idx = np.where(code == 1)[0]
    assert len(idx) >= 12
    g = np.array([0.0, 1.0])
    f_at = lambda xx: np.array([xx[1], np.sin(xx[0])])
    eps = 1e-5
    checked = 0
    for k in idx[:12]:
        Am = A[k].reshape(2, 18).T
        tau0, xh0 = _hit(x[k])
        # the oracle's own numbers first: hitting time (sample grid) and cosine at the hit
        assert abs(diag[k, 0] - tau0) <= 2e-3 + 1e-3 * tau0, (diag[k, 0], tau0)  # Euler at dt 1e-3 + the sample grid
        assert abs(diag[k, 1] - _ortho(xh0)) <= 5e-3
        taup, xhp = _hit(x[k] + eps * g)
        taum, xhm = _hit(x[k] - eps * g)
        dtau_g = (taup - taum) / (2 * eps)
        dort_g = (_ortho(xhp) - _ortho(xhm)) / (2 * eps)
        # TTS row: h = T - tau  ->  Lgh = -d tau . g ;  b = -Lfh - relaxTTS h with Lfh = -d tau . f
        np.testing.assert_allclose(Am[16, 0], -dtau_g, rtol=3e-2, atol=2e-4)
        del dort_g  # the reference's orthogonality row is not this derivative: see the module docstring and the next test
        # the same two gradients contracted with f(x0) sit in b
        tf_p, xf_p = _hit(x[k] + eps * f_at(x[k]))
        tf_m, xf_m = _hit(x[k] - eps * f_at(x[k]))
        dtau_f = (tf_p - tf_m) / (2 * eps)
        hreach = o.backTrajHorizon - diag[k, 0]
        np.testing.assert_allclose(b[k, 16], dtau_f - o.relaxTTS * hreach, rtol=3e-2, atol=5e-3)
        checked += 1
    assert checked >= 10


def test_ortho_row_is_the_reference_formula(oracle):
    model, variant = oracle.MODEL_IP_TB, oracle.VAR_TB
    o = oracle.default_options(model, variant)
    x, _ = oracle.make_batch(8, 96)
    A, b, code, diag = oracle.assemble_batch(model, variant, o, x)
    idx = np.where(code == 1)[0]
    g = np.array([0.0, 1.0])
    eps = 1e-6
    checked = 0
    for k in idx[:24]:
        th = diag[k, 0]  # time of the first sample inside the backup set
        xh = _flow(x[k], th)
        u = 10.0 * (np.pi / 10.0 - xh[1])
        uc = 2.0 * u / (UB - LB)
        bev = R_SAT * np.tan(np.pi / 8)
        if 1 - np.cos(np.pi / 4) * bev < abs(uc) < 1 + bev:
            continue  # on a bevel of the saturation the reference's DuSat convention enters: keep to the plain regions
        dsat = 1.0 if abs(uc) < 1 else 0.0
        DfCL = np.array([[0.0, 1.0], [np.cos(xh[0]), -10.0 * dsat]])
        Q = np.stack([(_flow(x[k] + eps * e, th) - _flow(x[k] - eps * e, th)) / (2 * eps) for e in np.eye(2)], axis=1)
        f = _fcl(xh)
        gh = np.array([1.0, 0.0])
        J = Q - np.outer(f, gh @ Q)
        row = (1.0 - gh @ f) / np.linalg.norm(f) * (gh @ DfCL @ J)
        Am = A[k].reshape(2, 18).T
        np.testing.assert_allclose(Am[17, 0], row @ g, rtol=5e-2, atol=2e-4)
        checked += 1
    assert checked >= 8


# ---- the double integrator of examples/DoubleIntegrator_implicit_tb.cpp: a CURVED backup set (disc of radius 0.01),
# so the hit moves along the boundary with x0 and the Hessian term of the rows is live

def _di_fcl(x):
    return np.array([x[1], _sat(-10.0 * x[0] - 20.0 * x[1], -1.0, 1.0)])


def _di_hit(x0, dt=1e-3, tmax=2.2):
    hb = lambda xx: 1e-4 - xx @ xx
    rk = lambda xx, h: (lambda k1: (lambda k2: (lambda k3: xx + h / 6 * (k1 + 2 * k2 + 2 * k3 + _di_fcl(xx + h * k3)))(
        _di_fcl(xx + 0.5 * h * k2)))(_di_fcl(xx + 0.5 * h * k1)))(_di_fcl(xx))
    x, t = np.array(x0, dtype=float), 0.0
    assert hb(x) < 0
    while t < tmax:
        xn = rk(x, dt)
        if hb(xn) >= 0:
            lo, hi = 0.0, dt
            for _ in range(50):
                m = 0.5 * (lo + hi)
                if hb(rk(x, m)) >= 0:
                    hi = m
                else:
                    lo = m
            return t + hi
        x, t = xn, t + dt
    raise AssertionError("backup set not reached")


def test_tts_row_with_a_curved_backup_set(oracle):
    model, variant = oracle.CONFIGS[12]
    o = oracle.default_options(model, variant)
    d = oracle.dims(model, variant, o)
    assert (d.nv, d.nc, d.npBT) == (2, 18, 2101)
    x, _ = oracle.make_batch(12, 160)
    A, b, code, diag = oracle.assemble_batch(model, variant, o, x)
    idx = [k for k in np.where(code == 1)[0] if 0.2 < diag[k, 0] < 1.9]  # away from the horizon's end and from t = 0
    assert len(idx) >= 12
    g = np.array([0.0, 1.0])
    eps = 1e-6
    checked = 0
    for k in idx[:16]:
        Am = A[k].reshape(2, 18).T
        tau0 = _di_hit(x[k])
        assert abs(diag[k, 0] - tau0) <= 2e-3 + 2e-3 * tau0, (diag[k, 0], tau0)
        dtau_g = (_di_hit(x[k] + eps * g) - _di_hit(x[k] - eps * g)) / (2 * eps)
        np.testing.assert_allclose(Am[16, 0], -dtau_g, rtol=3e-2, atol=2e-4)
        f0 = np.array([x[k][1], 0.0])
        dtau_f = (_di_hit(x[k] + eps * f0) - _di_hit(x[k] - eps * f0)) / (2 * eps)
        hreach = o.backTrajHorizon - diag[k, 0]
        np.testing.assert_allclose(b[k, 16], dtau_f - o.relaxTTS * hreach, rtol=3e-2, atol=5e-3)
        checked += 1
    assert checked >= 12
