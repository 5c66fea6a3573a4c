"""Closed-loop rollout (T x [filter + plant Euler step] per launch, examples/DoubleIntegrator.cpp:81-116) on the GPU.

A free-running comparison against a CPU closed loop is meaningless near the states where the filter's gain
blows up (a 1e-16 difference in u is amplified step after step), so every step is checked on the states the
device itself went through (logged): the oracle's exact filter on x_t must give the logged u_t and rc_t, and the
logged x_{t+1} must be the plant's Euler step of (x_t, u_t) bit for bit.  The fused rollout must also equal T
single-step launches bitwise.
"""
import numpy as np
import pytest
import torch

import gpu_util

pytestmark = pytest.mark.gpu


def _rollout(hip, B, T, dt, x, udes, solver=None):
    flt = hip.Filter(hip.MODEL_DOUBLE_INTEGRATOR, hip.EXPLICIT, solver=solver)
    dev = torch.device("cuda:0")
    tx, tu = torch.from_numpy(x.copy()).to(dev), torch.from_numpy(udes).to(dev)
    uact = torch.zeros((1, B), dtype=torch.float64, device=dev)
    relax = torch.zeros((1, B), dtype=torch.float64, device=dev)
    nfail = torch.zeros(B, dtype=torch.int32, device=dev)
    xlog = torch.zeros((T, 2, B), dtype=torch.float64, device=dev)
    ulog = torch.zeros((T, 1, B), dtype=torch.float64, device=dev)
    rclog = torch.zeros((T, B), dtype=torch.int32, device=dev)
    flt.rollout(T, dt, tx, tu, uact, relax, nfail, xlog, ulog, rclog)
    torch.cuda.synchronize()
    out = dict(x=tx.cpu().numpy(), uact=uact.cpu().numpy(), relax=relax.cpu().numpy(), nfail=nfail.cpu().numpy(),
               xlog=xlog.cpu().numpy(), ulog=ulog.cpu().numpy(), rclog=rclog.cpu().numpy())
    flt.close()
    return out


def test_every_step_matches_oracle_on_logged_states(hip, oracle):
    from asif_amd import workloads
    B, T, dt = 2048, 40, 0.001
    x, u = workloads.make_batch(2, B)
    out = _rollout(hip, B, T, dt, x, u)
    assert np.array_equal(out["xlog"][0], x)
    model, variant = oracle.CONFIGS[2]
    o = oracle.default_options(model, variant)
    prev_u = np.zeros(B)
    for t in range(T):
        xt = out["xlog"][t]
        ua, rl, rc = oracle.filter_batch(model, variant, o, np.ascontiguousarray(xt.T), np.ascontiguousarray(u.T),
                                         oracle.SOLVER_EXACT, None, 8, uact_init=prev_u[:, None])
        assert np.array_equal(out["rclog"][t], rc), f"step {t}: rc mismatches {(out['rclog'][t] != rc).sum()}"
        # a failed call keeps the previous input (uact_init carries it into the oracle call the same way)
        assert np.abs(out["ulog"][t, 0] - ua[:, 0]).max() <= 1e-6
        ut = out["ulog"][t, 0]
        # plant step of examples/DoubleIntegrator.cpp:96-110 with f = (x1, 0), g = (0, 1), no FMA contraction
        x0n = xt[0] + dt * ((0.0 + xt[1]) + ut * 0.0)
        x1n = xt[1] + dt * ((0.0 + 0.0) + ut * 1.0)
        nxt = out["xlog"][t + 1] if t + 1 < T else out["x"]
        assert np.array_equal(nxt[0], x0n) and np.array_equal(nxt[1], x1n)
        prev_u = ut
    assert np.array_equal(out["nfail"], (out["rclog"] != 1).sum(axis=0))
    assert np.array_equal(out["uact"][0], out["ulog"][-1, 0])


def test_rollout_equals_single_step_launches(hip):
    from asif_amd import workloads
    B, T, dt = 4096, 25, 0.002
    x, u = workloads.make_batch(2, B, first=5000)
    out = _rollout(hip, B, T, dt, x, u, solver=hip.default_solver(warm_start=0))  # cold working sets: bitwise
    warm = _rollout(hip, B, T, dt, x, u)                                          # default: warm working sets
    assert np.array_equal(warm["rclog"], out["rclog"]) and np.abs(warm["ulog"] - out["ulog"]).max() <= 1e-9
    flt = hip.Filter(hip.MODEL_DOUBLE_INTEGRATOR, hip.EXPLICIT)
    dev = torch.device("cuda:0")
    tx, tu = torch.from_numpy(x.copy()).to(dev), torch.from_numpy(u).to(dev)
    uact = torch.zeros((1, B), dtype=torch.float64, device=dev)
    relax = torch.zeros((1, B), dtype=torch.float64, device=dev)
    rc = torch.zeros(B, dtype=torch.int32, device=dev)
    for t in range(T):
        flt.filter(tx, tu, uact, relax, rc)
        torch.cuda.synchronize()
        assert np.array_equal(rc.cpu().numpy(), out["rclog"][t])
        assert np.array_equal(uact.cpu().numpy()[0], out["ulog"][t, 0])
        xt, ut = tx.cpu().numpy(), uact.cpu().numpy()[0]
        xn = np.stack([xt[0] + dt * ((0.0 + xt[1]) + ut * 0.0), xt[1] + dt * ((0.0 + 0.0) + ut * 1.0)])
        tx = torch.from_numpy(xn).to(dev)
    assert np.array_equal(tx.cpu().numpy(), out["x"])
    flt.close()


def test_rollout_rejects_other_variants(hip):
    flt = hip.Filter(hip.MODEL_INVERTED_PENDULUM_ROBUST, hip.ROBUST)
    dev = torch.device("cuda:0")
    x = torch.zeros((2, 4), dtype=torch.float64, device=dev)
    u = torch.zeros((1, 4), dtype=torch.float64, device=dev)
    r = torch.zeros((1, 4), dtype=torch.float64, device=dev)
    n = torch.zeros(4, dtype=torch.int32, device=dev)
    with pytest.raises(hip.AsifHipError):
        flt.rollout(3, 0.001, x, u, u.clone(), r, n)
    flt.close()


@pytest.mark.parametrize("cfg,T,dt", [(3, 6, 0.001), (9, 25, 0.01), (4, 12, 0.01), (10, 5, 0.001), (12, 12, 0.001), (8, 3, 0.001)])
def test_two_stage_filters_closed_loop(hip, oracle, cfg, T, dt):
    """Backup-trajectory classes (examples/InvertedPendulum_Implicit.cpp:113-136 and the like): T x (rows kernel, QP
    kernel, plant step) in stream order.  Checked on the logged states like the fused rollout: the oracle's exact
    filter on x_t gives the logged u_t / rc_t, and x_{t+1} is the plant's Euler step of (x_t, u_t)."""
    from asif_amd import workloads
    B = 512
    model, variant, _ = hip.CONFIGS[cfg]
    opts, oo, w = None, None, None
    omodel, ovariant = oracle.CONFIGS[cfg]
    oo = oracle.default_options(omodel, ovariant)
    if cfg == 10:
        w = workloads.make_learning()
        opts = hip.default_options(model, variant)
        for o in (opts, oo):
            o.x_unc[0], o.x_unc[1] = workloads.RB_X_UNC
        opts.use_learning = 1
        oo.set_learning(oracle.Learning.from_dict(w))
    flt = hip.Filter(model, variant, options=opts)
    if w is not None:
        flt.set_learning(w)
    d = flt.dims
    x, u = workloads.make_batch(cfg, B)
    dev = torch.device("cuda:0")
    tx, tu = torch.from_numpy(x.copy()).to(dev), torch.from_numpy(u).to(dev)
    uact = torch.zeros((1, B), dtype=torch.float64, device=dev)
    relax = torch.zeros((d.nrelax, B), dtype=torch.float64, device=dev)
    nfail = torch.full((B,), 99, dtype=torch.int32, device=dev)
    xlog = torch.zeros((T, d.nx, B), dtype=torch.float64, device=dev)
    ulog = torch.zeros((T, 1, B), dtype=torch.float64, device=dev)
    rclog = torch.zeros((T, B), dtype=torch.int32, device=dev)
    flt.rollout(T, dt, tx, tu, uact, relax, nfail, xlog, ulog, rclog)
    torch.cuda.synchronize()
    xlog, ulog, rclog = xlog.cpu().numpy(), ulog.cpu().numpy(), rclog.cpu().numpy()
    xend = tx.cpu().numpy()
    assert np.array_equal(xlog[0], x)
    # the plant the examples integrate: f, g of the model at the state the filter saw
    def plant(xt):
        if cfg in (3, 10, 8):
            return np.stack([xt[1], np.sin(xt[0])]), np.stack([0 * xt[0], 1 + 0 * xt[0]])
        if cfg in (9, 12):
            return np.stack([xt[1], 0 * xt[0]]), np.stack([0 * xt[0], 1 + 0 * xt[0]])
        return None, None
    for t in range(T):
        xt = xlog[t]
        ua, rl, rc = oracle.filter_batch(omodel, ovariant, oo, np.ascontiguousarray(xt.T), np.ascontiguousarray(u.T),
                                         oracle.SOLVER_EXACT, None, 8, uact_init=np.zeros((B, 1)))
        assert np.array_equal(rclog[t], rc), f"step {t}: rc mismatches {(rclog[t] != rc).sum()}"
        assert np.abs(ulog[t, 0] - ua[:, 0]).max() <= 1e-6
        nxt = xlog[t + 1] if t + 1 < T else xend
        f, g = plant(xt)
        if f is not None:
            np.testing.assert_allclose(nxt, xt + dt * (f + g * ulog[t, 0]), rtol=0, atol=1e-15)
        else:
            assert np.abs(nxt - xt).max() < 1.0 and np.abs(nxt - xt).max() > 0.0
    assert np.array_equal(nfail.cpu().numpy(), (rclog < 0).sum(axis=0))
    assert np.array_equal(uact.cpu().numpy()[0], ulog[-1, 0])
    # and it is T single filter launches with the plant step in between
    tx2 = torch.from_numpy(x.copy()).to(dev)
    ua2 = torch.zeros((1, B), dtype=torch.float64, device=dev)
    rl2 = torch.zeros((d.nrelax, B), dtype=torch.float64, device=dev)
    rc2 = torch.zeros(B, dtype=torch.int32, device=dev)
    flt.filter(tx2, tu, ua2, rl2, rc2)
    torch.cuda.synchronize()
    assert np.array_equal(ua2.cpu().numpy()[0], ulog[0, 0]) and np.array_equal(rc2.cpu().numpy(), rclog[0])
    flt.close()
