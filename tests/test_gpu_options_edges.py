"""Edge cases across the four filters: ragged batches (not a multiple of the wave size), padded leading
dimension (ld > B), non-default options checked against the oracle, updateOptions() semantics (including
the TB horizon quirk), special states, and argument validation at the C ABI."""
import ctypes as C

import numpy as np
import pytest
import torch

import gpu_util

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("cfg,B", [(2, 1), (2, 65), (3, 33), (4, 129), (5, 7)])
def test_ragged_batches(hip, oracle, cfg, B):
    out = gpu_util.run_filter(cfg, B, uact_init=3.0)
    ua, rl, rc = gpu_util.oracle_filter(oracle, cfg, out["x"], out["udes"], uact_init=3.0)
    assert np.array_equal(out["rc"], rc)
    assert np.abs(out["uact"] - ua).max() <= 1e-6


@pytest.mark.parametrize("cfg", [2, 3, 4, 5])
def test_padded_leading_dimension(hip, cfg):
    """Arrays with ld > B: instances beyond B must not be touched, results equal the dense call."""
    from asif_amd import workloads
    model, variant, _ = hip.CONFIGS[cfg]
    B, ld = 100, 160
    x, u = workloads.make_batch(cfg, B)
    dense = gpu_util.run_filter(cfg, B)
    flt = hip.Filter(model, variant)
    d = flt.dims
    dev = torch.device("cuda:0")
    tx = torch.full((d.nx, ld), 1e9, dtype=torch.float64, device=dev)
    tu = torch.full((d.nu, ld), 1e9, dtype=torch.float64, device=dev)
    tx[:, :B] = torch.from_numpy(x).to(dev)
    tu[:, :B] = torch.from_numpy(u).to(dev)
    uact = torch.full((d.nu, ld), -5.0, dtype=torch.float64, device=dev)
    relax = torch.full((d.nrelax, ld), -5.0, dtype=torch.float64, device=dev)
    rc = torch.full((ld,), 77, dtype=torch.int32, device=dev)
    hip.check(flt.lib.asif_hip_filter_batch(flt.handle, B, ld, C.c_void_p(tx.data_ptr()), C.c_void_p(tu.data_ptr()),
                                            C.c_void_p(uact.data_ptr()), C.c_void_p(relax.data_ptr()),
                                            C.c_void_p(rc.data_ptr()), None, None))
    torch.cuda.synchronize()
    assert np.array_equal(rc[:B].cpu().numpy(), dense["rc"])
    ok = np.isin(dense["rc"], (1, 2))
    assert np.array_equal(uact[:, :B].cpu().numpy()[:, ok], dense["uact"][:, ok])
    assert torch.all(rc[B:] == 77) and torch.all(uact[:, B:] == -5.0) and torch.all(relax[:, B:] == -5.0)


def _with_options(hip, oracle, cfg, **kw):
    model, variant, _ = hip.CONFIGS[cfg]
    o = hip.default_options(model, variant)
    oo = oracle.default_options(*oracle.CONFIGS[cfg])
    for k, v in kw.items():
        if k in ("lb", "ub"):
            getattr(o, k)[0] = v
            getattr(oo, k)[0] = v
        else:
            setattr(o, k, v)
            setattr(oo, k, v)
    return o, oo


@pytest.mark.parametrize("cfg,B,kw", [
    (2, 2048, dict(relaxLb=2.0, relaxCost=5.0, lb=-0.4, ub=0.7)),
    (3, 96, dict(relaxLb=3.0, relaxReachLb=1.0, backTrajHorizon=0.75, backTrajDt=0.005, satSharpness=0.4, lb=-2.0, ub=1.0)),
    (4, 1024, dict(relaxTTS=5.0, relaxMinOrtho=10.0, backTrajHorizon=1.5, backTrajMinOrtho=0.01, relaxCost=25.0)),
    (5, 1024, dict(pMin=0.5, pMax=1.5, relaxLb=1.0, relaxCost=3.0)),
    # saturation constants outside the fast Euler step's preconditions (DevOptions::satFastOk): the trajectory kernels
    # run their generic step -- a sharpness below 2^-100 (no bevel to speak of), and a large one (bevelStart < 0)
    (3, 96, dict(satSharpness=1e-31, backTrajHorizon=0.75, backTrajDt=0.005)),
    (3, 96, dict(satSharpness=4.0, backTrajHorizon=0.75, backTrajDt=0.005)),
    (4, 512, dict(satSharpness=1e-31, backTrajHorizon=1.0)),
    # a coarse trajectory step: the angle moves by up to 0.06 per step for the faster states, beyond what the carried
    # sin / cos accept (kTrigCarryMaxStep) -- those blocks are repeated on the generic step, the others stay on the fast one
    (3, 192, dict(backTrajHorizon=2.0, backTrajDt=0.04)),
    (8, 96, dict(backTrajHorizon=2.0, backTrajDt=0.04)),
    # Options::inf beyond OSQP's own infinity (1e30): "no bound" to the reference whatever its size -- the TB class's
    # inert rows (b = -inf) and the relaxation's upper bounds must not read as broken data: instances
    # inside the backup set keep rc 2 with u = clamp(uDes)
    (4, 1024, dict(inf=float("inf"))),
    (4, 1024, dict(inf=1e300)),
    (3, 64, dict(inf=float("inf"))),
    (2, 1024, dict(inf=float("inf"))),
    (5, 512, dict(inf=1e300)),
    # a sharpness whose bevelStop sits within rounding of 1: not the fast step's clamp, the generic chain
    (3, 96, dict(satSharpness=3e-15, backTrajHorizon=0.75, backTrajDt=0.005)),
    (12, 256, dict(satSharpness=1e-12, backTrajHorizon=1.0)),
    # an input pinned by its bounds (lb == ub): the solver's entry eliminates it before the method runs, so the rows
    # kernels do not solve in place (k_tb.hip: launch_tb, k_implicit.hip: fuse_mode) -- two launches, same answers
    (4, 1024, dict(lb=0.3, ub=0.3)),
    (3, 64, dict(lb=-0.2, ub=-0.2)),
    (9, 512, dict(lb=0.1, ub=0.1)),
])
def test_non_default_options(hip, oracle, cfg, B, kw):
    o, oo = _with_options(hip, oracle, cfg, **kw)
    out = gpu_util.run_filter(cfg, B, options=o, uact_init=2.5)
    model, variant = oracle.CONFIGS[cfg]
    d = oracle.dims(model, variant, oo)
    ua, rl, rc = oracle.filter_batch(model, variant, oo, np.ascontiguousarray(out["x"].T),
                                     np.ascontiguousarray(out["udes"].T), oracle.SOLVER_EXACT,
                                     uact_init=np.full((B, d.nu), 2.5))
    assert np.array_equal(out["rc"], rc), f"{(out['rc'] != rc).sum()} rc mismatches"
    assert np.abs(out["uact"].T - ua).max() <= 1e-6
    if "inf" in kw and cfg == 4:
        inside = rc == 2
        assert inside.sum() > 100
        lb, ub = oo.lb[0], oo.ub[0]
        assert np.array_equal(out["uact"][0, inside], np.clip(out["udes"][0, inside], lb, ub))


def test_update_options_and_tb_horizon_quirk(hip, oracle):
    """initialize() sizes the TB trajectory with (1+backTrajExtend), updateOptions() without it
    (src/asif_implicit_tb.cpp:177 vs :377): 316 samples before, 301 after, for the same options."""
    flt = hip.Filter(hip.MODEL_SEGWAY, hip.IMPLICIT_TB)
    assert flt.dims.npBT == 316
    flt.update_options(flt.options)
    assert flt.dims.npBT == 301
    flt.close()
    # explicit filter: updateOptions moves relaxLb in cost and LOWER bound; the batched handle applies the
    # options as a fresh initialize() would (delta pinned at the new relaxLb)
    f2 = hip.Filter(hip.MODEL_DOUBLE_INTEGRATOR, hip.EXPLICIT)
    o = hip.default_options(hip.MODEL_DOUBLE_INTEGRATOR, hip.EXPLICIT)
    o.relaxLb = 2.0
    f2.update_options(o)
    dev = torch.device("cuda:0")
    x = torch.tensor([[0.2], [0.1]], dtype=torch.float64, device=dev)
    u = torch.tensor([[0.5]], dtype=torch.float64, device=dev)
    ua = torch.zeros((1, 1), dtype=torch.float64, device=dev)
    rl = torch.zeros((1, 1), dtype=torch.float64, device=dev)
    rc = torch.zeros(1, dtype=torch.int32, device=dev)
    f2.filter(x, u, ua, rl, rc)
    torch.cuda.synchronize()
    assert rc.item() == 1 and abs(rl.item() - 2.0) < 1e-12 and abs(ua.item() - 0.5) < 1e-12


def test_special_states(hip, oracle):
    # pendulum: origin, on the safety boundary, outside it; segway: origin (inside the backup set)
    x = np.array([[0.0, np.pi, -np.pi, 3.5, 0.1], [0.0, 0.0, 0.5, 0.0, -np.pi]])
    u = np.array([[0.0, 1.0, -1.0, 0.3, 1.5]])
    out = gpu_util.run_filter(3, 5, x=x, udes=u, uact_init=4.0)
    ua, rl, rc = gpu_util.oracle_filter(oracle, 3, x, u, uact_init=4.0)
    assert np.array_equal(out["rc"], rc) and np.abs(out["uact"] - ua).max() <= 1e-6
    xs = np.zeros((4, 3))
    xs[:, 1] = [0.1, 0.0, 0.0, 0.0]
    xs[:, 2] = [0.0, 0.0, 0.3, 0.0]
    us = np.array([[0.0, 25.0, -25.0]])
    out = gpu_util.run_filter(4, 3, x=xs, udes=us, uact_init=4.0)
    ua, rl, rc = gpu_util.oracle_filter(oracle, 4, xs, us, uact_init=4.0)
    assert np.array_equal(out["rc"], rc) and np.abs(out["uact"] - ua).max() <= 1e-6
    assert out["rc"][0] == 2 and abs(out["uact"][0, 0]) == 0.0


def test_argument_validation(hip):
    flt = hip.Filter(hip.MODEL_DOUBLE_INTEGRATOR, hip.EXPLICIT)
    lib = flt.lib
    p = C.c_void_p(64)
    assert lib.asif_hip_filter_batch(flt.handle, -1, 4, p, p, p, p, p, None, None) == -1
    assert lib.asif_hip_filter_batch(flt.handle, 8, 4, p, p, p, p, p, None, None) == -1   # ld < B
    assert lib.asif_hip_filter_batch(flt.handle, 8, 8, None, p, p, p, p, None, None) == -1
    assert lib.asif_hip_filter_batch(None, 8, 8, p, p, p, p, p, None, None) == -1
    assert lib.asif_hip_filter_batch(flt.handle, 0, 0, None, None, None, None, None, None, None) == 0  # empty batch
    o = hip.default_options(hip.MODEL_INVERTED_PENDULUM_ROBUST, hip.ROBUST)
    o.nHalfPlanes = 9
    with pytest.raises(hip.AsifHipError):
        hip.Filter(hip.MODEL_INVERTED_PENDULUM_ROBUST, hip.ROBUST, options=o)
    s = hip.default_solver(lanes_per_qp=3)
    bad = hip.Filter(hip.MODEL_DOUBLE_INTEGRATOR, hip.EXPLICIT, solver=s)
    dev = torch.device("cuda:0")
    t = torch.zeros((2, 4), dtype=torch.float64, device=dev)
    with pytest.raises(hip.AsifHipError):
        bad.filter(t, t[:1], t[:1].clone(), t[:1].clone(), torch.zeros(4, dtype=torch.int32, device=dev))


@pytest.mark.parametrize("cfg,B", [(2, 300000), (2, 1000), (5, 270001), (9, 5000)])
def test_host_buffer_entry_equals_device_entry(hip, cfg, B):
    """asif_hip_filter_batch_host (H2D, kernels, D2H; large batches of the single-kernel filters in chunks on two
    streams) must give bit for bit what asif_hip_filter_batch gives on resident buffers, untouched slots included."""
    import ctypes as C
    import torch
    from asif_amd import workloads
    model, variant, _ = hip.CONFIGS[cfg]
    flt = hip.Filter(model, variant)
    d = flt.dims
    x, u = workloads.make_batch(cfg, B)
    dev = torch.device("cuda:0")
    tx, tu = torch.from_numpy(x).to(dev), torch.from_numpy(u).to(dev)
    uact = torch.full((d.nu, B), 7.0, dtype=torch.float64, device=dev)
    relax = torch.full((d.nrelax, B), -7.0, dtype=torch.float64, device=dev)
    rc = torch.zeros(B, dtype=torch.int32, device=dev)
    flt.filter(tx, tu, uact, relax, rc)
    torch.cuda.synchronize()
    for pinned in (False, True):
        hx, hu = torch.from_numpy(x.copy()), torch.from_numpy(u.copy())
        hua = torch.full((d.nu, B), 7.0, dtype=torch.float64)
        hrl = torch.full((d.nrelax, B), -7.0, dtype=torch.float64)
        hrc = torch.zeros(B, dtype=torch.int32)
        bufs = [hx, hu, hua, hrl, hrc]
        if pinned:
            bufs = [t.pin_memory() for t in bufs]
        hip.check(flt.lib.asif_hip_filter_batch_host(flt.handle, B, *[C.c_void_p(t.data_ptr()) for t in bufs]))
        assert np.array_equal(bufs[4].numpy(), rc.cpu().numpy())
        assert np.array_equal(bufs[2].numpy(), uact.cpu().numpy())
        assert np.array_equal(bufs[3].numpy(), relax.cpu().numpy())
    flt.close()


def test_critical_sample_selection_on_plateaus_and_ties(hip, oracle):
    """The two-pass critical-sample search (block minima, then the exact per-sample selection inside the chosen
    blocks) must keep the single-pass rule: smallest margin first, ties -> earlier sample.  At the origin the
    backup trajectory never moves, so every margin is bit-identical and the critical samples are 0..K-1; a state
    that converges to the origin has a long plateau of equal margins behind its transient."""
    from asif_amd import workloads
    # pendulum (K = 10, blocks of 16) and implicit double integrator (K = 4, blocks of 4)
    for cfg, K in ((3, 10), (9, 4)):
        x = np.zeros((2, 64))
        x[0, 1::2] = 0.3          # every other instance starts off the origin: mixed waves
        x[1, 2::4] = -0.2
        out = gpu_util.run_assemble(cfg, 64, x=x)
        idx = out["diag"][:K].astype(int)
        assert np.array_equal(idx[:, 0], np.arange(K))
        model, variant = oracle.CONFIGS[cfg]
        o = oracle.default_options(model, variant)
        A, b, _, _ = oracle.assemble_batch(model, variant, o, np.ascontiguousarray(x.T))
        np.testing.assert_allclose(out["A"].T, A, rtol=1e-9, atol=1e-11)
        for i in range(64):
            oracle.assemble(model, variant, o, x[:, i])
            assert np.array_equal(idx[:, i], oracle.last_crit_idx()), (cfg, i)
    # a horizon that is not a multiple of the block size (ragged last block) and shorter than one block
    for horizon_steps in (5, 17, 100):
        od = hip.default_options(1, 1)
        oo = oracle.default_options(oracle.MODEL_IP, oracle.VAR_IMPLICIT)
        for o in (od, oo):
            o.backTrajHorizon = horizon_steps * o.backTrajDt
        xs, _ = workloads.make_batch(3, 256)
        out = gpu_util.run_assemble(3, 256, options=od, x=xs)
        assert out["dims"].npBT == max(horizon_steps + 1, 10)
        A, b, _, _ = oracle.assemble_batch(oracle.MODEL_IP, oracle.VAR_IMPLICIT, oo, np.ascontiguousarray(xs.T))
        np.testing.assert_allclose(out["A"].T, A, rtol=1e-9, atol=1e-11)
        np.testing.assert_allclose(out["b"].T, b, rtol=1e-9, atol=1e-11)


@pytest.mark.parametrize("cfg,B", [(2, 4096), (3, 256), (4, 1024), (5, 512)])
def test_filter_call_is_graph_capturable(hip, cfg, B):
    """Once a first call has sized the handle's staging buffers, a filter call only enqueues kernels on the
    caller's stream: it can be captured into a HIP graph and replayed (streams and graphs instead of a
    tracing compiler).  The replay on new inputs must equal a direct call bit for bit."""
    from asif_amd import workloads
    model, variant, _ = hip.CONFIGS[cfg]
    flt = hip.Filter(model, variant)
    d = flt.dims
    dev = torch.device("cuda:0")
    x, u = workloads.make_batch(cfg, B)
    x2, u2 = workloads.make_batch(cfg, B, first=B)
    tx, tu = torch.from_numpy(x).to(dev), torch.from_numpy(u).to(dev)
    uact = torch.zeros((d.nu, B), dtype=torch.float64, device=dev)
    relax = torch.zeros((d.nrelax, B), dtype=torch.float64, device=dev)
    rc = torch.zeros(B, dtype=torch.int32, device=dev)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        flt.filter(tx, tu, uact, relax, rc)  # sizes the staging buffers
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        flt.filter(tx, tu, uact, relax, rc)
    tx.copy_(torch.from_numpy(x2))
    tu.copy_(torch.from_numpy(u2))
    uact.zero_()
    relax.zero_()
    rc.zero_()
    g.replay()
    torch.cuda.synchronize()
    got = (uact.cpu().numpy().copy(), relax.cpu().numpy().copy(), rc.cpu().numpy().copy())
    ref = gpu_util.run_filter(cfg, B, x=x2, udes=u2)
    assert np.array_equal(got[2], ref["rc"]) and np.array_equal(got[0], ref["uact"]) and np.array_equal(got[1], ref["relax"])
    flt.close()
