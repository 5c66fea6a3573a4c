#!/usr/bin/env python3
"""Parity soak (GPU box): device filter() against the oracle's exact answer on MANY more seeded instances than the tests
and the bench line check -- consecutive ranges of the same generators (workloads.make_batch(cfg, B, first)).  Prints per
config: instances, rc mismatches, max |uAct - u_ref| over the instances that solved a QP, rc histogram.
   python tools/soak_parity.py [seconds per config, default 20] [cfg ...] [--polish=0|1|2] [--integrator=1 [--tol=1e-8]]
--integrator=1: the trajectory filters with the reference's USE_ODEINT integrator (dopri5 dense output) on both sides; the
two adaptive controllers see inputs that differ in the last bits, so this is parity at the integrator's tolerance, not
bit for bit (tests/test_gpu_implicit_dopri.py, test_gpu_tb_dopri.py)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import gpu_util  # noqa: E402
import oracle_lib as O  # noqa: E402

O.build()
args = [a for a in sys.argv[1:] if not a.startswith("--")]
dopri = "--integrator=1" in sys.argv[1:]
tol = ([float(a.split("=")[1]) for a in sys.argv[1:] if a.startswith("--tol=")] or [1e-8])[0]
polish = [int(a.split("=")[1]) for a in sys.argv[1:] if a.startswith("--polish=")]
budget = float(args[0]) if len(args) > 0 else 20.0
cfgs = [int(c) for c in args[1:]] or [2, 3, 4, 5, 8, 9, 11, 12]
solver = None
if polish:  # asif_hip_solver::polish: 2 = dual active-set stage first (default), 1 = ADMM + active-set finish, 0 = ADMM alone
    from asif_amd import capi
    solver = capi.default_solver(polish=polish[0])
    print(f"solver mode polish = {polish[0]}", flush=True)
if dopri:
    print(f"integrator = 1 (dopri5 dense output), tolerances {tol:g}", flush=True)
CHUNK = {2: 1 << 20, 3: 8192, 4: 1 << 17, 5: 1 << 18, 8: 4096, 9: 1 << 17, 11: 1 << 19, 12: 1 << 16}
threads = min(os.cpu_count() or 8, 16)  # the GPU box's CPU share for one GPU
print(f"host threads {threads}; budget {budget:.0f} s of oracle time per config", flush=True)
for cfg in cfgs:
    model, variant = O.CONFIGS[cfg]
    oo = O.default_options(model, variant)
    od = None
    if dopri:
        from asif_amd import capi
        od = capi.default_options(*capi.CONFIGS[cfg][:2])
        for o in (od, oo):
            o.integrator = 1
            o.backTrajAbsTol = o.backTrajRelTol = tol
    B, first, n, bad, worst = CHUNK[cfg], 1 << 24, 0, 0, 0.0  # ranges the tests and the bench do not touch
    hist = {}
    t0 = last_note = time.perf_counter()
    while time.perf_counter() - t0 < budget:
        out = gpu_util.run_filter(cfg, B, first=first, uact_init=7.0, relax_init=-7.0, solver=solver, **({'options': od} if od is not None else {}))
        d = out["dims"]
        ua, rl, rc = O.filter_batch(model, variant, oo, np.ascontiguousarray(out["x"].T), np.ascontiguousarray(out["udes"].T),
                                    O.SOLVER_EXACT, None, threads, uact_init=np.full((B, d.nu), 7.0))
        bad += int((out["rc"] != rc).sum())
        ok = (rc == 1) | (rc == 2)
        if ok.any():
            worst = max(worst, float(np.abs(out["uact"].T[ok] - ua[ok]).max()))
        fb = ~ok
        if fb.any():  # failures: what the class leaves in uAct (untouched, or the saturated backup controller)
            worst = max(worst, float(np.abs(out["uact"].T[fb] - ua[fb]).max()))
        for k, c in zip(*np.unique(rc, return_counts=True)):
            hist[int(k)] = hist.get(int(k), 0) + int(c)
        n += B
        first += B
        if time.perf_counter() - last_note > 60.0:  # (a GPU box takes a silent command for hung after 7 minutes)
            last_note = time.perf_counter()
            print(f"  config {cfg}: {n} instances so far, rc mismatches {bad}", flush=True)
    print(f"config {cfg}: {n} instances, rc mismatches {bad}, max |uAct - u_ref| {worst:.3e}, rc {dict(sorted(hist.items()))}", flush=True)
