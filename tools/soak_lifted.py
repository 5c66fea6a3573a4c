"""Soak of the wave-level QP kernels on FRESH instances of the lifted problems the reference's robust and realizable
classes hand their solver (asif_hip_qp_solve_batch on qp_inv.hpp / qp_lds.hpp) against the oracle's filter on the same
states: 18 x 12 (C5's states from offset `first`), 22 x 15 (shipped half-planes), 38 x 29 / 62 x 47 / 86 x 65.
   python tools/soak_lifted.py [batches of 8192 for 18x12] [batches of 2048 for 22x15] [batches for the realizable kernels]
Prints status mismatches and the largest |u - u_ref| per shape.  Test infrastructure (uses the oracle); not product code."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import oracle_lib as O
from asif_amd import capi
from test_gpu_qp_generic import _solve
O.build()
n18, n22, nrz = (int(v) for v in (sys.argv[1:4] + ["8", "8", "2"])[:3])

class Hip:  # the little of tests/conftest.py's fixture _solve needs
    qp_solve_batch = staticmethod(capi.qp_solve_batch)
    default_solver = staticmethod(capi.default_solver)

def report(name, n, bad, err):
    print(f"{name}: {n} instances, {bad} status mismatches, max|u - u_ref| {err:.2e}", flush=True)

model, variant = O.CONFIGS[5]
o = O.default_options(model, variant)
d = O.dims(model, variant, o)
tot = bad = 0; err = 0.0
for k in range(n18):
    B = 8192
    x, u = O.make_batch(5, B, first=(k + 1) * 1000003)
    A, b, code, _ = O.assemble_batch(model, variant, o, x)
    Hd, c, lb, ub = (np.zeros((B, d.nv)) for _ in range(4))
    for i in range(B):
        Hd[i], c[i], lb[i], ub[i], be = O.qp_static(model, variant, o, u[i])
    ua, rl, rc = O.filter_batch(model, variant, o, x, u, O.SOLVER_EXACT)
    sol, st, it = _solve(Hip, Hd, c, A, b, lb, ub, be)
    bad += int(((st == 1) != (rc == 1)).sum()); ok = (rc == 1) & (st == 1)
    err = max(err, np.abs(sol[ok, 0].clip(o.lb[0], o.ub[0]) - ua[ok, 0]).max(), np.abs(sol[ok, 1] - rl[ok, 0]).max()); tot += B
report("18x12", tot, bad, err)

hp = O.load_halfplanes(); z = O.RobustData(hp)
tot = bad = 0; err = 0.0
for k in range(n22):
    B = 2048
    x, u = O.make_batch_robust_data(hp, B, first=(k + 1) * 1000003)
    ua, rl, rc = z.filter(x, u); A, b, code, sel = z.assemble(x)
    Hd, c, lb, ub = (np.zeros((B, z.nv)) for _ in range(4))
    for i in range(B):
        Hd[i], c[i], lb[i], ub[i], be = z.qp_static(u[i])
    sol, st, it = _solve(Hip, Hd, c, A, b, lb, ub, be)
    bad += int(((st == 1) != (rc == 1)).sum()); ok = (rc == 1) & (st == 1)
    err = max(err, np.abs(sol[ok, 0].clip(-20, 20) - ua[ok, 0]).max()); tot += B
report("22x15", tot, bad, err)

for name in ("100Hz", "10Hz_50pt", "10Hz"):
    kdat = O.load_kernel(name); z = O.Realizable(kdat)
    tot = bad = 0; err = 0.0
    for k in range(nrz):
        B = 512
        x, u = O.make_batch_realizable(kdat, B, first=(k + 1) * 1000003)
        ua, rl, rc = z.filter(x, u); A, b, code, info = z.assemble(x)
        keep = code == 1
        x, u, ua, rc, A, b = x[keep], u[keep], ua[keep], rc[keep], A[keep], b[keep]
        n = len(x)
        Hd, c, lb, ub = (np.zeros((n, z.nv)) for _ in range(4))
        for i in range(n):
            Hd[i], c[i], lb[i], ub[i], be = z.qp_static(u[i])
        sol, st, it = _solve(Hip, Hd, c, A, b, lb, ub, be)
        bad += int(((st == 1) != (rc == 1)).sum()); ok = (rc == 1) & (st == 1)
        err = max(err, np.abs(sol[ok, 0].clip(-20, 20) - ua[ok, 0]).max()); tot += n
    report(f"{z.nv}x{z.nc} ({name})", tot, bad, err)
