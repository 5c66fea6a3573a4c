"""Scratch: status mismatches of the LDS kernel on the robust-data 22x15 problems."""
import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as oracle
from asif_amd import capi as hip
from test_gpu_qp_generic import _solve
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
bad = np.where((st == 1) != (rc == 1))[0]
print("mismatches", bad, "st", st[bad], "rc", rc[bad], "it", it[bad])
print("status hist", dict(zip(*np.unique(st, return_counts=True))), "iters mean/max", it.mean(), it.max())
ok = (rc == 1) & (st == 1)
print("max err u", np.abs(sol[ok, 0].clip(-20, 20) - ua[ok, 0]).max())
for i in bad[:4]:
    print(i, "x", x[i], "u", u[i], "sol", sol[i, :2], "ua", ua[i])
np.savez(os.path.join(ROOT, "gpurun_out", "lds_bad.npz"), idx=bad, Hd=Hd[bad], c=c[bad], A=A[bad], b=b[bad], lb=lb[bad], ub=ub[bad], be=be)
