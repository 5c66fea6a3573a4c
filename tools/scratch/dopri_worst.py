#!/usr/bin/env python3
"""Scratch: where does the dopri5 soak's largest |uAct - u_ref| on config 3 come from?  Finds the worst instance of the
first chunks and prints both sides' rows for it."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import gpu_util, oracle_lib as O
from asif_amd import capi
O.build()
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
tol = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-8
model, variant = O.CONFIGS[cfg]
oo = O.default_options(model, variant); od = capi.default_options(*capi.CONFIGS[cfg][:2])
for o in (od, oo):
    o.integrator = 1; o.backTrajAbsTol = o.backTrajRelTol = tol
B, first = 8192, 1 << 24
worst = (0.0, None)
for it in range(40):
    out = gpu_util.run_filter(cfg, B, first=first, uact_init=7.0, relax_init=-7.0, options=od)
    ua, rl, rc = O.filter_batch(model, variant, oo, np.ascontiguousarray(out["x"].T), np.ascontiguousarray(out["udes"].T), O.SOLVER_EXACT, None, 16, uact_init=np.full((B, 1), 7.0))
    ok = rc == 1
    dv = np.where(ok, np.abs(out["uact"][0] - ua[:, 0]), 0.0)
    k = int(dv.argmax())
    if dv[k] > worst[0]:
        worst = (float(dv[k]), (first + k, out["x"][:, k].copy(), out["udes"][:, k].copy(), out["uact"][:, k].copy(), ua[k].copy(), out["relax"][:, k].copy(), rl[k].copy()))
    first += B
    print(it, worst[0], flush=True)
    if worst[0] > 3e-5: break
d, (idx, x, u, uad, uao, rld, rlo) = worst
print("worst", d, "instance", idx, "x", x.tolist(), "uDes", u.tolist(), "device", uad.tolist(), rld.tolist(), "oracle", uao.tolist(), rlo.tolist())
xs = np.ascontiguousarray(x[:, None]); us = np.ascontiguousarray(u[:, None])
rows = gpu_util.run_assemble(cfg, 1, options=od, x=xs)
A, b, code, diag = O.assemble_batch(model, variant, oo, np.ascontiguousarray(xs.T))
Ad, bd = rows["A"][:, 0], rows["b"][:, 0]
nc = b.shape[1]; nv = A.shape[1] // nc
print("nc", nc, "nv", nv)
Ao = A[0].reshape(nv, nc).T; Adv = Ad.reshape(nv, nc).T
for r in range(nc):
    print(r, "device", Adv[r].tolist(), bd[r], "| oracle", Ao[r].tolist(), b[0][r], "| diff", float(np.abs(Adv[r] - Ao[r]).max()), float(abs(bd[r] - b[0][r])))
print("diag device", rows["diag"][:, 0].tolist()); print("diag oracle", diag[0].tolist())
