"""Developer probe (qp_inv.hpp's section timers; see the comment below): time of asif_hip_qp_solve_batch on the lifted problems ASIFrealizable hands to its solver
(38 x 29 on the 100 Hz kernel: qp_lds.hpp, one wave per QP), problems built through the oracle's assembly.
Not product code, not a test.   python tools/dev_rz_time.py [100Hz|10Hz_50pt|10Hz] [copies]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import oracle_lib as O  # noqa: E402
from asif_amd import capi  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "100Hz"
copies = int(sys.argv[2]) if len(sys.argv) > 2 else 8
O.build()
k = O.load_kernel(name)
z = O.Realizable(k)
B0 = 1024
x, u = O.make_batch_realizable(k, B0)
A, b, code, info = z.assemble(x)
keep = code == 1
x, u, A, b = x[keep], u[keep], A[keep], b[keep]
n = len(x)
Hd, c, lb, ub = (np.zeros((n, z.nv)) for _ in range(4))
for i in range(n):
    Hd[i], c[i], lb[i], ub[i], be = z.qp_static(u[i])
rep = lambda a: np.tile(a, (copies, 1))
dev = torch.device("cuda:0")
t = lambda a: torch.from_numpy(np.ascontiguousarray(rep(a).T)).to(dev)
B = n * copies
sol = torch.zeros((z.nv, B), dtype=torch.float64, device=dev)
st = torch.zeros(B, dtype=torch.int32, device=dev)
it = torch.zeros(B, dtype=torch.int32, device=dev)
args = (t(Hd), t(c), t(A), t(b), t(lb), t(ub), sol, st, it)
# a build with -DASIF_INV_PROFILE (tools/build_variant.sh invprof "k_qp.hip" "-DASIF_INV_PROFILE") writes qp_inv.hpp's section
# timers and counts where the solution goes
for _ in range(2):
    capi.qp_solve_batch(*args, be=be)
torch.cuda.synchronize()
tt = sol[:10].cpu().numpy()
nn = it.cpu().numpy()
names = ["gradient", "inverse upkeep", "direction", "phi'(1)", "line search", "outer update", "prologue"]
tot = tt[:7].sum(0)
print(f"{name}: nv {z.nv} nc {z.nc}: newton mean {nn.mean():.2f}, cycles/QP mean {tot.mean():.0f}")
for k in range(7):
    print(f"{names[k]:14s} {tt[k].mean():12.0f} cycles  {100 * tt[k].sum() / tot.sum():5.1f} %   per newton {tt[k].sum() / nn.sum():9.0f}")
print(f"rebuilds per QP {tt[7].mean():.2f}, rank-one steps for rows {tt[8].mean():.1f}, for bounds {tt[9].mean():.1f}  (per Newton step: {tt[7].sum() / nn.sum():.2f}, {tt[8].sum() / nn.sum():.1f}, {tt[9].sum() / nn.sum():.1f})")
