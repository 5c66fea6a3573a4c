"""Scratch: time asif_hip_qp_solve_batch on the 22 x 15 problems of DoubleIntegrator_Robust (shipped half-planes)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import oracle_lib as O
O.build()
from asif_amd import capi
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
hp = O.load_halfplanes(); z = O.RobustData(hp)
n0 = min(B, 2048)
x, u = O.make_batch_robust_data(hp, n0)
A, b, code, sel = z.assemble(x)
Hd, c, lb, ub = (np.zeros((n0, z.nv)) for _ in range(4))
for i in range(n0): Hd[i], c[i], lb[i], ub[i], be = z.qp_static(u[i])
rep = (B + n0 - 1) // n0
tile = lambda a: np.tile(a, (rep, 1))[:B]
dev = torch.device("cuda:0")
t = lambda a: torch.from_numpy(np.ascontiguousarray(tile(a).T)).to(dev)
args = [t(Hd), t(c), t(A), t(b), t(lb), t(ub)]
sol = torch.zeros((z.nv, B), dtype=torch.float64, device=dev); st = torch.zeros(B, dtype=torch.int32, device=dev); it = torch.zeros(B, dtype=torch.int32, device=dev)
for _ in range(5): capi.qp_solve_batch(*args, sol, st, it, be=be)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
K = 30
e0.record()
for _ in range(K): capi.qp_solve_batch(*args, sol, st, it, be=be)
e1.record(); torch.cuda.synchronize()
import hashlib
h = hashlib.sha256(sol.cpu().numpy().tobytes() + st.cpu().numpy().tobytes() + it.cpu().numpy().tobytes()).hexdigest()[:12]
print(f"22x15 B={B}: {e0.elapsed_time(e1) / K * 1e3:.1f} us per launch, solved {(st == 1).sum().item()}, newton mean {it.float().mean().item():.2f} max {it.max().item()}, sha {h}")
