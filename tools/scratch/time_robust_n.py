"""Scratch: asif_hip_qp_solve_batch on ASIFrobust's lifted problem for N = 1..8 safety functions, 8 192 problems each
(ASIF_HIP_QP_INV_EXACT=0 for the padded grid)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import oracle_lib as O
O.build()
from asif_amd import capi
from test_gpu_qp_lds import _robust_qps_with_n_halfplanes
dev = torch.device("cuda:0")
for N in range(1, 9):
    d, q, _ = _robust_qps_with_n_halfplanes(O, N, 1024)
    B = 8192
    t = lambda a: torch.from_numpy(np.ascontiguousarray(np.tile(a, (8, 1)).T)).to(dev)
    args = [t(a) for a in q[:6]]
    sol = torch.zeros((d.nv, B), dtype=torch.float64, device=dev); st = torch.zeros(B, dtype=torch.int32, device=dev); it = torch.zeros(B, dtype=torch.int32, device=dev)
    for _ in range(3): capi.qp_solve_batch(*args, sol, st, it, be=q[6])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    K = 20
    e0.record()
    for _ in range(K): capi.qp_solve_batch(*args, sol, st, it, be=q[6])
    e1.record(); torch.cuda.synchronize()
    print(f"N={N} {d.nv}x{d.nc}: {e0.elapsed_time(e1) / K * 1e3:.1f} us per 8192, newton mean {it.float().mean().item():.2f} max {it.max().item()}", flush=True)
