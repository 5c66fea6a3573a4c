"""Scratch: per-section cycle counts of the LDS kernel (variant library built with -DASIF_LDS_PROFILE, which writes
its section timers where the solution goes).  Sections: 0 gradient, 1 factor, 2 solve, 3 line search, 4 outer
update, 5 rest, 6 build.  Not product code, not a test."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from asif_amd import capi  # noqa: E402
# (the profiling build is named by ASIF_HIP_LIB: tools/build_variant.sh <name> "k_qp.hip" "-DASIF_..._PROFILE")
import bench  # noqa: E402

dev = torch.device("cuda:0")
B = int(os.environ.get("B", 8192))
P = bench.qp_problem(5, B, dev)
sol = torch.zeros((P["nv"], B), dtype=torch.float64, device=dev)
st = torch.zeros(B, dtype=torch.int32, device=dev)
it = torch.zeros(B, dtype=torch.int32, device=dev)
for _ in range(2):
    capi.qp_solve_batch(P["Hd"], P["c"], P["A"], P["b"], P["lb"], P["ub"], sol, st, it, be=P["be"])
torch.cuda.synchronize()
t = sol[:7].cpu().numpy()
n = it.cpu().numpy()
names = ["gradient", "factor", "solve", "line search", "outer update", "rest", "build"]
tot = t.sum(0)
print("newton mean", n.mean(), "cycles/QP mean", tot.mean())
for k in range(7):
    print(f"{names[k]:14s} {t[k].mean():12.0f} cycles  {100 * t[k].sum() / tot.sum():5.1f} %   per newton {t[k].sum() / n.sum():9.0f}")
