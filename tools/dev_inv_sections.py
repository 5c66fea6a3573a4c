"""Scratch: per-section cycle counts of the half-wave QP kernel (variant library built with -DASIF_INV_PROFILE, which
writes its section timers where the solution goes).  Not product code, not a test.
   build:  cd asif_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -DASIF_INV_PROFILE -c k_qp.hip -o /tmp/kqp_prof.o
           hipcc --offload-arch=gfx950 -shared -fPIC -o ../libasif_invprof.so $(ls build/*.o | grep -v k_qp.o) /tmp/kqp_prof.o"""
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
t = sol[:7].cpu().numpy()[:, 0::2]  # one set of timers per wave (both halves carry the same)
n = it.cpu().numpy()
nw = np.maximum(n[0::2], n[1::2])
names = ["gradient", "K^-1 upkeep", "direction", "line search f(1)", "full search + step", "outer update", "setup"]
tot = t.sum(0)
print("newton mean per QP", n.mean(), "per wave", nw.mean(), "cycles/wave mean", tot.mean(), "max", tot.max())
for k in range(7):
    print(f"{names[k]:20s} {t[k].mean():12.0f} cycles  {100 * t[k].sum() / tot.sum():5.1f} %   per wave-newton {t[k].sum() / nw.sum():9.0f}")
c = sol[7:10].cpu().numpy()  # per QP: rebuilds of the inverse, rank-one steps for rows, for bounds
print("per QP: rebuilds mean %.2f max %d; rank-one steps rows mean %.1f max %d, bounds mean %.1f max %d" % (c[0].mean(), c[0].max(), c[1].mean(), c[1].max(), c[2].mean(), c[2].max()))
easy = n == 4
print("  the 4-step problems: rebuilds %.2f, rows %.1f, bounds %.1f" % (c[0][easy].mean(), c[1][easy].mean(), c[2][easy].mean()))
hard = np.argsort(tot)[-3:]
for w in hard:
    print("wave", w, "newton", nw[w], "cycles", tot[w], {names[k]: int(t[k][w]) for k in range(7)})
