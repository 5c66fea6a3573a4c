"""Developer probe (a build with -DASIF_INV_WAVETIME: tools/build_variant.sh wt "k_qp.hip" "-DASIF_INV_WAVETIME"): when
each wave of the half-wave QP kernel started and how long it ran, on the seeded lifted 18 x 12 batch."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, ctypes as C
import bench
from asif_amd import capi
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
q = bench.qp_problem(5, B, dev)
nv, nc = q["nv"], q["nc"]
solver = capi.default_solver()
be = (C.c_uint8 * nc)(*[int(v) for v in q["be"]])
lib = capi.load()
p = lambda t: C.c_void_p(t.data_ptr())
sol = torch.zeros((nv, B), dtype=torch.float64, device=dev); st = torch.zeros(B, dtype=torch.int32, device=dev); it = torch.zeros(B, dtype=torch.int32, device=dev)
for _ in range(3):
    sol.zero_()
    lib.asif_hip_qp_solve_batch(0, C.byref(solver), C.c_int64(B), C.c_int64(B), nv, nc, p(q["Hd"]), p(q["c"]), p(q["A"]), p(q["b"]), p(q["lb"]), p(q["ub"]), C.cast(be, C.c_void_p), p(sol), p(st), p(it), None)
    torch.cuda.synchronize()
s = sol.cpu().numpy(); itn = it.cpu().numpy()
t0, d1 = s[0], s[1]
base = t0.min()
us = lambda v: v / 100.0
print("starts %.1f .. %.1f us, last end %.1f us" % (us(t0.min() - base), us(t0.max() - base), us((t0 + d1).max() - base)))
hard = itn > 4
print("  duration of waves whose problems take 4 Newton steps: median %.1f  p99 %.1f  max %.1f us" % tuple(us(np.percentile(d1[~hard], [50, 99, 100]))))
print("  duration of waves with a longer problem:             median %.1f  p99 %.1f  max %.1f us" % tuple(us(np.percentile(d1[hard], [50, 99, 100]))))
hist, edges = np.histogram(us(t0 - base), bins=12)
print("  start-time histogram (us):", [int(e) for e in edges], hist.tolist())
