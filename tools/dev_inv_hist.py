"""Developer probe: Newton-step histogram of the lifted 18 x 12 problems (8192) on the half-wave kernel."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, ctypes as C
import bench
from asif_amd import capi
dev = torch.device("cuda:0")
B = 8192
q = bench.qp_problem(5, B, dev)
nv, nc = q["nv"], q["nc"]
sol = torch.zeros((nv, B), dtype=torch.float64, device=dev); st = torch.zeros(B, dtype=torch.int32, device=dev); it = torch.zeros(B, dtype=torch.int32, device=dev)
solver = capi.default_solver()
be = (C.c_uint8 * nc)(*[int(v) for v in q["be"]])
lib = capi.load()
p = lambda t: C.c_void_p(t.data_ptr())
r = lib.asif_hip_qp_solve_batch(0, C.byref(solver), C.c_int64(B), C.c_int64(B), nv, nc, p(q["Hd"]), p(q["c"]), p(q["A"]), p(q["b"]), p(q["lb"]), p(q["ub"]), C.cast(be, C.c_void_p), p(sol), p(st), p(it), None)
torch.cuda.synchronize()
itn = it.cpu().numpy()
print("hist", np.bincount(itn))
w = np.maximum(itn[0::2], itn[1::2])
print("mean per QP", itn.mean(), "mean of pair max", w.mean(), "sum of pair max / sum", w.sum() * 2 / itn.sum())
print("slowest", np.argsort(itn)[-8:], np.sort(itn)[-8:])
