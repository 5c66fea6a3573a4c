"""Developer probe: 50 calls of the lifted 18 x 12 batch (8 192) for rocprofv3 --kernel-trace --stats: the two launches of
qp_inv.hpp's two-phase form show as separate kernels."""
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
for _ in range(50):
    lib.asif_hip_qp_solve_batch(0, C.byref(solver), C.c_int64(B), C.c_int64(B), nv, nc, p(q["Hd"]), p(q["c"]), p(q["A"]), p(q["b"]), p(q["lb"]), p(q["ub"]), C.cast(be, C.c_void_p), p(sol), p(st), p(it), None)
torch.cuda.synchronize()
print("status", np.bincount(st.cpu().numpy() + 3), "newton max", int(it.max()))
