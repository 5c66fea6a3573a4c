"""Developer probe: the half-wave kernel on batches of EASY lifted 18 x 12 problems only (4 Newton steps each), at
batch sizes that put 1/2, 1, 2, 4 and 8 waves on every SIMD: separates what a wave costs alone from what co-resident
waves cost each other (occupancy, the CU's LDS pipe), without the tail of the hard instances."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, ctypes as C
import bench
from asif_amd import capi
dev = torch.device("cuda:0")
B0 = 8192
q = bench.qp_problem(5, B0, dev)
nv, nc = q["nv"], q["nc"]
solver = capi.default_solver()
be = (C.c_uint8 * nc)(*[int(v) for v in q["be"]])
lib = capi.load()
p = lambda t: C.c_void_p(t.data_ptr())

def run(qq, B, reps=0):
    sol = torch.zeros((nv, B), dtype=torch.float64, device=dev); st = torch.zeros(B, dtype=torch.int32, device=dev); it = torch.zeros(B, dtype=torch.int32, device=dev)
    call = lambda: lib.asif_hip_qp_solve_batch(0, C.byref(solver), C.c_int64(B), C.c_int64(B), nv, nc, p(qq["Hd"]), p(qq["c"]), p(qq["A"]), p(qq["b"]), p(qq["lb"]), p(qq["ub"]), C.cast(be, C.c_void_p), p(sol), p(st), p(it), None)
    assert call() == 0
    torch.cuda.synchronize()
    if not reps:
        return it.cpu().numpy()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); call(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return float(np.median(ts))

itn = run(q, B0)
easy = np.where(itn == 4)[0]
print("easy", len(easy), "of", B0)
for B in (256, 512, 1024, 2048, 4096, 8192, 16384, 32768):
    idx = torch.from_numpy(easy[np.arange(B) % len(easy)]).to(dev)
    qq = {k: (q[k][:, idx].contiguous() if torch.is_tensor(q[k]) and q[k].dim() == 2 else q[k]) for k in q}
    t = run(qq, B, 30)
    print(f"B {B:6d}  waves {B // 2:6d}  per SIMD {B / 2 / 1024:5.2f}   {t:8.1f} us   {t * 1e3 / B:7.1f} ns/QP")

# the seeded mix in its own order, and with its waves handed out longest first / longest last on EVERY XCD (block b
# runs pair (b % 8) * (npair / 8) + b / 8: qp_common's xcd_contiguous_index)
for B in (8192, 32768):
    base = np.arange(B) % B0
    npair = B // 2
    w = np.maximum(itn[base[0::2]], itn[base[1::2]])            # a wave's cost: its harder half
    slots = np.array([(b % 8) * (npair // 8) + b // 8 for b in range(npair)])  # dispatch order -> pair index
    def place(order):                                           # order: pairs in the order they should be dispatched
        o = np.empty(B, dtype=np.int64)
        o[2 * slots] = base[2 * order]
        o[2 * slots + 1] = base[2 * order + 1]
        return o
    orders = {"as seeded": base, "longest first": place(np.argsort(-w, kind="stable")), "longest last": place(np.argsort(w, kind="stable"))}
    for name, o in orders.items():
        idx = torch.from_numpy(o).to(dev)
        qq = {k: (q[k][:, idx].contiguous() if torch.is_tensor(q[k]) and q[k].dim() == 2 else q[k]) for k in q}
        print(f"B {B:6d}  {name:40s} {run(qq, B, 20):8.1f} us")
