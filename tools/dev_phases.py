"""Developer probe: C2 kernel time under settings that switch solver phases off (results are wrong, timing only).
Low-overhead launch loop as in bench.py (pre-marshalled ctypes call, one HIP event pair)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, ctypes as C
from asif_amd import capi, workloads
B = 65536
x, u = workloads.make_batch(2, B)
dev = torch.device("cuda:0")
tx = torch.from_numpy(x).to(dev); tu = torch.from_numpy(u).to(dev)
hip = C.CDLL("libamdhip64.so")
hip.hipEventRecord.argtypes = [C.c_void_p, C.c_void_p]
hip.hipEventElapsedTime.argtypes = [C.POINTER(C.c_float), C.c_void_p, C.c_void_p]
hip.hipEventSynchronize.argtypes = [C.c_void_p]
def run(**kw):
    s = capi.default_solver(**kw)
    flt = capi.Filter(capi.MODEL_DOUBLE_INTEGRATOR, capi.EXPLICIT, solver=s)
    uact = torch.zeros((1, B), dtype=torch.float64, device=dev); relax = torch.zeros((1, B), dtype=torch.float64, device=dev)
    rc = torch.zeros(B, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream()
    fn = flt.lib.asif_hip_filter_batch
    args = (flt.handle, B, tx.stride(0), C.c_void_p(tx.data_ptr()), C.c_void_p(tu.data_ptr()), C.c_void_p(uact.data_ptr()),
            C.c_void_p(relax.data_ptr()), C.c_void_p(rc.data_ptr()), None, C.c_void_p(stream.cuda_stream))
    for _ in range(10): fn(*args)
    torch.cuda.synchronize()
    ev = []
    for _ in range(2):
        e = C.c_void_p(); hip.hipEventCreate(C.byref(e)); ev.append(e)
    sp = C.c_void_p(stream.cuda_stream)
    n = 200
    hip.hipEventRecord(ev[0], sp)
    for _ in range(n): fn(*args)
    hip.hipEventRecord(ev[1], sp)
    torch.cuda.synchronize()
    ms = C.c_float(); hip.hipEventElapsedTime(C.byref(ms), ev[0], ev[1])
    return ms.value / n * 1e3
for name, kw in [("default (finish-first)", {}), ("presolve (assembly+clip)", dict(presolve=1)),
                 ("max_iter 0 (assembly+scale+factor, no finish)", dict(max_iter=0)),
                 ("max_iter 0, no scaling", dict(max_iter=0, scaling_iters=-1)),
                 ("no scaling", dict(scaling_iters=-1)),
                 ("rounds 1", dict(active_set_rounds=1, max_iter=2)), ("rounds 2", dict(active_set_rounds=2, max_iter=2)),
                 ("rounds 3", dict(active_set_rounds=3, max_iter=2)), ("rounds 4", dict(active_set_rounds=4, max_iter=2)),
                 ("rounds 6", dict(active_set_rounds=6, max_iter=2)),
                 ("refine 1", dict(refine_steps=1)), ("refine 0", dict(refine_steps=0, max_iter=2)),
                 ("polish 1", dict(polish=1))]:
    print(f"{name:48s} {run(**kw):7.2f} us")
