import sys, os, time, ctypes as C
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from asif_amd import capi, workloads
dev = torch.device("cuda:0")
for B in (65536, 8192, 262144):
    flt = capi.Filter(capi.MODEL_DOUBLE_INTEGRATOR, capi.EXPLICIT)
    x, u = workloads.make_batch(2, B)
    tx, tu = torch.from_numpy(x).to(dev), torch.from_numpy(u).to(dev)
    ua = torch.zeros(1, B, dtype=torch.float64, device=dev); rl = torch.zeros(1, B, dtype=torch.float64, device=dev)
    rc = torch.zeros(B, dtype=torch.int32, device=dev)
    fn = flt.lib.asif_hip_filter_batch
    st = torch.cuda.Stream(device=dev)
    args = (flt.handle, B, tx.stride(0), C.c_void_p(tx.data_ptr()), C.c_void_p(tu.data_ptr()), C.c_void_p(ua.data_ptr()),
            C.c_void_p(rl.data_ptr()), C.c_void_p(rc.data_ptr()), None, C.c_void_p(st.cuda_stream))
    for _ in range(50): fn(*args)
    torch.cuda.synchronize()
    K = 2000
    t0 = time.perf_counter()
    for _ in range(K): fn(*args)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("B", B, "host enqueue us/launch", (t1 - t0) / K * 1e6, "total us/step", (t2 - t0) / K * 1e6, flush=True)
    flt.close()
