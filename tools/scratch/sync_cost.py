"""Scratch: where a 20-step timed region spends its wall time -- graph launch call, GPU execution, and how long
torch.cuda.synchronize() takes to notice the end compared with spinning on hipEventQuery."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from asif_amd import capi, workloads  # noqa: E402

dev = torch.device("cuda:0")
B, K = 65536, 20
flt = capi.Filter(capi.MODEL_DOUBLE_INTEGRATOR, capi.EXPLICIT)
x, u = workloads.make_batch(2, B)
tx, tu = torch.from_numpy(x).to(dev), torch.from_numpy(u).to(dev)
ua = torch.zeros(1, B, dtype=torch.float64, device=dev); rl = torch.zeros(1, B, dtype=torch.float64, device=dev)
rc = torch.zeros(B, dtype=torch.int32, device=dev)
fn = flt.lib.asif_hip_filter_batch
st = torch.cuda.Stream(device=dev)
args = (flt.handle, B, tx.stride(0), C.c_void_p(tx.data_ptr()), C.c_void_p(tu.data_ptr()), C.c_void_p(ua.data_ptr()),
        C.c_void_p(rl.data_ptr()), C.c_void_p(rc.data_ptr()), None, C.c_void_p(st.cuda_stream))
for _ in range(5): fn(*args)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=st):
    for _ in range(K): fn(*args)
torch.cuda.synchronize()
hip = C.CDLL("libamdhip64.so")
hip.hipGraphUpload.argtypes = [C.c_void_p, C.c_void_p]; hip.hipGraphLaunch.argtypes = [C.c_void_p, C.c_void_p]
hip.hipEventRecord.argtypes = [C.c_void_p, C.c_void_p]; hip.hipEventQuery.argtypes = [C.c_void_p]
ex = C.c_void_p(g.raw_cuda_graph_exec()); sp = C.c_void_p(st.cuda_stream)
hip.hipGraphUpload(ex, sp); torch.cuda.synchronize()
ev = C.c_void_p(); hip.hipEventCreate(C.byref(ev))
for mode in ("sync", "spin", "sync", "spin"):
    ts = []
    for rep in range(20):
        torch.cuda.synchronize(); time.sleep(0.002)
        t0 = time.perf_counter()
        hip.hipGraphLaunch(ex, sp); hip.hipEventRecord(ev, sp)
        t1 = time.perf_counter()
        if mode == "spin":
            while hip.hipEventQuery(ev) != 0:
                pass
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        ts.append(((t1 - t0) * 1e6, (t2 - t0) * 1e6))
    ts.sort(key=lambda p: p[1])
    print(mode, "launch call %.1f us, until synchronize returns %.1f us (median of 20)" % ts[10], flush=True)
