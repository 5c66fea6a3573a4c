"""Scratch: time asif_hip_rollout_batch (T closed-loop steps per launch) on the C2 workload.  Not a test."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from asif_amd import capi, workloads
dev = torch.device("cuda:0")
B, T, dt = 65536, 100, 0.01
flt = capi.Filter(capi.MODEL_DOUBLE_INTEGRATOR, capi.EXPLICIT)
x, u = workloads.make_batch(2, B)
def run():
    tx, tu = torch.from_numpy(x).to(dev), torch.from_numpy(u).to(dev)
    ua = torch.zeros(1, B, dtype=torch.float64, device=dev); rl = torch.zeros(1, B, dtype=torch.float64, device=dev)
    nf = torch.zeros(B, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); flt.rollout(T, dt, tx, tu, ua, rl, nf); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1), int(nf.sum())
for _ in range(3): ms, nf = run()
print(json.dumps({"rollout_ms": ms, "T": T, "B": B, "filter_steps_per_s": B * T / ms * 1e3, "nfail_total": nf}))
