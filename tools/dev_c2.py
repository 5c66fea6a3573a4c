"""Developer probe (not a test, not the bench): C2 on the GPU -- parity vs the oracle, iteration
statistics and kernel timing for each lanes-per-QP mapping."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from asif_amd import capi, workloads
import oracle_lib as O, gpu_util

print(torch.cuda.get_device_name(0), torch.cuda.get_device_properties(0).gcnArchName)
B = 65536
x, u = workloads.make_batch(2, B)
ua, rl, rc = gpu_util.oracle_filter(O, 2, x, u, uact_init=0.0)
for lanes in (1, 2, 4):
    for K in (5, 10, 20):
        s = capi.default_solver(lanes_per_qp=lanes, check_interval=K)
        out = gpu_util.run_filter(2, B, solver=s)
        it = out["diag"][-1]
        mism = (out["rc"] != rc).sum()
        ok = (rc == 1) & (out["rc"] == 1)
        err = np.abs(out["uact"][0][ok] - ua[0][ok]).max()
        flt = capi.Filter(capi.MODEL_DOUBLE_INTEGRATOR, capi.EXPLICIT, solver=s)
        dev = torch.device("cuda:0")
        tx = torch.from_numpy(x).to(dev); tu = torch.from_numpy(u).to(dev)
        uact = torch.zeros((1, B), dtype=torch.float64, device=dev); relax = torch.zeros_like(uact)
        trc = torch.zeros(B, dtype=torch.int32, device=dev)
        for _ in range(5): flt.filter(tx, tu, uact, relax, trc)
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        n = 50
        e0.record()
        for _ in range(n): flt.filter(tx, tu, uact, relax, trc)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        print(f"lanes {lanes} K {K}: rc mism {mism} max|du| {err:.2e} iters feas mean {it[rc==1].mean():.1f} max {it[rc==1].max():.0f} "
              f"infeas mean {it[rc==-1].mean():.1f} max {it[rc==-1].max():.0f} | {ms*1e3:.1f} us/launch -> {B/ms/1e3:.1f} M solves/s")
