"""Developer probe: cost of the solver's parts on C2 (setup / iterations / finish), by switching them off."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from asif_amd import capi, workloads
B = 65536
x, u = workloads.make_batch(2, B)
dev = torch.device("cuda:0")
tx = torch.from_numpy(x).to(dev); tu = torch.from_numpy(u).to(dev)
uact = torch.zeros((1, B), dtype=torch.float64, device=dev); relax = torch.zeros_like(uact)
rc = torch.zeros(B, dtype=torch.int32, device=dev)
def t(**kw):
    flt = capi.Filter(capi.MODEL_DOUBLE_INTEGRATOR, capi.EXPLICIT, solver=capi.default_solver(**kw))
    for _ in range(3): flt.filter(tx, tu, uact, relax, rc)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): flt.filter(tx, tu, uact, relax, rc)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 30 * 1e3
print("empty-ish: max_iter 0                      %.1f us" % t(max_iter=0, polish=0, scaling_iters=0))
print("scaling 4 only (max_iter 0)               %.1f us" % t(max_iter=0, polish=0, scaling_iters=4))
print("scaling 1 only (max_iter 0)               %.1f us" % t(max_iter=0, polish=0, scaling_iters=1))
print("scal 4 + 2 iters + residual check, no fin %.1f us" % t(max_iter=2, polish=0, scaling_iters=4, check_interval=2))
print("scal 4 + 10 iters + residual check        %.1f us" % t(max_iter=10, polish=0, scaling_iters=4, check_interval=10))
print("scal 4 + 2 iters + finish rounds 0        %.1f us" % t(max_iter=2, polish=1, active_set_rounds=0, scaling_iters=4, check_interval=2))
print("scal 4 + 2 iters + finish rounds 1        %.1f us" % t(max_iter=2, polish=1, active_set_rounds=1, scaling_iters=4, check_interval=2))
print("scal 4 + 2 iters + finish rounds 2        %.1f us" % t(max_iter=2, polish=1, active_set_rounds=2, scaling_iters=4, check_interval=2))
print("scal 4 + 2 iters + finish rounds 12       %.1f us" % t(max_iter=2, polish=1, active_set_rounds=12, scaling_iters=4, check_interval=2))
for K in (2, 3, 4, 6, 8):
    print("default, K=%d                               %.1f us" % (K, t(check_interval=K)))
print("default refine 1                          %.1f us" % t(refine_steps=1))
print("default refine 3                          %.1f us" % t(refine_steps=3))
