"""Developer probe: how much of a trajectory filter's step is the bevel of the soft saturation (the divergent
sqrt + division of BackupLoop::saturateSoft, taken by a wave whenever ANY of its lanes is inside a bevel)?
Times filter() of one config at several satSharpness values (the bevel's width scales with it) and at an input range so
wide that no trajectory saturates.  Not product code, not a test.   python tools/dev_sat_time.py [cfg] [B]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from asif_amd import capi, workloads  # noqa: E402

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
model, variant, B0 = capi.CONFIGS[cfg]
B = int(sys.argv[2]) if len(sys.argv) > 2 else B0
dev = torch.device("cuda:0")
x, u = workloads.make_batch(cfg, B)
tx, tu = torch.from_numpy(x).to(dev), torch.from_numpy(u).to(dev)
for label, sharp, wide in (("example", None, False), ("sharpness 0.01", 0.01, False), ("sharpness 0.5", 0.5, False),
                           ("bounds +-1e3 (never saturates)", None, True)):
    o = capi.default_options(model, variant)
    if sharp is not None:
        o.satSharpness = sharp
    if wide:
        o.lb[0], o.ub[0] = -1e3, 1e3
    flt = capi.Filter(model, variant, options=o)
    d = flt.dims
    ua = torch.zeros((d.nu, B), dtype=torch.float64, device=dev)
    rl = torch.zeros((d.nrelax, B), dtype=torch.float64, device=dev)
    rc = torch.zeros(B, dtype=torch.int32, device=dev)
    dg = torch.zeros((d.ndiag, B), dtype=torch.float64, device=dev)
    for _ in range(3):
        flt.filter(tx, tu, ua, rl, rc, dg)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 20
    for _ in range(K):
        flt.filter(tx, tu, ua, rl, rc, dg)
    torch.cuda.synchronize()
    print(f"config {cfg} B {B} {label}: {(time.perf_counter() - t0) / K * 1e6:.1f} us per filter()")
    flt.close()
