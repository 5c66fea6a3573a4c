"""Developer probe: time one config's filter launch(es) and print iteration statistics."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from asif_amd import capi, workloads
cfg = int(sys.argv[1]); B = int(sys.argv[2]) if len(sys.argv) > 2 else capi.CONFIGS[cfg][2]
lanes_list = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0]
model, variant, _ = capi.CONFIGS[cfg]
x, u = workloads.make_batch(cfg, B)
dev = torch.device("cuda:0")
for lanes in lanes_list:
    flt = capi.Filter(model, variant, solver=capi.default_solver(lanes_per_qp=lanes))
    d = flt.dims
    tx = torch.from_numpy(x).to(dev); tu = torch.from_numpy(u).to(dev)
    uact = torch.zeros((d.nu, B), dtype=torch.float64, device=dev); relax = torch.zeros((d.nrelax, B), dtype=torch.float64, device=dev)
    rc = torch.zeros(B, dtype=torch.int32, device=dev); diag = torch.zeros((d.ndiag, B), dtype=torch.float64, device=dev)
    A = torch.zeros((d.nc * d.nv, B), dtype=torch.float64, device=dev); b = torch.zeros((d.nc, B), dtype=torch.float64, device=dev)
    code = torch.zeros(B, dtype=torch.int32, device=dev)
    for _ in range(2): flt.filter(tx, tu, uact, relax, rc, diag)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e2 = torch.cuda.Event(enable_timing=True)
    n = 5
    e0.record()
    for _ in range(n): flt.assemble(tx, A, b, code)
    e1.record()
    for _ in range(n): flt.filter(tx, tu, uact, relax, rc, diag)
    e2.record(); torch.cuda.synchronize()
    it = diag[-1].cpu().numpy(); r = rc.cpu().numpy()
    vals, cnt = np.unique(r, return_counts=True)
    print(f"cfg {cfg} B {B} lanes {lanes}: assemble {e0.elapsed_time(e1)/n*1e3:.1f} us, filter {e1.elapsed_time(e2)/n*1e3:.1f} us "
          f"-> {B/(e1.elapsed_time(e2)/n)/1e3:.2f} M inst/s | rc {dict(zip(vals.tolist(), cnt.tolist()))} | iters mean {it.mean():.1f} p99 {np.percentile(it,99):.0f} max {it.max():.0f}")
