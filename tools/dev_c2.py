"""Developer probe (not a test, not the bench): C2 on the GPU -- parity vs the oracle, iteration
statistics and kernel timing across solver settings."""
import sys, os, itertools
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from asif_amd import capi, workloads
import oracle_lib as O, gpu_util

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
B = int(sys.argv[2]) if len(sys.argv) > 2 else capi.CONFIGS[cfg][2]
lanes_list = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else [1, 2]
Ks = [int(v) for v in sys.argv[4].split(",")] if len(sys.argv) > 4 else [1, 2, 5]
rounds_list = [int(v) for v in sys.argv[5].split(",")] if len(sys.argv) > 5 else [12]
refine_list = [int(v) for v in sys.argv[6].split(",")] if len(sys.argv) > 6 else [2, 3]
scal_list = [int(v) for v in sys.argv[7].split(",")] if len(sys.argv) > 7 else [4]
ncheck = min(B, {2: 65536, 3: 512, 4: 8192, 5: 2048}[cfg])
x, u = workloads.make_batch(cfg, B)
ua, rl, rc = gpu_util.oracle_filter(O, cfg, x[:, :ncheck], u[:, :ncheck], uact_init=0.0)
model, variant, _ = capi.CONFIGS[cfg]
dev = torch.device("cuda:0")
for lanes, K, rounds, refine, scal in itertools.product(lanes_list, Ks, rounds_list, refine_list, scal_list):
    s = capi.default_solver(lanes_per_qp=lanes, check_interval=K, active_set_rounds=rounds, refine_steps=refine, scaling_iters=scal)
    out = gpu_util.run_filter(cfg, B, solver=s)
    it = out["diag"][-1]
    mism = (out["rc"][:ncheck] != rc).sum()
    ok = ((rc == 1) | (rc == 2)) & (out["rc"][:ncheck] == rc)
    err = np.abs(out["uact"][0][:ncheck][ok] - ua[0][ok]).max()
    flt = capi.Filter(model, variant, solver=s)
    d = flt.dims
    tx = torch.from_numpy(x).to(dev); tu = torch.from_numpy(u).to(dev)
    uact = torch.zeros((d.nu, B), dtype=torch.float64, device=dev); relax = torch.zeros((d.nrelax, B), dtype=torch.float64, device=dev)
    trc = torch.zeros(B, dtype=torch.int32, device=dev)
    for _ in range(3): flt.filter(tx, tu, uact, relax, trc)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n): flt.filter(tx, tu, uact, relax, trc)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print(f"cfg {cfg} lanes {lanes} K {K} rounds {rounds} refine {refine} scal {scal}: rc mism {mism} max|du| {err:.1e} iters mean {it.mean():.1f} "
          f"p99 {np.percentile(it, 99):.0f} max {it.max():.0f} | {ms*1e3:.1f} us/launch -> {B/ms/1e3:.1f} M inst/s", flush=True)
