"""Developer probe: per-lane and per-wave statistics of the active-set finish on C2."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from asif_amd import capi
import gpu_util
B = 65536
for K in (1, 2):
    for scal in (1, 2, 4):
        s = capi.default_solver(check_interval=K, scaling_iters=scal)
        out = gpu_util.run_filter(2, B, solver=s)
        r = out["diag"][0].reshape(-1, 64); f = out["diag"][1].reshape(-1, 64)
        print(f"K {K} scal {scal}: rounds/lane mean {r.mean():.2f}, per-wave max mean {r.max(1).mean():.2f}; farkas iters/lane mean {f.mean():.2f}, per-wave max mean {f.max(1).mean():.2f}, "
              f"per-wave sum-of-max bound n/a; rounds hist {np.bincount(out['diag'][0].astype(int))[:8]}  farkas hist {np.bincount(out['diag'][1].astype(int))[:12]}")
