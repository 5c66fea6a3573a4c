"""Developer probe: which instances does the active-set finish leave to ADMM, and for how long."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from asif_amd import capi
import gpu_util
cfg = int(sys.argv[1]); B = int(sys.argv[2])
for refine in (2, 3, 4):
    for rounds in (12,):
        s = capi.default_solver(check_interval=5, active_set_rounds=rounds, refine_steps=refine)
        out = gpu_util.run_filter(cfg, B, solver=s)
        it = out["diag"][-1]
        slow = np.where(it > 5)[0]
        print(f"cfg {cfg} refine {refine} rounds {rounds}: {len(slow)} instances beyond the first check; worst:")
        for i in slow[np.argsort(-it[slow])][:6]:
            print("   i", i, "iters", it[i], "rc", out["rc"][i], "x", repr(out["x"][:, i].tolist()), "udes", repr(out["udes"][:, i].tolist()), "u", out["uact"][0, i])
