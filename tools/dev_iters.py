"""Developer probe: how many instances the finish-first attempt decides before any ADMM iteration."""
import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import gpu_util
for cfg, B in ((2, 65536), (3, 4096), (4, 32768), (5, 8192), (9, 65536)):
    out = gpu_util.run_filter(cfg, B)
    it = out["diag"][-1]
    solved = np.isin(out["rc"], (1, 2, -1))
    print(f"cfg {cfg}: iterations == 0 for {(it[solved] == 0).mean() * 100:.3f} % of the QPs solved, max {it.max():.0f}, hist {np.unique(it[solved], return_counts=True)}")
