"""Scratch: where the wave-per-QP LDS kernel spends its time on the lifted 18 x 12 robust QP -- the same batch under
different solver settings (scaling iterations, Newton cap).  Not product code, not a test.
    python tools/dev_lds_time.py
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from asif_amd import capi  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    B = int(os.environ.get("B", 8192))
    P = bench.qp_problem(5, B, dev)
    sol = torch.zeros((P["nv"], B), dtype=torch.float64, device=dev)
    st = torch.zeros(B, dtype=torch.int32, device=dev)
    it = torch.zeros(B, dtype=torch.int32, device=dev)
    for kw in (dict(), dict(scaling_iters=-1), dict(scaling_iters=1), dict(scaling_iters=2), dict(scaling_iters=8), dict(max_iter=1)):
        s = capi.default_solver(**kw)
        for _ in range(2):
            capi.qp_solve_batch(P["Hd"], P["c"], P["A"], P["b"], P["lb"], P["ub"], sol, st, it, be=P["be"], solver=s)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            capi.qp_solve_batch(P["Hd"], P["c"], P["A"], P["b"], P["lb"], P["ub"], sol, st, it, be=P["be"], solver=s)
        e1.record()
        torch.cuda.synchronize()
        v, c = np.unique(st.cpu().numpy(), return_counts=True)
        print(json.dumps({"settings": kw, "ms": e0.elapsed_time(e1) / 5, "newton_mean": float(it.double().mean()),
                          "newton_max": int(it.max()), "status": {int(a): int(b) for a, b in zip(v, c)}}), flush=True)


if __name__ == "__main__":
    main()
