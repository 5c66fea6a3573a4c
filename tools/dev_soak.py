"""Scratch: the BASELINE configs at their full batch sizes (and the explicit filter on 3 x 2^20 instances) against the
oracle's exact solve: return codes identical on every instance, |u - u_ref| reported.  Not a test (needs a GPU and a
minute or two of host time on 16 threads)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import oracle_lib as O  # noqa: E402
from asif_amd import capi, workloads  # noqa: E402

O.build()
dev = torch.device("cuda:0")
total_mism = 0
for cfg, B, firsts in ((2, 1 << 20, (0, 7 << 20, 123 << 20)), (3, 16384, (0, 16384)), (4, 32768, (0, 32768, 65536)),
                       (5, 8192, (0, 8192, 16384, 24576))):
    model_g, variant_g, _ = capi.CONFIGS[cfg]
    flt = capi.Filter(model_g, variant_g)
    d = flt.dims
    model, variant = O.CONFIGS[cfg]
    o = O.default_options(model, variant)
    for first in firsts:
        x, u = workloads.make_batch(cfg, B, first=first)
        tx, tu = torch.from_numpy(x).to(dev), torch.from_numpy(u).to(dev)
        ua = torch.full((d.nu, B), 9.0, dtype=torch.float64, device=dev)
        rl = torch.zeros((max(d.nrelax, 1), B), dtype=torch.float64, device=dev)
        rc = torch.zeros(B, dtype=torch.int32, device=dev)
        flt.filter(tx, tu, ua, rl, rc)
        torch.cuda.synchronize()
        eu, er, erc = O.filter_batch(model, variant, o, np.ascontiguousarray(x.T), np.ascontiguousarray(u.T),
                                     O.SOLVER_EXACT, nthreads=16, uact_init=np.full((B, d.nu), 9.0))
        g_rc, g_u = rc.cpu().numpy(), ua.cpu().numpy().T
        mism = int((g_rc != erc).sum())
        total_mism += mism
        print("config", cfg, "first", first, "B", B, "rc mismatches", mism, "max |u - u_ref|",
              float(np.abs(g_u - eu).max()), "rc", dict(zip(*[a.tolist() for a in np.unique(g_rc, return_counts=True)])),
              flush=True)
    flt.close()
print("total rc mismatches", total_mism)
