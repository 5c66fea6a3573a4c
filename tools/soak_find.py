#!/usr/bin/env python3
"""Companion of soak_parity.py: scan the same ranges of one config and print the instances whose rc differs from the
oracle's (state, desired input, both answers).   python tools/soak_find.py <cfg> <chunks> [--polish=P]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import gpu_util  # noqa: E402
import oracle_lib as O  # noqa: E402

O.build()
cfg, chunks = int(sys.argv[1]), int(sys.argv[2])
solver = None
for a in sys.argv[3:]:
    if a.startswith("--polish="):
        from asif_amd import capi
        solver = capi.default_solver(polish=int(a.split("=")[1]))
CHUNK = {2: 1 << 20, 3: 8192, 4: 1 << 17, 5: 1 << 18, 8: 4096, 9: 1 << 17, 11: 1 << 19, 12: 1 << 16}
model, variant = O.CONFIGS[cfg]
oo = O.default_options(model, variant)
B, first = CHUNK[cfg], 1 << 24
np.set_printoptions(precision=17)
for c in range(chunks):
    out = gpu_util.run_filter(cfg, B, first=first, uact_init=7.0, relax_init=-7.0, solver=solver)
    d = out["dims"]
    ua, rl, rc = O.filter_batch(model, variant, oo, np.ascontiguousarray(out["x"].T), np.ascontiguousarray(out["udes"].T),
                                O.SOLVER_EXACT, None, min(os.cpu_count() or 8, 16), uact_init=np.full((B, d.nu), 7.0))
    for k in np.where(out["rc"] != rc)[0]:
        it = out["diag"][-1, k] if out["diag"].shape[0] else -1
        print(f"instance {first + k} (iterations {it}): x {out['x'][:, k].tolist()} uDes {out['udes'][:, k].tolist()} device rc {out['rc'][k]} uAct "
              f"{out['uact'][:, k].tolist()} relax {out['relax'][:, k].tolist()} | oracle rc {rc[k]} uAct {ua[k].tolist()} relax {rl[k].tolist()}", flush=True)
    first += B
print("scanned", chunks * B)
