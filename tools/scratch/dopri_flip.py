#!/usr/bin/env python3
"""Scratch (CPU only): the instance on which the dopri5 soak found its largest |uAct - u_ref| for config 3
(tools/scratch/dopri_worst.py), in the ORACLE alone under last-bit perturbations of the state: the forty safety rows
move by 1e-15, the end-of-horizon row by 4e-9 ... 2e-6 -- the adaptive controller's accept / grow decisions late in the
horizon (steps bounded by stability, error estimate hovering at its thresholds) flip, and each step sequence carries
its own global error.  The device's row sits among these."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O
O.build()
model, variant = O.CONFIGS[3]
oo = O.default_options(model, variant)
oo.integrator = 1
oo.backTrajAbsTol = oo.backTrajRelTol = float(sys.argv[1]) if len(sys.argv) > 1 else 1e-8
x0 = np.array([[1.037861610228091, 1.4332059189852728]])
A0, b0, _, _ = O.assemble_batch(model, variant, oo, x0)
nc = b0.shape[1]
R0 = A0[0].reshape(3, nc).T
print("end-of-horizon row", R0[40].tolist(), b0[0][40])
for k in range(1, 13):
    x = x0.copy()
    x[0, 0] = np.nextafter(x[0, 0], 10.0) if k % 2 else x[0, 0]
    x[0, 1] = x0[0, 1] * (1 + k * 2.3e-16)
    A, b, _, _ = O.assemble_batch(model, variant, oo, x)
    R = A[0].reshape(3, nc).T
    print(k, "safety rows move by %.2e, end-of-horizon row by %.2e (A) %.2e (b)" % (np.abs(R[:40] - R0[:40]).max(), np.abs(R[40] - R0[40]).max(), abs(b[0][40] - b0[0][40])))
