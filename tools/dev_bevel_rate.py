import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np
from asif_amd import capi
rng = np.random.default_rng(7)
tot = bad_q = bad_s = 0
worst = 0.0
for it in range(10):
    n = 4_000_000
    r = np.where(rng.random(n) < 0.5, 0.1, rng.uniform(0.01, 2.0, n))
    t = rng.uniform(0.0, 1.0, n)
    d = r * r * (0.5 + 0.5 * t)
    nn = np.sqrt(np.maximum(r * r - d, 0.0)) * rng.choice([-1.0, 1.0], n)
    sq, q = capi.math_probe(7, d, nn)
    ref = np.sqrt(d); qr = nn / ref
    bad_s += int((sq != ref).sum()); m = q != qr; bad_q += int(m.sum()); tot += n
    if m.any(): worst = max(worst, float((np.abs(q[m] - qr[m]) / np.spacing(np.abs(qr[m]))).max()))
print("samples", tot, "sqrt mismatches", bad_s, "quotient mismatches", bad_q, "worst (ulp)", worst)
