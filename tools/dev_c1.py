import sys, os, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, oracle_lib as O
exe = os.path.join(ROOT, "asif_amd", "host", "double_integrator")
out = subprocess.run([exe, "--steps", "2500"], capture_output=True, text=True)
rows = np.array([[float(v) for v in l.split(",")] for l in out.stdout.strip().split("\n")[1:]])
model, variant = O.CONFIGS[2]; o = O.default_options(model, variant)
xprev = np.vstack([[0.0, 0.0], rows[:-1, 1:3]])
ua, rl, rc = O.filter_batch(model, variant, o, xprev, np.ones((len(rows), 1)), O.SOLVER_EXACT)
err = np.abs(rows[:, 4] - ua[:, 0])
bad = np.where(err > 1e-6)[0]
print("steps", len(rows), "bad", len(bad), "max err", err.max(), "rc mismatch", (rows[:, 6] != rc).sum())
for k in bad[:10]:
    print(k, "x", xprev[k], "u_gpu", rows[k, 4], "u_ref", ua[k, 0], "relax", rows[k, 5], "rc", rows[k, 6], rc[k])
