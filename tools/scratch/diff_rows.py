#!/usr/bin/env python3
"""Scratch: where do the rows of two builds differ?  python tools/scratch/diff_rows.py libA.so libB.so cfg [B]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if sys.argv[1] == "--child":
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np, gpu_util
    from asif_amd import capi
    cfg = int(sys.argv[2]); B = int(sys.argv[3]) or capi.CONFIGS[cfg][2]
    r = gpu_util.run_assemble(cfg, B)
    np.savez(sys.argv[4], A=r["A"], b=r["b"], code=r["code"], diag=r["diag"], x=r["x"])
    sys.exit(0)
import numpy as np
la, lb, cfg = sys.argv[1], sys.argv[2], sys.argv[3]
B = sys.argv[4] if len(sys.argv) > 4 else "0"
out = []
for k, lib in enumerate((la, lb)):
    f = f"/tmp/rows_{k}.npz"
    subprocess.check_call([sys.executable, __file__, "--child", cfg, B, f], env=dict(os.environ, ASIF_HIP_LIB=os.path.abspath(lib)))
    out.append(np.load(f))
a, b = out
for key in ("A", "b", "diag", "code"):
    d = ~((a[key] == b[key]) | (np.isnan(a[key]) & np.isnan(b[key])))
    cols = np.where(d.reshape(-1, d.shape[-1]).any(0))[0] if d.ndim > 1 else np.where(d)[0]
    print(key, "differing instances", len(cols), cols[:10])
    if len(cols):
        i = cols[0]
        print("  instance", i, "x", a["x"][:, i], "code", a["code"][i], b["code"][i])
        print("  A", a[key].reshape(-1, a[key].shape[-1])[:, i][:12] if key != "code" else "", "\n  B", b[key].reshape(-1, b[key].shape[-1])[:, i][:12] if key != "code" else "")
        print("  diag A", a["diag"][:, i], "\n  diag B", b["diag"][:, i])
