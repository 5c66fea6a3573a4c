#!/usr/bin/env python3
"""Numeric A/B of two builds on one config (GPU box): max |difference| of rows, uAct, relax; rc equality.
   python tools/ab_diff.py <libA.so> <libB.so> <cfg> [batch]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if sys.argv[1] == "--child":
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np, gpu_util
    from asif_amd import capi
    cfg, B = int(sys.argv[2]), int(sys.argv[3])
    B = B or capi.CONFIGS[cfg][2]
    rows = gpu_util.run_assemble(cfg, B); flt = gpu_util.run_filter(cfg, B)
    np.savez(sys.argv[4], A=rows["A"], b=rows["b"], code=rows["code"], diag=rows["diag"], uact=flt["uact"], relax=flt["relax"], rc=flt["rc"])
    sys.exit(0)
import numpy as np
libs, cfg = sys.argv[1:3], sys.argv[3]
B = sys.argv[4] if len(sys.argv) > 4 else "0"
out = []
for k, lib in enumerate(libs):
    f = f"/tmp/ab_diff_{k}.npz"
    r = subprocess.run([sys.executable, __file__, "--child", cfg, B, f], env=dict(os.environ, ASIF_HIP_LIB=os.path.abspath(lib)), capture_output=True, text=True)
    if r.returncode: sys.exit(r.stderr[-1500:])
    out.append(np.load(f))
a, b = out
for k in ("A", "b", "diag", "uact", "relax"):
    d = np.abs(a[k] - b[k]); m = np.isfinite(d)
    rel = d[m] / (1e-300 + np.maximum(np.abs(a[k][m]), 1.0))
    print(f"{k:6s} max abs diff {d[m].max():.3e}  max rel (floor 1) {rel.max():.3e}  entries differing {(d[m] > 0).sum()} / {d.size}")
print("code equal", np.array_equal(a["code"], b["code"]), " rc equal", np.array_equal(a["rc"], b["rc"]))
if "--detail" in sys.argv:
    Bn = a["rc"].shape[0]
    dA = np.abs(a["A"] - b["A"]).reshape(-1, Bn); db = np.abs(a["b"] - b["b"]).reshape(-1, Bn)
    inst = np.where((dA > 0).any(0) | (db > 0).any(0))[0]
    for i in inst[:20]:
        print("instance", i, "code", a["code"][i], "idxHit", a["diag"][2, i], "crit idx", a["diag"][3:7, i], "A rows differing", np.where(dA[:, i] > 0)[0], "b rows", np.where(db[:, i] > 0)[0])
