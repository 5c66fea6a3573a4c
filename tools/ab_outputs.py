#!/usr/bin/env python3
"""Bitwise A/B of two builds of the library on the same inputs (GPU box): rows (asif_hip_assemble_batch), uAct, relax,
rc of the given configs at their BASELINE batch sizes.  Each build runs in its own process (ASIF_HIP_LIB).
   python tools/ab_outputs.py <libA.so> <libB.so> <cfg> [<cfg> ...] [--batch B]"""
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(cfgs, batch):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import gpu_util
    from asif_amd import capi
    out = {}
    for cfg in cfgs:
        B = batch or capi.CONFIGS[cfg][2]
        rows = gpu_util.run_assemble(cfg, B)
        flt = gpu_util.run_filter(cfg, B)
        out[str(cfg)] = {k: hashlib.sha256(v.tobytes()).hexdigest() for k, v in
                         (("A", rows["A"]), ("b", rows["b"]), ("code", rows["code"]), ("diag", rows["diag"]),
                          ("uact", flt["uact"]), ("relax", flt["relax"]), ("rc", flt["rc"]))}
    print(json.dumps(out))


if __name__ == "__main__":
    if sys.argv[1] == "--child":
        child([int(c) for c in sys.argv[3:]], int(sys.argv[2]))
        sys.exit(0)
    args = [a for a in sys.argv[1:] if not a.startswith("--batch")]
    batch = 0
    if "--batch" in sys.argv:
        batch = int(sys.argv[sys.argv.index("--batch") + 1])
        args = [a for a in args if a != str(batch)]
    libs, cfgs = args[:2], args[2:]
    res = []
    for lib in libs:
        env = dict(os.environ, ASIF_HIP_LIB=os.path.abspath(lib))
        o = subprocess.run([sys.executable, __file__, "--child", str(batch)] + cfgs, env=env, capture_output=True, text=True)
        if o.returncode != 0:
            sys.exit(o.stderr[-2000:])
        res.append(json.loads(o.stdout.strip().split("\n")[-1]))
    bad = 0
    for c in cfgs:
        for k in res[0][c]:
            same = res[0][c][k] == res[1][c][k]
            bad += not same
            print(f"config {c} {k:6s} {'identical' if same else 'DIFFERENT'} {res[0][c][k][:16]} {res[1][c][k][:16]}")
    sys.exit(1 if bad else 0)
