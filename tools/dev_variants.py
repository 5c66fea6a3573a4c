"""Scratch: time one filter config on alternative builds of libasif_hip.so and compare their outputs bit for bit.

    python tools/dev_variants.py --config 3 --libs asif_amd/libasif_hip.so asif_amd/csrc/build/variants/libasif_x.so

One child process per library (a process can hold one HIP library image of the same name safely); each prints
ms per step, a SHA-256 of (rows, uact, relax, rc) and the rc histogram.  Not product code, not a test.
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(cfg, lib, steps, batch):
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch
    from asif_amd import capi, workloads
    capi.LIB_PATH = os.path.abspath(lib)
    model, variant, B = capi.CONFIGS[cfg]
    B = batch or B
    dev = torch.device("cuda:0")
    if cfg == 10:
        import bench
        flt = capi.Filter(model, variant, options=bench.rb_options(capi, model, variant), device=0)
        flt.set_learning(workloads.make_learning())
    else:
        flt = capi.Filter(model, variant, device=0)
    d = flt.dims
    x, u = workloads.make_batch(cfg, B)
    x, u = torch.from_numpy(x).to(dev), torch.from_numpy(u).to(dev)
    uact = torch.zeros(d.nu, B, dtype=torch.float64, device=dev)
    relax = torch.zeros(max(d.nrelax, 1), B, dtype=torch.float64, device=dev)
    rc = torch.zeros(B, dtype=torch.int32, device=dev)
    A = torch.zeros(d.nc * d.nv, B, dtype=torch.float64, device=dev)
    b = torch.zeros(d.nc, B, dtype=torch.float64, device=dev)
    code = torch.zeros(B, dtype=torch.int32, device=dev)
    flt.assemble(x, A, b, code)
    for _ in range(3):
        flt.filter(x, u, uact, relax, rc)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        flt.filter(x, u, uact, relax, rc)
    e1.record()
    torch.cuda.synchronize()
    h = hashlib.sha256()
    for t in (A, b, code, uact, relax, rc):
        h.update(t.cpu().numpy().tobytes())
    if os.environ.get("DEV_VARIANTS_DUMP"):  # numeric comparison of two builds (bits may differ in the sign of a zero)
        os.makedirs(os.environ["DEV_VARIANTS_DUMP"], exist_ok=True)
        np.savez(os.path.join(os.environ["DEV_VARIANTS_DUMP"], f"c{cfg}_" + os.path.basename(lib) + ".npz"),
                 A=A.cpu().numpy(), b=b.cpu().numpy(), code=code.cpu().numpy(), uact=uact.cpu().numpy(),
                 relax=relax.cpu().numpy(), rc=rc.cpu().numpy())
    vals, cnt = np.unique(rc.cpu().numpy(), return_counts=True)
    print(json.dumps({"lib": lib, "config": cfg, "batch": B, "ms_per_step": e0.elapsed_time(e1) / steps,
                      "sha256": h.hexdigest()[:16], "rc": {int(v): int(c) for v, c in zip(vals, cnt)}}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=3)
    ap.add_argument("--libs", nargs="+", required=True)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--child", default="")
    a = ap.parse_args()
    if a.child:
        child(a.config, a.child, a.steps, a.batch)
        return
    for lib in a.libs:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--config", str(a.config), "--steps", str(a.steps),
                            "--batch", str(a.batch), "--libs", "x", "--child", lib], timeout=300)
        if r.returncode:
            print(json.dumps({"lib": lib, "rc": r.returncode}), flush=True)


if __name__ == "__main__":
    main()
