"""Developer probe: are a config's rows / outputs for the first n instances the same bits whether they are computed as a
batch of n or as the prefix of a larger batch (which may take another kernel path: two-role vs fused pass, LDS modes,
waves per workgroup)?   python tools/dev_prefix_bits.py <cfg> <n> <big>"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import gpu_util  # noqa: E402

cfg, n, big = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
ra, rb = gpu_util.run_assemble(cfg, n), gpu_util.run_assemble(cfg, big)
fa, fb = gpu_util.run_filter(cfg, n), gpu_util.run_filter(cfg, big)
for k in ("A", "b", "code", "diag"):
    x, y = ra[k], rb[k][..., :n]
    same = np.array_equal(x, y, equal_nan=True)
    print(f"config {cfg} {k:5s} {'identical' if same else 'DIFFERENT'}" + ("" if same else f"  {int((x != y).sum())} entries, max |diff| {np.nanmax(np.abs(x - y)):.3g}"))
for k in ("uact", "relax", "rc"):
    x, y = fa[k], fb[k][..., :n]
    same = np.array_equal(x, y, equal_nan=True)
    print(f"config {cfg} {k:5s} {'identical' if same else 'DIFFERENT'}" + ("" if same else f"  {int((x != y).sum())} entries"))
