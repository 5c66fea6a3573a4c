"""The HOST affine arithmetic the C++ mirror classes hand to user callbacks (asif_amd/host/include/asif_affine.h:
`AAF`, `interval` with libaffa's interface) replayed on the golden instruction programs that the reference's own
libaffa produced (tests/golden/affa_programs.json): centres, bounds and every coefficient bit for bit.
Pure host C++ (g++), no GPU."""
import json
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden", "affa_programs.json")


def _build(tmp):
    if os.environ.get("ASIF_SAN_DIR"):  # tests/test_sanitizers.py: the build of `make -C tests san`
        return os.path.join(os.environ["ASIF_SAN_DIR"], "host_aaf_san")
    exe = os.path.join(tmp, "host_aaf")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off",
                           "-I" + os.path.join(ROOT, "asif_amd", "host", "include"),
                           os.path.join(ROOT, "tests", "host_aaf_driver.cpp"), "-o", exe])
    return exe


def test_host_aaf_equals_libaffa_on_golden_programs(tmp_path):
    exe = _build(str(tmp_path))
    with open(GOLD) as f:
        cases = json.load(f)["cases"]
    checked = 0
    for c in cases:
        if c["rc"] != 0:
            continue
        lines = [f"{len(c['prog'])} {c['nreg']}"]
        for op, d, a, b, i0, i1 in c["prog"]:
            lines.append(f"{int(op)} {int(d)} {int(a)} {int(b)} {float(i0).hex()} {float(i1).hex()}")
        out = subprocess.run([exe], input="\n".join(lines) + "\n", capture_output=True, text=True, timeout=60)
        assert out.returncode == 0, out.stderr
        rows = out.stdout.strip().split("\n")
        assert len(rows) == c["nreg"]
        for r, row in enumerate(rows):
            v = row.split()
            center, lo, hi, n = float.fromhex(v[0]), float.fromhex(v[1]), float.fromhex(v[2]), int(v[3])
            coef = [float.fromhex(t) for t in v[4:]]
            assert n == c["n"][r], (r, c["prog"])
            same = lambda a, b: (a == b) or (np.isnan(a) and np.isnan(b))  # noqa: E731
            assert same(center, c["center"][r]) and same(lo, c["lo"][r]) and same(hi, c["hi"][r]), (r, c["prog"])
            assert all(same(x, y) for x, y in zip(coef, c["coef"][r])), (r, c["prog"])
            checked += 1
    assert checked >= 400
