#!/usr/bin/env python3
"""Fixtures for the robust filter on the data the reference ships.  Runs only where /root/reference exists.
TEST INFRASTRUCTURE.

  asif_amd/data/robust_halfplanes.json   SafetySetData of include/KernelData_{70-135kg,70-75kg}.h (numbers only)
  tests/golden/affa_di_robust_lie.json  interval Lie derivatives of examples/DoubleIntegrator_Robust.cpp at seeded
                                        point states for the npSSmax = 5 smallest-h half-planes, computed by the
                                        REFERENCE's libaffa (oracle/_ref, ref_di_robust_lie)

    python oracle/gen_robust_golden.py
"""
import ctypes as C
import json
import os
import re
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402

REF_INC = "/root/reference/include"
GOLD = os.path.join(ROOT, "tests", "golden")
NAMES = ["70-135kg", "70-75kg"]


def parse_header(path):
    txt = open(path).read()
    body = txt[txt.index("{", txt.index("SafetySetData")):txt.rindex("};") + 1]
    body = re.sub(r"//.*", "", body).replace("{", "[").replace("}", "]")
    return json.loads(body)


def main():
    O.build(force=True)
    rl = O.ref_lib()
    assert rl is not None, "oracle/_ref not built (reference tree missing?)"
    sets = {n: parse_header(os.path.join(REF_INC, "KernelData_%s.h" % n)) for n in NAMES}
    with open(os.path.join(os.path.dirname(GOLD), "..", "asif_amd", "data", "robust_halfplanes.json"), "w") as f:
        json.dump({"source": "SafetySetData of include/KernelData_{%s}.h of the reference (data only)" % ",".join(NAMES),
                   "generator": "oracle/gen_robust_golden.py", "sets": sets}, f)
    d = O.RbDesc()
    O.lib().or_rb_default(C.byref(d))
    par = [d.mMin, d.mMax, d.Klo, d.Khi, d.Flo, d.Fhi]
    hp = O.load_halfplanes("70-135kg")
    x, _ = O.make_batch_robust_data(hp, 96)
    x = np.vstack([x, [[0.0, 0.0], [2.9, 0.0], [-0.5, 2.95]]])
    cases = []
    for xi in x:
        h = 1.0 - hp[:, 0] * xi[0] - hp[:, 1] * xi[1]
        sel = np.argsort(h, kind="stable")[:5]
        rows = np.ascontiguousarray(hp[sel])
        out = np.zeros((5, 4))
        xi = np.ascontiguousarray(xi)
        r = rl.ref_di_robust_lie(O._p(xi), 5, O._p(rows), *[C.c_double(p) for p in par], O._p(out))
        assert r == 0
        cases.append({"x": xi.tolist(), "sel": sel.tolist(), "lie": out.tolist()})
    with open(os.path.join(GOLD, "affa_di_robust_lie.json"), "w") as f:
        json.dump({"source": "reference libaffa via oracle/ref_affa_shim.cpp (ref_di_robust_lie)",
                   "generator": "oracle/gen_robust_golden.py", "set": "70-135kg", "params": par, "cases": cases}, f)
    print("wrote", {n: len(v) for n, v in sets.items()}, "half-planes and", len(cases), "Lie-derivative cases")


if __name__ == "__main__":
    main()
