#!/usr/bin/env python3
"""Generate the realizable-filter fixtures.  Runs only where /root/reference exists.  TEST INFRASTRUCTURE.

  asif_amd/data/realizable_kernels.json   the polytope DATA of include/RealizableKernelData_*.h (numbers only:
                                         vertices, facet vertex indexes, normals, active sets, the two limits)
  tests/golden/affa_rz_facet_lie.json    interval Lie derivatives over every (facet, active constraint) pair
                                         and point-state dynamics midpoints, computed by the REFERENCE's libaffa
                                         (oracle/_ref, ref_rz_facet_lie / ref_rz_point_dynamics)

    python oracle/gen_realizable_golden.py
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
NAMES = ["100Hz", "100Hz_50pt", "10Hz", "10Hz_50pt"]


def parse_header(path):
    txt = open(path).read()
    body = txt[txt.index("=", txt.index("kernel_t")) + 1:txt.rindex("};") + 1]
    body = re.sub(r"//.*", "", body).replace("{", "[").replace("}", "]")
    vertices, facets, max_crit, max_act = json.loads(body)
    return {"vertices": vertices, "facetVertices": [f[0] for f in facets], "facetNormals": [f[1] for f in facets],
            "facetActive": [f[2] for f in facets], "maxCriticalFacets": max_crit, "maxActiveConstraints": max_act}


def main():
    O.build(force=True)
    rl = O.ref_lib()
    assert rl is not None, "oracle/_ref not built (reference tree missing?)"
    kernels = {n: parse_header(os.path.join(REF_INC, "RealizableKernelData_%s.h" % n)) for n in NAMES}
    with open(os.path.join(os.path.dirname(GOLD), "..", "asif_amd", "data", "realizable_kernels.json"), "w") as f:
        json.dump({"source": "include/RealizableKernelData_{%s}.h of the reference (data only)" % ",".join(NAMES),
                   "generator": "oracle/gen_realizable_golden.py", "kernels": kernels}, f)
    d = O.RzDesc()
    O.lib().or_rz_default(C.byref(d))
    par = [d.mMin, d.mMax, d.Klo, d.Khi, d.Flo, d.Fhi]
    out = {"source": "reference libaffa via oracle/ref_affa_shim.cpp (ref_rz_facet_lie, ref_rz_point_dynamics)",
           "generator": "oracle/gen_realizable_golden.py", "params": par, "tables": {}, "points": []}
    for n in NAMES:
        k = O.load_kernel(n)
        nF, nA = k["facetVertices"].shape[0], k["maxActiveConstraints"]
        tab = np.zeros((nF, nA, 4))
        r = rl.ref_rz_facet_lie(nF, nA, O._p(k["vertices"]), O._p(k["facetVertices"], C.c_int32),
                                O._p(k["facetNormals"]), O._p(k["facetActive"], C.c_int32),
                                *[C.c_double(p) for p in par], O._p(tab))
        assert r == 0
        out["tables"][n] = tab.tolist()
    x, _ = O.make_batch_realizable(O.load_kernel("100Hz"), 64)
    for xi in np.vstack([x, [[0.0, 0.0], [2.9, -0.0], [-1.0, 3.0]]]):
        f, g = np.zeros(2), np.zeros(2)
        xi = np.ascontiguousarray(xi)
        rl.ref_rz_point_dynamics(O._p(xi), *[C.c_double(p) for p in par], O._p(f), O._p(g))
        out["points"].append({"x": xi.tolist(), "f": f.tolist(), "g": g.tolist()})
    with open(os.path.join(GOLD, "affa_rz_facet_lie.json"), "w") as f:
        json.dump(out, f)
    print("wrote", len(kernels), "kernels,", sum(len(t) for t in out["tables"].values()), "facet rows,",
          len(out["points"]), "point cases")


if __name__ == "__main__":
    main()
