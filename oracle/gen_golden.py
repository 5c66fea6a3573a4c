#!/usr/bin/env python3
"""Generate tests/golden/*.json from the REFERENCE's libaffa (oracle/_ref, built by `make -C oracle ref`
from /root/reference/lib/libaffa/src, unmodified).  Runs only where /root/reference exists; the
JSON it writes is what travels to the GPU box.  TEST INFRASTRUCTURE.

    python oracle/gen_golden.py
"""
import json
import os
import random
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
CAP = 48


def random_program(rng, nreg=10, length=22):
    prog = []
    # seed registers: a mix of proper intervals, point intervals and plain constants
    for r in range(nreg):
        k = rng.random()
        if k < 0.5:
            lo = rng.uniform(-3, 3)
            hi = lo + rng.choice([0.0, 1e-12, rng.uniform(0.01, 1.5), rng.uniform(2.0, 8.0)])
            prog.append((O.OPS["INTERVAL"], r, 0, 0, lo, hi))
        elif k < 0.7:
            v = rng.uniform(-2, 2)
            prog.append((O.OPS["INTERVAL"], r, 0, 0, v, v))
        else:
            prog.append((O.OPS["CONST"], r, 0, 0, rng.uniform(-2, 2), 0.0))
    names = ["ADD", "SUB", "MUL", "DIV", "INV", "NEG", "SCALE", "SIN", "COPY"]
    weights = [4, 3, 4, 1, 1, 1, 2, 2, 1]
    for _ in range(length):
        op = rng.choices(names, weights)[0]
        prog.append((O.OPS[op], rng.randrange(nreg), rng.randrange(nreg), rng.randrange(nreg),
                     rng.uniform(-2, 2), 0.0))
    return prog


def targeted_programs():
    I, C_, ADD, SUB, MUL, DIV, INV, NEG, SCALE, SIN = (O.OPS[k] for k in (
        "INTERVAL", "CONST", "ADD", "SUB", "MUL", "DIV", "INV", "NEG", "SCALE", "SIN"))
    progs = []
    # sin: wide (>= 2pi), tiny (< 1e-10), ordinary, across an extremum
    progs.append([(I, 0, 0, 0, -4.0, 4.0), (SIN, 1, 0, 0, 0, 0), (I, 2, 0, 0, 0.3, 0.3 + 5e-11), (SIN, 3, 2, 0, 0, 0),
                  (I, 4, 0, 0, 1.0, 2.2), (SIN, 5, 4, 0, 0, 0), (I, 6, 0, 0, -0.2, 0.4), (SIN, 7, 6, 0, 0, 0)])
    # inv: positive, negative, straddling zero; division
    progs.append([(I, 0, 0, 0, 70.0, 135.0), (INV, 1, 0, 0, 0, 0), (I, 2, 0, 0, -3.0, -1.5), (INV, 3, 2, 0, 0, 0),
                  (I, 4, 0, 0, -1.0, 2.0), (INV, 5, 4, 0, 0, 0), (I, 6, 0, 0, 5.6, 5.8), (DIV, 7, 6, 0, 0, 0)])
    # the DoubleIntegrator_Robust expression shapes: -F*x/m and K/m with shared symbol m
    progs.append([(I, 0, 0, 0, 70.0, 135.0), (I, 1, 0, 0, 21.0, 21.0), (I, 2, 0, 0, 0.7, 0.7), (NEG, 3, 1, 0, 0, 0),
                  (MUL, 4, 3, 2, 0, 0), (DIV, 5, 4, 0, 0, 0), (I, 6, 0, 0, 5.6, 5.8), (DIV, 7, 6, 0, 0, 0),
                  (ADD, 8, 5, 7, 0, 0), (SUB, 9, 5, 7, 0, 0)])
    # cancellation between shared symbols and constants without symbols
    progs.append([(I, 0, 0, 0, 1.0, 3.0), (SUB, 1, 0, 0, 0, 0), (C_, 2, 0, 0, 2.5, 0), (MUL, 3, 0, 2, 0, 0),
                  (ADD, 4, 3, 0, 0, 0), (SCALE, 5, 4, 0, -0.5, 0), (MUL, 6, 5, 5, 0, 0)])
    return progs


def run_ref(prog, nreg):
    r, out = O.af_run_reference(prog, nreg, CAP)
    rec = {"prog": [list(p) for p in prog], "nreg": nreg, "rc": r}
    if r == 0:
        rec["center"] = out["center"].tolist()
        rec["n"] = out["n"].tolist()
        rec["lo"] = out["lo"].tolist()
        rec["hi"] = out["hi"].tolist()
        rec["idx"] = [out["idx"][k, :out["n"][k]].tolist() for k in range(nreg)]
        rec["coef"] = [out["coef"][k, :out["n"][k]].tolist() for k in range(nreg)]
    return rec


def main():
    O.build(force=True)
    assert O.ref_lib() is not None, "oracle/_ref not built (reference tree missing?)"
    os.makedirs(GOLD, exist_ok=True)
    rng = random.Random(20261003)
    recs = []
    for prog in targeted_programs():
        recs.append(run_ref(prog, 10))
    tries = 0
    while len(recs) < 64 and tries < 1000:
        tries += 1
        rec = run_ref(random_program(rng), 10)
        if rec["rc"] == 0 and max(rec["n"]) <= 40:
            recs.append(rec)
    with open(os.path.join(GOLD, "affa_programs.json"), "w") as f:
        json.dump({"source": "reference lib/libaffa/src via oracle/ref_affa_shim.cpp (ref_affa_run)",
                   "generator": "oracle/gen_golden.py", "cap": CAP, "cases": recs}, f)

    # interval Lie derivatives of the robust pendulum (src/asif_robust.cpp:282-337 on
    # examples/InvertedPendulum_Robust.cpp:53-70), N=4 box half-planes of SURVEY 8(d)
    import ctypes as C
    rl = O.ref_lib()
    o = O.default_options(O.MODEL_IP_ROBUST, O.VAR_ROBUST)
    hp = np.array(list(o.halfPlanes)[:8])
    x, _ = O.make_batch(5, 96)
    x = np.vstack([x, [[0.0, 0.0], [3.0, -3.0], [-1e-3, 2.5]]])
    cases = []
    for xi in x:
        h, flo, fhi, glo, ghi = (np.zeros(4) for _ in range(5))
        xi = np.ascontiguousarray(xi)
        rl.ref_ip_robust_lie(O._p(xi), C.c_double(o.pMin), C.c_double(o.pMax), O._p(hp), 4, O._p(h), O._p(flo),
                             O._p(fhi), O._p(glo), O._p(ghi))
        cases.append({"x": xi.tolist(), "h": h.tolist(), "Lfh_lo": flo.tolist(), "Lfh_hi": fhi.tolist(),
                      "Lgh_lo": glo.tolist(), "Lgh_hi": ghi.tolist()})
    with open(os.path.join(GOLD, "affa_ip_robust_lie.json"), "w") as f:
        json.dump({"source": "reference libaffa via oracle/ref_affa_shim.cpp (ref_ip_robust_lie)",
                   "generator": "oracle/gen_golden.py", "pMin": o.pMin, "pMax": o.pMax,
                   "halfPlanes": hp.tolist(), "cases": cases}, f)
    print("wrote", len(recs), "affine programs and", len(cases), "robust Lie-derivative cases")


if __name__ == "__main__":
    main()
