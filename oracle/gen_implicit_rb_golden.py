#!/usr/bin/env python3
"""Golden vectors for the interval safety set ASIFimplicitRB evaluates at the critical samples
(src/asif_implicit_robust.cpp:635-647): x_int[i] = interval(x[i]-x_unc[i], x[i]+x_unc[i]), the box safety set
of examples/InvertedPendulum_Implicit.cpp:31-37 / examples/DoubleIntegrator_implicit.cpp:32-38 written on
interval_t, h = h_int.convert().left().  Computed by the REFERENCE's libaffa (oracle/_ref) through the
instruction-program shim (ref_affa_run).  Runs only where /root/reference exists; the JSON travels.
TEST INFRASTRUCTURE.

    python oracle/gen_implicit_rb_golden.py
"""
import json
import math
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402


def program(x, unc, lo, hi):
    """h0 = -x0 + hi, h1 = x0 - lo, h2 = x1 - lo, h3 = -x1 + hi on registers 0..9 (h in 4, 6, 7, 9)."""
    I, C_, ADD, SUB, NEG = (O.OPS[k] for k in ("INTERVAL", "CONST", "ADD", "SUB", "NEG"))
    return [(I, 0, 0, 0, x[0] - unc[0], x[0] + unc[0]), (I, 1, 0, 0, x[1] - unc[1], x[1] + unc[1]),
            (NEG, 2, 0, 0, 0, 0), (C_, 3, 0, 0, hi, 0), (ADD, 4, 2, 3, 0, 0), (C_, 5, 0, 0, lo, 0),
            (SUB, 6, 0, 5, 0, 0), (SUB, 7, 1, 5, 0, 0), (NEG, 8, 1, 0, 0, 0), (ADD, 9, 8, 3, 0, 0)]


def main():
    O.build(force=True)
    assert O.ref_lib() is not None, "oracle/_ref not built (reference tree missing?)"
    rng = random.Random(20261010)
    cases = []
    for k in range(160):
        model = "pendulum" if k % 2 == 0 else "double_integrator"
        lo, hi = (-math.pi, math.pi) if model == "pendulum" else (-1.0, 1.0)
        x = [rng.uniform(-1.3, 1.3) * hi, rng.uniform(-1.3, 1.3) * hi]
        unc = [rng.choice([0.0, 1e-9, 0.01, 0.02, rng.uniform(0, 0.5)]),
               rng.choice([0.0, 1e-9, 0.01, 0.02, rng.uniform(0, 0.5)])]
        r, out = O.af_run_reference(program(x, unc, lo, hi), 10, 48)
        assert r == 0
        cases.append({"model": model, "x": x, "x_unc": unc,
                      "h_lo": [float(out["lo"][j]) for j in (4, 6, 7, 9)],
                      "h_hi": [float(out["hi"][j]) for j in (4, 6, 7, 9)]})
    path = os.path.join(ROOT, "tests", "golden", "affa_box_safety_interval.json")
    with open(path, "w") as f:
        json.dump({"source": "reference lib/libaffa/src via oracle/ref_affa_shim.cpp (ref_affa_run)",
                   "generator": "oracle/gen_implicit_rb_golden.py", "cases": cases}, f)
    print("wrote", len(cases), "interval safety-set cases")


if __name__ == "__main__":
    main()
