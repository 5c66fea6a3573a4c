"""The DEVICE affine arithmetic (asif_amd/csrc/affine_dev.hpp, behind the robust and realizable rows) replays
the instruction programs whose results the reference's own libaffa produced (tests/golden/affa_programs.json,
generated through oracle/ref_affa_shim.cpp): symbol counts and indexes identical, centres / bounds / coefficients
bit for bit for programs without sin(), to 1e-13 absolute (relative to max(1, |centre|)) where the device's libm sin()
is involved."""
import ctypes as C
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "affa_programs.json")


class Instr(C.Structure):
    _fields_ = [("op", C.c_int32), ("dst", C.c_int32), ("a", C.c_int32), ("b", C.c_int32),
                ("imm0", C.c_double), ("imm1", C.c_double)]


def _close(a, b, ulps):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    same = (a == b) | (np.isnan(a) & np.isnan(b))
    if ulps == 0:
        return bool(np.all(same))
    fin = np.isfinite(a) & np.isfinite(b)
    tol = ulps * np.spacing(np.maximum(np.abs(a), np.abs(b)))
    return bool(np.all(same | (fin & (np.abs(a - b) <= tol))))


def _near(a, b, atol):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    same = (a == b) | (np.isnan(a) & np.isnan(b))
    with np.errstate(invalid="ignore"):
        return bool(np.all(same | (np.abs(a - b) <= atol)))


def test_device_affine_forms_equal_libaffa_on_golden_programs(hip):
    lib = hip.load()
    with open(GOLD) as f:
        cases = json.load(f)["cases"]
    n_exact = n_sin = 0
    for c in cases:
        if c["rc"] != 0:
            continue
        prog = (Instr * len(c["prog"]))(*[Instr(int(p[0]), int(p[1]), int(p[2]), int(p[3]), float(p[4]), float(p[5]))
                                          for p in c["prog"]])
        nreg = c["nreg"]
        center, lo, hi = (np.zeros(nreg) for _ in range(3))
        n = np.zeros(nreg, dtype=np.int32)
        idx = np.zeros((nreg, 16), dtype=np.uint32)
        coef = np.zeros((nreg, 16))
        vp = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
        r = lib.asif_hip_affine_replay(0, prog, len(c["prog"]), nreg, vp(center), vp(n), vp(lo), vp(hi), vp(idx),
                                       vp(coef))
        assert r == 0, (r, c["prog"])
        has_sin = any(int(p[0]) == 9 for p in c["prog"])
        assert n.tolist() == c["n"], c["prog"]
        for k in range(nreg):
            assert idx[k, :n[k]].tolist() == c["idx"][k], (k, c["prog"])
            # the radius of a sin() form is a max of residuals of nearly equal size: a last-bit difference of libm
            # can move it by more than a few ulp of itself, so compare it against the size of the form
            scale = max(1.0, abs(c["center"][k])) if np.isfinite(c["center"][k]) else 1.0
            # a sin() form's centre and radius are differences of nearly equal terms: a last-bit difference between
            # the device's and glibc's sin() shows up at 1e-16 absolute, not at an ulp of the (possibly tiny) result
            assert (_near(center[k], c["center"][k], 1e-13 * scale) if has_sin else _close(center[k], c["center"][k], 0)), \
                (k, c["prog"])
            if has_sin:
                assert _near(coef[k, :n[k]], c["coef"][k], 1e-13 * scale), (k, c["prog"])
                assert _near([lo[k], hi[k]], [c["lo"][k], c["hi"][k]], 1e-12 * scale), (k, c["prog"])
            else:
                assert _close(coef[k, :n[k]], c["coef"][k], 0), (k, c["prog"])
                assert _close([lo[k], hi[k]], [c["lo"][k], c["hi"][k]], 0), (k, c["prog"])
        n_sin += has_sin
        n_exact += not has_sin
    assert n_exact >= 5 and n_exact + n_sin >= 60


def test_replay_argument_validation(hip):
    lib = hip.load()
    prog = (Instr * 1)(Instr(0, 20, 0, 0, 1.0, 0.0))
    z = np.zeros(64)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    zi = np.zeros(64, dtype=np.int32)
    assert lib.asif_hip_affine_replay(0, prog, 1, 4, vp(z), vp(zi), vp(z), vp(z), vp(zi), vp(z)) != 0  # dst out of range
    assert lib.asif_hip_affine_replay(0, prog, 1, 99, vp(z), vp(zi), vp(z), vp(z), vp(zi), vp(z)) != 0  # too many registers
