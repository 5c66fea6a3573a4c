"""ctypes view of oracle/liboracle.so (and oracle/_ref when it exists).

TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
import this module; the product (asif_amd/) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "liboracle.so")
# tests/test_sanitizers.py re-runs host cases with the builds of `make -C tests san` (address + undefined-behaviour
# sanitizers): ASIF_SAN_DIR names their directory
SAN_DIR = os.environ.get("ASIF_SAN_DIR")
if SAN_DIR:
    LIB_PATH = os.path.join(SAN_DIR, "liboracle_san.so")
REF_LIB_PATH = os.path.join(ORACLE_DIR, "_ref", "libaffa_ref.so")

MODEL_DI, MODEL_IP, MODEL_SEGWAY, MODEL_IP_ROBUST, MODEL_IP_TB = 0, 1, 2, 3, 4
VAR_EXPLICIT, VAR_IMPLICIT, VAR_TB, VAR_ROBUST = 0, 1, 2, 3
VAR_IMPLICIT_RB = 5  # src/asif_implicit_robust.cpp
SOLVER_EXACT, SOLVER_ADMM = 0, 1

# config id (BASELINE.json configs[]) -> (model, variant)
CONFIGS = {2: (MODEL_DI, VAR_EXPLICIT), 3: (MODEL_IP, VAR_IMPLICIT), 4: (MODEL_SEGWAY, VAR_TB),
           5: (MODEL_IP_ROBUST, VAR_ROBUST),
           8: (MODEL_IP_TB, VAR_TB),     # examples/InvertedPendulum_ImplicitTB.cpp (not a BASELINE.json config)
           9: (5, VAR_IMPLICIT),         # examples/DoubleIntegrator_implicit.cpp   (not a BASELINE.json config)
           10: (MODEL_IP, VAR_IMPLICIT_RB),  # ASIFimplicitRB on the pendulum model
           11: (6, VAR_EXPLICIT),        # class ASIF on the synthetic two-input model (no reference example has nu > 1)
           12: (7, VAR_TB)}              # examples/DoubleIntegrator_implicit_tb.cpp (not a BASELINE.json config)
MODEL_P2 = 6


class Learning(C.Structure):
    """or_learning == LearningData of include/asif_learning_utils.h:8-32 (weights column-major)."""
    DIMS = ("d_drift_in", "d_act_in", "d_drift_hidden", "d_act_hidden", "d_drift_hidden_2", "d_act_hidden_2",
            "d_drift_out", "d_act_out")
    PTRS = ("w_1_drift", "w_2_drift", "w_3_drift", "b_1_drift", "b_2_drift", "b_3_drift",
            "w_1_act", "w_2_act", "w_3_act", "b_1_act", "b_2_act", "b_3_act")
    _fields_ = [(n, C.c_uint32) for n in DIMS] + [(n, C.POINTER(C.c_double)) for n in PTRS]

    @classmethod
    def from_dict(cls, w):
        """w: dict with the DIMS as ints and the PTRS as float64 arrays (asif_amd.workloads.make_learning)."""
        L = cls()
        L._keep = {}
        for n in cls.DIMS:
            setattr(L, n, int(w[n]))
        for n in cls.PTRS:
            a = np.ascontiguousarray(w[n], dtype=np.float64)
            L._keep[n] = a
            setattr(L, n, a.ctypes.data_as(C.POINTER(C.c_double)))
        return L


class Options(C.Structure):
    _fields_ = [(n, C.c_double) for n in (
        "relaxCost", "relaxLb", "relaxReachLb", "relaxTTS", "relaxMinOrtho", "backTrajHorizon",
        "backTrajExtend", "backTrajDt", "backTrajMinOrtho", "satSharpness", "inf")] + [
        ("lb", C.c_double * 2), ("ub", C.c_double * 2), ("pMin", C.c_double), ("pMax", C.c_double),
        ("nHalfPlanes", C.c_int32), ("halfPlanes", C.c_double * 16),
        ("backContDt", C.c_double), ("x_unc", C.c_double * 4), ("n_debug", C.c_int32),
        ("use_learning", C.c_int32), ("learning", C.POINTER(Learning)),
        ("npSSmax", C.c_int32), ("integrator", C.c_int32), ("backTrajAbsTol", C.c_double),
        ("backTrajRelTol", C.c_double)]

    def set_learning(self, L):
        self._learning_keep = L
        self.learning = C.pointer(L) if L is not None else None
        self.use_learning = 1 if L is not None else 0


class Dims(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("nx", "nu", "npSS", "npBS", "npBTSS", "nv", "nc", "nrelax", "npBT")]


class AdmmSettings(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("rho", "sigma", "alpha", "eps_abs", "eps_rel", "eps_prim_inf",
                                          "eps_dual_inf")] + [
        (n, C.c_int) for n in ("scaling", "adaptive_rho", "adaptive_rho_interval", "check_termination",
                               "max_iter")] + [("adaptive_rho_tolerance", C.c_double), ("reduced_kkt", C.c_int),
                                               ("polish", C.c_int), ("scaling_pow2", C.c_int)]


class AfInstr(C.Structure):
    _fields_ = [("op", C.c_int), ("dst", C.c_int), ("a", C.c_int), ("b", C.c_int), ("imm0", C.c_double),
                ("imm1", C.c_double)]


OPS = dict(CONST=0, INTERVAL=1, ADD=2, SUB=3, MUL=4, DIV=5, INV=6, NEG=7, SCALE=8, SIN=9, COPY=10)


def build(force=False):
    """Compile liboracle.so (gcc) and, when /root/reference is present, oracle/_ref."""
    if SAN_DIR:
        return  # the sanitizer build is made by tests/Makefile
    if force or not os.path.exists(LIB_PATH) or any(
            os.path.getmtime(os.path.join(ORACLE_DIR, f)) > os.path.getmtime(LIB_PATH)
            for f in os.listdir(ORACLE_DIR) if f.endswith((".c", ".h"))):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s", "liboracle.so"])
    if os.path.isdir("/root/reference/lib/libaffa/src") and (force or not os.path.exists(REF_LIB_PATH)):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s", "ref"])


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(LIB_PATH)
        _lib.or_rng_uniform.restype = C.c_double
        _lib.or_rng_uniform.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64]
        for f in ("or_filter_batch", "or_assemble_batch", "or_qp_solve_batch"):
            getattr(_lib, f).restype = C.c_int64
    return _lib


def ref_lib():
    """The reference's libaffa behind oracle/ref_affa_shim.cpp, or None when not built."""
    if not os.path.exists(REF_LIB_PATH):
        return None
    return C.CDLL(REF_LIB_PATH)


def _p(a, t=C.c_double):
    return a.ctypes.data_as(C.POINTER(t))


def default_options(model, variant):
    o = Options()
    lib().or_default_options(model, variant, C.byref(o))
    return o


def dims(model, variant, o):
    d = Dims()
    r = lib().or_get_dims(model, variant, C.byref(o), C.byref(d))
    assert r == 0
    return d


def admm_settings(**kw):
    s = AdmmSettings()
    lib().or_admm_default_settings(C.byref(s))
    for k, v in kw.items():
        setattr(s, k, v)
    return s


def make_batch(cfg, B, first=0):
    model, variant = CONFIGS[cfg]
    d = dims(model, variant, default_options(model, variant))
    x = np.zeros((B, d.nx))
    u = np.zeros((B, d.nu))
    lib().or_make_batch(cfg, C.c_int64(B), C.c_int64(first), _p(x), _p(u))
    return x, u


def filter_batch(model, variant, o, x, udes, solver=SOLVER_EXACT, settings=None, nthreads=1, uact_init=None):
    d = dims(model, variant, o)
    B = x.shape[0]
    x = np.ascontiguousarray(x, dtype=np.float64)
    udes = np.ascontiguousarray(udes, dtype=np.float64).reshape(B, d.nu)
    uact = np.full((B, d.nu), np.nan) if uact_init is None else np.array(uact_init, dtype=np.float64).reshape(B, d.nu)
    relax = np.full((B, d.nrelax), np.nan)
    rc = np.zeros(B, dtype=np.int32)
    sp = C.byref(settings) if settings is not None else None
    n = lib().or_filter_batch(model, variant, C.byref(o), solver, sp, C.c_int64(B), _p(x), _p(udes), _p(uact),
                              _p(relax), _p(rc, C.c_int32), nthreads)
    assert n == B
    return uact, relax, rc


def assemble_batch(model, variant, o, x):
    d = dims(model, variant, o)
    B = x.shape[0]
    x = np.ascontiguousarray(x, dtype=np.float64)
    A = np.zeros((B, d.nc * d.nv))
    b = np.zeros((B, d.nc))
    code = np.zeros(B, dtype=np.int32)
    diag = np.zeros((B, 8))
    n = lib().or_assemble_batch(model, variant, C.byref(o), C.c_int64(B), _p(x), _p(A), _p(b), _p(code, C.c_int32),
                                _p(diag))
    assert n == B
    return A, b, code, diag


def qp_static(model, variant, o, udes):
    d = dims(model, variant, o)
    Hd, c, lb, ub = (np.zeros(d.nv) for _ in range(4))
    be = np.zeros(d.nc, dtype=np.uint8)
    ud = np.ascontiguousarray(np.atleast_1d(udes), dtype=np.float64)
    lib().or_qp_static(model, variant, C.byref(o), _p(ud), _p(Hd), _p(c), _p(lb), _p(ub), _p(be, C.c_uint8))
    return Hd, c, lb, ub, be


def qp_solve_batch(nv, nc, Hd, c, A, b, lb, ub, be=None, solver=SOLVER_ADMM, settings=None):
    """AoS inputs: Hd,c,lb,ub [B,nv]; A [B,nc*nv] col-major per instance; b [B,nc]."""
    B = c.shape[0]
    arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in (Hd, c, A, b, lb, ub)]
    sol = np.full((B, nv), np.nan)
    status = np.zeros(B, dtype=np.int32)
    iters = np.zeros(B, dtype=np.int32)
    bep = _p(np.ascontiguousarray(be, dtype=np.uint8), C.c_uint8) if be is not None else None
    sp = C.byref(settings) if settings is not None else None
    n = lib().or_qp_solve_batch(nv, nc, solver, sp, C.c_int64(B), *[_p(a) for a in arrs], bep, _p(sol),
                                _p(status, C.c_int32), _p(iters, C.c_int32))
    assert n == B
    return sol, status, iters


def filter_explicit_lie(model, o, x, udes, lfh, lgh):
    """One ASIF::filter(x, uDes, uAct, Lfh, Lgh, relax) per row of x (AoS), exact optimum; returns uact, relax, rc."""
    d = dims(model, VAR_EXPLICIT, o)
    B = x.shape[0]
    arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in (x, np.reshape(udes, (B, d.nu)), lfh, lgh)]
    ua = np.full((B, d.nu), np.nan)
    rl = np.full((B, 1), np.nan)
    rc = np.zeros(B, dtype=np.int32)
    for i in range(B):
        rc[i] = lib().or_filter_explicit_lie(model, C.byref(o), _p(arrs[0][i]), _p(arrs[1][i]), _p(arrs[2][i]),
                                             _p(arrs[3][i]), _p(ua[i]), _p(rl[i]))
    return ua, rl, rc


def last_kept_rows(cap=8):
    idx = np.zeros(cap, dtype=np.int32)
    n = lib().or_last_kept_rows(_p(idx, C.c_int32), cap)
    return idx[:n].copy()


def last_crit_idx(cap=16):
    idx = np.zeros(cap, dtype=np.int32)
    n = lib().or_last_crit_idx(_p(idx, C.c_int32), cap)
    return idx[:n].copy()


def assemble(model, variant, o, x):
    """One or_assemble on the calling thread (so or_last_crit_idx / or_rb_last_learning refer to it)."""
    d = dims(model, variant, o)
    x = np.ascontiguousarray(x, dtype=np.float64)
    A = np.zeros(d.nc * d.nv)
    b = np.zeros(d.nc)
    diag = np.zeros(8)
    code = lib().or_assemble(model, variant, C.byref(o), _p(x), _p(A), _p(b), _p(diag))
    return A, b, code


def rb_safety_lo(model, o, x):
    d = dims(model, VAR_IMPLICIT_RB, o)
    x = np.ascontiguousarray(x, dtype=np.float64)
    h = np.zeros(d.npSS)
    assert lib().or_rb_safety_lo(model, C.byref(o), _p(x), _p(h)) == 0
    return h


def rb_last_learning():
    dh = np.zeros(4)  # OR_MAX_NX
    lf = C.c_double()
    lg = np.zeros(2)  # OR_MAX_NU: the oracle copies the whole array (found by tests/test_sanitizers.py: this was one short)
    lib().or_rb_last_learning(_p(dh), C.byref(lf), _p(lg))
    return dh, lf.value, lg


def _run_program(fn, prog, nreg, cap=48):
    arr = (AfInstr * len(prog))(*[AfInstr(*p) for p in prog])
    center = np.zeros(nreg)
    n = np.zeros(nreg, dtype=np.int32)
    lo = np.zeros(nreg)
    hi = np.zeros(nreg)
    idx = np.zeros(nreg * cap, dtype=np.uint32)
    coef = np.zeros(nreg * cap)
    r = fn(arr, len(prog), nreg, cap, _p(center), _p(n, C.c_int32), _p(lo), _p(hi), _p(idx, C.c_uint32), _p(coef))
    return r, dict(center=center, n=n, lo=lo, hi=hi, idx=idx.reshape(nreg, cap), coef=coef.reshape(nreg, cap))


def af_run_oracle(prog, nreg, cap=48):
    return _run_program(lib().or_af_run, prog, nreg, cap)


def af_run_reference(prog, nreg, cap=48):
    rl = ref_lib()
    assert rl is not None
    return _run_program(rl.ref_affa_run, prog, nreg, cap)


# ------------------------------------------------------------------ realizable (oracle/or_realizable.c)
class RzDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("nx", "nu", "nVertices", "nFacets", "maxCriticalFacets",
                                         "maxActiveConstraints", "npSSmax")] + [
        ("vertices", C.POINTER(C.c_double)), ("facetVertices", C.POINTER(C.c_int32)),
        ("facetNormals", C.POINTER(C.c_double)), ("facetActive", C.POINTER(C.c_int32)),
        ("uncertaintyBounds", C.c_double * 4)] + [
        (n, C.c_double) for n in ("relaxDes", "relaxOffset", "relaxCost", "inf")] + [
        ("lb", C.c_double * 2), ("ub", C.c_double * 2)] + [
        (n, C.c_double) for n in ("mMin", "mMax", "Klo", "Khi", "Flo", "Fhi")]


def load_kernel(name):
    """Kernel polytope data of include/RealizableKernelData_<name>.h, from asif_amd/data/realizable_kernels.json
    (model input data, the same file the product reads)."""
    import json
    with open(os.path.join(ROOT, "asif_amd", "data", "realizable_kernels.json")) as f:
        k = json.load(f)["kernels"][name]
    return dict(vertices=np.array(k["vertices"], dtype=np.float64),
                facetVertices=np.array(k["facetVertices"], dtype=np.int32),
                facetNormals=np.array(k["facetNormals"], dtype=np.float64),
                facetActive=np.array(k["facetActive"], dtype=np.int32),
                maxCriticalFacets=int(k["maxCriticalFacets"]), maxActiveConstraints=int(k["maxActiveConstraints"]))


class Realizable:
    """or_rz handle for one kernel + the example's options (overridable through **kw)."""

    def __init__(self, kernel, **kw):
        L = lib()
        L.or_rz_create.restype = C.c_void_p
        for f in ("or_rz_filter_batch", "or_rz_assemble_batch"):
            getattr(L, f).restype = C.c_int64
        d = RzDesc()
        L.or_rz_default(C.byref(d))
        self._keep = {k: np.ascontiguousarray(kernel[k]) for k in ("vertices", "facetVertices", "facetNormals",
                                                                    "facetActive")}
        d.nVertices = self._keep["vertices"].shape[0]
        d.nFacets = self._keep["facetVertices"].shape[0]
        d.maxCriticalFacets = kernel["maxCriticalFacets"]
        d.maxActiveConstraints = kernel["maxActiveConstraints"]
        d.vertices = _p(self._keep["vertices"])
        d.facetVertices = _p(self._keep["facetVertices"], C.c_int32)
        d.facetNormals = _p(self._keep["facetNormals"])
        d.facetActive = _p(self._keep["facetActive"], C.c_int32)
        for k, v in kw.items():
            if k in ("uncertaintyBounds", "lb", "ub"):
                for i, vi in enumerate(v):
                    getattr(d, k)[i] = vi
            else:
                setattr(d, k, v)
        self.desc = d
        self.h = C.c_void_p(L.or_rz_create(C.byref(d)))
        assert self.h.value, "or_rz_create failed"
        nv, nc, npSS, npSSmax = (C.c_int() for _ in range(4))
        L.or_rz_dims(self.h, C.byref(nv), C.byref(nc), C.byref(npSS), C.byref(npSSmax))
        self.nv, self.nc, self.npSS, self.npSSmax = nv.value, nc.value, npSS.value, npSSmax.value
        self.nFacets, self.nA, self.maxCrit = d.nFacets, d.maxActiveConstraints, d.maxCriticalFacets

    def __del__(self):
        if getattr(self, "h", None) is not None and self.h.value:
            lib().or_rz_destroy(self.h)
            self.h = None

    def table(self):
        t = np.zeros((self.nFacets, self.nA, 4))
        bb = np.zeros((self.nFacets, 2, 2))
        lib().or_rz_table(self.h, _p(t), _p(bb))
        return t, bb

    def assemble(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        B = x.shape[0]
        A = np.zeros((B, self.nc * self.nv))
        b = np.zeros((B, self.nc))
        code = np.zeros(B, dtype=np.int32)
        stride = 1 + self.maxCrit + self.npSSmax
        info = np.zeros((B, stride), dtype=np.int32)
        n = lib().or_rz_assemble_batch(self.h, C.c_int64(B), _p(x), _p(A), _p(b), _p(code, C.c_int32),
                                       _p(info, C.c_int32), stride)
        assert n == B
        return A, b, code, info

    def qp_static(self, udes):
        Hd, c, lb, ub = (np.zeros(self.nv) for _ in range(4))
        be = np.zeros(self.nc, dtype=np.uint8)
        ud = np.ascontiguousarray(np.atleast_1d(udes), dtype=np.float64)
        lib().or_rz_qp_static(self.h, _p(ud), _p(Hd), _p(c), _p(lb), _p(ub), _p(be, C.c_uint8))
        return Hd, c, lb, ub, be

    def filter(self, x, udes, solver=SOLVER_EXACT, settings=None):
        x = np.ascontiguousarray(x, dtype=np.float64)
        B = x.shape[0]
        udes = np.ascontiguousarray(udes, dtype=np.float64).reshape(B, 1)
        uact = np.full((B, 1), np.nan)
        relax = np.full((B, 2), np.nan)
        rc = np.zeros(B, dtype=np.int32)
        sp = C.byref(settings) if settings is not None else None
        n = lib().or_rz_filter_batch(self.h, solver, sp, C.c_int64(B), _p(x), _p(udes), _p(uact), _p(relax),
                                     _p(rc, C.c_int32))
        assert n == B
        return uact, relax, rc


def make_batch_realizable(kernel, B, first=0, seed=6):
    """Config 6 workload: half the states uniform over 1.05 x the kernel's bounding box, half within
    +-2 % (radially) of a random point of a random facet, so that the critical-facet rows are exercised;
    uDes uniform in [-20, 20].  AoS x [B,2], uDes [B,1]."""
    V, FV = kernel["vertices"], kernel["facetVertices"]
    vmax = np.abs(V).max(axis=0)
    nF = FV.shape[0]
    x = np.zeros((B, 2))
    u = np.zeros((B, 1))
    r = lib().or_rng_uniform
    for k in range(B):
        i = first + k
        if r(seed, i, 3) < 0.5:
            fi = min(int(r(seed, i, 4) * nF), nF - 1)
            t = r(seed, i, 5)
            s = 1.0 + 0.02 * (2.0 * r(seed, i, 0) - 1.0)
            x[k] = s * (t * V[FV[fi, 0]] + (1.0 - t) * V[FV[fi, 1]])
        else:
            x[k, 0] = 1.05 * vmax[0] * (2.0 * r(seed, i, 0) - 1.0)
            x[k, 1] = 1.05 * vmax[1] * (2.0 * r(seed, i, 1) - 1.0)
        u[k, 0] = -20.0 + 40.0 * r(seed, i, 2)
    return x, u


# ------------------------------------------- robust filter on shipped half-planes (oracle/or_robust_data.c)
class RbDesc(C.Structure):
    _fields_ = [("N", C.c_int32), ("npSSmax", C.c_int32), ("halfPlanes", C.POINTER(C.c_double))] + [
        (n, C.c_double) for n in ("relaxCost", "relaxLb", "inf")] + [("lb", C.c_double * 2), ("ub", C.c_double * 2)] + [
        (n, C.c_double) for n in ("mMin", "mMax", "Klo", "Khi", "Flo", "Fhi")]


def load_halfplanes(name="70-135kg"):
    """SafetySetData of include/KernelData_<name>.h ([N,2]), from asif_amd/data/robust_halfplanes.json."""
    import json
    with open(os.path.join(ROOT, "asif_amd", "data", "robust_halfplanes.json")) as f:
        return np.array(json.load(f)["sets"][name], dtype=np.float64)


class RobustData:
    """or_rb handle: ASIFrobust of examples/DoubleIntegrator_Robust.cpp on a half-plane set."""

    def __init__(self, halfplanes, **kw):
        L = lib()
        L.or_rb_create.restype = C.c_void_p
        for f in ("or_rb_filter_batch", "or_rb_assemble_batch"):
            getattr(L, f).restype = C.c_int64
        d = RbDesc()
        L.or_rb_default(C.byref(d))
        self._hp = np.ascontiguousarray(halfplanes, dtype=np.float64)
        d.N = self._hp.shape[0]
        d.halfPlanes = _p(self._hp)
        for k, v in kw.items():
            if k in ("lb", "ub"):
                getattr(d, k)[0] = v[0]
            else:
                setattr(d, k, v)
        self.desc = d
        self.h = C.c_void_p(L.or_rb_create(C.byref(d)))
        assert self.h.value, "or_rb_create failed"
        nv, nc, m = (C.c_int() for _ in range(3))
        L.or_rb_dims(self.h, C.byref(nv), C.byref(nc), C.byref(m))
        self.nv, self.nc, self.npSSmax, self.N = nv.value, nc.value, m.value, d.N

    def __del__(self):
        if getattr(self, "h", None) is not None and self.h.value:
            lib().or_rb_destroy(self.h)
            self.h = None

    def assemble(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        B = x.shape[0]
        A = np.zeros((B, self.nc * self.nv))
        b = np.zeros((B, self.nc))
        code = np.zeros(B, dtype=np.int32)
        sel = np.zeros((B, self.npSSmax), dtype=np.int32)
        n = lib().or_rb_assemble_batch(self.h, C.c_int64(B), _p(x), _p(A), _p(b), _p(code, C.c_int32), _p(sel, C.c_int32))
        assert n == B
        return A, b, code, sel

    def qp_static(self, udes):
        Hd, c, lb, ub = (np.zeros(self.nv) for _ in range(4))
        be = np.zeros(self.nc, dtype=np.uint8)
        ud = np.ascontiguousarray(np.atleast_1d(udes), dtype=np.float64)
        lib().or_rb_qp_static(self.h, _p(ud), _p(Hd), _p(c), _p(lb), _p(ub), _p(be, C.c_uint8))
        return Hd, c, lb, ub, be

    def filter(self, x, udes, solver=SOLVER_EXACT, settings=None):
        x = np.ascontiguousarray(x, dtype=np.float64)
        B = x.shape[0]
        udes = np.ascontiguousarray(udes, dtype=np.float64).reshape(B, 1)
        uact = np.full((B, 1), np.nan)
        relax = np.full((B, 1), np.nan)
        rc = np.zeros(B, dtype=np.int32)
        sp = C.byref(settings) if settings is not None else None
        n = lib().or_rb_filter_batch(self.h, solver, sp, C.c_int64(B), _p(x), _p(udes), _p(uact), _p(relax),
                                     _p(rc, C.c_int32))
        assert n == B
        return uact, relax, rc


def make_batch_robust_data(halfplanes, B, first=0, seed=7):
    """Config 7 workload: states uniform over 1.1 x the bounding box of the polygon the half-planes cut out
    (|x_k| <= 1.1 / min_i |a_ik| is too wide; the support is taken from 1/|a| per axis), uDes in [-20, 20]."""
    a = np.abs(halfplanes)
    ext = 1.1 * np.array([1.0 / a[:, 0].max(), 1.0 / a[:, 1].max()])
    x = np.zeros((B, 2))
    u = np.zeros((B, 1))
    r = lib().or_rng_uniform
    for k in range(B):
        i = first + k
        x[k, 0] = ext[0] * (2.0 * r(seed, i, 0) - 1.0)
        x[k, 1] = ext[1] * (2.0 * r(seed, i, 1) - 1.0)
        u[k, 0] = -20.0 + 40.0 * r(seed, i, 2)
    return x, u
