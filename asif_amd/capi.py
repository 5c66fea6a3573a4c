"""ctypes binding of libasif_hip.so, the C ABI declared in include/asif_hip.h.

This is plumbing for tests and bench.py: device memory comes from torch tensors (FP64, contiguous,
SoA [component][B]) whose data_ptr() is handed to the library; the kernels run on torch's current
stream.  There is no fallback: if the shared library is missing or no gfx950 device is usable,
every call raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ASIF_HIP_LIB") or os.path.join(_HERE, "libasif_hip.so")  # env: a developer build to try

MODEL_DOUBLE_INTEGRATOR, MODEL_INVERTED_PENDULUM, MODEL_SEGWAY, MODEL_INVERTED_PENDULUM_ROBUST = 0, 1, 2, 3
EXPLICIT, IMPLICIT, IMPLICIT_TB, ROBUST = 0, 1, 2, 3

# BASELINE.json configs[k] -> (model, variant, default batch)
CONFIGS = {
    2: (MODEL_DOUBLE_INTEGRATOR, EXPLICIT, 65536),
    3: (MODEL_INVERTED_PENDULUM, IMPLICIT, 16384),
    4: (MODEL_SEGWAY, IMPLICIT_TB, 32768),
    5: (MODEL_INVERTED_PENDULUM_ROBUST, ROBUST, 8192),
    # not a BASELINE.json config: examples/InvertedPendulum_ImplicitTB.cpp (11 551-step backup trajectory)
    8: (6, IMPLICIT_TB, 16384),
    # not a BASELINE.json config: examples/DoubleIntegrator_implicit.cpp (201-step trajectory, npBTSS = 4)
    9: (7, IMPLICIT, 65536),
    # not a BASELINE.json config: class ASIFimplicitRB on the pendulum model, workload of
    # asif_amd.workloads (x_unc = RB_X_UNC, seeded networks of make_learning())
    10: (MODEL_INVERTED_PENDULUM, 5, 16384),
    # not a BASELINE.json config and not a reference example: class ASIF on the synthetic two-input model
    11: (8, EXPLICIT, 65536),
    # not a BASELINE.json config: examples/DoubleIntegrator_implicit_tb.cpp (2 101-step trajectory under ASIFimplicitTB)
    12: (9, IMPLICIT_TB, 65536),
}
MODEL_PLANAR_TWO_INPUT = 8
IMPLICIT_RB = 5
MODEL_INVERTED_PENDULUM_TB = 6
MODEL_DOUBLE_INTEGRATOR_IMPLICIT = 7
MODEL_DOUBLE_INTEGRATOR_TB = 9

EXPORTS = [
    "asif_hip_version", "asif_hip_error_string", "asif_hip_device_count", "asif_hip_default_options",
    "asif_hip_default_solver", "asif_hip_create", "asif_hip_destroy", "asif_hip_get_dims",
    "asif_hip_update_options", "asif_hip_filter_batch", "asif_hip_assemble_batch", "asif_hip_qp_solve_batch",
    "asif_hip_filter_batch_host", "asif_hip_default_realizable_options", "asif_hip_create_realizable",
    "asif_hip_update_realizable_options", "asif_hip_realizable_tables", "asif_hip_default_robust_data_options",
    "asif_hip_create_robust_data", "asif_hip_update_robust_data_options", "asif_hip_rollout_batch",
    "asif_hip_set_learning", "asif_hip_affine_replay", "asif_hip_qp_solve_batch_dense",
    "asif_hip_partition", "asif_hip_create_multi", "asif_hip_multi_destroy", "asif_hip_multi_size",
    "asif_hip_multi_handle", "asif_hip_multi_update_options", "asif_hip_filter_batch_host_multi",
    "asif_hip_filter_batch_lie",
    "asif_hip_math_probe", "asif_hip_qp_solve_batch_warm",
]

MODEL_DOUBLE_INTEGRATOR_SAMPLED = 4
MODEL_DOUBLE_INTEGRATOR_ROBUST = 5
REALIZABLE = 4


class RobustDataOptions(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("relaxCost", "relaxLb", "inf")] + [
        ("lb", C.c_double * 2), ("ub", C.c_double * 2), ("npSSmax", C.c_int32)] + [
        (n, C.c_double) for n in ("mMin", "mMax", "Klo", "Khi", "Flo", "Fhi")]


class KernelData(C.Structure):
    """asif_hip_kernel_data: ASIFrealizable::kernel_t flattened (host arrays)."""
    _fields_ = [(n, C.c_int32) for n in ("nx", "nVertices", "nFacets", "maxCriticalFacets", "maxActiveConstraints")] + [
        ("vertices", C.POINTER(C.c_double)), ("facetVertices", C.POINTER(C.c_int32)),
        ("facetNormals", C.POINTER(C.c_double)), ("facetActive", C.POINTER(C.c_int32))]


class RealizableOptions(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("relaxDes", "relaxOffset", "relaxCost", "inf")] + [
        ("lb", C.c_double * 2), ("ub", C.c_double * 2), ("uncertaintyBounds", C.c_double * 4),
        ("npSSmax", C.c_int32)] + [(n, C.c_double) for n in ("mMin", "mMax", "Klo", "Khi", "Flo", "Fhi")]


class Options(C.Structure):
    _fields_ = [(n, C.c_double) for n in (
        "relaxCost", "relaxLb", "relaxReachLb", "relaxTTS", "relaxMinOrtho", "backTrajHorizon",
        "backTrajExtend", "backTrajDt", "backTrajMinOrtho", "satSharpness", "inf")] + [
        ("lb", C.c_double * 2), ("ub", C.c_double * 2), ("pMin", C.c_double), ("pMax", C.c_double),
        ("nHalfPlanes", C.c_int32), ("halfPlanes", C.c_double * 16),
        # ASIFimplicitRB extras (include/asif_implicit_robust.h:24-37)
        ("backContDt", C.c_double), ("x_unc", C.c_double * 4), ("n_debug", C.c_int32), ("use_learning", C.c_int32),
        # class ASIF's npSSmax; ASIFimplicit's USE_ODEINT build (integrator 1) and its tolerances
        ("npSSmax", C.c_int32), ("integrator", C.c_int32), ("backTrajAbsTol", C.c_double),
        ("backTrajRelTol", C.c_double)]


class LearningData(C.Structure):
    """asif_hip_learning_data == LearningData of include/asif_learning_utils.h:8-32 (host pointers)."""
    DIMS = ("d_drift_in", "d_act_in", "d_drift_hidden", "d_act_hidden", "d_drift_hidden_2", "d_act_hidden_2",
            "d_drift_out", "d_act_out")
    PTRS = ("w_1_drift", "w_2_drift", "w_3_drift", "b_1_drift", "b_2_drift", "b_3_drift",
            "w_1_act", "w_2_act", "w_3_act", "b_1_act", "b_2_act", "b_3_act")
    _fields_ = [(n, C.c_uint32) for n in DIMS] + [(n, C.POINTER(C.c_double)) for n in PTRS]

    @classmethod
    def from_dict(cls, w):
        import numpy as np
        L = cls()
        L._keep = {}
        for n in cls.DIMS:
            setattr(L, n, int(w[n]))
        for n in cls.PTRS:
            a = np.ascontiguousarray(w[n], dtype=np.float64)
            L._keep[n] = a
            setattr(L, n, a.ctypes.data_as(C.POINTER(C.c_double)))
        return L


class Solver(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("rho", "sigma", "alpha", "eps_abs", "eps_rel", "eps_prim_inf",
                                          "eps_dual_inf", "adaptive_rho_tolerance")] + [
        (n, C.c_int32) for n in ("max_iter", "check_interval", "scaling_iters", "polish", "active_set_rounds",
                                 "refine_steps", "adaptive_rho", "lanes_per_qp", "presolve", "warm_start",
                                 "adaptive_rho_interval")]


class Dims(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("nx", "nu", "npSS", "npBS", "npBTSS", "nv", "nc", "nrelax", "npBT",
                                         "ndiag")]


class AsifHipError(RuntimeError):
    pass


_lib = None


def load():
    """dlopen libasif_hip.so (built in-tree by `make -C asif_amd/csrc` / __graft_entry__.build())."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise AsifHipError(f"{LIB_PATH} is missing: run `make -C asif_amd/csrc` (there is no CPU fallback)")
        # torch brings its own libamdhip64; it must be the one already in the process when this library's
        # HIP symbols are resolved, otherwise two HIP runtimes fight over the device (create -> ENODEVICE)
        import torch  # noqa: F401
        lib = C.CDLL(LIB_PATH)
        lib.asif_hip_error_string.restype = C.c_char_p
        lib.asif_hip_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.POINTER(Options),
                                        C.POINTER(Solver), C.c_int]
        lib.asif_hip_destroy.argtypes = [C.c_void_p]
        lib.asif_hip_get_dims.argtypes = [C.c_void_p, C.POINTER(Dims)]
        lib.asif_hip_update_options.argtypes = [C.c_void_p, C.POINTER(Options)]
        vp, i64 = C.c_void_p, C.c_int64
        lib.asif_hip_filter_batch.argtypes = [vp, i64, i64, vp, vp, vp, vp, vp, vp, vp]
        lib.asif_hip_assemble_batch.argtypes = [vp, i64, i64, vp, vp, vp, vp, vp, vp]
        lib.asif_hip_qp_solve_batch.argtypes = [C.c_int, C.POINTER(Solver), i64, i64, C.c_int32, C.c_int32, vp, vp,
                                                vp, vp, vp, vp, vp, vp, vp, vp, vp]
        lib.asif_hip_qp_solve_batch_warm.argtypes = [C.c_int, C.POINTER(Solver), i64, i64, C.c_int32, C.c_int32, vp,
                                                     vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, C.c_int32, vp]
        lib.asif_hip_filter_batch_host.argtypes = [vp, i64, vp, vp, vp, vp, vp]
        lib.asif_hip_default_realizable_options.argtypes = [C.c_int, C.POINTER(RealizableOptions)]
        lib.asif_hip_create_realizable.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(KernelData),
                                                   C.POINTER(RealizableOptions), C.POINTER(Solver), C.c_int]
        lib.asif_hip_update_realizable_options.argtypes = [vp, C.POINTER(RealizableOptions)]
        lib.asif_hip_realizable_tables.argtypes = [vp, vp, vp]
        lib.asif_hip_default_robust_data_options.argtypes = [C.c_int, C.POINTER(RobustDataOptions)]
        lib.asif_hip_create_robust_data.argtypes = [C.POINTER(C.c_void_p), C.c_int, vp, C.c_int32,
                                                    C.POINTER(RobustDataOptions), C.POINTER(Solver), C.c_int]
        lib.asif_hip_update_robust_data_options.argtypes = [vp, C.POINTER(RobustDataOptions)]
        lib.asif_hip_rollout_batch.argtypes = [vp, i64, i64, C.c_int32, C.c_double, vp, vp, vp, vp, vp, vp, vp, vp, vp]
        lib.asif_hip_set_learning.argtypes = [vp, C.POINTER(LearningData)]
        _lib = lib
    return _lib


def check(code):
    if code != 0:
        raise AsifHipError(f"asif_hip error {code}: {load().asif_hip_error_string(code).decode()}")


def default_options(model, variant):
    o = Options()
    check(load().asif_hip_default_options(model, variant, C.byref(o)))
    return o


def default_solver(**overrides):
    s = Solver()
    check(load().asif_hip_default_solver(C.byref(s)))
    for k, v in overrides.items():
        setattr(s, k, v)
    return s


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _stream():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class Filter:
    """One (model, variant, options) filter on one GPU: the batched analogue of constructing an
    ASIF* object of the reference and calling initialize(lb, ub, opts)."""

    def __init__(self, model, variant, options=None, solver=None, device=0):
        self.lib = load()
        self.model, self.variant, self.device = model, variant, device
        self.options = options if options is not None else default_options(model, variant)
        self.solver = solver if solver is not None else default_solver()
        h = C.c_void_p()
        check(self.lib.asif_hip_create(C.byref(h), model, variant, C.byref(self.options), C.byref(self.solver),
                                       device))
        self.handle = h
        d = Dims()
        check(self.lib.asif_hip_get_dims(self.handle, C.byref(d)))
        self.dims = d

    def close(self):
        if self.handle:
            self.lib.asif_hip_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def update_options(self, options):
        check(self.lib.asif_hip_update_options(self.handle, C.byref(options)))
        self.options = options
        check(self.lib.asif_hip_get_dims(self.handle, C.byref(self.dims)))

    def set_learning(self, weights):
        """ASIFimplicitRB::learning_data_: weights = dict of LearningData's fields (numpy), or None to clear."""
        if weights is None:
            check(self.lib.asif_hip_set_learning(self.handle, None))
        else:
            L = LearningData.from_dict(weights)
            check(self.lib.asif_hip_set_learning(self.handle, C.byref(L)))

    def filter(self, x, udes, uact, relax, rc, diag=None):
        """x [nx,B], udes [nu,B] -> uact [nu,B], relax [nrelax,B], rc int32[B]; all CUDA tensors, in place."""
        B = x.shape[1]
        check(self.lib.asif_hip_filter_batch(self.handle, B, x.stride(0), _ptr(x), _ptr(udes), _ptr(uact),
                                             _ptr(relax), _ptr(rc), _ptr(diag), _stream()))

    def filter_lie(self, x, udes, lfh, lgh, uact, relax, rc, diag=None):
        """class ASIF with caller-supplied Lie derivatives lfh [nc,B], lgh [nc*nu,B] (src/asif.cpp:130-165)."""
        B = x.shape[1]
        check(self.lib.asif_hip_filter_batch_lie(self.handle, B, x.stride(0), _ptr(x), _ptr(udes), _ptr(lfh), _ptr(lgh),
                                                 _ptr(uact), _ptr(relax), _ptr(rc), _ptr(diag), _stream()))

    def rollout(self, T, dt, x, udes, uact, relax, nfail, xlog=None, ulog=None, rclog=None):
        """T closed-loop steps (filter + plant Euler step) in one launch; x, uact, relax are updated in place."""
        B = x.shape[1]
        check(self.lib.asif_hip_rollout_batch(self.handle, B, x.stride(0), T, dt, _ptr(x), _ptr(udes), _ptr(uact),
                                              _ptr(relax), _ptr(nfail), _ptr(xlog), _ptr(ulog), _ptr(rclog), _stream()))

    def assemble(self, x, A, b, code, diag=None):
        B = x.shape[1]
        check(self.lib.asif_hip_assemble_batch(self.handle, B, x.stride(0), _ptr(x), _ptr(A), _ptr(b), _ptr(code),
                                               _ptr(diag), _stream()))


def math_probe(kind, a, b=None, device=0):
    """asif_hip_math_probe: numpy float64 in, (out0, out1) out."""
    import numpy as np
    a = np.ascontiguousarray(a, dtype=np.float64)
    n = a.size
    o0, o1 = np.zeros(n), np.zeros(n)
    p = lambda v: v.ctypes.data_as(C.c_void_p) if v is not None else None
    bb = np.ascontiguousarray(b, dtype=np.float64) if b is not None else None
    check(load().asif_hip_math_probe(device, kind, C.c_int64(n), p(a), p(bb), p(o0), p(o1)))
    return o0, o1


def partition(B, nblocks, r):
    """(first, count) of block r when B instances are cut into nblocks contiguous blocks (asif_hip_partition)."""
    first, count = C.c_int64(), C.c_int64()
    check(load().asif_hip_partition(C.c_int64(B), nblocks, r, C.byref(first), C.byref(count)))
    return first.value, count.value


class MultiFilter:
    """The same filter on several GPUs of one node behind one call (asif_hip_create_multi): HOST buffers in, HOST
    buffers out, contiguous blocks of the batch per device, no collective."""

    def __init__(self, model, variant, devices, options=None, solver=None):
        self.lib = load()
        self.lib.asif_hip_multi_handle.restype = C.c_void_p
        self.options = options if options is not None else default_options(model, variant)
        self.solver = solver if solver is not None else default_solver()
        devs = (C.c_int32 * len(devices))(*devices)
        h = C.c_void_p()
        check(self.lib.asif_hip_create_multi(C.byref(h), model, variant, C.byref(self.options), C.byref(self.solver),
                                             len(devices), devs))
        self.handle = h
        d = Dims()
        check(self.lib.asif_hip_get_dims(C.c_void_p(self.lib.asif_hip_multi_handle(self.handle, 0)), C.byref(d)))
        self.dims = d

    def close(self):
        if self.handle:
            self.lib.asif_hip_multi_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def filter_host(self, x, udes, uact, relax, rc):
        """numpy arrays, C-contiguous: x [nx,B], udes [nu,B] -> uact [nu,B], relax [nrelax,B], rc int32[B], in place."""
        B = x.shape[1]
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        check(self.lib.asif_hip_filter_batch_host_multi(self.handle, C.c_int64(B), p(x), p(udes), p(uact), p(relax),
                                                        p(rc)))


def default_realizable_options(model=MODEL_DOUBLE_INTEGRATOR_SAMPLED, **overrides):
    o = RealizableOptions()
    check(load().asif_hip_default_realizable_options(model, C.byref(o)))
    for k, v in overrides.items():
        if k in ("lb", "ub", "uncertaintyBounds"):
            for i, vi in enumerate(v):
                getattr(o, k)[i] = vi
        else:
            setattr(o, k, v)
    return o


class RealizableFilter(Filter):
    """ASIFrealizable on a polytopic kernel: `kernel` is a dict of numpy arrays vertices [nV,2] f64,
    facetVertices [nF,2] i32, facetNormals [nF,2] f64, facetActive [nF,nA] i32 plus the two limits
    maxCriticalFacets, maxActiveConstraints (ASIFrealizable::kernel_t)."""

    def __init__(self, kernel, options=None, solver=None, device=0, model=MODEL_DOUBLE_INTEGRATOR_SAMPLED):
        import numpy as np
        self.lib = load()
        self.model, self.variant, self.device = model, REALIZABLE, device
        self.options = options if options is not None else default_realizable_options(model)
        self.solver = solver if solver is not None else default_solver()
        v = np.ascontiguousarray(kernel["vertices"], dtype=np.float64)
        fv = np.ascontiguousarray(kernel["facetVertices"], dtype=np.int32)
        fn = np.ascontiguousarray(kernel["facetNormals"], dtype=np.float64)
        fa = np.ascontiguousarray(kernel["facetActive"], dtype=np.int32)
        k = KernelData(v.shape[1], v.shape[0], fv.shape[0], int(kernel["maxCriticalFacets"]),
                       int(kernel["maxActiveConstraints"]), v.ctypes.data_as(C.POINTER(C.c_double)),
                       fv.ctypes.data_as(C.POINTER(C.c_int32)), fn.ctypes.data_as(C.POINTER(C.c_double)),
                       fa.ctypes.data_as(C.POINTER(C.c_int32)))
        h = C.c_void_p()
        check(self.lib.asif_hip_create_realizable(C.byref(h), model, C.byref(k), C.byref(self.options),
                                                  C.byref(self.solver), device))
        self.handle = h
        self.nFacets, self.nActive, self.maxCrit = fv.shape[0], fa.shape[1], int(kernel["maxCriticalFacets"])
        d = Dims()
        check(self.lib.asif_hip_get_dims(self.handle, C.byref(d)))
        self.dims = d

    def update_options(self, options):
        check(self.lib.asif_hip_update_realizable_options(self.handle, C.byref(options)))
        self.options = options
        check(self.lib.asif_hip_get_dims(self.handle, C.byref(self.dims)))

    def tables(self):
        """(table [nF,nA,4], bbox [nF,2,2]) as built on the device."""
        import numpy as np
        t = np.zeros((self.nFacets, self.nActive, 4))
        bb = np.zeros((self.nFacets, 2, 2))
        check(self.lib.asif_hip_realizable_tables(self.handle, C.c_void_p(t.ctypes.data), C.c_void_p(bb.ctypes.data)))
        return t, bb


def default_robust_data_options(model=MODEL_DOUBLE_INTEGRATOR_ROBUST, **overrides):
    o = RobustDataOptions()
    check(load().asif_hip_default_robust_data_options(model, C.byref(o)))
    for k, v in overrides.items():
        if k in ("lb", "ub"):
            getattr(o, k)[0] = v[0]
        else:
            setattr(o, k, v)
    return o


class RobustDataFilter(Filter):
    """ASIFrobust on a half-plane data set [N,2] (numpy f64), npSSmax rows kept per call:
    the reference's examples/DoubleIntegrator_Robust.cpp."""

    def __init__(self, halfplanes, options=None, solver=None, device=0, model=MODEL_DOUBLE_INTEGRATOR_ROBUST):
        import numpy as np
        self.lib = load()
        self.model, self.variant, self.device = model, ROBUST, device
        self.options = options if options is not None else default_robust_data_options(model)
        self.solver = solver if solver is not None else default_solver()
        hp = np.ascontiguousarray(halfplanes, dtype=np.float64)
        h = C.c_void_p()
        check(self.lib.asif_hip_create_robust_data(C.byref(h), model, C.c_void_p(hp.ctypes.data), hp.shape[0],
                                                   C.byref(self.options), C.byref(self.solver), device))
        self.handle = h
        d = Dims()
        check(self.lib.asif_hip_get_dims(self.handle, C.byref(d)))
        self.dims = d

    def update_options(self, options):
        check(self.lib.asif_hip_update_robust_data_options(self.handle, C.byref(options)))
        self.options = options
        check(self.lib.asif_hip_get_dims(self.handle, C.byref(self.dims)))


def qp_solve_batch(Hd, c, A, b, lb, ub, sol, status, iters=None, be=None, solver=None, device=0):
    """Pre-assembled QPs, SoA CUDA tensors: Hd,c,lb,ub [nv,B]; A [nc*nv,B]; b [nc,B]."""
    nv, B = c.shape
    nc = b.shape[0]
    s = solver if solver is not None else default_solver()
    bep = None
    if be is not None:
        arr = (C.c_uint8 * nc)(*[int(v) for v in be])
        bep = C.cast(arr, C.c_void_p)
    check(load().asif_hip_qp_solve_batch(device, C.byref(s), B, c.stride(0), nv, nc, _ptr(Hd), _ptr(c), _ptr(A),
                                         _ptr(b), _ptr(lb), _ptr(ub), bep, _ptr(sol), _ptr(status), _ptr(iters),
                                         _stream()))


def qp_solve_batch_warm(Hd, H, c, A, b, lb, ub, sol, status, warm_x, warm_y, warm_in, iters=None, be=None,
                        solver=None, device=0):
    """asif_hip_qp_solve_batch_warm: Hd [nv,B] or H [nv*nv,B] (the other None); warm_x [nv,B], warm_y [nc+nv,B]
    are written by every call and read first when warm_in is true."""
    nv, B = c.shape
    nc = b.shape[0]
    s = solver if solver is not None else default_solver()
    bep = None
    if be is not None:
        arr = (C.c_uint8 * nc)(*[int(v) for v in be])
        bep = C.cast(arr, C.c_void_p)
    check(load().asif_hip_qp_solve_batch_warm(device, C.byref(s), B, c.stride(0), nv, nc, _ptr(Hd), _ptr(H), _ptr(c),
                                              _ptr(A), _ptr(b), _ptr(lb), _ptr(ub), bep, _ptr(sol), _ptr(status),
                                              _ptr(iters), _ptr(warm_x), _ptr(warm_y), 1 if warm_in else 0,
                                              _stream()))


def qp_solve_batch_dense(H, c, A, b, lb, ub, sol, status, iters=None, be=None, solver=None, device=0):
    """Same with a full cost matrix H [nv*nv,B] (column-major inside an instance, upper triangle read)."""
    nv, B = c.shape
    nc = b.shape[0]
    s = solver if solver is not None else default_solver()
    bep = None
    if be is not None:
        arr = (C.c_uint8 * nc)(*[int(v) for v in be])
        bep = C.cast(arr, C.c_void_p)
    check(load().asif_hip_qp_solve_batch_dense(device, C.byref(s), B, c.stride(0), nv, nc, _ptr(H), _ptr(c), _ptr(A),
                                               _ptr(b), _ptr(lb), _ptr(ub), bep, _ptr(sol), _ptr(status),
                                               _ptr(iters), _stream()))
