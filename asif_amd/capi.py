"""ctypes binding of libasif_hip.so, the C ABI declared in include/asif_hip.h.

This is plumbing for tests and bench.py: device memory comes from torch tensors (FP64, contiguous,
SoA [component][B]) whose data_ptr() is handed to the library; the kernels run on torch's current
stream.  There is no fallback: if the shared library is missing or no gfx950 device is usable,
every call raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libasif_hip.so")

MODEL_DOUBLE_INTEGRATOR, MODEL_INVERTED_PENDULUM, MODEL_SEGWAY, MODEL_INVERTED_PENDULUM_ROBUST = 0, 1, 2, 3
EXPLICIT, IMPLICIT, IMPLICIT_TB, ROBUST = 0, 1, 2, 3

# BASELINE.json configs[k] -> (model, variant, default batch)
CONFIGS = {
    2: (MODEL_DOUBLE_INTEGRATOR, EXPLICIT, 65536),
    3: (MODEL_INVERTED_PENDULUM, IMPLICIT, 16384),
    4: (MODEL_SEGWAY, IMPLICIT_TB, 32768),
    5: (MODEL_INVERTED_PENDULUM_ROBUST, ROBUST, 8192),
}

EXPORTS = [
    "asif_hip_version", "asif_hip_error_string", "asif_hip_device_count", "asif_hip_default_options",
    "asif_hip_default_solver", "asif_hip_create", "asif_hip_destroy", "asif_hip_get_dims",
    "asif_hip_update_options", "asif_hip_filter_batch", "asif_hip_assemble_batch", "asif_hip_qp_solve_batch",
    "asif_hip_filter_batch_host",
]


class Options(C.Structure):
    _fields_ = [(n, C.c_double) for n in (
        "relaxCost", "relaxLb", "relaxReachLb", "relaxTTS", "relaxMinOrtho", "backTrajHorizon",
        "backTrajExtend", "backTrajDt", "backTrajMinOrtho", "satSharpness", "inf")] + [
        ("lb", C.c_double * 1), ("ub", C.c_double * 1), ("pMin", C.c_double), ("pMax", C.c_double),
        ("nHalfPlanes", C.c_int32), ("halfPlanes", C.c_double * 16)]


class Solver(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("rho", "sigma", "alpha", "eps_abs", "eps_rel", "eps_prim_inf",
                                          "eps_dual_inf", "adaptive_rho_tolerance")] + [
        (n, C.c_int32) for n in ("max_iter", "check_interval", "scaling_iters", "polish", "active_set_rounds",
                                 "refine_steps", "adaptive_rho", "lanes_per_qp")]


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
        lib.asif_hip_filter_batch_host.argtypes = [vp, i64, vp, vp, vp, vp, vp]
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

    def filter(self, x, udes, uact, relax, rc, diag=None):
        """x [nx,B], udes [nu,B] -> uact [nu,B], relax [nrelax,B], rc int32[B]; all CUDA tensors, in place."""
        B = x.shape[1]
        check(self.lib.asif_hip_filter_batch(self.handle, B, x.stride(0), _ptr(x), _ptr(udes), _ptr(uact),
                                             _ptr(relax), _ptr(rc), _ptr(diag), _stream()))

    def assemble(self, x, A, b, code, diag=None):
        B = x.shape[1]
        check(self.lib.asif_hip_assemble_batch(self.handle, B, x.stride(0), _ptr(x), _ptr(A), _ptr(b), _ptr(code),
                                               _ptr(diag), _stream()))


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
