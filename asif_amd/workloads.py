"""Seeded synthetic batches of the benchmark configs.

RNG: one splitmix64 step of (seed * 2**32 + k), k = i*16 + j for instance i, coordinate j;
r = (z >> 11) * 2**-53.  Seeds: C2 1, C3 2, C4 3, C5 4.  numpy only; returns SoA arrays
x [nx, B], udes [nu, B] ready to upload.
"""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def uniform(seed, i, j):
    """r in [0,1) for instances i (uint64 array) and coordinate j."""
    with np.errstate(over="ignore"):
        z = (np.uint64(seed) << np.uint64(32)) + (i.astype(np.uint64) * np.uint64(16) + np.uint64(j))
        z = z + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def make_batch(cfg, B, first=0):
    i = np.arange(first, first + B, dtype=np.uint64)
    if cfg == 2:    # DoubleIntegrator explicit: x in [-1.2,1.2]^2 (safe box is +-1), uDes in [-1.5,1.5]
        x = np.stack([-1.2 + 2.4 * uniform(1, i, 0), -1.2 + 2.4 * uniform(1, i, 1)])
        u = (-1.5 + 3.0 * uniform(1, i, 2))[None, :]
    elif cfg == 3:  # InvertedPendulum implicit
        x = np.stack([-1.5 + 3.0 * uniform(2, i, 0), -1.5 + 3.0 * uniform(2, i, 1)])
        u = (-1.5 + 3.0 * uniform(2, i, 2))[None, :]
    elif cfg == 4:  # segway TB: x_j = 0.05 * xBound_j * (2r - 1)
        xb = [3.0, 3.0, np.pi / 6, np.pi]
        x = np.stack([0.05 * xb[j] * (2.0 * uniform(3, i, j) - 1.0) for j in range(4)])
        u = (-5.0 + 10.0 * uniform(3, i, 4))[None, :]
    elif cfg == 5:  # robust pendulum
        x = np.stack([-3.0 + 6.0 * uniform(4, i, 0), -3.0 + 6.0 * uniform(4, i, 1)])
        u = (-1.5 + 3.0 * uniform(4, i, 2))[None, :]
    elif cfg == 8:  # pendulum TB (examples/InvertedPendulum_ImplicitTB.cpp): around and inside the backup set
        x = np.stack([-1.4 + 3.0 * uniform(8, i, 0), -1.4 + 2.8 * uniform(8, i, 1)])
        u = (-1.5 + 3.0 * uniform(8, i, 2))[None, :]
    elif cfg == 9:  # double integrator implicit (examples/DoubleIntegrator_implicit.cpp); |x| <= 0.4, see or_make_batch
        x = np.stack([-0.4 + 0.8 * uniform(9, i, 0), -0.4 + 0.8 * uniform(9, i, 1)])
        u = (-1.5 + 3.0 * uniform(9, i, 2))[None, :]
    elif cfg == 10:  # pendulum under ASIFimplicitRB (no reference example; C3's distribution, own seed)
        x = np.stack([-1.5 + 3.0 * uniform(10, i, 0), -1.5 + 3.0 * uniform(10, i, 1)])
        u = (-1.5 + 3.0 * uniform(10, i, 2))[None, :]
    elif cfg == 11:  # synthetic two-input model under class ASIF (not a reference example): nx = 2, nu = 2
        x = np.stack([-1.6 + 3.2 * uniform(11, i, 0), -1.6 + 3.2 * uniform(11, i, 1)])
        u = np.stack([-1.5 + 3.0 * uniform(11, i, 2), -1.5 + 3.0 * uniform(11, i, 3)])
    elif cfg == 12:  # double integrator TB (examples/DoubleIntegrator_implicit_tb.cpp): see or_make_batch
        x = np.stack([-0.04 + 0.08 * uniform(12, i, 0), -0.04 + 0.08 * uniform(12, i, 1)])
        u = (-1.5 + 3.0 * uniform(12, i, 2))[None, :]
    else:
        raise ValueError(f"unknown config {cfg}")
    return np.ascontiguousarray(x), np.ascontiguousarray(u)


RB_X_UNC = (0.02, 0.01)  # state uncertainty radius of config 10 (Options::x_unc, include/asif_implicit_robust.h:25)


def make_learning(nx=2, nu=1, hidden=(16, 16), amp=0.2, bias=0.05, seed=11):
    """Seeded weights for LearningData (include/asif_learning_utils.h:8-32): two ReLU layers and a linear
    one for the drift residual (1 output) and for the actuation residual (nu outputs), inputs [x; Dh] (2 nx).
    The reference ships no weights; these are synthetic, of that architecture.  Matrices are column-major
    [rows x cols] flattened, as matrixVectorMultiply reads them."""
    h1, h2 = hidden
    w = dict(d_drift_in=2 * nx, d_act_in=2 * nx, d_drift_hidden=h1, d_act_hidden=h1, d_drift_hidden_2=h2,
             d_act_hidden_2=h2, d_drift_out=1, d_act_out=nu)
    shapes = [("w_1_drift", h1 * 2 * nx, amp), ("b_1_drift", h1, bias), ("w_2_drift", h2 * h1, amp),
              ("b_2_drift", h2, bias), ("w_3_drift", 1 * h2, amp), ("b_3_drift", 1, bias),
              ("w_1_act", h1 * 2 * nx, amp), ("b_1_act", h1, bias), ("w_2_act", h2 * h1, amp),
              ("b_2_act", h2, bias), ("w_3_act", nu * h2, amp), ("b_3_act", nu, bias)]
    k = 0
    for name, n, a in shapes:
        idx = np.arange(k, k + n, dtype=np.uint64)
        w[name] = np.ascontiguousarray(a * (2.0 * uniform(seed, idx, 0) - 1.0))
        k += n
    return w


def load_kernel(name="100Hz", path=None):
    """Polytope data of the reference's include/RealizableKernelData_<name>.h (numbers only, the package's own
    asif_amd/data/realizable_kernels.json) as the dict asif_amd.capi.RealizableFilter takes."""
    import json
    import os
    if path is None:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "realizable_kernels.json")
    with open(path) as f:
        k = json.load(f)["kernels"][name]
    return dict(vertices=np.array(k["vertices"], dtype=np.float64),
                facetVertices=np.array(k["facetVertices"], dtype=np.int32),
                facetNormals=np.array(k["facetNormals"], dtype=np.float64),
                facetActive=np.array(k["facetActive"], dtype=np.int32),
                maxCriticalFacets=int(k["maxCriticalFacets"]), maxActiveConstraints=int(k["maxActiveConstraints"]))


def load_halfplanes(name="70-135kg", path=None):
    """SafetySetData of the reference's include/KernelData_<name>.h ([N,2] float64), kept as numbers in the
    package's own asif_amd/data/robust_halfplanes.json."""
    import json
    import os
    if path is None:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "robust_halfplanes.json")
    with open(path) as f:
        return np.array(json.load(f)["sets"][name], dtype=np.float64)


def make_batch_robust_data(halfplanes, B, first=0, seed=7):
    """Config 7 (robust filter on the shipped half-planes): states uniform over 1.1 x the extent of the polygon
    along each axis (so part of the batch starts outside), uDes uniform in [-20, 20].  SoA x [2,B], udes [1,B]."""
    i = np.arange(first, first + B, dtype=np.uint64)
    a = np.abs(halfplanes)
    ext = 1.1 * np.array([1.0 / a[:, 0].max(), 1.0 / a[:, 1].max()])
    x = np.stack([ext[0] * (2.0 * uniform(seed, i, 0) - 1.0), ext[1] * (2.0 * uniform(seed, i, 1) - 1.0)])
    u = (-20.0 + 40.0 * uniform(seed, i, 2))[None, :]
    return np.ascontiguousarray(x), np.ascontiguousarray(u)


def make_batch_realizable(kernel, B, first=0, seed=6):
    """Config 6 (realizable filter, sampled double integrator): half the states uniform over 1.05 x the
    kernel's bounding box (some start outside -> rc -2), half within +-2 % radially of a random point of a random
    facet (critical-facet rows active); uDes uniform in [-20, 20].  SoA x [2,B], udes [1,B]."""
    i = np.arange(first, first + B, dtype=np.uint64)
    V, FV = kernel["vertices"], kernel["facetVertices"]
    vmax = np.abs(V).max(axis=0)
    nF = FV.shape[0]
    r0, r1, r3 = uniform(seed, i, 0), uniform(seed, i, 1), uniform(seed, i, 3)
    fi = np.minimum((uniform(seed, i, 4) * nF).astype(np.int64), nF - 1)
    t = uniform(seed, i, 5)
    s = 1.0 + 0.02 * (2.0 * r0 - 1.0)
    near = s[:, None] * (t[:, None] * V[FV[fi, 0]] + (1.0 - t)[:, None] * V[FV[fi, 1]])
    wide = np.stack([1.05 * vmax[0] * (2.0 * r0 - 1.0), 1.05 * vmax[1] * (2.0 * r1 - 1.0)], axis=1)
    x = np.where((r3 < 0.5)[:, None], near, wide).T
    u = (-20.0 + 40.0 * uniform(seed, i, 2))[None, :]
    return np.ascontiguousarray(x), np.ascontiguousarray(u)
