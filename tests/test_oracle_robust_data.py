"""CPU tests of the robust-filter oracle on the data the reference ships (oracle/or_robust_data.c:
examples/DoubleIntegrator_Robust.cpp + include/KernelData_70-135kg.h, npSSmax = 5 of 100 half-planes)."""
import json
import os

import numpy as np

import oracle_lib as O

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_rows_match_reference_libaffa_golden():
    with open(os.path.join(GOLD, "affa_di_robust_lie.json")) as f:
        g = json.load(f)
    hp = O.load_halfplanes(g["set"])
    z = O.RobustData(hp)
    assert (z.nv, z.nc, z.npSSmax, z.N) == (22, 15, 5, 100)  # src/asif_robust.cpp:20-22
    x = np.array([c["x"] for c in g["cases"]])
    A, b, code, sel = z.assemble(x)
    assert (code == 1).all()
    A = A.reshape(len(x), z.nv, z.nc).transpose(0, 2, 1)  # [B, row, col]
    for i, c in enumerate(g["cases"]):
        assert list(sel[i]) == c["sel"]
        for s in range(5):
            col, row = 2 + 4 * s, 3 * s
            lie = c["lie"][s]
            a0, a1 = hp[c["sel"][s]]
            assert A[i, row, 1] == 1.0 - a0 * x[i, 0] - a1 * x[i, 1]
            # bit-identical to libaffa: lo/hi of Lgh, lo/hi of Lfh
            assert A[i, row, col] == lie[0] and A[i, row, col + 2] == -lie[1]
            assert A[i, row, col + 1] == lie[2] and A[i, row, col + 3] == -lie[3]
            # fixed structure (:103-133)
            assert A[i, row + 1, 0] == -1.0 and A[i, row + 1, col] == 1.0 and A[i, row + 1, col + 2] == -1.0
            assert A[i, row + 2, col + 1] == 1.0 and A[i, row + 2, col + 3] == -1.0 and b[i, row + 2] == 1.0


def test_static_part_and_elimination_against_full_qp():
    hp = O.load_halfplanes()
    z = O.RobustData(hp)
    x, u = O.make_batch_robust_data(hp, 512)
    Hd, c, lb, ub, be = z.qp_static(u[0])
    assert list(be) == [0, 1, 1] * 5
    assert Hd[0] == 1.0 and Hd[1] == 50.0 and not Hd[2:].any()
    assert c[0] == -2.0 * u[0, 0] and c[1] == -2.0 * 50.0 * 5.0 and not c[2:].any()
    assert (lb[0], ub[0], lb[1], ub[1]) == (-20.0, 20.0, 5.0, 1e20) and not lb[2:].any() and (ub[2:] == 1e20).all()
    ua, rl, rc = z.filter(x, u)
    assert set(np.unique(rc)) <= {1, -1} and (rc == 1).sum() > 300 and (rc == -1).sum() > 50
    idx = np.where(rc == 1)[0][:48]
    s = O.admm_settings(eps_abs=1e-9, eps_rel=1e-9, max_iter=400000, sigma=1e-3)
    ua2, rl2, rc2 = z.filter(x[idx], u[idx], solver=O.SOLVER_ADMM, settings=s)
    ok = rc2 == 1
    assert ok.mean() > 0.9
    assert np.abs(ua2[ok] - ua[idx][ok]).max() < 1e-5
    assert np.abs(rl2[ok] - rl[idx][ok]).max() < 1e-5


def test_filter_properties():
    hp = O.load_halfplanes()
    z = O.RobustData(hp)
    x, u = O.make_batch_robust_data(hp, 2048)
    ua, rl, rc = z.filter(x, u)
    ok = rc == 1
    assert (ua[ok] >= -20.0).all() and (ua[ok] <= 20.0).all() and (rl[ok] >= 5.0 - 1e-12).all()
    assert np.isnan(ua[~ok]).all()
    # deep inside the set nothing binds: u = uDes and the relaxation sits at its lower bound
    ua, rl, rc = z.filter(np.zeros((4, 2)), np.array([[-15.0], [-1.0], [3.0], [19.0]]))
    assert (rc == 1).all() and np.abs(ua[:, 0] - [-15.0, -1.0, 3.0, 19.0]).max() < 1e-12 and np.abs(rl - 5.0).max() < 1e-12
    # both kernels of the reference load
    z2 = O.RobustData(O.load_halfplanes("70-75kg"), mMax=75.0)
    ua, rl, rc = z2.filter(x[:256], u[:256])
    assert (rc == 1).sum() > 150
