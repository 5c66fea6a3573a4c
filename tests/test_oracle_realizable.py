"""CPU tests of the realizable-filter oracle (oracle/or_realizable.c) against the fixtures made from the
reference's libaffa and kernel data (tests/golden/affa_rz_facet_lie.json; asif_amd/data/realizable_kernels.json)."""
import json
import os

import numpy as np
import pytest

import oracle_lib as O

GOLD = os.path.join(os.path.dirname(__file__), "golden")
KERNELS = ["100Hz", "100Hz_50pt", "10Hz", "10Hz_50pt"]
# nv = nu + npSS*2*(nu+1) + 1, nc = npSS*(nu+2) + npSSmax  (src/asif_realizable.cpp:19-22), npSSmax = 2
DIMS = {"100Hz": (38, 29, 9), "100Hz_50pt": (38, 29, 9), "10Hz": (86, 65, 21), "10Hz_50pt": (62, 47, 15)}


@pytest.fixture(scope="module")
def gold():
    with open(os.path.join(GOLD, "affa_rz_facet_lie.json")) as f:
        return json.load(f)


@pytest.mark.parametrize("name", KERNELS)
def test_dims_and_facet_table_match_reference_libaffa(name, gold):
    z = O.Realizable(O.load_kernel(name))
    assert (z.nv, z.nc, z.npSS) == DIMS[name]
    table, bbox = z.table()
    ref = np.array(gold["tables"][name])
    assert table.shape == ref.shape
    assert np.array_equal(table, ref)  # bit-exact: same operations in the same order as libaffa
    k = O.load_kernel(name)
    v = k["vertices"][k["facetVertices"]]  # [nF, 2 vertices, 2 comps]
    assert np.array_equal(bbox[:, :, 0], v.min(axis=1)) and np.array_equal(bbox[:, :, 1], v.max(axis=1))


def test_barrier_rows_use_reference_point_dynamics(gold):
    k = O.load_kernel("100Hz")
    z = O.Realizable(k)
    pts = gold["points"]
    x = np.array([p["x"] for p in pts])
    A, b, code, info = z.assemble(x)
    A = A.reshape(len(pts), z.nv, z.nc).transpose(0, 2, 1)  # [B, row, col]
    n = k["facetNormals"]
    for i, p in enumerate(pts):
        f, g = p["f"], p["g"]
        h = 1.0 - n[:, 0] * x[i, 0] - n[:, 1] * x[i, 1]
        order = np.argsort(h, kind="stable")[:2]
        assert list(info[i, 1 + z.maxCrit:]) == list(order)
        for r, fi in enumerate(order):
            Lfh = 0.0 + (-n[fi, 0]) * f[0] + (-n[fi, 1]) * f[1]
            Lgh = 0.0 + (-n[fi, 0]) * g[0] + (-n[fi, 1]) * g[1]
            row = 3 * z.npSS + r
            assert A[i, row, 0] == Lgh
            assert A[i, row, z.nv - 1] == 1.0
            hh = 1.0
            for j in range(2):
                hh -= n[fi, j] * x[i, j]
            assert b[i, row] == -Lfh - 10.0 * (hh - 0.0)


def test_fixed_structure_and_critical_rows():
    k = O.load_kernel("100Hz")
    z = O.Realizable(k)
    x, u = O.make_batch_realizable(k, 1024)
    A, b, code, info = z.assemble(x)
    A = A.reshape(-1, z.nv, z.nc).transpose(0, 2, 1)
    table, _ = z.table()
    Hd, c, lb, ub, be = z.qp_static(u[0])
    assert list(be[:3 * z.npSS]) == [0, 1, 1] * z.npSS and not be[3 * z.npSS:].any()
    assert Hd[0] == 1.0 and Hd[-1] == 100.0 and not Hd[1:-1].any() and c[0] == -2.0 * u[0, 0]
    assert lb[0] == -20.0 and ub[0] == 20.0 and not lb[1:].any() and (ub[1:] == 1e20).all()
    assert (info[:, 0] > 0).sum() > 100
    for i in range(x.shape[0]):
        nCrit = info[i, 0]
        s = 0
        for cidx in range(nCrit):
            for j in range(z.nA):
                t = table[info[i, 1 + cidx], j]
                col = 1 + 4 * s
                assert A[i, 3 * s, col] == t[0] and A[i, 3 * s, col + 2] == -t[1]
                assert A[i, 3 * s, col + 1] == t[2] and A[i, 3 * s, col + 3] == -t[3]
                s += 1
        for s in range(z.npSS):
            col = 1 + 4 * s
            assert A[i, 3 * s + 1, 0] == -1.0 and A[i, 3 * s + 1, col] == 1.0 and A[i, 3 * s + 1, col + 2] == -1.0
            assert A[i, 3 * s + 2, col + 1] == 1.0 and A[i, 3 * s + 2, col + 3] == -1.0 and b[i, 3 * s + 2] == 1.0
            if s >= nCrit * z.nA:
                assert not A[i, 3 * s].any()
        # code -1 <=> no critical facet while some h < 0 (:602-605)
        h = 1.0 - k["facetNormals"] @ x[i]
        assert code[i] == (-1 if (nCrit == 0 and (h < 0).any()) else 1)


def test_critical_facets_agree_with_brute_force_sampling():
    k = O.load_kernel("100Hz")
    z = O.Realizable(k)
    x, _ = O.make_batch_realizable(k, 512)
    _, _, _, info = z.assemble(x)
    V, FV = k["vertices"], k["facetVertices"]
    unc = np.array([0.031, 0.028])
    t = np.linspace(0.0, 1.0, 2001)[:, None]
    for i in range(x.shape[0]):
        found = []
        for fi in range(FV.shape[0]):
            p = t * V[FV[fi, 0]] + (1 - t) * V[FV[fi, 1]]
            if (np.abs(p - x[i]) <= unc).all(axis=1).any():
                found.append(fi)
        got = [f for f in info[i, 1:1 + z.maxCrit] if f >= 0]
        # sampling can only miss grazing contacts, never invent one; the filter keeps the first maxCrit in facet order
        assert set(found[:z.maxCrit]) <= set(got) or len(got) == z.maxCrit
        assert set(f for f in got if f in found) == set(got) or len(set(got) - set(found)) <= 1


@pytest.mark.parametrize("name", ["100Hz", "10Hz_50pt"])
def test_multiplier_elimination_matches_full_qp(name):
    """u*, delta* of the exact reduced problem vs OSQP-style ADMM on the full nv x nc problem the reference assembles."""
    k = O.load_kernel(name)
    z = O.Realizable(k)
    x, u = O.make_batch_realizable(k, 1024)
    ua, relax, rc = z.filter(x, u)
    _, _, code, info = z.assemble(x)
    assert set(np.unique(rc)) <= {1, -1, -2} and ((rc == -2) == (code == -1)).all()
    idx = np.concatenate([np.where((rc == 1) & (info[:, 0] > 0))[0][:40], np.where((rc == 1) & (info[:, 0] == 0))[0][:10]])
    s = O.admm_settings(eps_abs=1e-9, eps_rel=1e-9, max_iter=400000, sigma=1e-3)
    ua2, relax2, rc2 = z.filter(x[idx], u[idx], solver=O.SOLVER_ADMM, settings=s)
    ok = rc2 == 1
    assert ok.mean() > 0.9
    assert np.abs(ua2[ok] - ua[idx][ok]).max() < 1e-5
    assert np.abs(relax2[ok, 1] - relax[idx][ok, 1]).max() < 1e-5
    # the non-unique multiplier the reference hands out as relax[0]: the oracle reports its smallest feasible value
    assert (relax2[ok, 0] >= relax[idx][ok, 0] - 1e-5).all()


def test_filter_properties():
    k = O.load_kernel("100Hz")
    z = O.Realizable(k)
    x, u = O.make_batch_realizable(k, 4096)
    ua, relax, rc = z.filter(x, u)
    ok = rc == 1
    assert ok.sum() > 2000 and (rc == -2).sum() > 100
    assert (ua[ok] >= -20.0).all() and (ua[ok] <= 20.0).all() and (relax[ok, 1] >= 0).all()
    assert np.isnan(ua[~ok]).all()  # outputs untouched on failure (:324-326,350-351)
    # far inside the kernel nothing binds: u = uDes
    xin = np.zeros((8, 2))
    ud = np.linspace(-19, 19, 8)[:, None]
    ua, relax, rc = z.filter(xin, ud)
    assert (rc == 1).all() and np.abs(ua - ud).max() < 1e-12 and np.abs(relax[:, 1]).max() < 1e-12


def test_barrier_rows_are_lie_derivatives_of_the_smallest_margins(oracle):
    """The barrier rows of ASIFrealizable (src/asif_realizable.cpp:530-600) checked without the oracle's own
    arithmetic: with npSSmax = 4 rows per state, (f, g) are over-determined by  Lgh_i = -n_i . g,
    -b_i - relaxDes (h_i - relaxOffset) = -n_i . f  -- they must be consistent across the four rows, g must be
    (0, K/m) and f = (x1, -F x1 / m) inside the hull of the interval parameters (the reference takes midpoints of
    affine forms), and the rows must belong to the four facets with the smallest margins h_i = 1 - n_i . x."""
    k = oracle.load_kernel("100Hz")
    z = oracle.Realizable(k, npSSmax=4)
    d = z.desc
    n = k["facetNormals"]
    B = 200
    x, u = oracle.make_batch_realizable(k, B)
    A, b, code, info = z.assemble(x)
    base = z.npSS * 3
    for i in range(B):
        Am = A[i].reshape(z.nv, z.nc).T
        sel = info[i, 1 + z.maxCrit:1 + z.maxCrit + 4]
        h = 1.0 - n @ x[i]
        assert set(sel.tolist()) == set(np.argsort(h, kind="stable")[:4].tolist())
        N = -n[sel]
        lgh = Am[base:base + 4, 0]
        lfh = -(b[i, base:base + 4] + d.relaxDes * (h[sel] - d.relaxOffset))
        # g = (0, g1), f = (x1, f1): one scalar each, fixed by the row with the largest n_1 and checked on the others
        j = int(np.argmax(np.abs(N[:, 1])))
        g1 = lgh[j] / N[j, 1]
        f1 = (lfh[j] - N[j, 0] * x[i, 1]) / N[j, 1]
        assert np.abs(N[:, 1] * g1 - lgh).max() <= 1e-13
        assert np.abs(N[:, 0] * x[i, 1] + N[:, 1] * f1 - lfh).max() <= 1e-11 * (1 + np.abs(lfh).max())
        assert d.Klo / d.mMax - 1e-12 <= g1 <= d.Khi / d.mMin + 1e-12
        lo, hi = sorted((-d.Flo * x[i, 1] / d.mMin, -d.Fhi * x[i, 1] / d.mMax))
        assert lo - 1e-9 <= f1 <= hi + 1e-9
        assert np.all(Am[base:base + 4, z.nv - 1] == 1.0)
