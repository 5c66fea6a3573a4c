"""Realizable filter (ASIFrealizable) on the GPU, through the C ABI.

 - the device-built facet tables must be bit-identical to the reference's libaffa (golden fixture);
 - the assembled nc x nv rows must equal the oracle's (critical facets, barrier facets, every entry);
 - u*, delta* <= 1e-6 from the exact optimum of the QP the reference assembles, rc identical (1 / -1 / -2).
"""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
KERNELS = ["100Hz", "100Hz_50pt", "10Hz", "10Hz_50pt"]


def _run(hip, kernel, x, udes, options=None, solver=None, assemble=False, uact_init=7.0, relax_init=-7.0):
    flt = hip.RealizableFilter(kernel, options=options, solver=solver)
    d = flt.dims
    B = x.shape[1]
    dev = torch.device("cuda:0")
    tx, tu = torch.from_numpy(x).to(dev), torch.from_numpy(udes).to(dev)
    diag = torch.zeros((d.ndiag, B), dtype=torch.float64, device=dev)
    if assemble:
        A = torch.zeros((d.nc * d.nv, B), dtype=torch.float64, device=dev)
        b = torch.zeros((d.nc, B), dtype=torch.float64, device=dev)
        code = torch.zeros(B, dtype=torch.int32, device=dev)
        flt.assemble(tx, A, b, code, diag)
        torch.cuda.synchronize()
        out = dict(A=A.cpu().numpy(), b=b.cpu().numpy(), code=code.cpu().numpy())
    else:
        uact = torch.full((d.nu, B), uact_init, dtype=torch.float64, device=dev)
        relax = torch.full((d.nrelax, B), relax_init, dtype=torch.float64, device=dev)
        rc = torch.zeros(B, dtype=torch.int32, device=dev)
        flt.filter(tx, tu, uact, relax, rc, diag)
        torch.cuda.synchronize()
        out = dict(uact=uact.cpu().numpy(), relax=relax.cpu().numpy(), rc=rc.cpu().numpy())
    out.update(diag=diag.cpu().numpy(), dims=d, tables=flt.tables())
    flt.close()
    return out


@pytest.mark.parametrize("name", KERNELS)
def test_device_tables_bit_identical_to_reference_libaffa(hip, name):
    from asif_amd import workloads
    k = workloads.load_kernel(name)
    flt = hip.RealizableFilter(k)
    table, bbox = flt.tables()
    with open(os.path.join(GOLD, "affa_rz_facet_lie.json")) as f:
        ref = np.array(json.load(f)["tables"][name])
    assert np.array_equal(table, ref)
    v = k["vertices"][k["facetVertices"]]
    assert np.array_equal(bbox[:, :, 0], v.min(axis=1)) and np.array_equal(bbox[:, :, 1], v.max(axis=1))
    npSS = k["maxCriticalFacets"] * k["maxActiveConstraints"]
    assert (flt.dims.nv, flt.dims.nc, flt.dims.nrelax) == (1 + 4 * npSS + 1, 3 * npSS + 2, 2)
    flt.close()


@pytest.mark.parametrize("name", ["100Hz", "10Hz_50pt"])
def test_rows_match_oracle(hip, oracle, name):
    from asif_amd import workloads
    k = workloads.load_kernel(name)
    B = 1024
    x, u = workloads.make_batch_realizable(k, B)
    out = _run(hip, k, x, u, assemble=True)
    z = oracle.Realizable(oracle.load_kernel(name))
    A, b, code, info = z.assemble(np.ascontiguousarray(x.T))
    assert np.array_equal(out["code"], code)
    d = out["diag"]
    assert np.array_equal(d[0].astype(np.int32), info[:, 0])
    assert np.array_equal(d[1:1 + z.maxCrit].T.astype(np.int32), info[:, 1:1 + z.maxCrit])
    assert np.array_equal(d[1 + z.maxCrit:1 + z.maxCrit + z.npSSmax].T.astype(np.int32), info[:, 1 + z.maxCrit:])
    assert np.array_equal(out["A"].T, A)  # no FMA contraction in the assembly: bit-identical rows
    assert np.array_equal(out["b"].T, b)


@pytest.mark.parametrize("name", KERNELS)
def test_filter_matches_exact_optimum(hip, oracle, name):
    from asif_amd import workloads
    k = workloads.load_kernel(name)
    B = 16384
    x, u = workloads.make_batch_realizable(k, B)
    out = _run(hip, k, x, u)
    z = oracle.Realizable(oracle.load_kernel(name))
    ua, rl, rc = z.filter(np.ascontiguousarray(x.T), np.ascontiguousarray(u.T))
    assert np.array_equal(out["rc"], rc), f"rc mismatches {(out['rc'] != rc).sum()}"
    ok = rc == 1
    assert ok.sum() > B // 2 and (rc == -2).sum() > B // 20
    assert np.abs(out["uact"][0, ok] - ua[ok, 0]).max() <= 1e-6
    assert np.abs(out["relax"][:, ok] - rl[ok].T).max() <= 1e-6
    # where the reference leaves its outputs untouched (rc -1, -2) so does the device path
    assert np.all(out["uact"][0, ~ok] == 7.0) and np.all(out["relax"][:, ~ok] == -7.0)


def test_infeasible_critical_rows_give_rc_minus_one(hip, oracle):
    """Tight input bounds make the critical-facet interval for u empty for some boundary states."""
    from asif_amd import workloads
    k = workloads.load_kernel("100Hz")
    B = 8192
    x, u = workloads.make_batch_realizable(k, B)
    o = hip.default_realizable_options(lb=[-0.5], ub=[0.5])
    out = _run(hip, k, x, u, options=o)
    z = oracle.Realizable(oracle.load_kernel("100Hz"), lb=[-0.5], ub=[0.5])
    ua, rl, rc = z.filter(np.ascontiguousarray(x.T), np.ascontiguousarray(u.T))
    assert (rc == -1).sum() > 20
    assert np.array_equal(out["rc"], rc)
    ok = rc == 1
    assert np.abs(out["uact"][0, ok] - ua[ok, 0]).max() <= 1e-6


@pytest.mark.parametrize("npSSmax", [0, 1, 4])
def test_other_barrier_row_counts(hip, oracle, npSSmax):
    from asif_amd import workloads
    k = workloads.load_kernel("100Hz_50pt")
    B = 4096
    x, u = workloads.make_batch_realizable(k, B)
    o = hip.default_realizable_options(npSSmax=npSSmax, relaxDes=3.0, relaxOffset=0.05, relaxCost=20.0)
    out = _run(hip, k, x, u, options=o)
    z = oracle.Realizable(oracle.load_kernel("100Hz_50pt"), npSSmax=npSSmax, relaxDes=3.0, relaxOffset=0.05,
                          relaxCost=20.0)
    assert (out["dims"].nv, out["dims"].nc) == (z.nv, z.nc)
    ua, rl, rc = z.filter(np.ascontiguousarray(x.T), np.ascontiguousarray(u.T))
    assert np.array_equal(out["rc"], rc)
    ok = rc == 1
    assert np.abs(out["uact"][0, ok] - ua[ok, 0]).max() <= 1e-6
    if npSSmax > 0:
        assert np.abs(out["relax"][1, ok] - rl[ok, 1]).max() <= 1e-6


def test_update_options_and_model_parameters(hip, oracle):
    from asif_amd import workloads
    k = workloads.load_kernel("100Hz")
    B = 4096
    x, u = workloads.make_batch_realizable(k, B)
    flt = hip.RealizableFilter(k)
    kw = dict(relaxDes=4.0, relaxCost=30.0, mMin=60.0, mMax=90.0, Flo=20.0, Fhi=26.0, uncertaintyBounds=[0.05, 0.04, 0, 0])
    flt.update_options(hip.default_realizable_options(**kw))
    dev = torch.device("cuda:0")
    tx, tu = torch.from_numpy(x).to(dev), torch.from_numpy(u).to(dev)
    uact = torch.zeros((1, B), dtype=torch.float64, device=dev)
    relax = torch.zeros((2, B), dtype=torch.float64, device=dev)
    rc = torch.zeros(B, dtype=torch.int32, device=dev)
    flt.filter(tx, tu, uact, relax, rc)
    torch.cuda.synchronize()
    kw["uncertaintyBounds"] = [0.05, 0.04]
    z = oracle.Realizable(oracle.load_kernel("100Hz"), **kw)
    assert np.array_equal(flt.tables()[0], z.table()[0])
    ua, rl, rco = z.filter(np.ascontiguousarray(x.T), np.ascontiguousarray(u.T))
    assert np.array_equal(rc.cpu().numpy(), rco)
    ok = rco == 1
    assert np.abs(uact.cpu().numpy()[0, ok] - ua[ok, 0]).max() <= 1e-6
    flt.close()


def test_bad_arguments(hip):
    from asif_amd import workloads
    k = workloads.load_kernel("100Hz")
    with pytest.raises(hip.AsifHipError):
        hip.RealizableFilter(k, options=hip.default_realizable_options(npSSmax=5))
    bad = dict(k)
    bad["facetActive"] = k["facetActive"].copy()
    bad["facetActive"][3, 1] = 1000
    with pytest.raises(hip.AsifHipError):
        hip.RealizableFilter(bad)
    with pytest.raises(hip.AsifHipError):
        hip.RealizableFilter(k, model=hip.MODEL_DOUBLE_INTEGRATOR)
