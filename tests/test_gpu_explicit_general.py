"""Class ASIF beyond the shipped single-input example: the optional paths of src/asif.cpp on the device
-- npSSmax < npSS row selection (:250-268), caller-supplied Lie derivatives (:130-165, 287-292) and nu = 2 on a
synthetic two-input model (no example of the reference has more than one input; src/asif.cpp is written for any nu).
Oracle: the same paths restated in oracle/or_assembly.c (assemble_explicit_lie), exact optimum by enumeration."""
import numpy as np
import pytest
import torch

import gpu_util
from asif_amd import workloads

pytestmark = pytest.mark.gpu

U_TOL = 1e-6


def _run(hip, cfg, B, options=None, solver=None, lie=None):
    model, variant, _ = hip.CONFIGS[cfg]
    flt = hip.Filter(model, variant, options=options, solver=solver)
    d = flt.dims
    x, udes = workloads.make_batch(cfg, B)
    dev = torch.device("cuda:0")
    tx, tu = torch.from_numpy(x).to(dev), torch.from_numpy(udes).to(dev)
    uact = torch.full((d.nu, B), 7.0, dtype=torch.float64, device=dev)
    relax = torch.full((d.nrelax, B), -7.0, dtype=torch.float64, device=dev)
    rc = torch.zeros(B, dtype=torch.int32, device=dev)
    if lie is None:
        flt.filter(tx, tu, uact, relax, rc)
    else:
        flt.filter_lie(tx, tu, torch.from_numpy(lie[0]).to(dev), torch.from_numpy(lie[1]).to(dev), uact, relax, rc)
    A = torch.zeros((d.nc * d.nv, B), dtype=torch.float64, device=dev)
    b = torch.zeros((d.nc, B), dtype=torch.float64, device=dev)
    code = torch.zeros(B, dtype=torch.int32, device=dev)
    diag = torch.zeros((d.ndiag, B), dtype=torch.float64, device=dev)
    flt.assemble(tx, A, b, code, diag)
    torch.cuda.synchronize()
    flt.close()
    return dict(x=x, udes=udes, uact=uact.cpu().numpy(), relax=relax.cpu().numpy(), rc=rc.cpu().numpy(),
                A=A.cpu().numpy(), b=b.cpu().numpy(), kept=diag.cpu().numpy()[:d.nc].astype(int), dims=d)


@pytest.mark.parametrize("cfg,keep", [(2, 1), (2, 2), (2, 3), (11, 2), (11, 4)])
def test_row_selection(hip, oracle, cfg, keep):
    B = 4096
    od = hip.default_options(*hip.CONFIGS[cfg][:2])
    od.npSSmax = keep
    out = _run(hip, cfg, B, options=od)
    assert out["dims"].nc == keep
    model, variant = oracle.CONFIGS[cfg]
    oo = oracle.default_options(model, variant)
    oo.npSSmax = keep
    assert oracle.dims(model, variant, oo).nc == keep
    A, b, code, _ = oracle.assemble_batch(model, variant, oo, np.ascontiguousarray(out["x"].T))
    np.testing.assert_allclose(out["A"].T, A, rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(out["b"].T, b, rtol=1e-12, atol=1e-12)
    for i in (0, 1, 17, B - 1):  # which safety functions were kept, in row order
        oracle.assemble(model, variant, oo, out["x"][:, i])
        assert list(oracle.last_kept_rows()) == list(out["kept"][:, i])
    ua, rl, rc = oracle.filter_batch(model, variant, oo, np.ascontiguousarray(out["x"].T),
                                     np.ascontiguousarray(out["udes"].T), uact_init=np.full((B, out["dims"].nu), 7.0))
    assert np.array_equal(out["rc"], rc)
    assert np.abs(out["uact"] - ua.T).max() <= U_TOL
    assert len(np.unique(rc)) == 2


def test_two_input_model(hip, oracle):
    B = 16384
    out = _run(hip, 11, B)
    d = out["dims"]
    assert (d.nu, d.nv, d.nc, d.npSS) == (2, 3, 5, 5)
    model, variant = oracle.CONFIGS[11]
    oo = oracle.default_options(model, variant)
    A, b, code, _ = oracle.assemble_batch(model, variant, oo, np.ascontiguousarray(out["x"].T))
    np.testing.assert_allclose(out["A"].T, A, rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(out["b"].T, b, rtol=1e-12, atol=1e-12)
    ua, rl, rc = oracle.filter_batch(model, variant, oo, np.ascontiguousarray(out["x"].T),
                                     np.ascontiguousarray(out["udes"].T), uact_init=np.full((B, 2), 7.0))
    assert np.array_equal(out["rc"], rc), f"{(out['rc'] != rc).sum()} rc mismatches"
    assert (rc == 1).sum() > 1000 and (rc == -1).sum() > 100
    assert np.abs(out["uact"] - ua.T).max() <= U_TOL
    ok = rc == 1
    assert np.abs(out["relax"][0, ok] - rl[ok, 0]).max() <= U_TOL
    assert np.all(out["uact"][:, ~ok] == 7.0) and np.all(out["relax"][0, ~ok] == -7.0)  # untouched on failure
    assert np.all(np.abs(out["uact"][:, ok]) <= 1.0)                                    # inputSaturate per input


@pytest.mark.parametrize("cfg,keep", [(2, 0), (2, 2), (11, 0), (11, 3)])
def test_caller_supplied_lie_derivatives(hip, oracle, cfg, keep):
    B = 2048
    model, variant = oracle.CONFIGS[cfg]
    od = hip.default_options(*hip.CONFIGS[cfg][:2])
    oo = oracle.default_options(model, variant)
    if keep:
        od.npSSmax = oo.npSSmax = keep
    d = oracle.dims(model, variant, oo)
    rng = np.random.default_rng(cfg * 10 + keep)
    lfh = rng.normal(0, 1.0, (d.nc, B))
    lgh = rng.normal(0, 1.0, (d.nc * d.nu, B))
    out = _run(hip, cfg, B, options=od, lie=(lfh, lgh))
    ua, rl, rc = oracle.filter_explicit_lie(model, oo, np.ascontiguousarray(out["x"].T),
                                            np.ascontiguousarray(out["udes"].T), np.ascontiguousarray(lfh.T),
                                            np.ascontiguousarray(lgh.T))
    assert np.array_equal(out["rc"], rc), f"{(out['rc'] != rc).sum()} rc mismatches"
    ok = rc == 1
    assert ok.sum() > 100 and (~ok).sum() > 20
    assert np.abs(out["uact"][:, ok] - ua[ok].T).max() <= U_TOL
