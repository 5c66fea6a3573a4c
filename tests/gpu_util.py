"""Helpers shared by the gpu-marked parity tests: run the HIP path through the C ABI on seeded
batches and fetch the oracle's answer for the same inputs."""
import numpy as np
import torch

from asif_amd import capi, workloads


def run_filter(cfg, B, first=0, solver=None, options=None, uact_init=0.0, relax_init=0.0, x=None, udes=None,
               learning=None, model=None, variant=None):
    if model is None:
        model, variant, _ = capi.CONFIGS[cfg]
    flt = capi.Filter(model, variant, options=options, solver=solver)
    if learning is not None:
        flt.set_learning(learning)
    d = flt.dims
    if x is None:
        x, udes = workloads.make_batch(cfg, B, first)
    B = x.shape[1]
    dev = torch.device("cuda:0")
    tx = torch.from_numpy(x).to(dev)
    tu = torch.from_numpy(udes).to(dev)
    uact = torch.full((d.nu, B), float(uact_init), dtype=torch.float64, device=dev)
    relax = torch.full((d.nrelax, B), float(relax_init), dtype=torch.float64, device=dev)
    rc = torch.zeros(B, dtype=torch.int32, device=dev)
    diag = torch.zeros((d.ndiag, B), dtype=torch.float64, device=dev)
    flt.filter(tx, tu, uact, relax, rc, diag)
    torch.cuda.synchronize()
    out = dict(x=x, udes=udes, uact=uact.cpu().numpy(), relax=relax.cpu().numpy(), rc=rc.cpu().numpy(),
               diag=diag.cpu().numpy(), dims=d)
    flt.close()
    return out


def run_assemble(cfg, B, first=0, options=None, x=None, learning=None, model=None, variant=None):
    if model is None:
        model, variant, _ = capi.CONFIGS[cfg]
    flt = capi.Filter(model, variant, options=options)
    if learning is not None:
        flt.set_learning(learning)
    d = flt.dims
    if x is None:
        x, _ = workloads.make_batch(cfg, B, first)
    B = x.shape[1]
    dev = torch.device("cuda:0")
    tx = torch.from_numpy(x).to(dev)
    A = torch.zeros((d.nc * d.nv, B), dtype=torch.float64, device=dev)
    b = torch.zeros((d.nc, B), dtype=torch.float64, device=dev)
    code = torch.zeros(B, dtype=torch.int32, device=dev)
    diag = torch.zeros((d.ndiag, B), dtype=torch.float64, device=dev)
    flt.assemble(tx, A, b, code, diag)
    torch.cuda.synchronize()
    out = dict(x=x, A=A.cpu().numpy(), b=b.cpu().numpy(), code=code.cpu().numpy(), diag=diag.cpu().numpy(), dims=d)
    flt.close()
    return out


def oracle_filter(oracle, cfg, x, udes, uact_init=0.0, relax_init=0.0, solver=0, settings=None):
    """Oracle answer for SoA inputs; untouched slots keep the init value like the device path."""
    model, variant = oracle.CONFIGS[cfg]
    o = oracle.default_options(model, variant)
    B = x.shape[1]
    d = oracle.dims(model, variant, o)
    ua, rl, rc = oracle.filter_batch(model, variant, o, np.ascontiguousarray(x.T), np.ascontiguousarray(udes.T),
                                     solver, settings, uact_init=np.full((B, d.nu), float(uact_init)))
    rl = np.where(np.isnan(rl), relax_init, rl)
    return ua.T, rl.T, rc
