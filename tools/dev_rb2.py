"""Developer probe: robust-data filter, alternative options (the failing test's)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from asif_amd import capi, workloads
import oracle_lib as O
hp = workloads.load_halfplanes("70-75kg")
B = 4096
x, u = workloads.make_batch_robust_data(hp, B)
kw = dict(npSSmax=8, relaxCost=20.0, relaxLb=1.0, mMax=75.0, Flo=20.0, Fhi=26.0, lb=[-10.0], ub=[15.0])
for refine in (2, 3, 4):
    flt = capi.RobustDataFilter(hp, options=capi.default_robust_data_options(**kw), solver=capi.default_solver(refine_steps=refine))
    dev = torch.device("cuda:0"); d = flt.dims
    uact = torch.full((1, B), 7.0, dtype=torch.float64, device=dev); relax = torch.full((1, B), -7.0, dtype=torch.float64, device=dev)
    rc = torch.zeros(B, dtype=torch.int32, device=dev); diag = torch.zeros((d.ndiag, B), dtype=torch.float64, device=dev)
    flt.filter(torch.from_numpy(x).to(dev), torch.from_numpy(u).to(dev), uact, relax, rc, diag)
    torch.cuda.synchronize()
    rcd = rc.cpu().numpy(); it = diag.cpu().numpy()[-1]
    z = O.RobustData(O.load_halfplanes("70-75kg"), **kw)
    ua, rl, rco = z.filter(np.ascontiguousarray(x.T), np.ascontiguousarray(u.T))
    bad = np.where(rcd != rco)[0]
    print("refine", refine, "mismatches", len(bad), "iters max", it.max(), "mean", it.mean())
    for i in bad[:6]:
        print("  ", i, "x", x[:, i], "u", u[0, i], "dev rc", rcd[i], "iters", it[i], "oracle rc", rco[i], "u*", ua[i], "delta*", rl[i], "dev u", uact[0, i].item())
