"""Developer probe: realizable filter iteration statistics."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from asif_amd import capi, workloads
import oracle_lib as O
k = workloads.load_kernel("100Hz")
B = 65536
x, u = workloads.make_batch_realizable(k, B)
flt = capi.RealizableFilter(k)
dev = torch.device("cuda:0"); d = flt.dims
uact = torch.zeros((1, B), dtype=torch.float64, device=dev); relax = torch.zeros((2, B), dtype=torch.float64, device=dev)
rc = torch.zeros(B, dtype=torch.int32, device=dev); diag = torch.zeros((d.ndiag, B), dtype=torch.float64, device=dev)
flt.filter(torch.from_numpy(x).to(dev), torch.from_numpy(u).to(dev), uact, relax, rc, diag)
torch.cuda.synchronize()
it = diag.cpu().numpy()[-1]; rcd = rc.cpu().numpy()
print("iters hist", np.unique(it, return_counts=True))
z = O.Realizable(O.load_kernel("100Hz"))
ua, rl, rco = z.filter(np.ascontiguousarray(x.T), np.ascontiguousarray(u.T))
bad = np.where(it > 2)[0]
print("n slow", len(bad), "rc mism", (rcd != rco).sum())
A, b, code, info = z.assemble(np.ascontiguousarray(x.T[bad[:5]]))
for kk, i in enumerate(bad[:5]):
    Ak = A[kk].reshape(z.nv, z.nc).T
    print(i, "x", x[:, i], "u", u[0, i], "iters", it[i], "u*", ua[i], "relax", rl[i], "ncrit", info[kk, 0])
    print("   barrier rows", Ak[27, 0], b[kk, 27], "|", Ak[28, 0], b[kk, 28])
