"""Developer probe: robust-data filter mismatches against the oracle."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from asif_amd import capi, workloads
import oracle_lib as O
hp = workloads.load_halfplanes()
B = 8192
x, u = workloads.make_batch_robust_data(hp, B)
lanes = int(sys.argv[1]) if len(sys.argv) > 1 else 0
flt = capi.RobustDataFilter(hp, solver=capi.default_solver(lanes_per_qp=lanes))
dev = torch.device("cuda:0")
d = flt.dims
uact = torch.full((1, B), 7.0, dtype=torch.float64, device=dev); relax = torch.full((1, B), -7.0, dtype=torch.float64, device=dev)
rc = torch.zeros(B, dtype=torch.int32, device=dev); diag = torch.zeros((d.ndiag, B), dtype=torch.float64, device=dev)
flt.filter(torch.from_numpy(x).to(dev), torch.from_numpy(u).to(dev), uact, relax, rc, diag)
torch.cuda.synchronize()
rcd = rc.cpu().numpy(); it = diag.cpu().numpy()[-1]
z = O.RobustData(O.load_halfplanes())
ua, rl, rco = z.filter(np.ascontiguousarray(x.T), np.ascontiguousarray(u.T))
bad = np.where(rcd != rco)[0]
print("mismatches", len(bad), "iters max", it.max(), "mean", it.mean())
A, b, code, sel = z.assemble(np.ascontiguousarray(x.T[bad]))
for k, i in enumerate(bad[:8]):
    print(i, "x", x[:, i], "u", u[0, i], "dev rc", rcd[i], "iters", it[i], "oracle rc", rco[i], "u*", ua[i], "delta*", rl[i])
    Ak = A[k].reshape(z.nv, z.nc).T
    for s in range(5):
        col = 2 + 4 * s
        print("   row", s, "h", Ak[3 * s, 1], "Lgh", Ak[3 * s, col], -Ak[3 * s, col + 2], "Lfh lo", Ak[3 * s, col + 1])
