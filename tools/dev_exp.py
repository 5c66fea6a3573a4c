"""Developer probe: finish-first (polish 2) against finish-at-checks (polish 1/3) on every path: parity + time."""
import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch, ctypes as C
from asif_amd import capi, workloads
import gpu_util, oracle_lib as O
hip = C.CDLL("libamdhip64.so")
hip.hipEventRecord.argtypes = [C.c_void_p, C.c_void_p]
hip.hipEventElapsedTime.argtypes = [C.POINTER(C.c_float), C.c_void_p, C.c_void_p]
def timeit(flt, x, u, B, n=100):
    dev = torch.device("cuda:0"); d = flt.dims
    tx = torch.from_numpy(x).to(dev); tu = torch.from_numpy(u).to(dev)
    uact = torch.zeros((d.nu, B), dtype=torch.float64, device=dev); relax = torch.zeros((d.nrelax, B), dtype=torch.float64, device=dev)
    rc = torch.zeros(B, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream(); fn = flt.lib.asif_hip_filter_batch
    args = (flt.handle, B, tx.stride(0), C.c_void_p(tx.data_ptr()), C.c_void_p(tu.data_ptr()), C.c_void_p(uact.data_ptr()),
            C.c_void_p(relax.data_ptr()), C.c_void_p(rc.data_ptr()), None, C.c_void_p(stream.cuda_stream))
    for _ in range(5): fn(*args)
    torch.cuda.synchronize()
    ev = []
    for _ in range(2):
        e = C.c_void_p(); hip.hipEventCreate(C.byref(e)); ev.append(e)
    sp = C.c_void_p(stream.cuda_stream)
    hip.hipEventRecord(ev[0], sp)
    for _ in range(n): fn(*args)
    hip.hipEventRecord(ev[1], sp); torch.cuda.synchronize()
    ms = C.c_float(); hip.hipEventElapsedTime(C.byref(ms), ev[0], ev[1])
    return ms.value / n * 1e3, uact.cpu().numpy(), rc.cpu().numpy()
for cfg, B, n in ((2, 65536, 100), (5, 8192, 100), (4, 32768, 10), (3, 16384, 5), (9, 65536, 20)):
    model, variant, _ = capi.CONFIGS[cfg]
    x, u = workloads.make_batch(cfg, B)
    m = min(B, {2: 65536, 5: 8192, 4: 8192, 3: 256, 9: 8192}[cfg])
    ua, rl, rco = gpu_util.oracle_filter(O, cfg, x[:, :m], u[:, :m])
    for pol in (1, 2):
        flt = capi.Filter(model, variant, solver=capi.default_solver(polish=pol))
        t, uact, rc = timeit(flt, x, u, B, n)
        ok = (rco == 1) | (rco == 2)
        print(f"cfg {cfg} polish {pol}: {t:9.2f} us  rc mism {(rc[:m] != rco).sum()}  err {np.abs(uact[:, :m] - ua)[:, ok].max():.1e}")
        flt.close()
k = workloads.load_kernel("100Hz"); x, u = workloads.make_batch_realizable(k, 65536)
z = O.Realizable(O.load_kernel("100Hz")); ua, rl, rco = z.filter(np.ascontiguousarray(x.T), np.ascontiguousarray(u.T))
for pol in (1, 2):
    flt = capi.RealizableFilter(k, solver=capi.default_solver(polish=pol)); t, uact, rc = timeit(flt, x, u, 65536)
    print(f"cfg 6 polish {pol}: {t:9.2f} us  rc mism {(rc != rco).sum()}  err {np.abs(uact[0] - ua[:, 0])[rco == 1].max():.1e}"); flt.close()
hp = workloads.load_halfplanes(); x, u = workloads.make_batch_robust_data(hp, 8192)
z = O.RobustData(O.load_halfplanes()); ua, rl, rco = z.filter(np.ascontiguousarray(x.T), np.ascontiguousarray(u.T))
for pol in (1, 2):
    flt = capi.RobustDataFilter(hp, solver=capi.default_solver(polish=pol)); t, uact, rc = timeit(flt, x, u, 8192)
    print(f"cfg 7 polish {pol}: {t:9.2f} us  rc mism {(rc != rco).sum()}  err {np.abs(uact[0] - ua[:, 0])[rco == 1].max():.1e}"); flt.close()
