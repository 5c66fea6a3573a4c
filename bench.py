#!/usr/bin/env python3
"""bench.py -- QP solves/s of the batched CBF-QP filter() hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config 2..12|qp] [--batch B] [--kernel 100Hz]
                    [--presolve 1] [--polish 0|1|2] [--no-cpu-baseline] [--no-pcie]
                    [--shape c2|c3|c4|c5full] [--lanes 64]       (with --config qp; the realizable filter's 38 x 29 ...
                    86 x 65 and the shipped half-planes' 22 x 15: tools/dev_rz_time.py, tools/scratch/bench_rd22.py)

One "step" = one pass of the filter (constraint assembly + in-kernel solve + saturation + return
code) over one batch of synthetic states that is already resident in HBM when the timed region starts.
`value` is therefore KERNEL-ONLY (inputs and outputs in HBM); the transfer-inclusive rate of SURVEY 8(d)
(H2D + kernels + D2H of the same batch from host buffers) is reported beside it as `pcie_inclusive`, never as `value`.
Default workload: BASELINE.json configs[1] -- DoubleIntegrator explicit CBF, batch 65 536 per GPU.
N > 1: one process per GPU, the batch axis is sharded -- every rank owns its own slice of the seeded instance
stream (weak scaling), no data-path collective and no RCCL at all: the barrier around the timed region and the
max-over-ranks of the elapsed time go over a gloo (TCP, host) process group.  Started by torchrun (RANK / LOCAL_RANK /
WORLD_SIZE in the environment) or, when those are absent and --gpus N > 1, by this script spawning its N ranks as
child processes before anything touches a GPU.
--config qp: asif_hip_qp_solve_batch on pre-assembled problems of one shape (SURVEY 8d's second byte model).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from asif_amd import capi, dist, workloads  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
SHADER_CLOCK_HZ = 2.4e9  # MI355X peak engine clock
N_SIMD = 256 * 4
VALU_ISSUE_CYCLES = 4    # one wave's FP64 / FP32 vector instruction occupies its SIMD's issue for 4 cycles
                         # (MI355X_MICROARCH.md, row "vector-instruction ISSUE cost"; FP64 FMA: 16 lanes per cycle)
PROFILE_ROUNDS = ("r04", "r03")  # committed rocprofv3 summaries, newest first: a config not re-profiled this round keeps its last one

# algorithmic HBM bytes per instance of the state->input path: read 8(nx+nu), write 8(nu+nrelax)+4
# the dominant kernel of each config's step, as its name appears in a rocprofv3 kernel trace
DOMINANT_KERNEL = {2: "explicit_light_kernel", 3: "implicit_rows_kernel", 4: "tb_rows_kernel", 5: "qp_policy_kernel",
                   6: "realizable_filter_kernel", 7: "robust_data", 8: "tb_rows_kernel", 9: "implicit_rows_kernel",
                   10: "implicit_rows_kernel", 11: "explicit", 12: "tb_rows_kernel"}
ALG_BYTES = {2: 44, 3: 52, 4: 60, 5: 44, 6: 52, 7: 44, 8: 44, 9: 52, 10: 52, 11: 60, 12: 44}
IMPLICIT_RB_CFG = 10  # ASIFimplicitRB on the pendulum model; not a BASELINE.json config
REALIZABLE_CFG = 6  # ASIFrealizable on the sampled double integrator; not a BASELINE.json config
ROBUST_DATA_CFG = 7  # ASIFrobust on the shipped data: examples/DoubleIntegrator_Robust.cpp + KernelData_70-135kg.h
WORKLOAD = {
    2: "C2 DoubleIntegrator explicit CBF (ASIF::filter), seeded x in U[-1.2,1.2]^2, uDes in U[-1.5,1.5]",
    3: "C3 InvertedPendulum_Implicit (ASIFimplicit::filter, 5001-step backup trajectory)",
    4: "C4 segway_implicit_tb (ASIFimplicitTB::filter, 316-step backup trajectory), one GPU's share",
    5: "C5 InvertedPendulum_Robust (ASIFrobust::filter, affine-arithmetic rows, nv=18 nc=12)",
    6: "C6 DoubleIntegrator_RealizableSampled (ASIFrealizable::filter, polytopic kernel, facet search + interval rows)",
    7: "C7 DoubleIntegrator_Robust (ASIFrobust::filter on the shipped 100 half-planes, 5 kept per call, nv=22 nc=15)",
    8: "C8 InvertedPendulum_ImplicitTB (ASIFimplicitTB::filter, 11 551-step backup trajectory)",
    9: "C9 DoubleIntegrator_implicit (ASIFimplicit::filter, 201-step backup trajectory, npBTSS 4, nv=3 nc=17)",
    10: "C10 ASIFimplicitRB::filter on the InvertedPendulum_Implicit model (backup input held 10 steps, interval "
        "margins under x_unc, two 4-16-16-1 ReLU residual networks with seeded weights)",
    11: "C11 class ASIF on a synthetic two-input model (nx=2, nu=2, five half-planes; no reference example has nu > 1)",
    12: "C12 DoubleIntegrator_implicit_tb (ASIFimplicitTB::filter, 2 101-step backup trajectory, nv=2 nc=18)",
}


def rb_options(lib_module, model, variant):
    """Config 10's options on either side (asif_amd.capi or the oracle's ctypes view): the example's options
    plus x_unc and use_learning; the caller attaches the weights of workloads.make_learning()."""
    o = lib_module.default_options(model, variant)
    for i, v in enumerate(workloads.RB_X_UNC):
        o.x_unc[i] = v
    o.use_learning = 1
    return o


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def host_cores():
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, 16))  # the GPU box's CPU share for one GPU is 16


def cpu_baseline_handle(cfg, data_name, gpu_uact, gpu_rc, x, udes):
    """cpu_baseline() for configs 6 and 7: the oracle's ASIFrealizable / ASIFrobust-on-data restatement with its
    OSQP-style ADMM on the full nv x nc problem (the QP the reference hands to OSQP), python threads over ctypes
    calls (GIL released)."""
    from concurrent.futures import ThreadPoolExecutor
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    O.build()
    if cfg == REALIZABLE_CFG:
        z, what = O.Realizable(O.load_kernel(data_name)), "ASIFrealizable"
    else:
        z, what = O.RobustData(O.load_halfplanes(data_name)), "ASIFrobust (DoubleIntegrator_Robust data)"
    cores = host_cores()
    xa, ua = np.ascontiguousarray(x.T), np.ascontiguousarray(udes.T)
    probe = min(2000, xa.shape[0])
    t = time.perf_counter()
    z.filter(xa[:probe], ua[:probe], solver=O.SOLVER_ADMM)
    per = (time.perf_counter() - t) / probe
    n = int(max(cores * 16, min(15.0 / per * cores, xa.shape[0])))
    cuts = [n * j // cores for j in range(cores + 1)]
    t = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        list(ex.map(lambda j: z.filter(xa[cuts[j]:cuts[j + 1]], ua[cuts[j]:cuts[j + 1]], solver=O.SOLVER_ADMM),
                    range(cores)))
    dt = time.perf_counter() - t
    m = min(xa.shape[0], 65536)
    ue, _, rc = z.filter(xa[:m], ua[:m])
    ok = rc == 1
    err = float(np.abs(gpu_uact[0, :m][ok] - ue[ok, 0]).max()) if ok.any() else 0.0
    return ({"value": n / dt, "unit": "QP solves/s", "cores": cores, "kind": "port",
             "cpu_model": cpu_model(), "single_thread_value": 1.0 / per,
             "sample": f"{n} instances of the same seeded workload, {what} restatement + OSQP-style ADMM on "
                       f"the full {z.nv}x{z.nc} QP (eps 1e-3, max_iter 2000, cold start, {per * 1e6:.1f} us per "
                       f"filter() on one core) over {cores} host threads"},
            {"max_abs_u_err_vs_exact": err, "rc_mismatches": int((rc != gpu_rc[:m]).sum()), "checked_instances": m})


def cpu_baseline(cfg, gpu_uact, gpu_rc, x, udes, seconds=15.0):
    """Rank 0, N=1 only: the oracle's OSQP-style restatement (OSQP 0.6 defaults, cold start) timed on
    the host cores over a bounded sample of the same workload (`seconds` of CPU work in total), plus the parity numbers
    the metric asks for (max|u - u_ref| against the exact optimum, rc mismatches on the FULL batch) with the oracle as
    the checker, and SURVEY 8(c)'s second deviation figure: max|u_gpu - u_cpuADMM(eps 1e-3)|, the "OSQP-like envelope"
    -- how far an OSQP-default-tolerance answer sits from what the device returns, on the sampled instances."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    O.build()
    model, variant = O.CONFIGS[cfg]
    o = O.default_options(model, variant)
    if cfg == IMPLICIT_RB_CFG:
        o = rb_options(O, model, variant)
        o.set_learning(O.Learning.from_dict(workloads.make_learning()))
    cores = host_cores()
    probe = {2: 20000, 3: 16, 4: 256, 5: 2000, 8: 16, 9: 512, 10: 16, 11: 20000, 12: 64}[cfg]
    xs, us = O.make_batch(cfg, probe)
    t = time.perf_counter()
    O.filter_batch(model, variant, o, xs, us, O.SOLVER_ADMM, None, 1)
    per = (time.perf_counter() - t) / probe
    n = int(max(cores * 16, min(seconds / per, 4e6)))  # about `seconds` of CPU work in total
    xs, us = O.make_batch(cfg, n)
    t = time.perf_counter()
    ua_admm, _, rc_admm = O.filter_batch(model, variant, o, xs, us, O.SOLVER_ADMM, None, cores,
                                         uact_init=np.zeros((n, gpu_uact.shape[0])))
    dt = time.perf_counter() - t
    m = min(x.shape[1], {2: 65536, 3: 16384, 4: 32768, 5: 8192, 8: 16384, 9: 65536, 10: 16384, 11: 65536, 12: 65536}[cfg])  # the full batch
    ua, _, rc = O.filter_batch(model, variant, o, np.ascontiguousarray(x[:, :m].T),
                               np.ascontiguousarray(udes[:, :m].T), O.SOLVER_EXACT, None, cores,
                               uact_init=np.zeros((m, gpu_uact.shape[0])))
    ok = (rc == 1) | (rc == 2)
    err = float(np.abs(gpu_uact[:, :m].T[ok] - ua[ok]).max()) if ok.any() else 0.0
    mism = int((rc != gpu_rc[:m]).sum())
    # the sample is the head of the same seeded stream as the device batch (rank 0: first = 0)
    k = min(n, m)
    both = np.isin(rc_admm[:k], (1, 2)) & np.isin(gpu_rc[:k], (1, 2))
    env = float(np.abs(gpu_uact[:, :k].T[both] - ua_admm[:k][both]).max()) if both.any() else None
    return ({"value": n / dt, "unit": "QP solves/s", "cores": cores, "kind": "port",
             "cpu_model": cpu_model(), "single_thread_value": 1.0 / per,
             "sample": f"{n} instances of the same seeded workload, OSQP-style ADMM restatement "
                       f"(eps 1e-3, max_iter 2000, cold start, {per * 1e6:.2f} us per filter() on one core) "
                       f"over {cores} host threads; OSQP itself is not in the image, this is the oracle's restatement"},
            {"max_abs_u_err_vs_exact": err, "rc_mismatches": mism, "checked_instances": m,
             "osqp_like_envelope": {"max_abs_u_gpu_minus_u_admm_eps1e-3": env, "instances": int(both.sum()),
                                    "rc_differs_from_gpu": int((rc_admm[:k] != gpu_rc[:k]).sum()),
                                    "note": "the OSQP-style restatement at OSQP's default tolerances against the device's "
                                            "answer; not an error of either: eps 1e-3 without polish is this far from the "
                                            "exact optimum"}})


def pcie_inclusive(flt, x, udes, rc_dev, uact_dev, reps=20):
    """The same batch handed over as HOST buffers (asif_hip_filter_batch_host: H2D, kernels, D2H, blocking).
    Reported beside the headline, never as `value` (t_kernel_wall incl. transfers)."""
    import ctypes as C
    d = flt.dims
    B = x.shape[1]
    fn = flt.lib.asif_hip_filter_batch_host
    res = {}
    for kind in ("pinned", "pageable"):
        bufs = [torch.from_numpy(np.ascontiguousarray(x)), torch.from_numpy(np.ascontiguousarray(udes)),
                torch.zeros((d.nu, B), dtype=torch.float64), torch.zeros((d.nrelax, B), dtype=torch.float64),
                torch.zeros(B, dtype=torch.int32)]
        if kind == "pinned":
            bufs = [t.pin_memory() for t in bufs]
        args = (flt.handle, B) + tuple(C.c_void_p(t.data_ptr()) for t in bufs)
        for _ in range(3):
            capi.check(fn(*args))
        t0 = time.perf_counter()
        for _ in range(reps):
            capi.check(fn(*args))
        dt = (time.perf_counter() - t0) / reps
        same = bool(np.array_equal(bufs[4].numpy(), rc_dev))
        ok = np.isin(rc_dev, (1, 2))
        same = same and bool(np.array_equal(bufs[2].numpy()[:, ok], uact_dev.cpu().numpy()[:, ok]))
        res[kind] = {"value": B / dt, "unit": "QP solves/s", "ms_per_call": dt * 1e3, "bitwise_equal_to_device_path": same}
    return res


def profile_numbers(tag):
    """Counter-derived numbers of the committed rocprofv3 passes of this command (profiles/<round>/): PMC passes cannot
    run inside this process.  Returns (traffic_bytes_per_step, sq summary dict or None, source string)."""
    traffic, sq, src = None, None, None
    for rnd in PROFILE_ROUNDS:
        base = os.path.join(ROOT, "profiles", rnd)
        f = os.path.join(base, f"{tag}_pmc_summary.json")
        if traffic is None and os.path.isfile(f):
            traffic = json.load(open(f)).get("traffic_bytes_per_step")
            src = f"profiles/{rnd}/{tag}_pmc_summary.json"
        f = os.path.join(base, f"{tag}_sq_summary.json")
        if sq is None and os.path.isfile(f):
            sq = json.load(open(f))
            sq["_source"] = f"profiles/{rnd}/{tag}_sq_summary.json"
    return traffic, sq, src


def rocprof_direct(tag, kernel_substr):
    """AverageNs of the dominant kernel in the committed rocprofv3 --kernel-trace --stats summary of the DIRECT-launch
    leg of this command alone (tools/prof_direct.sh: --graph 0 --no-pcie --no-cpu-baseline, nothing else in the
    process): (average ns, calls, file) or None.  A reader recomputes roofline.frac_rocprof from that CSV."""
    import csv
    for rnd in PROFILE_ROUNDS:
        f = os.path.join(ROOT, "profiles", rnd, f"{tag}_direct_kernel_stats.csv")
        if os.path.isfile(f):
            for r in csv.DictReader(open(f)):
                if kernel_substr in r["Name"]:
                    return float(r["AverageNs"]), int(r["Calls"]), f"profiles/{rnd}/{tag}_direct_kernel_stats.csv"
    return None


def valu_roofline(sq, step_ms, tag):
    """The bound that binds this path: vector-ALU issue.  issued wave-instructions (SQ_INSTS_VALU, all kernels of one
    step, from the committed PMC pass) x 4 issue cycles / (SIMDs x shader cycles of the measured step)."""
    if not sq or not sq.get("insts_valu_per_step") or step_ms <= 0:
        return None
    avail = N_SIMD * SHADER_CLOCK_HZ * step_ms * 1e-3
    used = sq["insts_valu_per_step"] * VALU_ISSUE_CYCLES
    slots = None
    if sq.get("insts_salu_per_step") is not None:
        # a wave issues one instruction per turn whatever its type: with a single wave on a SIMD (this path's batch
        # sizes) scalar instructions take issue turns away from vector ones instead of overlapping with them
        slots = (sq["insts_valu_per_step"] + sq["insts_salu_per_step"]) * VALU_ISSUE_CYCLES / avail
    return {"bound": "valu-issue", "insts_valu_per_step": sq["insts_valu_per_step"],
            "insts_salu_per_step": sq.get("insts_salu_per_step"), "frac_valu_plus_salu": slots,
            "waves_per_step": sq.get("waves_per_step"), "insts_valu_per_wave": sq.get("insts_valu_per_wave"),
            "issue_cycles_per_inst": VALU_ISSUE_CYCLES, "simd_cycles_available": avail, "frac": used / avail,
            "counters_source": f"{sq.get('_source', tag)} (rocprofv3 --pmc, separate pass of the "
                               "same command); duration measured live",
            "single_wave_ceiling": 4.0 / 5.16,
            "note": "frac = share of all SIMD issue slots of the chip (4 cycles per vector instruction: the 78.6 TFLOP/s "
                    "FP64 peak) used by vector instructions over the step.  Measured on this part "
                    "(profiles/r03/fp64_peak_microbench.txt, simd_mix_microbench.txt): ONE wave on a SIMD issues an "
                    "independent v_fma_f64 every 5.16 cycles (59 TFLOP/s with a wave on every SIMD; 72 TFLOP/s needs four "
                    "or more waves per SIMD) and a dependent one every 6.25, so 0.78 is the ceiling of a kernel whose batch "
                    "gives a SIMD one wave, and a branch instruction costs such a wave 45-55 cycles"}


def spawn_ranks(n):
    """--gpus N without a launcher: start the N ranks as fresh child processes (nothing has touched a GPU yet in this
    one), hand them the torchrun environment, pass rank 0's JSON line through."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for pr in procs:
        rc = rc or pr.wait()
    sys.exit(rc)


# --config qp: (filter config whose rows are solved, nv, nc, algorithmic bytes per instance by SURVEY 8(d)'s formula
# read 8 (nv + nv + nc nv + nc + 2 nv) + write 8 nv + 4.  The formula gives 180 / 1436 / 516 / 2548; the survey's table
# prints 164 and 500 for the two nv = 2 shapes (16 bytes short of its own formula) -- the formula is used.
QP_SHAPES = {
    "c2": (2, 2, 4, 180), "c3": (3, 3, 41, 1436), "c4": (4, 2, 18, 516), "c5full": (5, 18, 12, 2548),
}


def qp_problem(cfg, B, dev, first=0):
    """Pre-assembled QPs of a filter config, built with the product's own rows kernel (asif_hip_assemble_batch) and the
    cost / bounds the class's initialize() / updateCost() set -- no oracle involved."""
    model, variant, _ = capi.CONFIGS[cfg]
    flt = capi.Filter(model, variant, device=dev.index)
    d, o = flt.dims, flt.options
    x, udes = workloads.make_batch(cfg, B, first=first)
    tx = torch.from_numpy(x).to(dev)
    A = torch.zeros((d.nc * d.nv, B), dtype=torch.float64, device=dev)
    b = torch.zeros((d.nc, B), dtype=torch.float64, device=dev)
    code = torch.zeros(B, dtype=torch.int32, device=dev)
    flt.assemble(tx, A, b, code)
    torch.cuda.synchronize()
    flt.close()
    keep = np.isin(code.cpu().numpy(), (1, 2))  # TB: instances that never reach the backup set have no QP
    idx = np.where(keep)[0]
    sel = torch.from_numpy(idx[np.arange(B) % len(idx)]).to(dev)  # tiled back up to B problems
    A, b = A[:, sel].contiguous(), b[:, sel].contiguous()
    u = torch.from_numpy(udes).to(dev)[:, sel]
    nv = d.nv
    Hd = torch.zeros((nv, B), dtype=torch.float64, device=dev)
    c = torch.zeros((nv, B), dtype=torch.float64, device=dev)
    lb = torch.zeros((nv, B), dtype=torch.float64, device=dev)
    ub = torch.full((nv, B), o.inf, dtype=torch.float64, device=dev)
    Hd[0], c[0], lb[0], ub[0] = 1.0, -2.0 * u[0], o.lb[0], o.ub[0]
    be = None
    if cfg == 2:    # src/asif.cpp:84-98: delta pinned
        Hd[1], c[1], lb[1], ub[1] = o.relaxCost, -2.0 * o.relaxCost * o.relaxLb, o.relaxLb, o.relaxLb
    elif cfg == 3:  # src/asif_implicit.cpp:237-254
        Hd[1:3] = o.relaxCost
        c[1], c[2] = -2.0 * o.relaxCost * o.relaxLb, -2.0 * o.relaxCost * o.relaxReachLb
        lb[1], lb[2] = o.relaxLb, o.relaxReachLb
    elif cfg == 4:  # src/asif_implicit_tb.cpp:198-210
        Hd[1], c[1], lb[1] = o.relaxCost, -2.0 * o.relaxCost * o.relaxLb, o.relaxLb
    elif cfg == 5:  # src/asif_robust.cpp:89-148: zero cost on the multipliers, two of three rows equalities
        Hd[1], c[1], lb[1] = o.relaxCost, -2.0 * o.relaxCost * o.relaxLb, o.relaxLb
        be = [(i % 3) != 0 for i in range(d.nc)]
    return dict(Hd=Hd, c=c, A=A, b=b, lb=lb, ub=ub, be=be, x=x[:, idx[np.arange(B) % len(idx)]],
                udes=udes[:, idx[np.arange(B) % len(idx)]], nv=nv, nc=d.nc)


_native = [False, None]


def native_loop():
    """asif_amd/host/libbench_loop.so (bench_loop.c), built on demand with gcc; None when that is not possible."""
    if _native[0]:
        return _native[1]
    _native[0] = True
    import ctypes as C
    import subprocess
    so = os.path.join(ROOT, "asif_amd", "host", "libbench_loop.so")
    try:
        if not os.path.exists(so):  # normally built by __graft_entry__.build(); ranks may race here: build aside, rename
            tmp = f"{so}.{os.getpid()}.tmp"
            subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", os.path.join(ROOT, "asif_amd", "host", "bench_loop.c"),
                                   "-o", tmp])
            os.replace(tmp, so)
        _native[1] = C.CDLL(so)
    except Exception as e:
        print(f"bench.py: native step loop unavailable ({e}); stepping from Python", file=sys.stderr)
    return _native[1]


def upload_graph(graph, stream):
    """The instantiated graph is handed to the device before the timed region (hipGraphUpload: preparation, nothing runs),
    so that the timed replay does not include it.  Returns the raw executable-graph handle, or None."""
    if graph is None or os.environ.get("BENCH_GRAPH_UPLOAD", "1") == "0":
        return None
    import ctypes as C
    try:
        ex = graph.raw_cuda_graph_exec()
        hip = C.CDLL("libamdhip64.so")
        hip.hipGraphUpload.argtypes = [C.c_void_p, C.c_void_p]
        if hip.hipGraphUpload(C.c_void_p(ex), C.c_void_p(stream.cuda_stream)) != 0:
            return None
        torch.cuda.synchronize()
        return ex
    except Exception:
        return None


def launch_graph(graph, graph_exec, stream):
    import ctypes as C
    if graph_exec is not None:
        hip = C.CDLL("libamdhip64.so")
        hip.hipGraphLaunch.argtypes = [C.c_void_p, C.c_void_p]
        if hip.hipGraphLaunch(C.c_void_p(graph_exec), C.c_void_p(stream.cuda_stream)) == 0:
            return
    with torch.cuda.stream(stream):
        graph.replay()


def bench_qp(args, grp, dev):
    """asif_hip_qp_solve_batch on one shape: the QPWrapperAbstract path (initialize + solve + getSolution, cold start)
    for a batch of pre-assembled problems resident in HBM.  Algorithmic bytes per instance:
    read 8 (nv + nv + nc nv + nc + 2 nv), write 8 nv + 4."""
    import ctypes as C
    cfg, nv, nc, alg = QP_SHAPES[args.shape]
    B = args.batch or {"c2": 65536, "c3": 16384, "c4": 32768, "c5full": 8192}[args.shape]
    first, count = grp.shard(B)
    q = qp_problem(cfg, count, dev, first)
    assert alg == 8 * (nv + nv + nc * nv + nc + 2 * nv) + 8 * nv + 4
    solver = capi.default_solver(lanes_per_qp=args.lanes)
    if args.polish >= 0:
        solver.polish = args.polish
    sol = torch.zeros((nv, B), dtype=torch.float64, device=dev)
    status = torch.zeros(B, dtype=torch.int32, device=dev)
    iters = torch.zeros(B, dtype=torch.int32, device=dev)
    bep = None
    if q["be"] is not None:
        arr = (C.c_uint8 * nc)(*[int(v) for v in q["be"]])
        bep = C.cast(arr, C.c_void_p)
    lib = capi.load()
    stream = torch.cuda.Stream(device=dev)  # a side stream: its launches can be captured into a graph
    stream.wait_stream(torch.cuda.current_stream())
    ptr = lambda t: C.c_void_p(t.data_ptr())
    call = (dev.index, C.byref(solver), C.c_int64(B), C.c_int64(q["c"].stride(0)), nv, nc, ptr(q["Hd"]), ptr(q["c"]),
            ptr(q["A"]), ptr(q["b"]), ptr(q["lb"]), ptr(q["ub"]), bep, ptr(sol), ptr(status), ptr(iters),
            C.c_void_p(stream.cuda_stream))

    def step():
        r = lib.asif_hip_qp_solve_batch(*call)
        if r != 0:
            capi.check(r)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    # --graph (see --help): the 2 x 4 shape's step is bound by the host's launch rate like the explicit filter's
    if args.graph < 0:
        args.graph = 1 if args.shape == "c2" and args.lanes != 64 else 0
    graph = None
    if args.graph:
        try:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=stream):
                for _ in range(args.steps):
                    step()
        except Exception as e:
            print(f"bench.py: graph capture failed ({e}); direct launches", file=sys.stderr)
            graph = None
        torch.cuda.synchronize()
    graph_exec = upload_graph(graph, stream)
    grp.barrier()
    hip = C.CDLL("libamdhip64.so")
    hip.hipEventRecord.argtypes = [C.c_void_p, C.c_void_p]
    hip.hipEventElapsedTime.argtypes = [C.POINTER(C.c_float), C.c_void_p, C.c_void_p]
    ev = [C.c_void_p(), C.c_void_p()]
    for e in ev:
        assert hip.hipEventCreate(C.byref(e)) == 0
    sptr = C.c_void_p(stream.cuda_stream)
    t0 = time.perf_counter()
    hip.hipEventRecord(ev[0], sptr)
    if graph is not None:
        launch_graph(graph, graph_exec, stream)
    else:
        for _ in range(args.steps):
            step()
    hip.hipEventRecord(ev[1], sptr)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    grp.barrier()
    elapsed = grp.max_over_ranks(t1 - t0)
    ms = C.c_float()
    assert hip.hipEventElapsedTime(C.byref(ms), ev[0], ev[1]) == 0
    step_ms = ms.value / max(args.steps, 1)
    st = status.cpu().numpy()
    it = iters.cpu().numpy()
    kernel = ("wave-per-QP, plain ADMM (admm_wave.hpp), then qp_inv.hpp / qp_lds.hpp for what it leaves at max_iter"
              if (args.lanes == 64 and solver.polish == 0)
              else "two QPs per wave, K_J^-1 by rank-one steps in registers (qp_inv.hpp)"
              if ((args.lanes == 64 or nv > 3) and nv <= 32 and nc <= 32 and os.environ.get("ASIF_HIP_QP_INV") != "0")
              else "wave-per-QP, factor in LDS (qp_lds.hpp)" if (args.lanes == 64 or nv > 3)
              else "in-register (gi_small.hpp + admm_small.hpp)")
    tag = f"qp_{args.shape}" + ("_wave" if args.lanes == 64 else "") + (f"_polish{solver.polish}" if args.polish >= 0 else "")
    traffic, sq, src = profile_numbers(tag)
    achieved = alg * B / (step_ms * 1e-3) / 1e9
    out = {"metric": "QP solves/sec (asif_hip_qp_solve_batch, pre-assembled)", "value": B * grp.world * args.steps / elapsed,
           "unit": "QP solves/s", "n_gpus": grp.world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": elapsed / max(args.steps, 1) * 1e3, "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": f"pre-assembled QPs of {WORKLOAD[cfg]}: nv={nv} nc={nc}", "batch_per_gpu": B,
                      "launch": (f"one HIP graph of {args.steps} launches, replayed once" if graph is not None
                                 else f"{args.steps} direct launches"),
                      "kernel": kernel, "solver": {"polish": solver.polish, "iterations_mean": float(it.mean()),
                                                   "iterations_max": int(it.max())},
                      "status_histogram": {str(int(k)): int(v) for k, v in zip(*np.unique(st, return_counts=True))}},
           "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": src,
                        "traffic_frac": (traffic / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
                        "kernel_avg_us": step_ms * 1e3, "algorithmic_bytes_per_launch": alg * B,
                        "algorithmic_bytes_per_instance": alg, "valu": valu_roofline(sq, step_ms, tag)},
           "cpu_baseline": None}
    if grp.rank == 0 and grp.world == 1 and not args.no_cpu_baseline:
        # the oracle as checker + baseline: exact optimum of the same problems (u, delta), OSQP-style ADMM timed
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib as O
        O.build()
        m = min(B, 2048)
        host = {k: np.ascontiguousarray(q[k][:, :m].cpu().numpy().T) for k in ("Hd", "c", "A", "b", "lb", "ub")}
        be = np.array(q["be"], dtype=np.uint8) if q["be"] is not None else None
        t = time.perf_counter()
        O.qp_solve_batch(nv, nc, host["Hd"], host["c"], host["A"], host["b"], host["lb"], host["ub"], be, O.SOLVER_ADMM)
        per = (time.perf_counter() - t) / m
        out["cpu_baseline"] = {"value": 1.0 / per, "unit": "QP solves/s", "cores": 1, "kind": "port",
                               "cpu_model": cpu_model(), "single_thread_value": 1.0 / per,
                               "sample": f"{m} of the same problems, OSQP-style ADMM restatement (eps 1e-3, max_iter 2000, "
                                         f"cold start) on one host thread"}
        m = min(B, 65536)  # parity: a larger head of the batch than the timed CPU sample
        host = {k: np.ascontiguousarray(q[k][:, :m].cpu().numpy().T) for k in ("Hd", "c", "A", "b", "lb", "ub")}
        if nv <= 3:
            ex, stex, _ = O.qp_solve_batch(nv, nc, host["Hd"], host["c"], host["A"], host["b"], host["lb"], host["ub"], be,
                                           O.SOLVER_EXACT)
            ok = stex == 1
            g = sol[:, :m].cpu().numpy().T
            out["parity"] = {"status_mismatches": int(((st[:m] == 1) != ok).sum()),
                             "max_abs_err_vs_exact": float(np.abs(g[ok] - ex[ok]).max()), "checked_instances": m}
        else:
            model, variant = O.CONFIGS[cfg]
            ua, rl, rc = O.filter_batch(model, variant, O.default_options(model, variant),
                                        np.ascontiguousarray(q["x"][:, :m].T), np.ascontiguousarray(q["udes"][:, :m].T),
                                        nthreads=host_cores())
            g = sol[:, :m].cpu().numpy().T
            out["parity"] = {"status_mismatches": int((st[:m] != rc).sum()),
                             "max_abs_err_vs_exact": float(np.abs(g[:, 0] - ua[:, 0]).max()), "checked_instances": m}
    if grp.rank == 0:
        print(json.dumps(out))
    grp.close()


def bench_filter(args, grp, dev, cfg, headline):
    """One filter config on this rank's slice of its seeded stream: W warm-up steps, then K timed steps bracketed by a
    barrier + synchronize on both sides, max of the elapsed time over ranks.  Returns the JSON record (rank 0 prints).
    headline: the full record of the default workload (extra legs: graph replay beside direct launches, PCIe-inclusive
    rate); otherwise the compact per-config record of the default line's `configs`."""
    import ctypes as C
    solver = capi.default_solver(lanes_per_qp=args.lanes, presolve=args.presolve)
    if args.polish >= 0:
        solver.polish = args.polish
    if cfg == REALIZABLE_CFG:
        default_b = 65536
        kernel = workloads.load_kernel(args.kernel)
        flt = capi.RealizableFilter(kernel, solver=solver, device=dev.index)
    elif cfg == ROBUST_DATA_CFG:
        default_b = 8192
        halfplanes = workloads.load_halfplanes(args.halfplanes)
        flt = capi.RobustDataFilter(halfplanes, solver=solver, device=dev.index)
    elif cfg == IMPLICIT_RB_CFG:
        model, variant, default_b = capi.CONFIGS[cfg]
        flt = capi.Filter(model, variant, options=rb_options(capi, model, variant), solver=solver, device=dev.index)
        flt.set_learning(workloads.make_learning())
    else:
        model, variant, default_b = capi.CONFIGS[cfg]
        flt = capi.Filter(model, variant, solver=solver, device=dev.index)
    B = (args.batch if headline else 0) or default_b
    d = flt.dims
    first, count = grp.shard(B)  # this rank's own slice of the seeded instance stream
    if cfg == REALIZABLE_CFG:
        x, udes = workloads.make_batch_realizable(kernel, count, first=first)
    elif cfg == ROBUST_DATA_CFG:
        x, udes = workloads.make_batch_robust_data(halfplanes, count, first=first)
    else:
        x, udes = workloads.make_batch(cfg, count, first=first)
    if args.state_scale != 1.0:
        x = x * args.state_scale  # NOT the SURVEY 8(d) workload any more; recorded in config.state_scale
    tx = torch.from_numpy(x).to(dev)
    tu = torch.from_numpy(udes).to(dev)
    uact = torch.zeros((d.nu, B), dtype=torch.float64, device=dev)
    relax = torch.zeros((d.nrelax, B), dtype=torch.float64, device=dev)
    rc = torch.zeros(B, dtype=torch.int32, device=dev)

    # Hot loop: one C-ABI call per step with every argument marshalled once (a Python-side wrapper per
    # call costs more than the explicit filter's kernel).  The launches go to a side stream so that they can be captured.
    fn = flt.lib.asif_hip_filter_batch
    stream = torch.cuda.Stream(device=dev)
    stream.wait_stream(torch.cuda.current_stream())
    call_args = (flt.handle, B, tx.stride(0), C.c_void_p(tx.data_ptr()), C.c_void_p(tu.data_ptr()),
                 C.c_void_p(uact.data_ptr()), C.c_void_p(relax.data_ptr()), C.c_void_p(rc.data_ptr()), None,
                 C.c_void_p(stream.cuda_stream))

    def step():
        r = fn(*call_args)
        if r != 0:
            capi.check(r)

    # K steps from native code (asif_amd/host/bench_loop.c: the same K C-ABI calls in a C loop): the library's callers
    # are C++ programs, and a ctypes call costs ~1.5 us of argument conversion per step
    native = native_loop()
    if native is not None:
        native.bench_loop_filter.argtypes = [C.c_void_p, C.c_int32] + list(fn.argtypes)
        fn_addr = C.cast(fn, C.c_void_p)

    def steps(k):
        if native is not None:
            r = native.bench_loop_filter(fn_addr, k, *call_args)
            if r != 0:
                capi.check(r)
        else:
            for _ in range(k):
                step()

    hip = C.CDLL("libamdhip64.so")
    hip.hipEventRecord.argtypes = [C.c_void_p, C.c_void_p]
    hip.hipEventElapsedTime.argtypes = [C.POINTER(C.c_float), C.c_void_p, C.c_void_p]
    sptr = C.c_void_p(stream.cuda_stream)

    def timed(graph_mode):
        """W warm-up steps, then exactly K steps between barrier + synchronize.  graph_mode: the K launches captured
        into one HIP graph BEFORE the timed region and replayed once inside it -- same K launches, same arguments,
        same stream order.  Returns (max-over-ranks wall seconds, device ms per step from HIP events on the launch
        stream, how the steps were launched)."""
        for _ in range(args.warmup):
            step()
        torch.cuda.synchronize()
        graph = None
        if graph_mode:
            try:
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, stream=stream):
                    steps(args.steps)
            except Exception as e:  # capture not possible (e.g. a first call still has to size staging buffers)
                print(f"bench.py: graph capture failed ({e}); direct launches", file=sys.stderr)
                graph = None
            torch.cuda.synchronize()
        graph_exec = upload_graph(graph, stream)
        # One event pair brackets the K back-to-back launches (raw hipEvent* through ctypes: a torch.cuda.Event.record
        # costs more host time than the explicit filter's kernel runs; an event between every two kernels would put a
        # barrier packet into the queue).  Average launch duration = device time between the two events / K.
        ev = []
        for _ in range(2):
            e = C.c_void_p()
            assert hip.hipEventCreate(C.byref(e)) == 0
            ev.append(e)
        grp.barrier()
        t0 = time.perf_counter()
        hip.hipEventRecord(ev[0], sptr)
        if graph is not None:
            launch_graph(graph, graph_exec, stream)
        else:
            steps(args.steps)
        hip.hipEventRecord(ev[1], sptr)
        torch.cuda.synchronize()  # this rank's K steps are done: stop its clock, then meet the others
        t1 = time.perf_counter()
        grp.barrier()
        elapsed = grp.max_over_ranks(t1 - t0)
        ms = C.c_float()
        assert hip.hipEventElapsedTime(C.byref(ms), ev[0], ev[1]) == 0
        for e in ev:
            hip.hipEventDestroy(e)
        how = (f"one HIP graph of {args.steps} filter launches, replayed once" if graph is not None
               else f"{args.steps} direct launches" + (" from a native loop" if native is not None else " from Python"))
        return elapsed, ms.value / max(args.steps, 1), how

    # --graph: -1 (default) = direct launches are the line's `value` (what a caller of the C ABI pays per call, and the
    # figure comparable across rounds); for the launch-bound explicit filter the same K steps are ALSO timed as one
    # replayed graph and reported beside it (value_graph_replay), never instead of it.
    launch_bound = cfg in (2, 11)
    use_graph = args.graph == 1
    elapsed, step_ms, how = timed(use_graph)
    other = None
    if headline and launch_bound and args.graph < 0:
        other = timed(True)

    # one extra, untimed call with the diagnostics buffer: ADMM iterations used per instance (last diag row)
    diag = torch.zeros((d.ndiag, B), dtype=torch.float64, device=dev)
    capi.check(fn(*(call_args[:8] + (C.c_void_p(diag.data_ptr()), call_args[9]))))
    torch.cuda.synchronize()
    iters_host = diag[d.ndiag - 1].cpu().numpy()
    rc_host = rc.cpu().numpy()
    solved_mask = np.isin(rc_host, (1, 2, -1))
    solved = int(solved_mask.sum())  # instances whose QP was solved (or found infeasible)
    # "QP solves" counts the instances whose filter() went through a QP (rc 1, 2, -1); TB instances that never reach the
    # backup set (rc -3) are filtered without one and count only in instances_per_s
    solved_all = grp.sum_over_ranks(solved)
    value = solved_all * args.steps / elapsed
    instances_per_s = B * grp.world * args.steps / elapsed
    alg_bytes = ALG_BYTES[cfg] * B
    achieved = alg_bytes / (step_ms * 1e-3) / 1e9 if step_ms > 0 else 0.0
    # HBM traffic and instruction counts per step: rocprofv3 PMC passes cannot run inside this process; the numbers
    # come from the committed summaries of the same command (profiles/<round>/) and are labelled as such
    tag = f"c{cfg}"
    if cfg == 2 and args.presolve:
        tag, default_b = "c2pre", 4194304  # the closed-form path was profiled at the HBM-bound batch size
    if args.polish >= 0 and args.polish != 2:
        tag += f"polish{args.polish}"
    traffic, sq, traffic_src = (None, None, None)
    if B == default_b and (cfg != REALIZABLE_CFG or args.kernel == "100Hz"):
        traffic, sq, traffic_src = profile_numbers(tag)
    direct = rocprof_direct(tag, DOMINANT_KERNEL.get(cfg, "asif")) if B == default_b else None
    working_set = (8 * (d.nx + 2 * d.nu + d.nrelax) + 4) * B
    note = ("HBM is the mandated roofline and not the one that binds: 44-60 algorithmic bytes against 10^3-10^6 FP64 "
            "operations per instance; see `valu` and DESIGN.md")
    if working_set < 256 * 2 ** 20:
        note += (f".  The {working_set / 1e6:.1f} MB this launch reads and writes are the SAME buffers at every timed step: "
                 "Infinity-Cache resident (256 MiB), so `achieved` here is a cache-resident rate, not an HBM rate; the "
                 "HBM figure of this kernel is the 16 M-instance run (738 MB working set) in DESIGN.md / profiles/")
    out = {
        "metric": "QP solves/sec (batched filter())",
        "value": value,
        "unit": "QP solves/s",
        "n_gpus": grp.world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / max(args.steps, 1) * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": WORKLOAD[cfg], "batch_per_gpu": B, "sharding": "instances, no collective",
                   "lanes_per_qp": args.lanes or "default",
                   "presolve": args.presolve,
                   "state_scale": args.state_scale,
                   "launch": how,
                   # how the QPs were decided: polish 2 runs the in-register dual active-set stage first, then OSQP-style
                   # ADMM iterations (with an active-set finish at every check) for what it leaves undecided
                   "solver": {"polish": solver.polish,
                              "admm_iterations_mean": float(iters_host[solved_mask].mean()) if solved else 0.0,
                              "admm_iterations_max": float(iters_host[solved_mask].max()) if solved else 0.0,
                              "decided_before_first_iteration": float((iters_host[solved_mask] == 0).mean()) if solved else 0.0},
                   "rc_histogram": {str(int(k)): int(v) for k, v in zip(*np.unique(rc_host, return_counts=True))},
                   "qp_solved_fraction": solved / B,
                   "value_definition": "kernel-only, inputs and outputs resident in HBM; instances whose filter() solved a "
                                       "QP (rc 1, 2, -1) per second; transfer-inclusive rate: pcie_inclusive"},
        "instances_per_s": instances_per_s,
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                     # the same fraction on the bytes the PMC pass saw move (where they exceed the algorithmic ones --
                     # the explicit filter at 16 M: old outputs of failed instances re-read -- this is what the memory system did)
                     "traffic_frac": (traffic / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
                     "kernel_avg_us": step_ms * 1e3, "algorithmic_bytes_per_launch": alg_bytes,
                     # the same fraction from the committed profile instead of this run's event pair: algorithmic
                     # bytes / AverageNs of the dominant kernel in the rocprofv3 kernel trace of the direct-launch leg
                     # alone (tools/prof_direct.sh) / peak -- recomputable from the named CSV
                     "frac_rocprof": (alg_bytes / direct[0] / HBM_PEAK_GBS) if direct else None,
                     "rocprof_kernel_avg_ns": direct[0] if direct else None,
                     "rocprof_calls": direct[1] if direct else None,
                     "rocprof_source": direct[2] if direct else None,
                     "valu": valu_roofline(sq, step_ms, tag), "note": note},
    }
    if other is not None:
        e2, ms2, how2 = other
        out["value_graph_replay"] = {"value": solved_all * args.steps / e2, "ms_per_step": e2 / max(args.steps, 1) * 1e3,
                                     "kernel_avg_us": ms2 * 1e3, "launch": how2,
                                     "note": "the same K steps captured into one HIP graph before the timed region and "
                                             "replayed once inside it; `value` is the direct-launch figure"}
    if grp.rank == 0 and grp.world == 1 and not args.no_cpu_baseline:
        if cfg in (REALIZABLE_CFG, ROBUST_DATA_CFG):
            base, parity = cpu_baseline_handle(cfg, args.kernel if cfg == REALIZABLE_CFG else args.halfplanes,
                                               uact.cpu().numpy(), rc_host, x, udes)
        else:
            base, parity = cpu_baseline(cfg, uact.cpu().numpy(), rc_host, x, udes,
                                        seconds=15.0 if headline else args.cpu_seconds)
        out["cpu_baseline"] = base
        out["parity"] = parity
    elif grp.rank == 0:
        out["cpu_baseline"] = None
    if headline and grp.rank == 0 and grp.world == 1 and not args.no_pcie:
        out["pcie_inclusive"] = pcie_inclusive(flt, x, udes, rc_host, uact)
    flt.close()
    return out


def compact(rec):
    """The per-config entry of the default line's `configs`: what VERDICT r2 asks for, nothing repeated from the top."""
    r = rec["roofline"]
    v = r.get("valu") or {}
    return {"workload": rec["config"]["workload"], "batch_per_gpu": rec["config"]["batch_per_gpu"],
            "value": rec["value"], "unit": rec["unit"], "instances_per_s": rec["instances_per_s"],
            "ms_per_step": rec["ms_per_step"], "launch": rec["config"]["launch"],
            "rc_histogram": rec["config"]["rc_histogram"], "solver": rec["config"]["solver"],
            "roofline": {"bound": "hbm", "achieved": r["achieved"], "peak": r["peak"], "unit": r["unit"], "frac": r["frac"],
                         "traffic": r["traffic"], "traffic_source": r["traffic_source"], "kernel_avg_us": r["kernel_avg_us"],
                         "algorithmic_bytes_per_launch": r["algorithmic_bytes_per_launch"],
                         "frac_rocprof": r.get("frac_rocprof"), "rocprof_kernel_avg_ns": r.get("rocprof_kernel_avg_ns"),
                         "rocprof_source": r.get("rocprof_source"),
                         "valu": v.get("frac"), "valu_plus_salu": v.get("frac_valu_plus_salu"),
                         "valu_counters_source": v.get("counters_source")},
            "parity": rec.get("parity"), "cpu_baseline": rec.get("cpu_baseline")}


def c1_single_agent(steps=2500):
    """BASELINE.json configs[0] ("single agent, CPU path, plumbing, no GPU"): the closed loop of
    examples/DoubleIntegrator.cpp:63-116 on the C++ mirror (asif_amd/host/double_integrator), timed per filter() call by
    the program itself, twice: with `QPSOLVER::HOST` (ASIF::QPWrapperHost: the product's dual active-set method on the
    calling thread -- the config as written, and the headline of this entry) and with the default solver (QPWrapperHip:
    one QP per control step on the GPU, launch + synchronisation).  Beside them the oracle's OSQP-style restatement on
    one host core for the same closed-loop states (what `CPU OSQP path` stands for here: OSQP itself is not in the
    image).  Both runs are checked step by step against the exact optimum on the states they were in."""
    import subprocess
    exe = os.path.join(ROOT, "asif_amd", "host", "double_integrator")
    try:
        if not os.path.exists(exe):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "asif_amd", "host"), "-s"])
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib as O
        O.build()
        model, variant = O.CONFIGS[2]
        o = O.default_options(model, variant)
        ud = np.ones((steps, 1))

        def run(solver):
            out = subprocess.run([exe, "--steps", str(steps), "--time", "--solver", solver], capture_output=True, text=True,
                                 timeout=300)
            if out.returncode != 0:
                return {"error": out.stderr[-300:]}, None
            t = json.loads([l for l in out.stderr.strip().split("\n") if l.startswith("{")][-1])
            rows = np.array([[float(v) for v in line.split(",")] for line in out.stdout.strip().split("\n")[1:]])
            xprev = np.vstack([[0.0, 0.0], rows[:-1, 1:3]])
            ue, _, rce = O.filter_batch(model, variant, o, xprev, ud, O.SOLVER_EXACT)
            return {"us_per_filter_median": t["median_us"], "us_per_filter_mean": t["mean_us"], "us_per_filter_p99": t["p99_us"],
                    "rc_mismatches_vs_exact": int((rows[:, 6].astype(int) != rce).sum()),
                    "max_abs_u_err_vs_exact": float(np.abs(rows[:, 4] - ue[:, 0]).max())}, (rows, xprev)

        host, hr = run("host")
        hip, _ = run("hip")
        if hr is None:
            return {"error": host.get("error")}
        rows, xprev = hr
        t0 = time.perf_counter()
        ua, _, rca = O.filter_batch(model, variant, o, xprev, ud, O.SOLVER_ADMM, None, 1)
        cpu_us = (time.perf_counter() - t0) / steps * 1e6
        return {"workload": "C1 examples/DoubleIntegrator.cpp closed loop, single agent, first 2500 steps "
                            "(asif_amd/host/double_integrator: ASIF::ASIF::filter, QPSOLVER::HOST = ASIF::QPWrapperHost, the "
                            "product's dual active-set method on the calling thread, no GPU)",
                "us_per_filter_single_agent": host["us_per_filter_median"], "us_per_filter_mean": host["us_per_filter_mean"],
                "us_per_filter_p99": host["us_per_filter_p99"], "steps": steps,
                "rc_mismatches_vs_exact": host["rc_mismatches_vs_exact"],
                "max_abs_u_err_vs_exact": host["max_abs_u_err_vs_exact"],
                "through_the_gpu_solver": dict(hip, solver="QPWrapperHip (default QPSOLVER): one launch + synchronisation per "
                                                          "control step"),
                "cpu_us": cpu_us, "cpu_kind": "port: oracle's OSQP-style ADMM restatement, one host core, cold start per step",
                "osqp_like_envelope": float(np.abs(rows[:, 4] - ua[:, 0])[rca == 1].max())}
    except Exception as e:  # the line must still be printed
        return {"error": repr(e)[:300]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="all",
                    help="all (default): C2 is the line's value, C3 / C4 (one GPU's share) / C5 ride in `configs`, C1 in `c1`; "
                         "2..12: one filter config; qp: pre-assembled QPs (see --shape)")
    ap.add_argument("--shape", default="c2", choices=sorted(QP_SHAPES), help="--config qp: which problems")
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--lanes", type=int, default=0)
    ap.add_argument("--kernel", default="100Hz", help="config 6: RealizableKernelData_<name> polytope")
    ap.add_argument("--halfplanes", default="70-135kg", help="config 7: KernelData_<name> half-plane set")
    ap.add_argument("--polish", type=int, default=-1,
                    help="asif_hip_solver.polish: 0 pure ADMM, 1 active-set finish at the checks, 2 (library default) also "
                         "once before the first iteration")
    ap.add_argument("--presolve", type=int, default=0,
                    help="asif_hip_solver.presolve (config 2: pinned variables eliminated and a one-variable closed form under polish "
                         "0 / 1 too -- the default mode, polish 2, takes that light kernel anyway; default 0)")
    ap.add_argument("--graph", type=int, default=-1,
                    help="0: K direct launches; 1: the K timed launches are captured into one HIP graph (before the timed "
                         "region) and replayed once inside it -- same K launches, same arguments, same stream order; "
                         "-1 (default): direct launches are `value`, and for the launch-bound explicit filter (config 2, "
                         "11: a launch costs the host 2.8-3.6 us, the kernel runs in about 2) the graph replay of the same "
                         "K steps is timed too and reported beside it as value_graph_replay")
    ap.add_argument("--cpu-seconds", type=float, default=5.0,
                    help="CPU work spent on the cpu_baseline sample of each config in `configs` (the headline's is 15 s)")
    ap.add_argument("--state-scale", type=float, default=1.0,
                    help="multiplies the seeded states (1.0 = the SURVEY 8(d) workload; e.g. 0.5 puts every C2 state inside the "
                         "safe set: no failed instance, the kernel's memory behaviour without the untouched-slot reads)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pcie", action="store_true", help="skip the host-buffer (PCIe-inclusive) leg")
    ap.add_argument("--no-c1", action="store_true", help="skip the single-agent C1 leg of --config all")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args.gpus)  # does not return
    if int(os.environ.get("WORLD_SIZE", "1")) != max(args.gpus, 1):
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={os.environ.get('WORLD_SIZE', '1')}: start it as "
                 f"`python -m torch.distributed.run --nnodes=1 --nproc-per-node {args.gpus} --master-addr 127.0.0.1 "
                 f"bench.py --gpus {args.gpus} ...` or without a launcher")
    # ASIF_BENCH_REHEARSAL=1: rehearse the N>1 code path on a one-GPU box (every rank on cuda:0); never set by
    # the driver.
    rehearsal = os.environ.get("ASIF_BENCH_REHEARSAL") == "1"
    grp = dist.Group(backend="gloo")  # barrier + max of the elapsed time over TCP on the host: no RCCL anywhere
    dev = torch.device("cuda", grp.local_rank if (grp.world > 1 and not rehearsal) else 0)
    torch.cuda.set_device(dev)
    if args.config == "qp":
        return bench_qp(args, grp, dev)

    if args.config != "all":
        out = bench_filter(args, grp, dev, int(args.config), headline=True)
    else:
        # The line: BASELINE.json configs[1] (C2) is `value`; configs[2..4] -- the pendulum-sized and the 8-GPU one, each
        # rank its own block (C4: 32 768 per GPU = 262 144 over 8) -- are timed the same way, K steps each between
        # barriers, and ride in `configs`; configs[0] (single agent, C++ class) in `c1` (rank 0 of a 1-GPU run).
        out = bench_filter(args, grp, dev, 2, headline=True)
        out["configs"] = {}
        for cfg in (3, 4, 5):
            out["configs"][f"c{cfg}"] = compact(bench_filter(args, grp, dev, cfg, headline=False))
        if grp.rank == 0 and grp.world == 1 and not args.no_c1 and not args.no_cpu_baseline:
            out["c1"] = c1_single_agent()
    if grp.rank == 0:
        print(json.dumps(out))
    grp.close()


if __name__ == "__main__":
    main()
