#!/usr/bin/env python3
"""Regenerates the measured tables of DESIGN.md section 6 (between the TABLE-* markers) from the committed records under
profiles/<round>/, newest round first (a record not taken again this round keeps its last one, and says which round it
is from).   python tools/make_design_table.py [--write]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROUNDS = ("r04", "r03")


def j(name):
    for rnd in ROUNDS:
        try:
            d = json.load(open(os.path.join(ROOT, "profiles", rnd, name)))
            d["_round"] = rnd
            return d
        except Exception:
            continue
    return None


def fmt_rate(v):
    for unit, k in (("G", 1e9), ("M", 1e6), ("k", 1e3)):
        if v >= k:
            return f"{v / k:.3g} {unit}"
    return f"{v:.3g}"


NAMES = {2: "C2 DI explicit", 3: "C3 pendulum implicit", 4: "C4 segway TB (one GPU's share)", 5: "C5 robust pendulum",
         6: "C6 realizable DI", 7: "C7 robust DI, shipped data", 8: "C8 pendulum TB", 9: "C9 DI implicit",
         10: "C10 pendulum, `ASIFimplicitRB`", 11: "C11 two-input model, class `ASIF`", 12: "C12 DI TB"}


def config_table():
    out = ["| config | batch / GPU | step | QP solves/s | HBM traffic per step (PMC) vs algorithmic | HBM `frac` | `roofline.valu` (+ SALU) | max\\|u−u_ref\\| | rc mismatches | OSQP-like envelope | CPU baseline (oracle, threads; 1 thread) |",
           "|---|---|---|---|---|---|---|---|---|---|---|"]
    for c in range(2, 13):
        b = j(f"c{c}_bench.json")
        if not b:
            continue
        r, cb, pa = b["roofline"], b["cpu_baseline"], b["parity"]
        v = r.get("valu") or {}
        step = b["ms_per_step"] * 1e3
        stepf = f"{step:.2f} µs" if step < 1000 else f"{step / 1e3:.3f} ms"
        tr = f"{r['traffic'] / 1e6:.3g} MB vs {r['algorithmic_bytes_per_launch'] / 1e6:.3g} MB" if r.get("traffic") else "—"
        vv = f"{v['frac']:.2f} ({v['frac_valu_plus_salu']:.2f})" if v.get("frac") else "—"
        env = (pa.get("osqp_like_envelope") or {}).get("max_abs_u_gpu_minus_u_admm_eps1e-3")
        envf = f"{env:.2g}" if env is not None else "—"
        extra = ""
        if abs(b["instances_per_s"] - b["value"]) > 1e-6 * b["value"]:
            extra = f" ({fmt_rate(b['instances_per_s'])} instances)"
        out.append(f"| {NAMES[c]}{'' if b['_round'] == ROUNDS[0] else ' (' + b['_round'] + ')'} | {b['config']['batch_per_gpu']:,} | **{stepf}** | {fmt_rate(b['value'])}{extra} | {tr} | "
                   f"{r['frac']:.2g} | {vv} | {pa['max_abs_u_err_vs_exact']:.1e} | {pa['rc_mismatches']} / {pa['checked_instances']:,} | "
                   f"{envf} | {fmt_rate(cb['value'])}/s on {cb['cores']}; {fmt_rate(cb['single_thread_value'])}/s |")
    return "\n".join(out)


def qp_table():
    out = ["| shape | batch | step | QP/s | kernel | algorithmic B / QP | HBM `frac` | PMC traffic per step | `roofline.valu` (+ SALU) | parity |",
           "|---|---|---|---|---|---|---|---|---|---|"]
    for s, label in (("c2", "2×4 (C2)"), ("c3", "3×41 (C3)"), ("c4", "2×18 (C4)"), ("c5full", "18×12 (C5, lifted)"),
                     ("c5full_wave_per_qp", "18×12, one wave per QP (`ASIF_HIP_QP_INV=0`)"),
                     ("c5full_wave_polish0", "18×12, plain ADMM wave + second pass (`polish = 0`)")):
        b = j(f"qp_{s}_bench.json")
        if not b:
            continue
        r = b["roofline"]
        v = r.get("valu") or {}
        pa = b.get("parity") or {}
        step = b["ms_per_step"] * 1e3
        stepf = f"{step:.2f} µs" if step < 1000 else f"{step / 1e3:.3f} ms"
        tr = f"{r['traffic'] / 1e6:.3g} MB vs {r['algorithmic_bytes_per_launch'] / 1e6:.3g} MB" if r.get("traffic") else "—"
        vv = f"{v['frac']:.2f} ({v['frac_valu_plus_salu']:.2f})" if v.get("frac") else "—"
        out.append(f"| {label}{'' if b['_round'] == ROUNDS[0] else ' (' + b['_round'] + ')'} | {b['config']['batch_per_gpu']:,} | {stepf} | {fmt_rate(b['value'])} | {b['config']['kernel']} | "
                   f"{r['algorithmic_bytes_per_instance']} | {r['frac']:.2g} | {tr} | {vv} | "
                   f"{pa.get('status_mismatches')} status mismatches, {pa.get('max_abs_err_vs_exact', 0):.1e} |")
    return "\n".join(out)


def default_line():
    d = j("default_driver_style_bench.json")
    if not d:
        return ""
    g = d.get("value_graph_replay") or {}
    out = [f"`python bench.py --gpus 1 --steps 20 --warmup 5` (the driver's command, `profiles/{d['_round']}/default_driver_style_bench.json`): "
           f"C2 **{fmt_rate(d['value'])} QP solves/s** ({d['ms_per_step'] * 1e3:.2f} µs per step, {d['config']['launch']}; the same 20 steps "
           f"replayed as one HIP graph: {fmt_rate(g.get('value', 0))}/s, {g.get('ms_per_step', 0) * 1e3:.2f} µs), "
           f"`roofline.frac` {d['roofline']['frac']:.3f} (kernel {d['roofline']['kernel_avg_us']:.2f} µs by the run's own event pair), "
           f"`roofline.frac_rocprof` {d['roofline'].get('frac_rocprof') or 0:.3f} (algorithmic bytes / AverageNs "
           f"{d['roofline'].get('rocprof_kernel_avg_ns') or 0:.0f} of `{d['roofline'].get('rocprof_source')}`, the direct-launch-only trace), "
           f"0 rc mismatches / 65 536."]
    for k, v in d.get("configs", {}).items():
        out.append(f"`configs.{k}`: {fmt_rate(v['value'])} QP solves/s, {v['ms_per_step'] * 1e3:.1f} µs per step, `roofline.frac` "
                   f"{v['roofline']['frac']:.2g}, traffic {((v['roofline']['traffic'] or 0) / 1e6):.3g} MB, `valu` {v['roofline']['valu'] or 0:.2f}, "
                   f"{v['parity']['rc_mismatches']} rc mismatches / {v['parity']['checked_instances']:,}, max|u−u_ref| {v['parity']['max_abs_u_err_vs_exact']:.1e}, "
                   f"CPU {fmt_rate(v['cpu_baseline']['value'])}/s on {v['cpu_baseline']['cores']} threads.")
    c1 = d.get("c1") or {}
    if "us_per_filter_single_agent" in c1:
        g = c1.get("through_the_gpu_solver") or {}
        out.append(f"`c1`: {c1['us_per_filter_single_agent']:.2f} µs per single-agent `filter()` with `QPSOLVER::HOST` (`QPWrapperHost`: the product's "
                   f"dual active-set method on the calling thread; median of 2 500 closed-loop steps, p99 {c1['us_per_filter_p99']:.2f} µs), "
                   f"{g.get('us_per_filter_median', 0):.1f} µs through the GPU solver (`QPWrapperHip`, p99 {g.get('us_per_filter_p99', 0):.1f}), against "
                   f"{c1['cpu_us']:.1f} µs for the oracle's OSQP-style restatement on one host core; "
                   f"{c1['rc_mismatches_vs_exact']} / {g.get('rc_mismatches_vs_exact')} rc mismatches, max|u−u_ref| {c1['max_abs_u_err_vs_exact']:.1e}, OSQP-like envelope {c1['osqp_like_envelope']:.2g}.")
    return "\n\n".join(out)


def main():
    blocks = {"CONFIGS": config_table(), "QP": qp_table(), "DEFAULT": default_line()}
    if "--write" not in sys.argv:
        for k, v in blocks.items():
            print(f"==== {k}\n{v}\n")
        return
    p = os.path.join(ROOT, "DESIGN.md")
    s = open(p).read()
    for k, v in blocks.items():
        a, b = f"<!-- TABLE-{k}-BEGIN -->", f"<!-- TABLE-{k}-END -->"
        i, e = s.index(a) + len(a), s.index(b)
        s = s[:i] + "\n" + v + "\n" + s[e:]
    open(p, "w").write(s)


if __name__ == "__main__":
    main()
