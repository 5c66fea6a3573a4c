#!/usr/bin/env python3
"""Turn gpurun_out/prof_<tag>/ (made by tools/prof_cfg.sh) into the committed summaries under profiles/<round>/:
   <tag>_kernel_stats.csv   per-kernel calls / total / avg / min / max ns (rocprofv3 --kernel-trace)
   <tag>_pmc_summary.json   HBM bytes per launch per kernel: FETCH_SIZE x2 + WRITE_SIZE, KB units
                            (MI355X_MICROARCH.md, HBM / rocprofv3 section: gfx950 FETCH_SIZE reports half of a
                            coalesced streaming read; WRITE_SIZE exact)
   python tools/summarize_prof.py r01 c2 c3 ..."""
import collections
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def pmc(tag, name):
    files = glob.glob(os.path.join(ROOT, "gpurun_out", f"prof_{tag}", f"pmc_{name}", "**", "*counter_collection.csv"),
                      recursive=True)
    agg = collections.defaultdict(list)
    if not files:
        return agg
    for r in csv.DictReader(open(max(files, key=os.path.getmtime))):
        agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return agg


def pmc_multi(tag, name):
    """several counters in one pass: {kernel: {counter: mean per launch}}, launches per kernel"""
    files = glob.glob(os.path.join(ROOT, "gpurun_out", f"prof_{tag}", f"pmc_{name}", "**", "*counter_collection.csv"),
                      recursive=True)
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    if not files:
        return {}
    for r in csv.DictReader(open(max(files, key=os.path.getmtime))):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in agg.items()}


def sq_summary(tag, out, calls_per_kernel):
    """<tag>_sq_summary.json: SQ counters per launch per kernel, and per bench step (kernels weighted by how often
    one step launches them, from the kernel trace of the same command)."""
    sq = pmc_multi(tag, "sq")
    sq2 = pmc_multi(tag, "sq2")
    kernels = {}
    for k in sq:
        if "asif" not in k:
            continue
        kernels[k] = dict(sq[k])
        kernels[k].update(sq2.get(k, {}))
    if not kernels:
        return
    steps = max(calls_per_kernel.values()) if calls_per_kernel else 1
    valu = waves = salu = 0.0
    for k, c in kernels.items():
        per_step = 1.0
        for name, calls in calls_per_kernel.items():
            if name.split("(")[0].strip() == k.split("(")[0].strip():
                per_step = calls / steps
        c["launches_per_step"] = per_step
        valu += c.get("SQ_INSTS_VALU", 0.0) * per_step
        waves += c.get("SQ_WAVES", 0.0) * per_step
        salu += c.get("SQ_INSTS_SALU", 0.0) * per_step
    with open(os.path.join(out, f"{tag}_sq_summary.json"), "w") as f:
        json.dump({"command": "tools/prof_cfg.sh (rocprofv3 --pmc SQ_*, two passes of <= 6 counters, no trace domain)",
                   "unit": "wave-level instruction counts / SQ cycles summed over the chip, mean per launch",
                   "kernels": kernels, "insts_valu_per_step": valu, "insts_salu_per_step": salu, "waves_per_step": waves,
                   "insts_valu_per_wave": valu / waves if waves else None}, f, indent=1)
    print(tag, "SQ_INSTS_VALU per step", round(valu), "waves", round(waves), "per wave", round(valu / waves) if waves else None)


def main():
    rnd = sys.argv[1]
    out = os.path.join(ROOT, "profiles", rnd)
    os.makedirs(out, exist_ok=True)
    for tag in sys.argv[2:]:
        db = os.path.join(ROOT, "gpurun_out", f"prof_{tag}", "trace", "t_results.db")
        with open(os.path.join(out, f"{tag}_kernel_stats.csv"), "w") as f:
            subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "rocprof_db_stats.py"), db], stdout=f)
        calls = {}
        for r in csv.DictReader(open(os.path.join(out, f"{tag}_kernel_stats.csv"))):
            if "asif" in r["Name"]:
                calls[r["Name"]] = int(r["Calls"])
        demangled = {}
        for name, c in calls.items():
            try:
                d = subprocess.check_output(["c++filt", name.replace(".kd", "")], text=True).strip()
            except Exception:
                d = name
            demangled[d] = c
        sq_summary(tag, out, demangled)
        fetch, write = pmc(tag, "fetch"), pmc(tag, "write")
        kernels = {}
        for k in fetch:
            if "asif" not in k or k not in write:
                continue
            fk, wk = sum(fetch[k]) / len(fetch[k]), sum(write[k]) / len(write[k])
            kernels[k] = {"FETCH_SIZE_KB_per_launch_raw": fk, "WRITE_SIZE_KB_per_launch_raw": wk,
                          "launches": len(fetch[k]), "traffic_bytes_per_launch": (2.0 * fk + wk) * 1024.0}
        steps = max(demangled.values()) if demangled else 1
        for k, v in kernels.items():
            v["launches_per_step"] = 1.0
            for name, c in demangled.items():
                if name.split("(")[0].strip() == k.split("(")[0].strip():
                    v["launches_per_step"] = c / steps
        total = sum(v["traffic_bytes_per_launch"] * v["launches_per_step"] for v in kernels.values())
        with open(os.path.join(out, f"{tag}_pmc_summary.json"), "w") as f:
            json.dump({"command": "tools/prof_cfg.sh (rocprofv3 --pmc FETCH_SIZE | --pmc WRITE_SIZE, separate passes)",
                       "correction": "counter unit KB; gfx950 FETCH_SIZE x2 for coalesced streaming reads; WRITE_SIZE exact",
                       "kernels": kernels, "traffic_bytes_per_step": total}, f, indent=1)
        print(tag, {k[:60]: round(v["traffic_bytes_per_launch"]) for k, v in kernels.items()}, "total", round(total))


if __name__ == "__main__":
    main()
