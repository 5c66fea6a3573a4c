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
    for r in csv.DictReader(open(max(files, key=os.path.getmtime))):
        agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return agg


def main():
    rnd = sys.argv[1]
    out = os.path.join(ROOT, "profiles", rnd)
    os.makedirs(out, exist_ok=True)
    for tag in sys.argv[2:]:
        db = os.path.join(ROOT, "gpurun_out", f"prof_{tag}", "trace", "t_results.db")
        with open(os.path.join(out, f"{tag}_kernel_stats.csv"), "w") as f:
            subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "rocprof_db_stats.py"), db], stdout=f)
        fetch, write = pmc(tag, "fetch"), pmc(tag, "write")
        kernels = {}
        for k in fetch:
            if "asif" not in k:
                continue
            fk, wk = sum(fetch[k]) / len(fetch[k]), sum(write[k]) / len(write[k])
            kernels[k] = {"FETCH_SIZE_KB_per_launch_raw": fk, "WRITE_SIZE_KB_per_launch_raw": wk,
                          "launches": len(fetch[k]), "traffic_bytes_per_launch": (2.0 * fk + wk) * 1024.0}
        total = sum(v["traffic_bytes_per_launch"] for v in kernels.values())
        with open(os.path.join(out, f"{tag}_pmc_summary.json"), "w") as f:
            json.dump({"command": "tools/prof_cfg.sh (rocprofv3 --pmc FETCH_SIZE | --pmc WRITE_SIZE, separate passes)",
                       "correction": "counter unit KB; gfx950 FETCH_SIZE x2 for coalesced streaming reads; WRITE_SIZE exact",
                       "kernels": kernels, "traffic_bytes_per_step": total}, f, indent=1)
        print(tag, {k[:60]: round(v["traffic_bytes_per_launch"]) for k, v in kernels.items()}, "total", round(total))


if __name__ == "__main__":
    main()
