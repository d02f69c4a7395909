#!/usr/bin/env python3
"""Turn one gpurun_out/<tag>/ session (tools/gpu_round.sh) into the small files committed under profiles/:
  profiles/<name>_kernel_stats.csv|md   rocprofv3 --kernel-trace --stats per-kernel summary
  profiles/<name>_pmc.json              FETCH_SIZE / WRITE_SIZE / SQ counters of the decode kernel, per launch
  profiles/traffic_latest.json          {"hbm_bytes_per_launch": ...} read by bench.py
  profiles/<name>_bench.json, _phase.json
Usage: summarize_round.py gpurun_out/<tag> <name>"""
import collections
import csv
import glob
import json
import os
import shutil
import sys


def main():
    src, name = sys.argv[1], sys.argv[2]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    dst = os.path.join(root, "profiles")
    os.makedirs(dst, exist_ok=True)
    # kernel stats from the kernel trace
    rows = collections.defaultdict(list)
    for f in glob.glob(os.path.join(src, "prof", "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    total = sum(sum(v) for v in rows.values()) or 1
    stats = sorted(rows.items(), key=lambda kv: -sum(kv[1]))
    with open(os.path.join(dst, name + "_kernel_stats.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "Percentage"])
        for k, v in stats:
            w.writerow([k, len(v), sum(v), "%.1f" % (sum(v) / len(v)), min(v), max(v), "%.3f" % (100.0 * sum(v) / total)])
    bench = {}
    bj = os.path.join(src, "bench.json")
    if os.path.exists(bj) and os.path.getsize(bj):
        bench = json.load(open(bj))
        shutil.copy(bj, os.path.join(dst, name + "_bench.json"))
    with open(os.path.join(dst, name + "_kernel_stats.md"), "w") as f:
        f.write("# %s: rocprofv3 --kernel-trace --stats -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline\n\n" % name)
        f.write("One MI355X, B=256, T=1000, V=29, beam=100.  bench.py's HIP-event duration of the decode kernel in the same session: "
                "%s ms (value %s utt/s).\n\n" % (bench.get("kernel_ms"), bench.get("value")))
        f.write("| kernel | calls | avg (us) | min (us) | max (us) | % of GPU time |\n|---|---|---|---|---|---|\n")
        for k, v in stats[:8]:
            f.write("| `%s` | %d | %.1f | %.1f | %.1f | %.2f |\n" % (k[:100], len(v), sum(v) / len(v) / 1e3, min(v) / 1e3, max(v) / 1e3, 100.0 * sum(v) / total))
    # PMC
    pmc = {}
    for d in ["pmc_FETCH_SIZE", "pmc_WRITE_SIZE", "pmc_sq"]:
        for f in glob.glob(os.path.join(src, d, "**", "*counter_collection.csv"), recursive=True):
            acc = collections.defaultdict(list)
            for r in csv.DictReader(open(f)):
                if "ctc_beam_decode" in r.get("Kernel_Name", ""):
                    acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            for k, v in acc.items():
                pmc[k] = sum(v) / len(v)
    if pmc:
        out = {"per_launch_mean": pmc, "note": "decode kernel only; FETCH_SIZE/WRITE_SIZE in KiB as reported by rocprofv3 (gfx950: FETCH_SIZE can "
               "under-report wide coalesced reads by 2x -- MI355X_MICROARCH.md; this kernel's reads are 4-16 B per lane, uncalibrated)"}
        if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
            out["hbm_bytes_per_launch"] = int((pmc["FETCH_SIZE"] + pmc["WRITE_SIZE"]) * 1024)
            json.dump({"hbm_bytes_per_launch": out["hbm_bytes_per_launch"], "source": name + "_pmc.json"}, open(os.path.join(dst, "traffic_latest.json"), "w"))
        json.dump(out, open(os.path.join(dst, name + "_pmc.json"), "w"), indent=1)
    pj = os.path.join(src, "phase.json")
    if os.path.exists(pj):
        shutil.copy(pj, os.path.join(dst, name + "_phase.json"))
    print(open(os.path.join(dst, name + "_kernel_stats.md")).read())
    print(json.dumps(pmc, indent=1))


if __name__ == "__main__":
    main()
