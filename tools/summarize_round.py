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
    # the other configurations (tools/gpu_round2.sh: prof_cfg = kernel trace, pmc_cfg_* = traffic of bench_configs.py)
    rows2 = collections.defaultdict(list)
    for f in glob.glob(os.path.join(src, "prof_cfg", "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows2[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    if rows2:
        total2 = sum(sum(v) for v in rows2.values()) or 1
        st2 = sorted(rows2.items(), key=lambda kv: -sum(kv[1]))
        with open(os.path.join(dst, name + "_other_configs_kernel_stats.csv"), "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "Percentage"])
            for k, v in st2:
                w.writerow([k, len(v), sum(v), "%.1f" % (sum(v) / len(v)), min(v), max(v), "%.3f" % (100.0 * sum(v) / total2)])
        with open(os.path.join(dst, name + "_other_configs_kernel_stats.md"), "w") as f:
            f.write("# %s: rocprofv3 --kernel-trace --stats -- python tools/bench_configs.py --only 234 (configs[2] per-GPU shape = wide-beam "
                    "run-time-layout kernel; configs[3] = prune_rows_wg_kernel + prune_resolve_kernel + pruned decode kernel; configs[4] shape without the LM)\n\n" % name)
            f.write("| kernel | calls | avg (us) | min (us) | max (us) | % of GPU time |\n|---|---|---|---|---|---|\n")
            for k, v in st2[:10]:
                f.write("| `%s` | %d | %.1f | %.1f | %.1f | %.2f |\n" % (k[:110], len(v), sum(v) / len(v) / 1e3, min(v) / 1e3, max(v) / 1e3, 100.0 * sum(v) / total2))
    cfgp = collections.defaultdict(dict)
    for d in ["pmc_cfg_FETCH_SIZE", "pmc_cfg_WRITE_SIZE"]:
        for f in glob.glob(os.path.join(src, d, "**", "*counter_collection.csv"), recursive=True):
            acc = collections.defaultdict(list)
            for r in csv.DictReader(open(f)):
                kn = r.get("Kernel_Name", "")
                if "ctc_beam_decode" in kn or "prune_" in kn:
                    acc[(kn[:100], r["Counter_Name"])].append(float(r["Counter_Value"]))
            for (kn, cn), v in acc.items():
                cfgp[kn][cn + "_KiB_per_launch_mean"] = sum(v) / len(v)
    if cfgp and os.path.exists(os.path.join(dst, name + "_pmc.json")):
        out = json.load(open(os.path.join(dst, name + "_pmc.json")))
        out["other_configs"] = cfgp
        out["other_configs_note"] = ("prune_rows_wg_kernel reads the rows with 16-B-per-lane loads: FETCH_SIZE under-counts those by 2x on gfx950 "
                                     "(MI355X_MICROARCH.md)")
        json.dump(out, open(os.path.join(dst, name + "_pmc.json"), "w"), indent=1)
    tj = os.path.join(src, "timeline.json")
    if os.path.exists(tj):
        shutil.copy(tj, os.path.join(dst, name + "_timeline.json"))
    pj = os.path.join(src, "phase.json")
    if os.path.exists(pj):
        shutil.copy(pj, os.path.join(dst, name + "_phase.json"))
    print(open(os.path.join(dst, name + "_kernel_stats.md")).read())
    print(json.dumps(pmc, indent=1))


if __name__ == "__main__":
    main()
