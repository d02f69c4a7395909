#!/usr/bin/env python3
"""Turns an evidence session's raw outputs (tools/gpu_profile.sh <tag> -> gpurun_out/<tag>: rocprofv3 CSVs, bench line, timelines) into the small files
committed under profiles/:  <tag>_kernel_stats.{csv,md} (bench.py), <tag>_other_configs_kernel_stats.md,
<tag>_extras_kernel_stats.md, <tag>_pmc.json (FETCH_SIZE / WRITE_SIZE per launch and kernel, with the calibration of the
counters against a copy of known size), <tag>_timeline.json, <tag>_bench.json; and points profiles/traffic_latest.json
at the new numbers.      python tools/summarize_profiles.py r04a"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def stats_md(src, dst_prefix, note):
    rows = list(csv.DictReader(open(src)))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    shutil.copy(src, dst_prefix + ".csv")
    with open(dst_prefix + ".md", "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats summary\n\n%s\n\n| kernel | calls | avg (us) | min (us) | max (us) | %% of GPU time |\n|---|---|---|---|---|---|\n" % note)
        for r in rows:
            f.write("| `%s` | %s | %.1f | %.1f | %.1f | %s |\n" % (r["Name"][:110], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3,
                                                               float(r["MaxNs"]) / 1e3, r["Percentage"]))


def counters(d):
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            acc[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
    return acc


def main():
    tag = sys.argv[1]
    src = os.path.join(ROOT, "gpurun_out", tag)
    dst = os.path.join(ROOT, "profiles", tag)
    import subprocess
    head = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True, cwd=ROOT).stdout.strip()
    stats_md(os.path.join(src, "prof", "trace_kernel_stats.csv"), dst + "_kernel_stats", "`python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-pmc` (configs[1]: B=256, T=1000, V=29, beam 100); tree " + head)
    stats_md(os.path.join(src, "prof_cfg", "trace_kernel_stats.csv"), dst + "_other_configs_kernel_stats", "`python tools/bench_configs.py --only 234 --reps 1` (configs[2] per-GPU shape: beam 500, T 2000; configs[3]: V=10000 pruned; configs[4] shape without LM)")
    stats_md(os.path.join(src, "prof_extras", "trace_kernel_stats.csv"), dst + "_extras_kernel_stats", "`python tools/profile_extras.py`: LM instantiation (configs[4] per-GPU shape, test.arpa), two-workgroups-per-CU build (512 utterances), the raw-logit kernels at B=64, T=500, V=10000 (prune_logits_wg_kernel, log_softmax_rows_wg_kernel, the one-wave pair they replace), expand_compact_kernel (256 x 100 x 1000), 1 GiB device copy (counter calibration)")
    pmc = {"unit": "KiB per launch as rocprofv3 reports FETCH_SIZE / WRITE_SIZE (separate --pmc passes)", "kernels": {}}
    cal = {}
    for group, sub in (("bench", "pmc_%s"), ("other_configs", "pmc_cfg_%s"), ("extras", "pmc_extras_%s")):
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            for (k, cn), v in counters(os.path.join(src, sub % c)).items():
                mean = sum(v) / len(v)
                if "__amd_rocclr_copyBuffer" in k and max(v) > 400000:  # the 1 GiB copies
                    big = [x for x in v if x > 400000]
                    cal[c] = sum(big) / len(big)
                if mean < 1000 or "rocclr" in k or "at::native" in k:
                    continue
                pmc["kernels"].setdefault(group + ": " + k[:120], {})[c + "_KiB"] = round(mean, 1)
    GiB_KiB = float(1 << 20)
    pmc["calibration"] = {"what": "torch dst.copy_(src) of 2^28 float32 = exactly 1 GiB read + 1 GiB written (__amd_rocclr_copyBuffer, 16 B per lane)",
                          "FETCH_SIZE_KiB_reported": cal.get("FETCH_SIZE"), "WRITE_SIZE_KiB_reported": cal.get("WRITE_SIZE"),
                          "fetch_factor": round(GiB_KiB / cal["FETCH_SIZE"], 4) if cal.get("FETCH_SIZE") else None,
                          "write_factor": round(GiB_KiB / cal["WRITE_SIZE"], 4) if cal.get("WRITE_SIZE") else None,
                          "conclusion": "WRITE_SIZE is exact; FETCH_SIZE reports half of the bytes read by 16-B-per-lane loads on gfx950 (as MI355X_MICROARCH.md says): kernels that read with 128-bit loads (prune_rows_wg_kernel, expand_compact_kernel's label reads) need x2, kernels that read 4-8 B per lane (the decode kernels' node walks, log_softmax's row reads) do not"}
    head = [v for k, v in pmc["kernels"].items() if k.startswith("bench: ") and "ctc_beam_decode_kernel" in k]
    if head:
        h = head[0]
        total = int((h.get("FETCH_SIZE_KiB", 0) + h.get("WRITE_SIZE_KiB", 0)) * 1024)
        pmc["hbm_bytes_per_launch"] = total
        lat = {"hbm_bytes_per_launch": total, "source": tag + "_pmc.json"}
        comp = [v for k, v in pmc["kernels"].items() if k.startswith("extras: ") and ("ctc_beam_decode_kernel<0, 0, 1, false, 1024, false, false>" in k or "ctc_beam_decode_kernel<0, 0, 1, false, 1024, 0, false>" in k)]
        if comp:  # the same kernel handing its results over in compact form (tools/profile_extras.py "expand")
            lat["compact_hbm_bytes_per_launch"] = int((comp[0].get("FETCH_SIZE_KiB", 0) + comp[0].get("WRITE_SIZE_KiB", 0)) * 1024)
            pmc["compact_hbm_bytes_per_launch"] = lat["compact_hbm_bytes_per_launch"]
        wide = [v for k, v in pmc["kernels"].items() if k.startswith("other_configs: ") and "ctc_beam_decode_kernel<0, 1, 0, false" in k]
        if wide:
            lat["wide_beam_hbm_bytes_per_launch"] = int((wide[0].get("FETCH_SIZE_KiB", 0) + wide[0].get("WRITE_SIZE_KiB", 0)) * 1024)
        prune = [v for k, v in pmc["kernels"].items() if k.startswith("other_configs: ") and "prune_rows_wg_kernel" in k]
        if prune:  # 128-bit loads: FETCH_SIZE x2 (calibration)
            lat["prune_hbm_bytes_per_launch"] = int((2 * prune[0].get("FETCH_SIZE_KiB", 0) + prune[0].get("WRITE_SIZE_KiB", 0)) * 1024)
        json.dump(lat, open(os.path.join(ROOT, "profiles", "traffic_latest.json"), "w"))
    json.dump(pmc, open(dst + "_pmc.json", "w"), indent=1)
    # instruction counters of the headline kernel (per wave: counter / SQ_WAVES)
    ins = counters(os.path.join(src, "pmc_insts"))
    per = {}
    for (k, cn), v in ins.items():
        if "ctc_beam_decode_kernel" in k:
            per.setdefault(k[:120], {})[cn] = sum(v) / len(v)
    if per:
        json.dump({"unit": "per launch (rocprofv3 --pmc, one pass)", "kernels": per}, open(dst + "_pmc_insts.json", "w"), indent=1)
    for name in ("timeline.json", "timeline_blank.json", "timeline_lm.json", "bench.json", "parity_sweep.json"):
        if os.path.exists(os.path.join(src, name)):
            shutil.copy(os.path.join(src, name), dst + "_" + name)
    if os.path.exists(os.path.join(src, "prof_extras.log")):
        last = [ln for ln in open(os.path.join(src, "prof_extras.log")) if ln.startswith("{")]
        if last:
            json.dump(json.loads(last[-1]), open(dst + "_extras_timings.json", "w"), indent=1)
    if os.path.exists(os.path.join(src, "pytest_gpu.log")):
        open(dst + "_pytest_gpu_tail.txt", "w").write("".join(open(os.path.join(src, "pytest_gpu.log")).readlines()[-16:]))
    print(json.dumps(pmc, indent=1)[:3000])


if __name__ == "__main__":
    main()
