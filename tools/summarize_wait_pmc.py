#!/usr/bin/env python3
"""gpurun_out/<tag>/pmc_wait_{a,b} (tools/gpu_wait_pmc.sh) -> profiles/<tag>_pmc_wait.json: per decode kernel, the mean of every SQ
counter per launch and the shares of wave-cycles parked / stalled at issue / issuing.   python tools/summarize_wait_pmc.py r05a"""
import collections
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for sub in ("pmc_wait_a", "pmc_wait_b"):
    for f in glob.glob(os.path.join(ROOT, "gpurun_out", tag, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "ctc_beam_decode_kernel" in r["Kernel_Name"]:
                acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {"what": "rocprofv3 --pmc (counters only), python tools/pmc_kinds.py: mean per launch over the launches of each decode kernel; "
               "WAIT_ANY = wave parked (s_waitcnt / barrier), WAIT_INST_ANY = stalled at issue, ACTIVE_INST_ANY = issuing (disjoint, sum ~ WAVE_CYCLES; "
               "SQ_* cycle counters are in units of 4 clocks per the gfx9 convention)",
       "tree": subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True, cwd=ROOT).stdout.strip(), "kernels": {}}
for k, cs in acc.items():
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    d = {c: round(v, 1) for c, v in sorted(m.items())}
    d["launches"] = max(len(v) for v in cs.values())
    wc = m.get("SQ_WAVE_CYCLES")
    if wc:
        d["share_of_wave_cycles"] = {c: round(m[c] / wc, 4) for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU",
                                                                      "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS", "SQ_LDS_BANK_CONFLICT") if c in m}
    out["kernels"][k[:100]] = d
dst = os.path.join(ROOT, "profiles", tag + "_pmc_wait.json")
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out, indent=1))
