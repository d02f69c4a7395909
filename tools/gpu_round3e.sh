#!/bin/bash
# bash tools/gpu_round3e.sh <tag>: GPU suite, bench line, rocprofv3 kernel stats + PMC traffic for the headline, the other
# configurations and the extras (LM kernel, log_softmax, expand_compact, counter calibration), barrier timeline.
TAG=${1:-r03e}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
if [ -z "$SKIP_TESTS" ]; then
( time timeout 900 python -m pytest tests -m gpu -x -q --durations=6 ) > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?"; tail -12 "$OUT/pytest_gpu.log"
( timeout 200 python tests/sweeps/gpu_stress_lm.py --n 150 --seed 61 ) > "$OUT/stress_lm.log" 2>&1; echo "stress lm rc=$? $(tail -1 $OUT/stress_lm.log | cut -c1-100)"
fi
( time timeout 900 python bench.py --steps 10 --warmup 3 ) > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc=$?"; python - <<PY
import json
d=json.load(open("$OUT/bench.json"))
print("value", d["value"], "kernel_ms", d["kernel_ms"], "e2e", d["e2e"]["ms_per_batch"], "pipelined", d["pipelined"].get("value"))
for k,v in d["other_configs"].items(): print(" ", k[:74], v.get("decode_kernel_ms"), v.get("call_ms"), v.get("error"))
PY
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -o trace -- python "$GRAFT_REPO_ROOT/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-extras > "$OUT/prof.log" 2>&1; echo "rocprof stats rc=$?"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_cfg" -o trace -- python "$GRAFT_REPO_ROOT/tools/bench_configs.py" --only 234 --reps 1 > "$OUT/prof_cfg.log" 2>&1; echo "rocprof cfg rc=$?"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_extras" -o trace -- python "$GRAFT_REPO_ROOT/tools/profile_extras.py" > "$OUT/prof_extras.log" 2>&1; echo "rocprof extras rc=$?"; tail -1 "$OUT/prof_extras.log"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OUT/pmc_$c" -o p -- python "$GRAFT_REPO_ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-extras > "$OUT/pmc_$c.log" 2>&1; echo "pmc $c rc=$?"
  timeout 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OUT/pmc_cfg_$c" -o p -- python "$GRAFT_REPO_ROOT/tools/bench_configs.py" --only 23 --reps 1 > "$OUT/pmc_cfg_$c.log" 2>&1; echo "pmc cfg $c rc=$?"
  timeout 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OUT/pmc_extras_$c" -o p -- python "$GRAFT_REPO_ROOT/tools/profile_extras.py" > "$OUT/pmc_extras_$c.log" 2>&1; echo "pmc extras $c rc=$?"
done
cd "$GRAFT_REPO_ROOT"
timeout 200 python tools/barrier_timeline.py --out "$OUT/timeline.json" > "$OUT/timeline.log" 2>&1; echo "timeline rc=$?"
timeout 200 python tools/barrier_timeline.py --lm tests/data/test.arpa --batch 128 --T 1500 --frames 3 --repeat 12 --out "$OUT/timeline_lm.json" > "$OUT/timeline_lm.log" 2>&1; echo "timeline lm rc=$?"
python3 - <<PY
import csv,glob,collections
for pat in ["pmc_*","pmc_cfg_*","pmc_extras_*"]:
  for d in sorted(glob.glob("$OUT/"+pat)):
    for f in glob.glob(d+"/**/*counter_collection.csv", recursive=True):
        acc=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            acc[(r.get('Kernel_Name','')[:60], r['Counter_Name'])].append(float(r['Counter_Value']))
        for k,v in acc.items():
            if sum(v)/len(v) > 1000: print(d.split('/')[-1], k,'n=%d'%len(v),'mean=%.6g'%(sum(v)/len(v)))
PY
