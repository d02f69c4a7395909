#!/bin/bash
# bash tools/gpu_round3b.sh <tag> [variants...]: GPU tests, wide-beam phase profile + PMC traffic, kernel variants (bench + timeline each)
TAG=${1:-r03b}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
if [ -z "$SKIP_TESTS" ]; then
( time timeout 900 python -m pytest tests -m gpu -x -q --durations=8 ) > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?"; tail -14 "$OUT/pytest_gpu.log"
fi
if [ -z "$SKIP_WIDE" ]; then
timeout 200 python tools/phase_profile.py --beam 500 --frames 600 --batch 256 --out "$OUT/phase_k500.json" > "$OUT/phase_k500.log" 2>&1; echo "phase k500 rc=$?"; python -c "
import json;d=json.load(open('$OUT/phase_k500.json'));print('k500 us/frame',d['us_per_frame']);[print('  %5.1f%%  %s'%(v,k)) for k,v in sorted(d['phases_percent'].items(),key=lambda kv:-kv[1])]"
timeout 200 python tools/bench_configs.py --only 2 --out "$OUT/cfg2.json" > "$OUT/cfg2.log" 2>&1; tail -1 "$OUT/cfg2.log"
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OUT/pmc_cfg_$c" -o p -- python "$GRAFT_REPO_ROOT/tools/bench_configs.py" --only 2 --reps 1 > "$OUT/pmc_cfg_$c.log" 2>&1; echo "pmc cfg $c rc=$?"
done
cd "$GRAFT_REPO_ROOT"
python3 - <<PY
import csv,glob,collections
for c in ["FETCH_SIZE","WRITE_SIZE"]:
    for f in glob.glob("$OUT/pmc_cfg_%s/**/*counter_collection.csv"%c, recursive=True):
        acc=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            acc[(r.get('Kernel_Name','')[:70], r['Counter_Name'])].append(float(r['Counter_Value']))
        for k,v in acc.items(): print(k,'n=%d'%len(v),'mean=%.6g'%(sum(v)/len(v)))
PY
fi
for v in "$@"; do
  export CTCDECODE_AMD_LIB=$GRAFT_REPO_ROOT/ctcdecode_amd/_lib/var_$v.so
  for rep in 1 2; do
    timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > "$OUT/bench_$v.json" 2> "$OUT/bench_$v.err"
    python -c "import json;d=json.load(open('$OUT/bench_$v.json'));print('$v: %.0f utt/s  kernel %.3f ms'%(d['value'],d['kernel_ms']))"
  done
  timeout 200 python tools/barrier_timeline.py --out "$OUT/timeline_$v.json" > "$OUT/timeline_$v.log" 2>&1
  python - <<PY
import json
d=json.load(open("$OUT/timeline_$v.json"))
print("$v timeline clocks/frame", d["clocks_per_frame"], " ".join("%d:%d"%(r["stamp"],r["max"]) for r in d["rows"]))
PY
done
