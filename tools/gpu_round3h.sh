#!/bin/bash
# bash tools/gpu_round3h.sh <tag>: GPU suite + bench + PMC traffic (headline, compact path via profile_extras)
TAG=${1:-r03h}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -m gpu -x -q --durations=5 ) > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?"; tail -10 "$OUT/pytest_gpu.log"
( timeout 200 python tests/sweeps/gpu_stress.py --n 200 --seed 91 ) > "$OUT/stress.log" 2>&1; echo "stress rc=$? $(tail -1 $OUT/stress.log | cut -c1-80)"
( time timeout 900 python bench.py --steps 10 --warmup 3 ) > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc=$?"; python - <<PY
import json
d=json.load(open("$OUT/bench.json"))
print("value", d["value"], "kernel_ms", d["kernel_ms"], "e2e", d["e2e"]["ms_per_batch"], "pipelined", d["pipelined"].get("value"), "compact", d.get("compact_results"))
for k,v in d["other_configs"].items(): print(" ", k[:74], v.get("decode_kernel_ms"), v.get("call_ms"), v.get("error"))
PY
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OUT/pmc_$c" -o p -- python "$GRAFT_REPO_ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-extras > "$OUT/pmc_$c.log" 2>&1; echo "pmc $c rc=$?"
  timeout 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OUT/pmc_extras_$c" -o p -- python "$GRAFT_REPO_ROOT/tools/profile_extras.py" --only expand,occ2 > "$OUT/pmc_extras_$c.log" 2>&1; echo "pmc extras $c rc=$?"
done
cd "$GRAFT_REPO_ROOT"
python3 - <<PY
import csv,glob,collections
for pat in ["pmc_*_SIZE","pmc_extras_*"]:
  for d in sorted(glob.glob("$OUT/"+pat)):
    for f in glob.glob(d+"/**/*counter_collection.csv", recursive=True):
        acc=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            acc[(r.get('Kernel_Name','')[:70], r['Counter_Name'])].append(float(r['Counter_Value']))
        for k,v in acc.items():
            if sum(v)/len(v) > 1000 and 'ctc_beam' in k[0]: print(d.split('/')[-1], k,'n=%d'%len(v),'mean=%.6g'%(sum(v)/len(v)))
PY
