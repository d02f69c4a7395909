#!/bin/bash
# bash tools/gpu_round3d.sh <tag>: full GPU suite, bench line, wide-beam time / phases / PMC traffic
TAG=${1:-r03d}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -m gpu -x -q --durations=6 ) > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?"; tail -12 "$OUT/pytest_gpu.log"
( timeout 200 python tests/sweeps/gpu_stress.py --n 250 --seed 51 ) > "$OUT/stress.log" 2>&1; echo "stress rc=$? $(tail -1 $OUT/stress.log | cut -c1-100)"
( time timeout 600 python bench.py --steps 10 --warmup 3 ) > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc=$?"; python - <<PY
import json
d=json.load(open("$OUT/bench.json"))
print("value", d["value"], "kernel_ms", d["kernel_ms"], "e2e", d["e2e"]["ms_per_batch"], "pipelined", d["pipelined"])
for k,v in d["other_configs"].items(): print(" ", k[:70], v.get("decode_kernel_ms"), v.get("call_ms"))
print(d["cpu_baseline"])
PY
timeout 200 python tools/phase_profile.py --beam 500 --frames 600 --batch 256 --out "$OUT/phase_k500.json" > "$OUT/phase_k500.log" 2>&1; python -c "
import json;d=json.load(open('$OUT/phase_k500.json'));print('k500 us/frame',d['us_per_frame']);[print('  %5.1f%%  %s'%(v,k)) for k,v in sorted(d['phases_percent'].items(),key=lambda kv:-kv[1])[:9]]"
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
            if 'ctc_beam' in r.get('Kernel_Name',''): acc[(r.get('Kernel_Name','')[:70], r['Counter_Name'])].append(float(r['Counter_Value']))
        for k,v in acc.items(): print(k,'n=%d'%len(v),'mean=%.6g'%(sum(v)/len(v)))
PY
