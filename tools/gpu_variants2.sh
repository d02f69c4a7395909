#!/bin/bash
# bash tools/gpu_variants2.sh <tag> v1 v2 ...: quick parity subset on the product library, then bench (x3) + barrier timeline per variant
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
if [ -z "$SKIP_TESTS" ]; then
( timeout 600 python -m pytest tests/test_gpu_decode.py tests/test_gpu_lm.py -x -q -k "not config3_shape and not config4" ) > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$? $(tail -1 $OUT/pytest_gpu.log)"
( timeout 200 python tests/sweeps/gpu_stress.py --n 200 --seed 71 ) > "$OUT/stress.log" 2>&1; echo "stress rc=$? $(tail -1 $OUT/stress.log | cut -c1-80)"
( timeout 200 python tests/sweeps/gpu_stress_lm.py --n 100 --seed 72 ) > "$OUT/stress_lm.log" 2>&1; echo "stress lm rc=$? $(tail -1 $OUT/stress_lm.log | cut -c1-80)"
fi
for rep in 1 2 3; do
for v in "$@"; do
  export CTCDECODE_AMD_LIB=$GRAFT_REPO_ROOT/ctcdecode_amd/_lib/var_$v.so
  timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > "$OUT/bench_$v.json" 2> "$OUT/bench_$v.err"
  python -c "import json;d=json.load(open('$OUT/bench_$v.json'));print('$v: %.0f utt/s  kernel %.3f ms'%(d['value'],d['kernel_ms']))"
done
done
for v in "$@"; do
  export CTCDECODE_AMD_LIB=$GRAFT_REPO_ROOT/ctcdecode_amd/_lib/var_$v.so
  timeout 200 python tools/barrier_timeline.py --out "$OUT/timeline_$v.json" > "$OUT/timeline_$v.log" 2>&1
  python - <<PY
import json
d=json.load(open("$OUT/timeline_$v.json"))
print("$v timeline clocks/frame", d["clocks_per_frame"], " ".join("%d:%d"%(r["stamp"],r["max"]) for r in d["rows"]))
PY
done
