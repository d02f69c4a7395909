#!/bin/bash
# Bench value of several builds of the library in ONE GPU-box session (names as given to tools/build_variants.sh).
# bash tools/gpu_variants.sh <tag> name1 name2 ...     (CHECK=1: also run the parity subset + stress per variant)
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
for v in "$@"; do
  export CTCDECODE_AMD_LIB=$GRAFT_REPO_ROOT/ctcdecode_amd/_lib/var_$v.so
  if [ -n "$CHECK" ]; then
    ( timeout 300 python -m pytest tests/test_gpu_decode.py -x -q -k "not reference and not 1000 and not wide" ) > "$OUT/pytest_$v.log" 2>&1; echo "$v pytest rc=$? $(tail -1 $OUT/pytest_$v.log)"
    ( timeout 200 python tests/sweeps/gpu_stress.py --n 200 ) > "$OUT/stress_$v.log" 2>&1; echo "$v stress rc=$? $(tail -1 $OUT/stress_$v.log | cut -c1-60)"
  fi
  for rep in 1 2; do
    timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > "$OUT/bench_$v.json" 2> "$OUT/bench_$v.err"
    python -c "import json;d=json.load(open('$OUT/bench_$v.json'));print('$v: %.0f utt/s  kernel %.3f ms'%(d['value'],d['kernel_ms']))"
  done
done
