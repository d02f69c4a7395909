#!/bin/bash
# bash tools/gpu_lm_check.sh <tag> [variants...]: LM-tier GPU tests + LM stress on the product library, then tools/gpu_lm_variants.sh
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
( timeout 500 python -m pytest tests/test_gpu_lm.py -x -q ) > "$OUT/pytest_lm.log" 2>&1; echo "pytest lm rc=$? $(tail -1 $OUT/pytest_lm.log)"
( timeout 200 python tests/sweeps/gpu_stress_lm.py --n 200 --seed 301 ) > "$OUT/stress_lm.log" 2>&1; echo "stress lm rc=$? $(tail -1 $OUT/stress_lm.log | cut -c1-90)"
( timeout 200 python tests/sweeps/gpu_stress_lm.py --n 100 --seed 302 --degenerate ) > "$OUT/stress_lm_deg.log" 2>&1; echo "stress lm degenerate rc=$? $(tail -1 $OUT/stress_lm_deg.log | cut -c1-90)"
( timeout 200 python tests/sweeps/gpu_stress.py --n 200 --seed 303 ) > "$OUT/stress.log" 2>&1; echo "stress rc=$? $(tail -1 $OUT/stress.log | cut -c1-90)"
bash tools/gpu_lm_variants.sh "$TAG" "$@" > "$OUT/variants.log" 2>&1; grep -v "^ *[0-9]* \(wait\|(tick\|state\|C\|D\)" "$OUT/variants.log" | grep -v amdgpu
