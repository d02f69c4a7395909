#!/bin/bash
# bash tools/gpu_round3_final.sh <tag>: the round's closing session on the GPU box -- full GPU suite, random stress (ordinary,
# degenerate, LM), the 3 x 512 full-size parity sweep against the real reference, then everything tools/gpu_round3e.sh measures
# (bench line, rocprofv3 kernel stats + PMC traffic for bench / other configs / extras, barrier timeline).
TAG=${1:-r03i}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -m gpu -x -q --durations=5 ) > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?"; tail -9 "$OUT/pytest_gpu.log"
( timeout 300 python tests/sweeps/gpu_stress.py --n 600 --seed 201 ) > "$OUT/stress.log" 2>&1; echo "stress rc=$? $(tail -1 $OUT/stress.log | cut -c1-90)"
( timeout 300 python tests/sweeps/gpu_stress.py --n 400 --seed 202 --degenerate ) > "$OUT/stress_degenerate.log" 2>&1; echo "stress degenerate rc=$? $(tail -1 $OUT/stress_degenerate.log | cut -c1-90)"
( timeout 300 python tests/sweeps/gpu_stress_lm.py --n 300 --seed 203 ) > "$OUT/stress_lm.log" 2>&1; echo "stress lm rc=$? $(tail -1 $OUT/stress_lm.log | cut -c1-90)"
( timeout 300 python tests/sweeps/gpu_stress_lm.py --n 200 --seed 204 --degenerate ) > "$OUT/stress_lm_degenerate.log" 2>&1; echo "stress lm degenerate rc=$? $(tail -1 $OUT/stress_lm_degenerate.log | cut -c1-90)"
( timeout 600 python tests/sweeps/parity_sweep.py --n ${PARITY_N:-512} --out "$OUT/parity_sweep.json" --head "$2" ) > "$OUT/parity_sweep.log" 2>&1; echo "parity sweep rc=$?"; cat "$OUT/parity_sweep.log" | cut -c1-200
SKIP_TESTS=1 bash tools/gpu_round3e.sh "$TAG"
