#!/bin/bash
# Round-3 GPU-box session: smoke, parity tests, degenerate-input stress, bench line.   bash tools/gpu_round3.sh <tag> [quick]
TAG=${1:-r03a}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
echo "host: $(nproc) cores" | tee "$OUT/host.txt"
( time timeout 180 python -c "import __graft_entry__ as g; g.smoke()" ) > "$OUT/smoke.log" 2>&1; echo "smoke rc=$?" | tee -a "$OUT/host.txt"; tail -2 "$OUT/smoke.log"
( time timeout 1200 python -m pytest tests -m gpu -x -q --durations=12 ) > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?" | tee -a "$OUT/host.txt"; tail -22 "$OUT/pytest_gpu.log"
if [ -z "$2" ]; then
  ( timeout 300 python tests/sweeps/gpu_stress.py --n 300 --seed 31 --degenerate ) > "$OUT/stress_degenerate.log" 2>&1; echo "stress degenerate rc=$? $(tail -1 $OUT/stress_degenerate.log | cut -c1-120)"
  ( timeout 300 python tests/sweeps/gpu_stress_lm.py --n 200 --seed 32 --degenerate ) > "$OUT/stress_lm_degenerate.log" 2>&1; echo "stress lm degenerate rc=$? $(tail -1 $OUT/stress_lm_degenerate.log | cut -c1-120)"
  ( timeout 300 python tests/sweeps/gpu_stress.py --n 300 --seed 33 ) > "$OUT/stress.log" 2>&1; echo "stress rc=$? $(tail -1 $OUT/stress.log | cut -c1-120)"
fi
( time timeout 600 python bench.py --steps 10 --warmup 3 ) > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc=$?"; cat "$OUT/bench.json"; tail -3 "$OUT/bench.err"
