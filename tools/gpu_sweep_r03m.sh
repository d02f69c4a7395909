#!/bin/bash
# closing random sweeps of the final tree (both tiers, ordinary + degenerate inputs) against the oracle
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03m; mkdir -p "$OUT"; cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
( timeout 100 python tests/sweeps/gpu_stress.py --n 500 --seed 501 ) > "$OUT/stress.log" 2>&1; echo "stress rc=$? $(tail -1 $OUT/stress.log | cut -c1-60)"
( timeout 60 python tests/sweeps/gpu_stress.py --n 300 --seed 502 --degenerate ) > "$OUT/stress_deg.log" 2>&1; echo "stress degenerate rc=$? $(tail -1 $OUT/stress_deg.log | cut -c1-60)"
( timeout 60 python tests/sweeps/gpu_stress_lm.py --n 300 --seed 503 ) > "$OUT/stress_lm.log" 2>&1; echo "stress lm rc=$? $(tail -1 $OUT/stress_lm.log | cut -c1-60)"
( timeout 40 python tests/sweeps/gpu_stress_lm.py --n 150 --seed 504 --degenerate ) > "$OUT/stress_lm_deg.log" 2>&1; echo "stress lm degenerate rc=$? $(tail -1 $OUT/stress_lm_deg.log | cut -c1-60)"
