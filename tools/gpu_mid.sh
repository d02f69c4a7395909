#!/bin/bash
# bash tools/gpu_mid.sh <tag>: a mid-size GPU check of the product library between kernel changes: fixture parity (both tiers), short
# random sweeps (ordinary + degenerate, both tiers), LM-tier kernel times
TAG=${1:-mid}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p "$OUT"; cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
( timeout 300 python -m pytest tests/test_gpu_lm.py tests/test_gpu_decode.py -x -q -k "fixtures or golden or degenerate or streaming or host_path" ) > "$OUT/pytest.log" 2>&1; echo "pytest rc=$? $(tail -1 $OUT/pytest.log)"
( timeout 120 python tests/sweeps/gpu_stress.py --n ${N:-150} --seed 411 ) > "$OUT/stress.log" 2>&1; echo "stress rc=$? $(tail -1 $OUT/stress.log | cut -c1-90)"
( timeout 120 python tests/sweeps/gpu_stress.py --n ${N:-150} --seed 412 --degenerate ) > "$OUT/stress_deg.log" 2>&1; echo "stress degenerate rc=$? $(tail -1 $OUT/stress_deg.log | cut -c1-90)"
( timeout 120 python tests/sweeps/gpu_stress_lm.py --n ${N:-150} --seed 413 ) > "$OUT/stress_lm.log" 2>&1; echo "stress lm rc=$? $(tail -1 $OUT/stress_lm.log | cut -c1-90)"
( timeout 120 python tests/sweeps/gpu_stress_lm.py --n 80 --seed 414 --degenerate ) > "$OUT/stress_lm_deg.log" 2>&1; echo "stress lm degenerate rc=$? $(tail -1 $OUT/stress_lm_deg.log | cut -c1-90)"
( timeout 120 python tools/lm_probe.py ) 2>&1 | grep -v amdgpu
