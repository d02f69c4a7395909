#!/bin/bash
# lean GPU check of the product library after a kernel change: fixtures (both tiers, all builds), edge cases, short random sweeps
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out/chk
( timeout 120 python -m pytest tests/test_gpu_lm.py tests/test_gpu_decode.py -x -q -k "fixtures or edge_cases or golden or degenerate_inputs or capability" ) 2>&1 | tail -1
( timeout 60 python tests/sweeps/gpu_stress.py --n 300 --seed 601 ) 2>&1 | tail -1 | cut -c1-40
( timeout 60 python tests/sweeps/gpu_stress.py --n 150 --seed 602 --degenerate ) 2>&1 | tail -1 | cut -c1-40
( timeout 60 python tests/sweeps/gpu_stress_lm.py --n 200 --seed 603 ) 2>&1 | tail -1 | cut -c1-40
( timeout 60 python tests/sweeps/gpu_stress_lm.py --n 80 --seed 604 --degenerate ) 2>&1 | tail -1 | cut -c1-40
