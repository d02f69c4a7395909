#!/bin/bash
# One kernel-tuning iteration on the GPU box: core parity tests, a short stress, the bench value, the barrier timeline.
# bash tools/gpu_iter.sh <tag> [pytest -k expression]
TAG=${1:-it}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
K=${2:-"not reference and not 1000 and not wide"}
( timeout 400 python -m pytest tests/test_gpu_decode.py tests/test_gpu_lm.py -x -q -k "$K" ) > "$OUT/pytest.log" 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 "$OUT/pytest.log"
[ $rc -ne 0 ] && exit 1
( timeout 200 python tests/sweeps/gpu_stress.py --n 300 ) > "$OUT/stress.log" 2>&1; echo "stress rc=$?"; tail -2 "$OUT/stress.log"
timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc=$?"
python -c "import json;d=json.load(open('$OUT/bench.json'));print('bench: %.0f utt/s  kernel %.3f ms'%(d['value'],d['kernel_ms']))"
timeout 200 python tools/barrier_timeline.py --out "$OUT/timeline.json" > "$OUT/timeline.log" 2>&1; echo "timeline rc=$?"; cut -c1-80 "$OUT/timeline.log" | tail -20
