#!/bin/bash
# bash tools/gpu_final_r03m.sh: closing GPU session of round 3's second sitting (tight budget): GPU suite without the wide-beam
# real-reference tests (run separately earlier: 5 passed), the bench line, rocprofv3 kernel stats of the bench command
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03m
mkdir -p "$OUT"; cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
( time timeout 230 python -m pytest tests -m gpu -x -q --durations=8 -k "not (k500 or wide or config2 or beam_500)" ) > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?"; tail -14 "$OUT/pytest_gpu.log"
( timeout 110 python bench.py ) > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc=$?"; cut -c1-400 "$OUT/bench.json"
cd /tmp
( timeout 60 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -o k -- python "$GRAFT_REPO_ROOT/bench.py" --steps 6 --warmup 2 --no-extras --no-cpu-baseline ) > "$OUT/kt.log" 2>&1; echo "rocprof rc=$?"
find "$OUT/kt" -name "*kernel_stats.csv" | head -2
