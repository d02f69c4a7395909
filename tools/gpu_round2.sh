#!/bin/bash
# Round-2 GPU-box session: smoke, parity tests, bench line, rocprofv3 kernel trace of bench.py AND of the other
# configurations (prune kernel, wide beam), PMC traffic passes for both, barrier timeline.
# bash tools/gpu_round2.sh <tag> [skip_tests]
TAG=${1:-r02a}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
echo "host: $(nproc) cores" | tee "$OUT/host.txt"
( time timeout 180 python -c "import __graft_entry__ as g; g.smoke()" ) > "$OUT/smoke.log" 2>&1; echo "smoke rc=$?" | tee -a "$OUT/host.txt"; tail -2 "$OUT/smoke.log"
if [ -z "$2" ]; then
  ( time timeout 900 python -m pytest tests -m gpu -x -q ) > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?" | tee -a "$OUT/host.txt"; tail -5 "$OUT/pytest_gpu.log"
fi
( time timeout 600 python bench.py --steps 10 --warmup 3 ) > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc=$?"; cat "$OUT/bench.json"; tail -3 "$OUT/bench.err"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -o trace -- python "$GRAFT_REPO_ROOT/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-extras > "$OUT/prof.log" 2>&1; echo "rocprof stats rc=$?"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_cfg" -o trace -- python "$GRAFT_REPO_ROOT/tools/bench_configs.py" --only 234 > "$OUT/prof_cfg.log" 2>&1; echo "rocprof cfg rc=$?"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OUT/pmc_$c" -o p -- python "$GRAFT_REPO_ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-extras > "$OUT/pmc_$c.log" 2>&1; echo "pmc $c rc=$?"
  timeout 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OUT/pmc_cfg_$c" -o p -- python "$GRAFT_REPO_ROOT/tools/bench_configs.py" --only 23 --reps 1 > "$OUT/pmc_cfg_$c.log" 2>&1; echo "pmc cfg $c rc=$?"
done
cd "$GRAFT_REPO_ROOT"
timeout 200 python tools/barrier_timeline.py --out "$OUT/timeline.json" > "$OUT/timeline.log" 2>&1; echo "timeline rc=$?"
find "$OUT" -name '*.csv' | head -30
