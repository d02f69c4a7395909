#!/bin/bash
# One GPU-box session: smoke, parity tests, bench, rocprofv3 kernel trace.  Usage (from the repo root on the box):
#   bash tools/gpu_round.sh [tag]      -> everything under gpurun_out/<tag>/
TAG=${1:-run}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
echo "== host: $(nproc) cores; $(rocm-smi --showproductname 2>/dev/null | grep -m1 -i 'card series' || true)" | tee "$OUT/host.txt"
( time timeout 600 python -c "import __graft_entry__ as g; g.build(); g.smoke()" ) > "$OUT/smoke.log" 2>&1; echo "smoke rc=$?" | tee -a "$OUT/host.txt"
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?" | tee -a "$OUT/host.txt"
tail -5 "$OUT/pytest_gpu.log"
for th in 256 512 1024; do
  timeout 600 python bench.py --steps 3 --warmup 1 --threads $th --no-cpu-baseline > "$OUT/bench_t$th.json" 2> "$OUT/bench_t$th.err"; echo "bench t=$th rc=$?"; cat "$OUT/bench_t$th.json"
done
( time timeout 900 python bench.py --steps 5 --warmup 2 ) > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc=$?"; cat "$OUT/bench.json"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/prof" -o trace -- python "$GRAFT_REPO_ROOT/bench.py" --steps 5 --warmup 2 --no-cpu-baseline > "$OUT/prof.log" 2>&1; echo "rocprof rc=$?"
find "$OUT/prof" -name '*stats*' | head; 
for f in $(find "$OUT/prof" -name '*kernel_stats*.csv' | head -2); do head -8 "$f"; done
