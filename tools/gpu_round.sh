#!/bin/bash
# Full GPU-box session for a round checkpoint: smoke, parity tests, bench (+cpu baseline), rocprofv3 kernel trace and
# PMC traffic passes.  bash tools/gpu_round.sh <tag>   -> gpurun_out/<tag>/ ; summaries are then copied to profiles/.
TAG=${1:-run}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
echo "host: $(nproc) cores" | tee "$OUT/host.txt"
( time timeout 120 python -c "import __graft_entry__ as g; g.build(); g.smoke()" ) > "$OUT/smoke.log" 2>&1; echo "smoke rc=$?" | tee -a "$OUT/host.txt"
( time timeout 400 python -m pytest tests -m gpu -x -q ) > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?" | tee -a "$OUT/host.txt"; tail -3 "$OUT/pytest_gpu.log"
( time timeout 400 python bench.py --steps 10 --warmup 3 ) > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc=$?"; cat "$OUT/bench.json"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -o trace -- python "$GRAFT_REPO_ROOT/bench.py" --steps 10 --warmup 3 --no-cpu-baseline > "$OUT/prof.log" 2>&1; echo "rocprof stats rc=$?"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OUT/pmc_$c" -o p -- python "$GRAFT_REPO_ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/pmc_$c.log" 2>&1; echo "pmc $c rc=$?"
done
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d "$OUT/pmc_sq" -o p -- python "$GRAFT_REPO_ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/pmc_sq.log" 2>&1; echo "pmc sq rc=$?"
cd "$GRAFT_REPO_ROOT"
timeout 120 python tools/phase_profile.py --out "$OUT/phase.json" > /dev/null 2>&1
find "$OUT" -name '*.csv' | head -20
