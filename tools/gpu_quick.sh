#!/bin/bash
# Short GPU-box session: parity tests, phase profile, bench at a few workgroup sizes.  bash tools/gpu_quick.sh <tag>
TAG=${1:-q}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
( timeout 60 python -c "import __graft_entry__ as g; g.smoke()" ) > "$OUT/smoke.log" 2>&1 || { echo "SMOKE FAILED"; tail -5 "$OUT/smoke.log"; exit 1; }
( timeout 240 python -m pytest tests -m gpu -x -q ) > "$OUT/pytest_gpu.log" 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 "$OUT/pytest_gpu.log"; [ $rc -ne 0 ] && exit 1
for th in ${THREADS:-256 512 1024}; do
  timeout 120 python tools/phase_profile.py --threads $th --out "$OUT/phase_t$th.json" > /dev/null 2> "$OUT/phase_t$th.err"
  python - <<PY
import json
d=json.load(open("$OUT/phase_t$th.json"))
print("threads $th: kernel %.2f ms (instrumented); us/utt:" % d["kernel_ms_instrumented"], {k.split()[0]:v for k,v in d["phases_us_per_utterance"].items() if v})
PY
  timeout 120 python bench.py --steps 3 --warmup 1 --threads $th --no-cpu-baseline > "$OUT/bench_t$th.json" 2> "$OUT/bench_t$th.err"
  python -c "import json;d=json.load(open('$OUT/bench_t$th.json'));print('  bench: %.0f utt/s  kernel %.2f ms'%(d['value'],d['kernel_ms']))"
done
