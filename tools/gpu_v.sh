#!/bin/bash
# bash tools/gpu_v.sh <reps> v1 v2 ...: interleaved kernel times of library variants (tools/raw_multi.py), then the barrier
# timeline of the LAST one (rows: stamp:max clocks of the slowest wave)
REPS=$1; shift
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/v
L=""; for v in "$@"; do L="$L ctcdecode_amd/_lib/var_$v.so"; done
python tools/raw_multi.py ${SHAPE:-256 1000 29 100} $REPS $L $EXTRA   # SHAPE="B T V beam"; EXTRA="--lm tests/data/test.arpa" / "--cu-sharing 1"
if [ -n "$TIMELINE" ]; then
for v in $TIMELINE; do
CTCDECODE_AMD_LIB=$GRAFT_REPO_ROOT/ctcdecode_amd/_lib/var_$v.so python tools/barrier_timeline.py --repeat 3 --frames 4 --out gpurun_out/v/tl_$v.json > gpurun_out/v/tl_$v.log 2>&1
python - <<PY
import json
d=json.load(open("gpurun_out/v/tl_$v.json"))
print("$v timeline clocks/frame", d["clocks_per_frame"], " ".join("%d:%d"%(r["stamp"],r["max"]) for r in d["rows"]))
PY
done
fi
