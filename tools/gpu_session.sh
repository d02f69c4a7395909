#!/bin/bash
# One GPU-box session of a kernel-tuning round (replaces the per-session scripts of earlier rounds):
#   bash tools/gpu_session.sh <tag> [check] [variants "v1 v2 ..."] [timeline "v ..."] [kinds "randn blank peaky"]
# check: fixtures + edge cases + short random sweeps of the product library; variants: interleaved kernel times of
# ctcdecode_amd/_lib/var_<v>.so (tools/build_variants.sh) through tools/raw_multi.py; timeline: barrier timeline of those builds.
TAG=$1; shift
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; OUT=gpurun_out/$TAG; mkdir -p $OUT
while [ $# -gt 0 ]; do
  case $1 in
    check)
      ( timeout ${CHECK_TIMEOUT:-300} python -m pytest tests/test_gpu_decode.py -x -q -k "${CHECK_K:-fixtures or edge_cases or golden or degenerate or capability or stream or host_path or stress}" ) 2>&1 | tail -3
      ( timeout 90 python tests/sweeps/gpu_stress.py --n ${STRESS_N:-400} --seed 701 ) 2>&1 | tail -1 | cut -c1-60
      ( timeout 90 python tests/sweeps/gpu_stress.py --n 150 --seed 702 --degenerate ) 2>&1 | tail -1 | cut -c1-60
      shift;;
    variants)
      L=""; for v in $2; do L="$L ctcdecode_amd/_lib/var_$v.so"; done
      for kind in ${KINDS:-randn}; do
        echo "== $kind"; timeout 200 python tools/raw_multi.py ${SHAPE:-256 1000 29 100} ${REPS:-10} $L --kind $kind $EXTRA 2>&1 | tail -$(echo $2 | wc -w) | tee -a $OUT/variants.log
      done
      shift 2;;
    timeline)
      for v in $2; do
        CTCDECODE_AMD_LIB=$GRAFT_REPO_ROOT/ctcdecode_amd/_lib/var_$v.so timeout 200 python tools/barrier_timeline.py --repeat 3 --frames 4 --kind ${TLKIND:-randn} --out $OUT/tl_${v}_${TLKIND:-randn}.json > $OUT/tl_${v}_${TLKIND:-randn}.log 2>&1
        echo "timeline $v:"; cut -c1-150 $OUT/tl_${v}_${TLKIND:-randn}.log | tail -${TLROWS:-32}
      done
      shift 2;;
    *) echo "unknown step $1"; shift;;
  esac
done
