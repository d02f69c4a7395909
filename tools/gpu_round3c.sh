#!/bin/bash
# bash tools/gpu_round3c.sh <tag>: parity subset, bench at B=256/512 for the product library and the 8-waves-per-SIMD build, wide-beam time
TAG=${1:-r03c}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
( time timeout 600 python -m pytest tests/test_gpu_decode.py tests/test_gpu_lm.py -x -q -k "not config3_shape and not north_star and not config4" ) > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?"; tail -4 "$OUT/pytest_gpu.log"
( timeout 200 python tests/sweeps/gpu_stress.py --n 200 --seed 41 ) > "$OUT/stress.log" 2>&1; echo "stress rc=$? $(tail -1 $OUT/stress.log | cut -c1-100)"
b() { # name, env..., -- args
  name=$1; shift
  timeout 200 env "$@" > "$OUT/bench_$name.json" 2> "$OUT/bench_$name.err"
  python -c "import json;d=json.load(open('$OUT/bench_$name.json'));print('$name: %.0f utt/s  step %.3f ms kernel %.3f ms'%(d['value'],d['ms_per_step'],d['kernel_ms']), d.get('pipelined',{}).get('value'))"
}
L=$GRAFT_REPO_ROOT/ctcdecode_amd/_lib
b main256 X=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline
b main512 X=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --batch 512
b q1_256 CTCDECODE_AMD_LIB=$L/var_q1.so python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras
b w8_256 CTCDECODE_AMD_LIB=$L/var_w8.so python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras
b w8_512 CTCDECODE_AMD_LIB=$L/var_w8.so python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --batch 512
b w8_512_excl CTCDECODE_AMD_LIB=$L/var_w8.so CTCD_LDS_FLOOR=90000 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --batch 512
b w8_256_shared CTCDECODE_AMD_LIB=$L/var_w8.so CTCD_LDS_FLOOR=0 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras
timeout 200 python tools/inflight_probe.py > "$OUT/inflight_main.log" 2>&1; tail -4 "$OUT/inflight_main.log"
CTCDECODE_AMD_LIB=$L/var_w8.so CTCD_LDS_FLOOR=0 timeout 200 python tools/inflight_probe.py > "$OUT/inflight_w8.log" 2>&1; tail -4 "$OUT/inflight_w8.log"
timeout 200 python tools/bench_configs.py --only 2 --out "$OUT/cfg2.json" > "$OUT/cfg2.log" 2>&1; tail -1 "$OUT/cfg2.log"
timeout 200 python tools/phase_profile.py --beam 500 --frames 600 --batch 256 --out "$OUT/phase_k500.json" > "$OUT/phase_k500.log" 2>&1; python -c "
import json;d=json.load(open('$OUT/phase_k500.json'));print('k500 us/frame',d['us_per_frame']);[print('  %5.1f%%  %s'%(v,k)) for k,v in sorted(d['phases_percent'].items(),key=lambda kv:-kv[1])[:9]]"
