#!/bin/bash
# edge cases on the GPU (incl. the all -0.0 rows added late) + one rocprofv3 --pmc pass: instructions per wave of the headline kernel
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03m; mkdir -p "$OUT"; cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
( timeout 60 python -m pytest tests/test_gpu_decode.py -x -q -k "edge_cases" ) 2>&1 | tail -1
cd /tmp
timeout 90 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d "$OUT/sq1" -o p -- python "$GRAFT_REPO_ROOT/bench.py" --steps 2 --warmup 1 --no-extras --no-cpu-baseline > "$OUT/sq1.log" 2>&1; echo "sq1 rc=$?"
python3 - <<PY
import csv,glob,collections
for f in glob.glob("$OUT/sq1/**/*counter_collection.csv", recursive=True):
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'ctc_beam_decode' in r.get('Kernel_Name',''):
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in acc.items(): print(k,'n=%d'%len(v),'mean=%.6g'%(sum(v)/len(v)))
PY
