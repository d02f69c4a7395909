#!/bin/bash
# rocprofv3 --pmc passes over bench.py (counters only; no trace domains mixed in).  bash tools/gpu_pmc.sh <tag> [threads]
TAG=${1:-pmc}; TH=${2:-1024}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p "$OUT"; cd /tmp; export TMPDIR=/tmp
run() { # name counters...
  name=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$OUT/$name" -o p -- python "$GRAFT_REPO_ROOT/bench.py" --steps 2 --warmup 1 --threads $TH --no-cpu-baseline > "$OUT/$name.log" 2>&1
  echo "$name rc=$?"
}
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD
run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT
run tcc1 FETCH_SIZE
run tcc2 WRITE_SIZE
find "$OUT" -name '*counter_collection.csv' | head
python3 - <<PY
import csv,glob,collections
for d in ["sq1","sq2","tcc1","tcc2"]:
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv"%d, recursive=True):
        acc=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if 'ctc_beam_decode' in r.get('Kernel_Name',''):
                acc[r['Counter_Name']].append(float(r['Counter_Value']))
        for k,v in acc.items(): print(d,k,'n=%d'%len(v),'mean=%.4g'%(sum(v)/len(v)))
PY
