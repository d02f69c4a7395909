#!/bin/bash
# Hardware counters of the decode kernel for several builds of the library (tools/build_variants.sh): VARIANTS="base x" TAG=r06d PASSES="A B" bash tools/pmc_cmp_variants.sh
TAG=${TAG:-r05h13}; export TAG
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQC\?_[A-Z_0-9]*" | sort -u | grep -i "ICACHE\|IFETCH\|INST_LEVEL\|SQC_" > $OUT/ic_counters.txt; head -40 $OUT/ic_counters.txt | tr '\n' ' '; echo
A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS"
B="SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_INSTS_SMEM SQ_WAIT_INST_LDS"
for v in ${VARIANTS:-base fn}; do
  for P in ${PASSES:-A B}; do
    C="$A"; [ $P = B ] && C="$B"
    CTCDECODE_AMD_LIB=$GRAFT_REPO_ROOT/ctcdecode_amd/_lib/var_$v.so timeout 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/${v}_$P -o p -- python $GRAFT_REPO_ROOT/tools/pmc_child.py 256 1000 29 100 0 > $OUT/${v}_$P.log 2>&1; echo "$v $P rc=$?"
  done
done
python - <<'PY'
import csv,glob,collections,os
out=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/'+os.environ['TAG']
for v in os.environ.get("VARIANTS","base fn").split():
    acc=collections.defaultdict(list)
    for f in glob.glob(out+'/%s_*/**/*counter_collection.csv'%v, recursive=True):
        for r in csv.DictReader(open(f)):
            if 'ctc_beam_decode_kernel' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
    print(v, {k: round(sum(x)/len(x)) for k,x in sorted(acc.items())})
PY
