#!/bin/bash
# Instruction-cache counters of the scorer's kernels by code size (tools/icache_probe.py):  TAG=r06w bash tools/icache_probe.sh
TAG=${TAG:-r06w}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for v in nolm lm2 lm1 rt wide; do
  timeout 200 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $OUT/ic_$v -o p -- python $GRAFT_REPO_ROOT/tools/icache_probe.py $v > $OUT/ic_$v.log 2>&1; echo "$v rc=$? $(grep kernel_ms $OUT/ic_$v.log)"
done
python - <<'PY'
import csv, glob, collections, os
out = os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/' + os.environ.get('TAG', 'r06w')
for v in "nolm lm2 lm1 rt wide".split():
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(out + '/ic_%s/**/*counter_collection.csv' % v, recursive=True):
        for r in csv.DictReader(open(f)):
            if 'ctc_beam_decode_kernel' in r['Kernel_Name']:
                acc[r['Kernel_Name'][:90]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, c in acc.items():
        m = {n: sum(x) / len(x) for n, x in c.items()}
        req = m.get('SQC_ICACHE_REQ', 0) or 1
        print(v, k, {n: round(x) for n, x in sorted(m.items())}, "miss rate %.5f" % (m.get('SQC_ICACHE_MISSES', 0) / req),
              "issue-stall share %.3f" % (m.get('SQ_WAIT_INST_ANY', 0) / (m.get('SQ_WAVE_CYCLES', 0) or 1)))
PY
