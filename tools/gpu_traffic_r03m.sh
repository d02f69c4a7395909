#!/bin/bash
# smoke() + the two rocprofv3 --pmc traffic passes (FETCH_SIZE, WRITE_SIZE; counters only) over the headline bench command
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03m; mkdir -p "$OUT"; cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
( timeout 40 python -c "import __graft_entry__ as g; g.smoke()" ) 2>&1 | tail -2
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 60 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OUT/tcc_$c" -o p -- python "$GRAFT_REPO_ROOT/bench.py" --steps 2 --warmup 1 --no-extras --no-cpu-baseline > "$OUT/tcc_$c.log" 2>&1; echo "$c rc=$?"
done
python3 - <<PY
import csv,glob,collections
for c in ["FETCH_SIZE","WRITE_SIZE"]:
    for f in glob.glob("$OUT/tcc_%s/**/*counter_collection.csv"%c, recursive=True):
        acc=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if 'ctc_beam_decode' in r.get('Kernel_Name',''):
                acc[r['Counter_Name']].append(float(r['Counter_Value']))
        for k,v in acc.items(): print(k,'n=%d'%len(v),'mean=%.8g'%(sum(v)/len(v)))
PY
