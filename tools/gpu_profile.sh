#!/bin/bash
# The evidence session of a round, ONE script (replaces the per-session gpu_round*.sh of rounds 1-3):
#   bash tools/gpu_profile.sh <tag> [suite] [bench] [stats] [pmc] [timeline] [sweeps]      (default: all but sweeps)
# suite: smoke() + pytest -m gpu; bench: the bench line; stats: rocprofv3 --kernel-trace --stats of bench.py, of the other
# configurations (tools/bench_configs.py: wide beam, prune, configs[4] shape) and of the extras (tools/profile_extras.py: LM
# kernel, two-workgroups-per-CU build, log_softmax, expand, the 1 GiB calibration copy); pmc: FETCH_SIZE / WRITE_SIZE passes of
# the same three commands (counters only, separate passes) + the instruction counters of the headline; timeline: barrier
# timelines (random and blank-dominated rows, LM kernel); sweeps: parity sweep against the real reference + random stress.
# Everything lands in gpurun_out/<tag>/; python tools/summarize_profiles.py <tag> turns it into the files under profiles/.
TAG=${1:-r04}; shift
STEPS="${*:-suite bench stats pmc timeline}"
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p "$OUT"; cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
echo "host: $(nproc) cores; steps: $STEPS" | tee "$OUT/host.txt"
has() { case " $STEPS " in *" $1 "*) return 0;; *) return 1;; esac; }
if has suite; then
  ( time timeout 180 python -c "import __graft_entry__ as g; g.smoke()" ) > "$OUT/smoke.log" 2>&1; echo "smoke rc=$?" | tee -a "$OUT/host.txt"; tail -2 "$OUT/smoke.log"
  ( time timeout ${SUITE_TIMEOUT:-900} python -m pytest tests -m gpu -x -q --durations=8 ) > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?" | tee -a "$OUT/host.txt"; tail -14 "$OUT/pytest_gpu.log"
fi
if has bench; then
  ( time timeout 600 python bench.py --steps 20 --warmup 3 ) > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc=$?"; cut -c1-600 "$OUT/bench.json"; tail -3 "$OUT/bench.err"
fi
cd /tmp
if has stats; then
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -o trace -- python "$GRAFT_REPO_ROOT/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-pmc > "$OUT/prof.log" 2>&1; echo "rocprof stats rc=$?"
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_cfg" -o trace -- python "$GRAFT_REPO_ROOT/tools/bench_configs.py" --only 234 --reps 1 > "$OUT/prof_cfg.log" 2>&1; echo "rocprof cfg rc=$?"
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_extras" -o trace -- python "$GRAFT_REPO_ROOT/tools/profile_extras.py" > "$OUT/prof_extras.log" 2>&1; echo "rocprof extras rc=$?"
fi
if has pmc; then
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OUT/pmc_$c" -o p -- python "$GRAFT_REPO_ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-pmc > "$OUT/pmc_$c.log" 2>&1; echo "pmc $c rc=$?"
    timeout 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OUT/pmc_cfg_$c" -o p -- python "$GRAFT_REPO_ROOT/tools/bench_configs.py" --only 23 --reps 1 > "$OUT/pmc_cfg_$c.log" 2>&1; echo "pmc cfg $c rc=$?"
    timeout 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OUT/pmc_extras_$c" -o p -- python "$GRAFT_REPO_ROOT/tools/profile_extras.py" > "$OUT/pmc_extras_$c.log" 2>&1; echo "pmc extras $c rc=$?"
  done
  timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d "$OUT/pmc_insts" -o p -- python "$GRAFT_REPO_ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-pmc > "$OUT/pmc_insts.log" 2>&1; echo "pmc insts rc=$?"
fi
cd "$GRAFT_REPO_ROOT"
if has timeline; then
  timeout 200 python tools/barrier_timeline.py --repeat 6 --frames 4 --out "$OUT/timeline.json" > "$OUT/timeline.log" 2>&1; echo "timeline rc=$?"; cut -c1-110 "$OUT/timeline.log" | tail -24
  timeout 200 python tools/barrier_timeline.py --repeat 4 --frames 4 --kind blank --out "$OUT/timeline_blank.json" > "$OUT/timeline_blank.log" 2>&1; echo "timeline (blank) rc=$?"
  timeout 200 python tools/barrier_timeline.py --repeat 6 --frames 3 --lm tests/data/test.arpa --out "$OUT/timeline_lm.json" > "$OUT/timeline_lm.log" 2>&1; echo "timeline (LM) rc=$?"
fi
if has sweeps; then
  ( timeout 900 python tests/sweeps/parity_sweep.py --out "$OUT/parity_sweep.json" ) 2>&1 | tail -4
  ( timeout 200 python tests/sweeps/gpu_stress.py --n 600 --seed 801 ) 2>&1 | tail -1 | cut -c1-80
  ( timeout 200 python tests/sweeps/gpu_stress.py --n 400 --seed 802 --degenerate ) 2>&1 | tail -1 | cut -c1-80
  ( timeout 200 python tests/sweeps/gpu_stress_lm.py --n 300 --seed 803 ) 2>&1 | tail -1 | cut -c1-80
  ( timeout 200 python tests/sweeps/gpu_stress_lm.py --n 200 --seed 804 --degenerate ) 2>&1 | tail -1 | cut -c1-80
fi
find "$OUT" -name '*kernel_stats.csv' | head
