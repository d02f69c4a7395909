#!/bin/bash
# bash tools/gpu_lm_variants.sh <tag> v1 v2 ...: LM-tier kernel time (configs[4] per-GPU shape, tests/data/test.arpa) and barrier
# timeline per variant build of the library (tools/build_variants.sh name:CTC_QUICK_BUILD=2,...)
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
for rep in 1 2; do
for v in "$@"; do
  export CTCDECODE_AMD_LIB=$GRAFT_REPO_ROOT/ctcdecode_amd/_lib/var_$v.so
  timeout 200 python - "$v" <<'PY'
import os, sys, torch, ctcdecode_amd
ROOT = os.environ["GRAFT_REPO_ROOT"]
labels = ["_", "'", " "] + [chr(ord("a") + i) for i in range(26)]
torch.manual_seed(3)
lp = torch.randn((128, 1500, 29)).log_softmax(-1).cuda()
for arpa in ("test.arpa", "chars.arpa"):
    dec = ctcdecode_amd.CTCBeamDecoder(labels, beam_width=100, log_probs_input=True, model_path=os.path.join(ROOT, "tests", "data", arpa), alpha=0.5, beta=1.0)
    dec.set_timing(True)
    ms = []
    for _ in range(4):
        r = dec.decode_device(lp); torch.cuda.synchronize(); ms.append(dec.last_kernel_ms())
    print("%s %s: kernel %.2f ms (%.2f us/frame) checksum %d" % (sys.argv[1], arpa, min(ms[1:]), min(ms[1:]) / 1.5, int(r[0].long().sum() + r[3].long().sum())), flush=True)
PY
done
done
for v in "$@"; do
  export CTCDECODE_AMD_LIB=$GRAFT_REPO_ROOT/ctcdecode_amd/_lib/var_$v.so
  timeout 200 python tools/barrier_timeline.py --lm tests/data/test.arpa --batch 128 --T 1500 --frames 3 --repeat 12 --out "$OUT/timeline_$v.json" > "$OUT/timeline_$v.log" 2>&1
  python - <<PY
import json
d=json.load(open("$OUT/timeline_$v.json"))
print("$v timeline clocks/frame", d["clocks_per_frame"], d["stamps_per_frame"]); print("\n".join("%2d %-50s max %5d med %5d min %5d" % (r["stamp"], r["what"][:50], r["max"], r["median"], r["min"]) for r in d["rows"]))
PY
done
