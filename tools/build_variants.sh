#!/bin/bash
# Builds variants of the HIP library for kernel experiments: ctcdecode_amd/_lib/var_<name>.so  (git-ignored; they travel
# to the GPU box).  Usage: tools/build_variants.sh name1:DEF1=1,DEF2=1 name2:DEF=3 "name3::-O2 -mllvm -some-flag" ...
# (name:defines:extra hipcc flags)
cd "$(dirname "$0")/.."
for spec in "$@"; do
  IFS=: read -r name defs flags <<< "$spec"
  CTCD_EXTRA_HIPCC_FLAGS="$flags" python - <<PY &
import sys
sys.path.insert(0, ".")
from ctcdecode_amd import _build
d = [x for x in "$defs".split(",") if x]
print(_build.build(defines=d, out=_build.LIB_DIR + "/var_$name.so"))
PY
done
wait
