#!/bin/bash
# gfx950 code bytes, registers and scratch of every kernel in the built library (or in the library given as $1).
# usage: bash tools/code_sizes.sh [lib.so] [> profiles/rNN_code_sizes.txt]
set -e
LIB=${1:-$(dirname "$0")/../ctcdecode_amd/_lib/libctcdecode_amd.so}
TMP=$(mktemp -d)
trap 'rm -rf "$TMP"' EXIT
BUNDLER=/opt/rocm/lib/llvm/bin/clang-offload-bundler
objcopy -O binary --only-section=.hip_fatbin "$LIB" "$TMP/fat.bin"
python3 - "$TMP/fat.bin" "$TMP" <<'PY'
import struct, sys
data = open(sys.argv[1], 'rb').read()
magic = b'__CLANG_OFFLOAD_BUNDLE__'
pos, n = 0, 0
while True:
    i = data.find(magic, pos)
    if i < 0:
        break
    cnt = struct.unpack_from('<Q', data, i + 24)[0]
    off = i + 32
    for _ in range(cnt):
        o, sz, tl = struct.unpack_from('<QQQ', data, off)
        triple = data[off + 24: off + 24 + tl].decode()
        off += 24 + tl
        if 'gfx950' in triple and sz:
            open('%s/co%03d.elf' % (sys.argv[2], n), 'wb').write(data[i + o: i + o + sz])
            n += 1
    pos = i + 24
PY
printf "%10s %6s %6s %8s  %s\n" code_B VGPR SGPR scratch kernel
for co in "$TMP"/co*.elf; do
  /opt/rocm/lib/llvm/bin/llvm-readelf -sW "$co" | awk '$4=="FUNC" && $3>0 {print $3, $8}' > "$TMP/funcs"
  /opt/rocm/lib/llvm/bin/llvm-readelf --notes "$co" > "$TMP/notes" 2>/dev/null || true
  while read -r size name; do
    v=$(awk -v k="$name" '$0 ~ "\\.name:" && $2==k {f=1} f && /\.vgpr_count:/ {print $2; exit}' "$TMP/notes")
    s=$(awk -v k="$name" '$0 ~ "\\.name:" && $2==k {f=1} f && /\.sgpr_count:/ {print $2; exit}' "$TMP/notes")
    p=$(awk -v k="$name" '$0 ~ "\\.name:" && $2==k {f=1} f && /\.private_segment_fixed_size:/ {print $2; exit}' "$TMP/notes")
    printf "%10s %6s %6s %8s  %s\n" "$size" "${v:--}" "${s:--}" "${p:--}" "$(echo "$name" | c++filt | sed 's/(ctcdk::KernelArgs)//; s/ctcdk:://' | cut -c1-110)"
  done < "$TMP/funcs"
done | sort -n | uniq
