#!/bin/bash
# A hash of the machine code of every kernel in the built library (or in the library given as $1): "this change leaves the north-star
# kernel as it was" is checked, not assumed.   usage: bash tools/kernel_hashes.sh [lib.so] | grep "1024, 0, false"
set -e
LIB=${1:-$(dirname "$0")/../ctcdecode_amd/_lib/libctcdecode_amd.so}
TMP=$(mktemp -d)
trap 'rm -rf "$TMP"' EXIT
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
for f in "$TMP"/co*.elf; do
  /opt/rocm/lib/llvm/bin/llvm-objdump -d --no-show-raw-insn --no-leading-addr "$f" 2>/dev/null | python3 -c '
import sys, re, hashlib, subprocess
cur=None; acc={}
for l in sys.stdin:
    m=re.match(r"^<(.*)>:$", l.strip())
    if m: cur=m.group(1); acc[cur]=hashlib.sha1(); continue
    if cur and l.strip():
        # branch targets are printed as absolute addresses: keep the mnemonic and register operands only
        acc[cur].update(re.sub(r"0x[0-9a-f]+|<[^>]*>|\s+", " ", l).encode())
for k,h in acc.items():
    if "ctc_beam_decode_kernel" in k or "tie_frame" in k:
        name=subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
        print(h.hexdigest()[:12], name[:110])
'
done | sort -k2
