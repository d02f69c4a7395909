#!/usr/bin/env python3
"""Recipe that produced the constants in ctcdecode_amd/csrc/exact_math.h.

The reference's float32 log_sum_exp (ctcdecode/src/decoder_utils.h:47-54) binds to
the host C library's expf/logf.  This script documents how their data tables and
operation order were read out of THIS image's glibc 2.35 libm (stripped, so by
address):

  readelf -sW --dyn-syms libm.so.6 | grep -E ' (expf|logf)@@'     # IFUNC resolvers
  objdump -d --start-address=<resolver> ...                       # -> *_fma / *_sse2 variants
  objdump -d --start-address=0x7aba0 --stop-address=0x7ac38       # __expf_fma main path
  objdump -d --start-address=0x7add0 --stop-address=0x7ae7a       # __logf_fma main path

The rip-relative loads in those bodies give the table addresses used below
(.rodata is mapped at file offset == vaddr in this build).  Run it to re-print
the tables; tests/native/exact_math_check.cpp is the actual guard (exhaustive
comparison of the restated routines against the live libm).
"""
import struct
import sys

LIBM = sys.argv[1] if len(sys.argv) > 1 else "/lib/x86_64-linux-gnu/libm.so.6"
EXP2F_DATA = 0xB2B80  # uint64 tab[32]; double shift_scaled, poly[3], shift, invln2_scaled, poly_scaled[3]
LOGF_DATA = 0xB2CE0   # struct {double invc, logc;} tab[16]; double ln2, poly[3]


def main():
    d = open(LIBM, "rb").read()
    f64 = lambda off: struct.unpack_from("<d", d, off)[0]
    u64 = lambda off: struct.unpack_from("<Q", d, off)[0]
    print("exp2f tab  :", ", ".join("0x%016x" % u64(EXP2F_DATA + 8 * i) for i in range(32)))
    names = ["shift_scaled", "poly0", "poly1", "poly2", "shift", "invln2_scaled", "poly_scaled0", "poly_scaled1", "poly_scaled2"]
    for i, n in enumerate(names):
        print("%-14s %s" % (n, f64(EXP2F_DATA + 256 + 8 * i).hex()))
    for i in range(16):
        print("logf tab[%2d] invc=%s logc=%s" % (i, f64(LOGF_DATA + 16 * i).hex(), f64(LOGF_DATA + 16 * i + 8).hex()))
    for i, n in enumerate(["ln2", "A0", "A1", "A2"]):
        print("%-4s %s" % (n, f64(LOGF_DATA + 256 + 8 * i).hex()))


if __name__ == "__main__":
    main()
