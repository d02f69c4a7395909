"""Builds ctcdecode_amd/_lib/libctcdecode_amd.so (HIP kernels + C ABI) for gfx950, in-tree.

The library must share the HIP runtime that PyTorch-ROCm has already loaded into the process (device pointers and
streams cross the boundary), so it is linked against the libamdhip64.so that ships inside torch/lib (SONAME
``libamdhip64.so``); /opt/rocm/lib is on the rpath as the fallback for a host program that does not use torch.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "_lib")
# CTCDECODE_AMD_LIB: load another build of the library (kernel experiments: tools/build_variants.sh)
LIB_PATH = os.environ.get("CTCDECODE_AMD_LIB") or os.path.join(LIB_DIR, "libctcdecode_amd.so")
SOURCES = ["ctcdecode_amd.hip"]
KERNEL_SOURCE = "decode_kernels.hip"  # compiled once per group of kernel instantiations (decode_kernel.h CTC_KERNEL_LIST), in parallel
KERNEL_GROUPS = 12
HEADERS = ["decode_kernels.hip", "decode_kernel.h", "beam_core.h", "stl_emul.h", "exact_math.h", "exact_math_f64.h", "exact_math_f64_tables.h", "lm_tables.h", "lm_build.h", "lm_callback.h", "compact_results.h", os.path.join("..", "..", "include", "ctcdecode_amd.h")]
KERNEL_HEADERS = ["decode_kernels.hip", "decode_kernel.h", "beam_core.h", "stl_emul.h", "exact_math.h", "exact_math_f64.h", "exact_math_f64_tables.h", "lm_tables.h", "compact_results.h"]
ROCM = os.environ.get("ROCM_HOME", "/opt/rocm")


def _hipcc():
    return shutil.which("hipcc") or os.path.join(ROCM, "bin", "hipcc")


def _torch_lib_dir():
    try:
        import torch

        d = os.path.join(os.path.dirname(torch.__file__), "lib")
        if os.path.exists(os.path.join(d, "libamdhip64.so")):
            return d
    except Exception:
        pass
    return None


def is_stale():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(p) > t for p in deps)


def build(force=False, verbose=False, defines=(), out=None, jobs=None):
    """defines/out: build a variant of the library (kernel experiments) next to the product one."""
    from concurrent.futures import ThreadPoolExecutor

    lib_path = out or LIB_PATH
    if not out and not force and not is_stale():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    stem = os.path.splitext(lib_path)[0] if out else os.path.join(LIB_DIR, "obj")
    base = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
            "-Wno-unused-result",
            # (lane-0 atomics on per-workgroup counters are already one per wave: the atomic optimizer's wave reduction
            #  around them is ~10 instructions of dead weight per site on every wave)
            "-mllvm", "-amdgpu-atomic-optimizer-strategy=None"] + ["-D" + d for d in defines] + os.environ.get("CTCD_EXTRA_HIPCC_FLAGS", "").split()
    if verbose:
        base.insert(1, "-Rpass-analysis=kernel-resource-usage")
    quick = any(d.split("=")[0] == "CTC_QUICK_BUILD" for d in defines)
    units = [(os.path.join(CSRC, src), stem + "." + os.path.splitext(src)[0] + ".o", []) for src in SOURCES]
    units += [(os.path.join(CSRC, KERNEL_SOURCE), "%s.kernels%02d.o" % (stem, g), ["-DCTC_KERNEL_GROUP=%d" % g])
              for g in range(2 if quick else KERNEL_GROUPS)]

    # the kernel groups depend on the headers decode_kernels.hip pulls in, not on the host file: an edit of ctcdecode_amd.hip alone
    # recompiles one unit instead of thirteen (product build only; variant builds always compile everything)
    kdeps = [os.path.join(CSRC, h) for h in KERNEL_HEADERS] + [os.path.abspath(__file__)]

    def fresh(u):
        src, obj, _ = u
        return (not out and not defines and not force and os.path.basename(src) == KERNEL_SOURCE and os.path.exists(obj)
                and all(os.path.getmtime(obj) > os.path.getmtime(d) for d in kdeps))

    def compile_one(u):
        src, obj, extra = u
        if fresh(u):
            return obj
        cmd = base + extra + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.run(cmd, check=True)
        return obj

    with ThreadPoolExecutor(max_workers=jobs or min(len(units), os.cpu_count() or 4)) as ex:
        objs = list(ex.map(compile_one, units))
    tl = _torch_lib_dir()
    libdirs = ([tl] if tl else []) + [os.path.join(ROCM, "lib")]
    link = ["g++", "-shared", "-o", lib_path] + objs
    for ld in libdirs:
        link += ["-L" + ld, "-Wl,-rpath," + ld, "-Wl,-rpath-link," + ld]
    link += ["-lamdhip64", "-lpthread"]
    if verbose:
        print(" ".join(link), file=sys.stderr)
    subprocess.run(link, check=True)
    return lib_path


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
