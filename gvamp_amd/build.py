"""Builds gvamp_amd/libgvamp.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libgvamp.so")
SOURCES = ["gv_kernels.hip", "gv_mfma.hip", "gv_capi.hip", "gv_solvers.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-I/opt/rocm/include"]


def kernel_src_hash():
    """sha256 of the streaming-kernel sources (what a persisted tuning pick and a PMC profile belong to; bench.py and
    scripts/pmc_summary.py compute the same)"""
    import hashlib
    h = hashlib.sha256()
    for f in ("gv_mfma.hip", "gv_mfma.h", "gv_pval_dev.h"):      # (gv_pval_dev.h is compiled into k_fin_pvals)
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def _newer(dst, srcs):
    if not os.path.exists(dst):
        return True
    t = os.path.getmtime(dst)
    return any(os.path.getmtime(s) > t for s in srcs)


def build_lib(force=False, verbose=False):
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    deps = srcs + [os.path.join(CSRC, "gv_internal.h"), os.path.join(ROOT, "include", "gvamp.h")]
    deps += [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".hpp"))]
    have_objs = all(os.path.exists(os.path.join(CSRC, os.path.basename(x) + ".o")) for x in srcs)
    if not force and have_objs and not _newer(LIB, deps):
        return LIB
    objs = []
    for s in srcs:
        o = os.path.join(CSRC, os.path.basename(s) + ".o")
        if force or _newer(o, deps):
            cmd = ["hipcc"] + FLAGS + ['-DGV_KERNEL_SRC_HASH="%s"' % kernel_src_hash()[:16], "-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            subprocess.check_call(cmd)
        objs.append(o)
    cmd = ["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + \
          ["-L/opt/rocm/lib", "-lrccl", "-Wl,-rpath,/opt/rocm/lib"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv, verbose=True))
