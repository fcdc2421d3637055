"""Builds libbot7hip.so (gfx950) in-tree with hipcc.  Cross-compiles without a GPU.

    python -m bot7_amd.build [--force]

One object per .hip translation unit (parallel), then one shared library next to this file.  The library
links only against the HIP runtime (libamdhip64.so.7); no torch, no rocBLAS/rocSOLVER; librccl.so.1 is dlopen'ed
at the first b7_comm_* call (csrc/comm.hip).
"""
import concurrent.futures
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
BUILD = os.path.join(HERE, "_build")
OUT = os.path.join(HERE, "libbot7hip.so")
SOURCES = ["api.hip", "sobol.hip", "covar.hip", "potrf.hip", "potrf_persist.hip", "posterior.hip", "score.hip", "extras.hip", "comm.hip"]
HEADERS = [os.path.join(CSRC, "b7_internal.h"), os.path.join(CSRC, "gemm_f64.h"), os.path.join(CSRC, "potrf_diag.h"),
           os.path.join(ROOT, "include", "bot7hip.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-fno-fast-math", "-Wall",
         "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-I/opt/rocm/include"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src):
    obj = os.path.join(BUILD, src.replace(".hip", ".o"))
    path = os.path.join(CSRC, src)
    if _stale(obj, [path] + HEADERS + [os.path.abspath(__file__)]):
        subprocess.check_call([HIPCC] + FLAGS + ["-c", path, "-o", obj])
        return obj, True
    return obj, False


def build(force=False, verbose=False):
    os.makedirs(BUILD, exist_ok=True)
    if force:
        for f in os.listdir(BUILD):
            os.remove(os.path.join(BUILD, f))
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(6, len(SOURCES))) as ex:
        results = list(ex.map(_compile, SOURCES))
    objs = [o for o, _ in results]
    if any(changed for _, changed in results) or _stale(OUT, objs):
        subprocess.check_call([HIPCC, "-shared", "-fPIC", "--offload-arch=gfx950", "-o", OUT] + objs +
                              ["-ldl", "-Wl,-rpath,/opt/rocm/lib", "-Wl,--no-undefined"])
        if verbose:
            print("built", OUT)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
