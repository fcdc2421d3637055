"""Builds libbot7hip.so (gfx950) in-tree with hipcc.  Cross-compiles without a GPU.

    python -m bot7_amd.build [--force]

One object per .hip translation unit (parallel), then one shared library next to this file, linked -Bsymbolic: its internal
calls bind to its own definitions, so that a second build of the library (the diagnostic one) can live in the same process.  The library
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
SOURCES = ["api.hip", "sobol.hip", "covar.hip", "potrf.hip", "potrf_persist.hip", "posterior.hip", "score.hip", "extras.hip", "comm.hip", "group.hip", "blr_small.hip", "gp_small.hip", "kpost_small.hip"]
# The DIAGNOSTIC build (tools/_build/libbot7hip_diag.so, -DB7_DIAG): the shipped sources + round 3's likelihood kernel kept as
# a bit-for-bit reference.  Only translation units that mention B7_DIAG are compiled a second time; the rest are shared.
DIAG_ONLY_SOURCES = ["nll_small.hip"]
DIAG_OUT = os.path.join(ROOT, "tools", "_build", "libbot7hip_diag.so")
HEADERS = [os.path.join(CSRC, "b7_internal.h"), os.path.join(CSRC, "gemm_f64.h"), os.path.join(CSRC, "potrf_diag.h"), os.path.join(CSRC, "comm_rccl.h"), os.path.join(CSRC, "ksx_exp.h"), os.path.join(CSRC, "exp_table.h"),
           os.path.join(ROOT, "include", "bot7hip.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-fno-fast-math", "-Wall",
         "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-I/opt/rocm/include"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


# per-file flags.  covar.hip: keep the MFMA accumulators of ksx_kernel in VGPRs (the default put them in AGPRs and read every
# result back with v_accvgpr_read: 2 of the kernel's 27 VALU instructions per output, and it is bound by VALU + MFMA issue)
EXTRA_FLAGS = {"covar.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"]}


def _compile(src, diag=False):
    obj = os.path.join(BUILD, ("diag_" if diag else "") + src.replace(".hip", ".o"))
    path = os.path.join(CSRC, src)
    if _stale(obj, [path] + HEADERS + [os.path.abspath(__file__)]):
        subprocess.check_call([HIPCC] + FLAGS + EXTRA_FLAGS.get(src, []) + (["-DB7_DIAG"] if diag else []) + ["-c", path, "-o", obj])
        return obj, True
    return obj, False


def _mentions_diag(src):
    return "B7_DIAG" in open(os.path.join(CSRC, src)).read()


def build_diag(verbose=False):
    """tools/_build/libbot7hip_diag.so: the library with its A/B switches, fault injector and RCCL override compiled in (-DB7_DIAG).
    Test infrastructure: tests load it BESIDE the shipped library (bot7_amd.Context(..., lib="diag"))."""
    build()
    os.makedirs(os.path.dirname(DIAG_OUT), exist_ok=True)
    todo = [s for s in SOURCES if _mentions_diag(s)] + DIAG_ONLY_SOURCES
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, len(todo))) as ex:
        res = dict(zip(todo, ex.map(lambda s: _compile(s, True), todo)))
    objs = [res[s][0] if s in res else os.path.join(BUILD, s.replace(".hip", ".o")) for s in SOURCES + DIAG_ONLY_SOURCES]
    if any(ch for _, ch in res.values()) or _stale(DIAG_OUT, objs):
        subprocess.check_call([HIPCC, "-shared", "-fPIC", "--offload-arch=gfx950", "-o", DIAG_OUT] + objs +
                              ["-ldl", "-Wl,-rpath,/opt/rocm/lib", "-Wl,--no-undefined", "-Wl,-Bsymbolic"])
        if verbose:
            print("built", DIAG_OUT)
    return DIAG_OUT


def posterior_isa():
    """The device ISA of posterior.hip as text (same compiler, same flags as the object that ships), cached."""
    out = os.path.join(BUILD, "posterior.s")
    path = os.path.join(CSRC, "posterior.hip")
    if _stale(out, [path] + HEADERS + [os.path.abspath(__file__)]):
        subprocess.check_call([HIPCC] + [f for f in FLAGS if f != "-fPIC"] + ["-S", "--cuda-device-only", "-Wno-unused-command-line-argument",
                                                                            path, "-o", out])
    return out


def check_agpr_discipline():
    """post_kernel_w4 keeps its 32 accumulator tiles in a0..a255 BY NUMBER inside inline asm; the compiler is told so
    only through clobber lists.  The build fails unless, in the ISA that ships, for both instantiations (1) no
    instruction outside those asm statements names an AGPR, (2) the kernel has no scratch, and (3) every accumulator tile
    is started from the literal 0 and read back.  Returns the counts per NJ."""
    import re
    text = open(posterior_isa()).read()
    found = {}
    for m in re.finditer(r"^(_ZN\S*post_kernel_w4(?:ILi(\d)E|tILi\d+ELi\d+E)\S*):[^\n]*\n(.*?)\n\s*s_endpgm", text, flags=re.S | re.M):
        nj = int(m.group(2)) if m.group(2) else 16   # 16: the tall shape (16 strips of rows x 2 of candidates: 32 tiles)
        ntiles = 32 if nj == 16 else 8 * nj
        in_asm, stats, bad = False, {"mfma": 0, "mfma_from_zero": 0, "acc_reads": 0, "scratch": 0}, []
        for line in m.group(3).split("\n"):
            code = line.split(";")[0]
            if "ASMSTART" in line:
                in_asm = True
                continue
            if "ASMEND" in line:
                in_asm = False
                continue
            if in_asm:
                stats["mfma"] += "v_mfma_f64_16x16x4_f64" in code
                stats["mfma_from_zero"] += bool(re.search(r"v_mfma_f64_16x16x4_f64 a\[[^\]]*\], v\[[^\]]*\], v\[[^\]]*\], 0\s*$", code))
                stats["acc_reads"] += "v_accvgpr_read_b32" in code
            else:
                if re.search(r"(?<![A-Za-z0-9_.])a(\[|\d)", code):
                    bad.append(line.strip())
                stats["scratch"] += "scratch_" in code
        if bad or stats["scratch"] or stats["mfma_from_zero"] != ntiles or stats["acc_reads"] != 8 * ntiles:
            raise RuntimeError("post_kernel_w4<%d>: AGPR discipline broken: %r, compiler-generated AGPR uses: %r"
                               % (nj, stats, bad[:5]))
        found[nj] = stats
    if not {2, 4} <= set(found):
        raise RuntimeError("post_kernel_w4<2> and <4> expected in the ISA of posterior.hip, found %r" % sorted(found))
    return found


def build_post_probe():
    """tools/_build/post_probe: both large-grid posterior kernels on synthetic operands against the host model of the
    arithmetic (ascending fma chains, fixed fold order); tests/test_gpu_parity.py runs it on the GPU box."""
    src = os.path.join(ROOT, "tools", "post_probe.hip")
    out = os.path.join(ROOT, "tools", "_build", "post_probe")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    if _stale(out, [src, os.path.join(ROOT, "tools", "post_kernel_w8.h"), os.path.join(CSRC, "posterior.hip")] + HEADERS):
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-fast-math", "-Wno-unused-value",
                               "-DB7_POST_NO_LAUNCHERS", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC,
                               "-I/opt/rocm/include", "-o", out, src, "-Wl,-rpath,/opt/rocm/lib"])
    return out


def build(force=False, verbose=False):
    os.makedirs(BUILD, exist_ok=True)
    if force:
        for f in os.listdir(BUILD):
            os.remove(os.path.join(BUILD, f))
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, len(SOURCES) + 2)) as ex:
        isa = ex.submit(check_agpr_discipline)
        probe = ex.submit(build_post_probe)
        results = list(ex.map(_compile, SOURCES))
        isa.result()
        probe.result()
    objs = [o for o, _ in results]
    if any(changed for _, changed in results) or _stale(OUT, objs):
        subprocess.check_call([HIPCC, "-shared", "-fPIC", "--offload-arch=gfx950", "-o", OUT] + objs +
                              ["-ldl", "-Wl,-rpath,/opt/rocm/lib", "-Wl,--no-undefined", "-Wl,-Bsymbolic"])
        if verbose:
            print("built", OUT)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
    if "--diag" in sys.argv:
        build_diag(verbose=True)
