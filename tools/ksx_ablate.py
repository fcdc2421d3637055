"""What bounds ksx_kernel?  DIAGNOSTIC builds of the library with parts of the kernel switched off (-DB7_KSX_ABLATE=bits:
1 no stores, 2 conflict-free table reads, 4 no exp, 8 one MFMA per tile, 16 stores to row-contiguous addresses), timed with tools/ksx_rate.py.  The results are
wrong on purpose; only the times mean anything.
    python tools/ksx_ablate.py build            (here: cross-compiles tools/_build/libbot7hip_ksxN.so)
    python tools/ksx_ablate.py run [d ...]      (on the GPU box)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VARIANTS = (0, 128)


def lib(v):
    return os.path.join(ROOT, "tools", "_build", "libbot7hip_ksx%d.so" % v)


def build():
    from bot7_amd import build as B
    B.build()
    os.makedirs(os.path.join(ROOT, "tools", "_build"), exist_ok=True)
    for v in VARIANTS:
        objs = []
        for src in B.SOURCES:
            obj = os.path.join(B.BUILD, src.replace(".hip", ".o"))
            if src == "covar.hip":
                obj = os.path.join(ROOT, "tools", "_build", "covar_ksx%d.o" % v)
                subprocess.check_call([B.HIPCC] + B.FLAGS + B.EXTRA_FLAGS.get(src, []) + ["-DB7_KSX_ABLATE=%d" % v, "-c", os.path.join(B.CSRC, src), "-o", obj])
            objs.append(obj)
        subprocess.check_call([B.HIPCC, "-shared", "-fPIC", "--offload-arch=gfx950", "-o", lib(v)] + objs +
                              ["-ldl", "-Wl,-rpath,/opt/rocm/lib", "-Wl,--no-undefined", "-Wl,-Bsymbolic"])
        print("built", lib(v), flush=True)


def run(dims):
    names = {0: "full", 1: "no stores", 2: "no table conflicts", 4: "no exp", 8: "one MFMA per tile", 16: "row-contiguous store addresses", 32: "non-temporal stores", 128: "one workgroup fewer per CU (LDS padded)"}
    for v in VARIANTS:
        label = " + ".join(names[b] for b in (1, 2, 4, 8, 16, 32, 128) if v & b) or "full"
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "ksx_rate.py")] + dims, capture_output=True, text=True,
                             env=dict(os.environ, BOT7HIP_LIB=lib(v)))
        print("== %s" % label)
        print("".join(l + "\n" for l in out.stdout.splitlines() if l.startswith("d =")), end="", flush=True)
        if out.returncode != 0:
            print(out.stderr[-500:])


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "build":
        build()
    else:
        run(sys.argv[2:] or ["2", "32", "64"])
