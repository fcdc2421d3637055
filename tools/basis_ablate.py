"""What bounds mlp_resident_kernel?  DIAGNOSTIC builds (-DB7_MLP_ABLATE=bits: 1 no feature stores, 2 one k-step per layer),
timed with tools/basis_rate.py.  Results are wrong on purpose.
    python tools/basis_ablate.py build | run"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VARIANTS = (0, 1, 2, 3)


def lib(v):
    return os.path.join(ROOT, "tools", "_build", "libbot7hip_mlp%d.so" % v)


def build():
    from bot7_amd import build as B
    B.build()
    for v in VARIANTS:
        objs = []
        for src in B.SOURCES:
            obj = os.path.join(B.BUILD, src.replace(".hip", ".o"))
            if src == "extras.hip":
                obj = os.path.join(ROOT, "tools", "_build", "extras_mlp%d.o" % v)
                subprocess.check_call([B.HIPCC] + B.FLAGS + B.EXTRA_FLAGS.get(src, []) + ["-DB7_MLP_ABLATE=%d" % v, "-c", os.path.join(B.CSRC, src), "-o", obj])
            objs.append(obj)
        subprocess.check_call([B.HIPCC, "-shared", "-fPIC", "--offload-arch=gfx950", "-o", lib(v)] + objs + ["-ldl", "-Wl,-rpath,/opt/rocm/lib", "-Wl,--no-undefined", "-Wl,-Bsymbolic"])
        print("built", lib(v), flush=True)


def run():
    names = {1: "no feature stores", 2: "one k-step per layer"}
    for v in VARIANTS:
        label = " + ".join(names[b] for b in (1, 2) if v & b) or "full"
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "basis_rate.py")], capture_output=True, text=True, env=dict(os.environ, BOT7HIP_LIB=lib(v)))
        print("== %s" % label)
        print("".join(l + "\n" for l in out.stdout.splitlines() if l.startswith("act None") or l.startswith("act Tanh")), end="", flush=True)


if __name__ == "__main__":
    build() if len(sys.argv) > 1 and sys.argv[1] == "build" else run()
