"""Where does nll_small_kernel's time go?  DIAGNOSTIC build (-DB7_NLL_STAMP): the kernel reads s_memtime at its phase
boundaries and workgroup b reports stamp b instead of its likelihood terms (the results are wrong on purpose).
    python tools/nll_small_stamps.py build     (here: cross-compiles tools/_build/libbot7hip_nllstamp.so)
    python tools/nll_small_stamps.py run       (on the GPU box)"""
import math
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "tools", "_build", "libbot7hip_nllstamp.so")
POISON = os.path.join(ROOT, "tools", "_build", "libbot7hip_nllpoison.so")  # -DB7_NLL_POISON: NaN above K's diagonal blocks' diagonals
NAMES = ["start", "hypers + loads landed, image zeroed", "image scattered", "half norms", "K tiles", "X zeroed", "diag_core 1",
         "z1", "L21", "A22 update, r2", "diag_core 2", "z2", "reductions"]


def build():
    from bot7_amd import build as B
    B.build()
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    for lib, define in ((LIB, "-DB7_NLL_STAMP"), (POISON, "-DB7_NLL_POISON")):
        objs = []
        B.build_diag()
        for src in B.SOURCES + B.DIAG_ONLY_SOURCES:   # round 3's kernel lives in the diagnostic build (B7_NLL_SMALL=2 selects it)
            obj = os.path.join(B.BUILD, ("diag_" if (B._mentions_diag(src) or src in B.DIAG_ONLY_SOURCES) else "") + src.replace(".hip", ".o"))
            if src == "nll_small.hip":
                obj = lib.replace(".so", ".o")
                subprocess.check_call([B.HIPCC] + B.FLAGS + B.EXTRA_FLAGS.get(src, []) + [define, "-DB7_DIAG", "-c", os.path.join(B.CSRC, src), "-o", obj])
            objs.append(obj)
        subprocess.check_call([B.HIPCC, "-shared", "-fPIC", "--offload-arch=gfx950", "-o", lib] + objs +
                              ["-ldl", "-Wl,-rpath,/opt/rocm/lib", "-Wl,--no-undefined", "-Wl,-Bsymbolic"])
        print("built", lib)


def run():
    os.environ["BOT7HIP_LIB"] = LIB
    os.environ["B7_NLL_SMALL"] = "2"
    import numpy as np
    import bot7_amd
    ctx = bot7_amd.Context(0)
    rng = np.random.default_rng(0)
    for d, N in ((6, 64), (6, 100), (32, 128)):
        X = rng.random((N, d))
        Y = rng.normal(size=(N, 1))
        ctx.gp_set_data(X, Y)
        B = 32
        for _ in range(3):
            nll = ctx.gp_nll_batch(np.full((B, d), 0.3), 1.0, 1e-3, 0.0)
        st = 2.0 * (np.asarray(nll) - 0.5 * N * math.log(2 * math.pi))
        diag = list(st[16:32])
        st = [s for s in st[:16] if s >= 0]
        names = NAMES if N > 64 else NAMES[:8] + NAMES[-1:]
        st = st[:len(names)]
        print("d %d N %d: total %.0f cycles" % (d, N, st[-1]))
        for k in range(1, len(st)):
            print("   %-40s %7.0f cycles" % (names[k] if k < len(names) else "?", st[k] - st[k - 1]))
        # the factor routine's first call, per 16-column step: wave 0's pivot chain | barrier | (VAR 0's sub-panel) | update + barrier
        prev = 0.0
        for kb in range(4):
            f, b_, s_, u = diag[4 * kb:4 * kb + 4]
            print("   diag_core step %d: chain %6.0f  barrier %5.0f  update + barrier %6.0f   (cycles)" % (kb, f - prev, b_ - f, u - b_))
            prev = u
        print("   diag_core tail (rest of the inverse doubling): %.0f cycles" % (st[6] - st[5] - prev))


if __name__ == "__main__":
    build() if sys.argv[1:] == ["build"] else run()
