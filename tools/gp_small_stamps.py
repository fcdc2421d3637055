"""Where does gp_small_kernel's time go?  DIAGNOSTIC build (-DB7_GS_STAMP): workgroup 0's wave 0 (the factor routine's wave)
and wave 4 (the first helper wave) record s_memtime at the kernel's phase boundaries.
    python tools/gp_small_stamps.py build     (here: cross-compiles tools/_build/libbot7hip_gsstamp.so)
    python tools/gp_small_stamps.py run       (on the GPU box)"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "tools", "_build", "libbot7hip_gsstamp.so")
NAMES = ["start", "hypers + loads landed, image zeroed", "image scattered, half norms, (fit: scaled observations out)", "K11 tiles",
         "(the diagnostic second sub-tile), barrier", "diag_core 1 (idle waves: K11 rest, K21, K22 entries)", "L11 out, L21 chains", "L21 image",
         "L21 L21' chains and subtraction / L21 inv(L11), stores", "barrier", "images for block 2", "diag_core 2 (helpers: z1, r2)",
         "z2 / inverse's off-diagonal block, stores", "reductions / alpha"]


def build():
    from bot7_amd import build as B
    B.build()
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    objs = []
    for src in B.SOURCES:
        obj = os.path.join(B.BUILD, src.replace(".hip", ".o"))
        if src == "gp_small.hip":
            obj = LIB.replace(".so", ".o")
            subprocess.check_call([B.HIPCC] + B.FLAGS + B.EXTRA_FLAGS.get(src, []) + ["-DB7_GS_STAMP", "-c", os.path.join(B.CSRC, src), "-o", obj])
        objs.append(obj)
    subprocess.check_call([B.HIPCC, "-shared", "-fPIC", "--offload-arch=gfx950", "-o", LIB] + objs +
                          ["-ldl", "-Wl,-rpath,/opt/rocm/lib", "-Wl,--no-undefined", "-Wl,-Bsymbolic"])
    print("built", LIB)


def run():
    os.environ["BOT7HIP_LIB"] = LIB
    import numpy as np
    import bot7_amd
    from bot7_amd import _lib
    ctx = bot7_amd.Context(0)
    L = _lib.load()
    rng = np.random.default_rng(0)
    buf = (C.c_ulonglong * 96)()

    def report(title):
        L.b7dbg_gs_stamps(buf)
        for wv, who in ((0, "wave 0"), (1, "wave 4")):
            st = [buf[32 * wv + i] for i in range(14)]
            print("  %s (%s): total %d cycles (s_memtime: shader clock; %.1f us at 2.4 GHz)" % (title, who, st[13] - st[0], (st[13] - st[0]) / 2400.0))
            print("     (diagnostic: the wave's K sub-tile a second time, code already fetched: %d cycles; SIMD of waves 0..7: %s)"
                  % (buf[32 * wv + 15] - buf[32 * wv + 14], [int(buf[16 + k]) for k in range(8)]))
            prev = st[0]
            for k in range(1, 14):
                if st[k] >= prev and st[k] != 0:
                    print("     %-62s %7d" % (NAMES[k], st[k] - prev))
                    prev = st[k]

    def diag_report():
        # the factor routine's own stamps: per 16-column step: chain | barrier | update + barrier  (slots 2 + 4 kb ..)
        for name, base, t0 in (("block 0", 40, buf[4]), ("block 1", 64, buf[10])):
            d = [buf[base + i] for i in range(24)]
            if d[2] == 0 or d[2] < t0:
                continue
            prev = t0
            out = []
            for kb in range(4):
                f, b_, s_, u = d[2 + 4 * kb:6 + 4 * kb]
                out.append("step %d: chain %d barrier %d update+barrier %d" % (kb, f - prev, b_ - f, u - b_))
                prev = u
            print("     diag_core %s: %s" % (name, "; ".join(out)))

    for d, N in ((2, 25), (6, 64), (6, 100), (32, 128)):
        X = rng.random((N, d))
        Y = rng.normal(size=(N, 1))
        ctx.gp_set_data(X, Y)
        for _ in range(3):
            ctx.gp_nll_batch(np.full((1, d), 0.3), 1.0, 1e-3, 0.0)
        ctx.sync()
        print("d %d N %d" % (d, N))
        report("likelihood")
        diag_report()
        ctx.grid_sobol(512, d, 1, download=False)
        for _ in range(3):
            ctx.eval_nominate([{"lenscale_sq": np.full(d, 0.3), "amp": 1.0, "noise": 1e-3, "mean": 0.0}], score="cb")
        ctx.sync()
        report("fit")
        L.b7dbg_gs_stamps(buf)
        if buf[20]:
            print("     fit tail (wave 0): diag 2 end -> chains of the inverse's block %d -> its image %d -> stores issued %d -> t products + butterflies %d -> "
                  "t in LDS + barrier %d -> alpha %d -> out %d" % (buf[23] - buf[11], buf[24] - buf[23], buf[12] - buf[24], buf[20] - buf[12],
                                                               buf[21] - buf[20], buf[22] - buf[21], buf[13] - buf[22]))


if __name__ == "__main__":
    build() if sys.argv[1:] == ["build"] else run()
