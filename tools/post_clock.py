"""In-kernel clock of post_kernel (MI355X_MICROARCH.md, DVFS give-back item 6): builds a DIAGNOSTIC copy of the library
with -DB7_POST_STAMPS (every workgroup stamps s_memtime / s_memrealtime at its start and end), runs the headline
posterior for >= 2 s back to back on the bench inputs, and prints the median in-kernel clock, the workgroup duration and
the MFMA issue efficiency that follows:  cycles a workgroup needs for its MFMAs alone / cycles it took.

    python tools/post_clock.py build        (here: cross-compiles tools/_build/libbot7hip_stamps.so)
    python tools/post_clock.py run [N d M]  (on the GPU box)"""
import ctypes as C
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tools", "_build", "libbot7hip_stamps.so")
sys.path.insert(0, ROOT)


def build():
    from bot7_amd import build as B
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    B.build()
    objs = []
    for src in B.SOURCES:
        obj = os.path.join(B.BUILD, src.replace(".hip", ".o"))
        if src == "posterior.hip":
            obj = os.path.join(os.path.dirname(OUT), "posterior_stamps.o")
            subprocess.check_call([B.HIPCC] + B.FLAGS + ["-DB7_POST_STAMPS", "-c", os.path.join(B.CSRC, src), "-o", obj])
        objs.append(obj)
    subprocess.check_call([B.HIPCC, "-shared", "-fPIC", "--offload-arch=gfx950", "-o", OUT] + objs +
                          ["-ldl", "-Wl,-rpath,/opt/rocm/lib", "-Wl,--no-undefined", "-Wl,-Bsymbolic"])
    print("built", OUT)


def run(N=2048, d=32, M=1 << 18):
    os.environ["BOT7HIP_LIB"] = OUT
    import numpy as np
    import bench
    import bot7_amd
    from bot7_amd import _lib
    from harness import benchmarks
    ctx = bot7_amd.Context(0)
    X_obs = bench.make_inputs(ctx, d, N, M, 0, M)
    Y = benchmarks.registry["ackley"](X_obs)
    amp = float(np.var(Y))
    ctx.gp_set_data(X_obs, Y)
    ctx.gp_fit_hyp(np.full(d, d / 8.0), amp, 1e-4 * amp, float(np.mean(Y)))
    t0 = time.time()
    n = 0
    while time.time() - t0 < 2.5:          # >= 2 s of back-to-back launches before the stamps that count
        ctx.gp_predict(download=False)
        n += 1
    ctx.sync()
    ctx.timer_start(0)
    ctx.gp_predict(download=False)
    ctx.timer_stop(0)
    wall_ms = ctx.timer_ms(0)
    L = _lib.load()
    Npad = (N + 127) // 128 * 128
    tall = Npad % 256 == 0                 # launch_post's rule for large grids (csrc/posterior.hip)
    cand = 128 if tall else 256            # candidates per workgroup
    nb = min(4096, M // cand)
    buf = (C.c_ulonglong * (4 * nb))()
    rc = L.b7dbg_post_stamps(buf, nb)
    assert rc == 0, rc
    a = np.frombuffer(buf, dtype=np.uint64).reshape(nb, 4).astype(np.float64)
    cyc, ticks = a[:, 2] - a[:, 0], a[:, 3] - a[:, 1]
    clk = cyc / ticks * 100e6
    # MFMAs per SIMD (= per wave) per workgroup: 128 per full 16-deep stage; in the diagonal block 16 x (strips left) per
    # stage.  128 x 256 shape: n-tiles of 8 strips x 4 candidate strips; tall shape: 16 strips x 2
    if tall:
        mfma = sum(t * 16 * 128 + 4 * 2 * sum(16 - q for q in range(16)) for t in range(Npad // 256))
    else:
        mfma = sum(t * 8 * 128 + 4 * 4 * sum(8 - q for q in range(8)) for t in range(Npad // 128))
    need = mfma * 64.0
    print("launches before the stamped one: %d; fit + predict wall of the stamped pass %.3f ms" % (n, wall_ms))
    print("shape: %s, %d candidates per workgroup, %d workgroups stamped" % ("256-row n-tiles" if tall else "128-row n-tiles", cand, nb))
    print("in-kernel clock: median %.3f GHz (min %.3f, max %.3f) over %d workgroups" % (np.median(clk) / 1e9, clk.min() / 1e9, clk.max() / 1e9, nb))
    print("workgroup duration: median %.1f us = %.3e cycles; MFMA-only cycles %.3e -> issue efficiency %.3f"
          % (np.median(ticks) / 100.0, np.median(cyc), need, need / np.median(cyc)))
    start = a[:, 1] - a[:, 1].min()
    end = a[:, 3] - a[:, 1].min()
    print("kernel span by stamps: %.3f ms; rounds: first-start spread %.1f us, last end %.3f ms" % (end.max() / 1e5, np.percentile(start, 20) / 100.0, end.max() / 1e5))
    ctx.close()


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "build":
        build()
    else:
        run(*[int(x) for x in sys.argv[2:5]])
