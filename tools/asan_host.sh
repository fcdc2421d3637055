#!/bin/bash
# Host-side AddressSanitizer + UBSan build of libbot7hip.so (device code as always: -Xarch_host keeps the sanitizer off the
# gfx950 compilation; GPU sanitizers are not available on this pool) and the CPU test files that go through the library
# under it.  CPU container only.   usage: bash tools/asan_host.sh
set -eu
R=$(cd "$(dirname "$0")/.." && pwd)
cd $R
python - <<'PY'
import os, subprocess, sys
sys.path.insert(0, os.getcwd())
from bot7_amd import build as B
from concurrent.futures import ThreadPoolExecutor
os.makedirs("tools/_build/asan", exist_ok=True)
def comp(src):
    obj = "tools/_build/asan/" + src.replace(".hip", ".o")
    subprocess.check_call([B.HIPCC] + B.FLAGS + B.EXTRA_FLAGS.get(src, []) + ["-Xarch_host", "-fsanitize=address,undefined", "-Xarch_host",
                          "-fno-omit-frame-pointer", "-g", "-c", os.path.join(B.CSRC, src), "-o", obj])
    return obj
with ThreadPoolExecutor(4) as ex:
    objs = list(ex.map(comp, B.SOURCES))
subprocess.check_call([B.HIPCC, "-shared", "-fPIC", "--offload-arch=gfx950", "-fsanitize=address,undefined", "-shared-libsan", "-o",
                       "tools/_build/libbot7hip_asan.so"] + objs + ["-ldl", "-Wl,-rpath,/opt/rocm/lib"])
PY
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:protect_shadow_gap=0 BOT7HIP_LIB=$R/tools/_build/libbot7hip_asan.so \
  python -m pytest tests/test_abi_and_host.py tests/test_sharded_loop.py tests/test_dist_gloo.py -x -q -m "not gpu" -p no:cacheprovider
