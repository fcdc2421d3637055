"""Builds the RCCL test double (tests/stub/rccl_shm_stub.cpp -> tests/stub/_build/librccl_shm_stub.so): several ranks on ONE GPU,
in several processes (shared memory) or in one (ncclCommInitAll).  Test infrastructure; loaded through B7_RCCL_LIB."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def stub_lib():
    src = os.path.join(ROOT, "tests", "stub", "rccl_shm_stub.cpp")
    out = os.path.join(ROOT, "tests", "stub", "_build", "librccl_shm_stub.so")
    if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        os.makedirs(os.path.dirname(out), exist_ok=True)
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "-std=c++17", "-fPIC", "-shared", "-I/opt/rocm/include", "-o", out, src,
                               "-Wl,-rpath,/opt/rocm/lib", "-lrt"])
    return out


if __name__ == "__main__":
    print(stub_lib())
