"""Builds tests/zstd_host/zs_host.cpp (the zstd level-17 encoder headers of agc_amd/csrc/zstd/ for the HOST) into
tests/zstd_host/_build/libzs_host.so.  TEST INFRASTRUCTURE ONLY: lets the CPU suite compare the encoder with libzstd 1.4.9;
the product compiles the same headers with hipcc and never loads this library."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT = os.path.join(HERE, "_build")
LIB = os.path.join(OUT, "libzs_host.so")


def build(force=False):
    import fcntl
    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, ".lock"), "w") as lock:  # (one build at a time: pytest-xdist workers)
        fcntl.flock(lock, fcntl.LOCK_EX)
        return _build(force)


def _build(force=False):
    src = os.path.join(HERE, "zs_host.cpp")
    deps = [src] + [os.path.join(ROOT, "agc_amd", "csrc", "zstd", h) for h in ("zs_common.h", "zs_opt.h", "zs_opt_sm.h", "zs_opt_grp.h", "zs_entropy.h", "zs_frame.h")]
    if force or not os.path.exists(LIB) or any(os.path.getmtime(d) > os.path.getmtime(LIB) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unknown-pragmas", src, "-o", LIB])
    return LIB


if __name__ == "__main__":
    print(build(force=True))
