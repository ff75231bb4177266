"""Builds the host pipeline against the CPU device stand-in (tests/devsim/agc_hip_sim.c) into tests/devsim/_build/.
TEST INFRASTRUCTURE ONLY: the same host sources as agc_amd/build.py:build_host(), linked with the simulator instead of
libagc_hip.so, so host logic is testable without a GPU.  The product build never touches this directory."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT = os.environ.get("AGC_DEVSIM_OUT") or os.path.join(HERE, "_build")  # (override: a scratch build next to a running fuzz campaign)
HOST = os.path.join(ROOT, "agc_amd", "csrc", "host")
SIM_HIP = os.path.join(OUT, "libagc_hip.so")
SIM_HOST = os.path.join(OUT, "libagc_host.so")
SIM_CLI = os.path.join(OUT, "agc_amd_sim")


def _newer(target, deps):
    return os.path.exists(target) and all(os.path.getmtime(d) <= os.path.getmtime(target) for d in deps)


class _Locked:
    """one build at a time (pytest-xdist workers may all find the stand-in stale at once)"""

    def __enter__(self):
        import fcntl
        os.makedirs(OUT, exist_ok=True)
        self.f = open(os.path.join(OUT, ".lock"), "w")
        fcntl.flock(self.f, fcntl.LOCK_EX)

    def __exit__(self, *a):
        import fcntl
        fcntl.flock(self.f, fcntl.LOCK_UN)
        self.f.close()


def build(force=False):
    with _Locked():
        return _build(force)


def _build(force=False):
    os.makedirs(OUT, exist_ok=True)
    sim_src = [os.path.join(HERE, "agc_hip_sim.c"), os.path.join(ROOT, "oracle", "agc_oracle.c")]
    hdr = os.path.join(ROOT, "include", "agc_hip.h")
    zsim = os.path.join(HERE, "zstd_sim.cpp")
    zdeps = [zsim] + [os.path.join(ROOT, "agc_amd", "csrc", "zstd", h) for h in ("zs_common.h", "zs_opt.h", "zs_opt_sm.h", "zs_entropy.h", "zs_frame.h", "zs_params.h")]
    if force or not _newer(SIM_HIP, sim_src + [hdr] + zdeps):
        objs = []
        for s_ in sim_src:
            o = os.path.join(OUT, os.path.basename(s_) + ".o")
            subprocess.check_call(["gcc", "-O2", "-std=c11", "-fPIC", "-Wall", "-c", s_, "-o", o])
            objs.append(o)
        zo = os.path.join(OUT, "zstd_sim.o")
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-Wall", "-c", zsim, "-o", zo])
        subprocess.check_call(["g++", "-shared"] + objs + [zo, "-o", SIM_HIP])
    host_src = [os.path.join(HOST, s) for s in ("compressor.cpp", "compressor_batch.cpp", "compressor_dist.cpp", "capi_host.cpp", "reader.cpp")]
    host_dep = host_src + [os.path.join(HOST, s) for s in ("compressor.h", "compressor_impl.h", "host_support.h", "reader.h", "archive_read.h")] + [SIM_HIP]
    common = ["g++", "-O2", "-std=c++17", "-fPIC", "-Wall", "-pthread"]
    if force or not _newer(SIM_HOST, host_dep):
        subprocess.check_call(common + ["-shared"] + host_src + ["-o", SIM_HOST, "-L" + OUT, "-lagc_hip", "-Wl,-rpath,$ORIGIN", "-lz", "-ldl"])
    cli_src = [os.path.join(HOST, "main.cpp")]
    if force or not _newer(SIM_CLI, cli_src + [os.path.join(HOST, "reader.h"), SIM_HOST]):
        subprocess.check_call(common + cli_src + ["-o", SIM_CLI, "-L" + OUT, "-lagc_host", "-lagc_hip", "-Wl,-rpath,$ORIGIN", "-lz", "-ldl"])
    return SIM_CLI


SIM_TWO = os.path.join(OUT, "two_ranks_one_process")


def build_two_ranks(force=False):
    """tests/devsim/two_ranks_one_process.cpp (two ranks of the multi-GPU mode in one process) against the stand-in libraries"""
    build(force)
    src = os.path.join(HERE, "two_ranks_one_process.cpp")
    with _Locked():
        if force or not _newer(SIM_TWO, [src, SIM_HOST]):
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", src, "-o", SIM_TWO, "-L" + OUT, "-lagc_host", "-lagc_hip",
                                   "-Wl,-rpath,$ORIGIN", "-lz", "-ldl"])
    return SIM_TWO


if __name__ == "__main__":
    print(build(force=True))
