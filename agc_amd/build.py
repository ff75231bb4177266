"""In-tree build of the HIP library (gfx950 only).  `python -m agc_amd.build` or build()."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")


class _Locked:
    """one build at a time per checkout (test workers of one run may all find the libraries stale at once)"""

    def __enter__(self):
        import fcntl
        import hashlib
        import tempfile
        self.f = open(os.path.join(tempfile.gettempdir(), "agc_amd_build_%s.lock" % hashlib.sha1(HERE.encode()).hexdigest()[:12]), "w")
        try:
            fcntl.flock(self.f, fcntl.LOCK_EX)
        except BaseException:
            self.f.close()
            raise

    def __exit__(self, *a):
        import fcntl
        fcntl.flock(self.f, fcntl.LOCK_UN)
        self.f.close()
LIB = os.path.join(HERE, "libagc_hip.so")
SOURCES = ["api.hip", "scan_kernels.hip", "pack_kernels.hip", "lz_kernels.hip", "splitters.hip", "zstd_kernels.hip", "seg_kernels.hip", "segments.hip", "dev_common.h", "sym_view.h",
           "zstd/zs_common.h", "zstd/zs_opt.h", "zstd/zs_opt_sm.h", "zstd/zs_opt_grp.h", "zstd/zs_entropy.h", "zstd/zs_frame.h", "zstd/zs_params.h"]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(HERE, "..", "include", "agc_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 ... -> agc_amd/libagc_hip.so (cross-compiles without a GPU)."""
    with _Locked():
        if not force and not _stale():
            return LIB
        hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-Wall",
               os.path.join(CSRC, "api.hip"), "-o", LIB + ".tmp"]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd, cwd=CSRC)
        os.replace(LIB + ".tmp", LIB)  # (never a half-written library under the final name)
        return LIB


HOST = os.path.join(CSRC, "host")
HOST_LIB = os.path.join(HERE, "libagc_host.so")
HOST_BIN = os.path.join(HERE, "bin", "agc_amd")
HOST_SOURCES = ["compressor.cpp", "compressor_batch.cpp", "compressor_dist.cpp", "compressor_impl.h", "compressor.h", "host_support.h",
                "capi_host.cpp", "main.cpp", "reader.cpp", "reader.h", "capi_read.cpp", "archive_read.h"]
HOST_LIB_SOURCES = ["compressor.cpp", "compressor_batch.cpp", "compressor_dist.cpp", "capi_host.cpp", "reader.cpp"]
READ_LIB = os.path.join(HERE, "libagc_read.so")


def build_read(force=False, verbose=False):
    """g++ for the read side (include/agc_read.h): libagc_read.so -- host only, no HIP dependency."""
    srcs = [os.path.join(HOST, s) for s in ("reader.cpp", "capi_read.cpp")]
    deps = srcs + [os.path.join(HOST, "reader.h"), os.path.join(HOST, "archive_read.h"), os.path.join(HERE, "..", "include", "agc_read.h")]
    with _Locked():
        if not force and os.path.exists(READ_LIB) and all(os.path.getmtime(d) <= os.path.getmtime(READ_LIB) for d in deps):
            return READ_LIB
        cmd = [os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-fPIC", "-Wall", "-shared"] + srcs + ["-o", READ_LIB + ".tmp", "-ldl", "-pthread"]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
        os.replace(READ_LIB + ".tmp", READ_LIB)  # (never a half-written library under the final name)
        return READ_LIB


def build_host(force=False, verbose=False):
    """g++ for the host-side compressor (C++17): libagc_host.so (links libagc_hip.so by $ORIGIN rpath)
    and the agc-compatible CLI agc_amd/bin/agc_amd.  zstd is dlopen'ed at run time."""
    build()
    build_read(force, verbose)
    deps = [os.path.join(HOST, s) for s in HOST_SOURCES] + [LIB]
    with _Locked():
        stale = force or not os.path.exists(HOST_LIB) or not os.path.exists(HOST_BIN) or \
            any(os.path.getmtime(d) > min(os.path.getmtime(HOST_LIB), os.path.getmtime(HOST_BIN)) for d in deps)
        if not stale:
            return HOST_LIB
        os.makedirs(os.path.dirname(HOST_BIN), exist_ok=True)
        cxx = os.environ.get("CXX", "g++")
        common = [cxx, "-O2", "-std=c++17", "-fPIC", "-Wall", "-pthread"]
        # (linked under a temporary name and renamed: a process that only loads / executes never meets a half-written file)
        cmd1 = common + ["-shared"] + [os.path.join(HOST, s) for s in HOST_LIB_SOURCES] + ["-o", HOST_LIB + ".tmp",
                         "-L" + HERE, "-lagc_hip", "-Wl,-rpath,$ORIGIN", "-Wl,-soname,libagc_host.so", "-lz", "-ldl"]
        if verbose:
            print(" ".join(cmd1), file=sys.stderr)
        subprocess.check_call(cmd1)
        os.replace(HOST_LIB + ".tmp", HOST_LIB)
        cmd2 = common + [os.path.join(HOST, "main.cpp"), "-o", HOST_BIN + ".tmp", "-L" + HERE, "-lagc_host", "-lagc_hip",
                         "-Wl,-rpath,$ORIGIN/..", "-lz", "-ldl"]
        if verbose:
            print(" ".join(cmd2), file=sys.stderr)
        subprocess.check_call(cmd2)
        os.replace(HOST_BIN + ".tmp", HOST_BIN)
        return HOST_LIB


def write_build_log():
    """agc_amd/build.log: what was built with what -- compiler versions, the command lines' flags, size and sha256 of every
    artefact (the .so files are git-ignored and ship prebuilt with a snapshot: this is their record)"""
    import hashlib
    import time
    lines = ["# agc_amd build record (agc_amd/build.py: write_build_log)", "written: " + time.strftime("%Y-%m-%d %H:%M:%S UTC", time.gmtime())]
    for tool, arg in ((os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--version"), (os.environ.get("CXX", "g++"), "--version")):
        try:
            out = subprocess.run([tool, arg], capture_output=True, text=True, timeout=60).stdout.strip().splitlines()
            lines.append(f"{tool}: " + " | ".join(x.strip() for x in out[:3]))
        except Exception as e:  # noqa: BLE001
            lines.append(f"{tool}: {e}")
    lines.append("libagc_hip.so: hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Wall csrc/api.hip (one translation unit: it includes the *.hip files)")
    lines.append("libagc_host.so / bin/agc_amd / libagc_read.so: g++ -O2 -std=c++17 -fPIC -Wall -pthread csrc/host/*.cpp, linked against libagc_hip.so ($ORIGIN rpath), -lz -ldl")
    for path in (LIB, HOST_LIB, READ_LIB, HOST_BIN):
        if os.path.exists(path):
            h = hashlib.sha256(open(path, "rb").read()).hexdigest()
            lines.append(f"{os.path.relpath(path, HERE)}: {os.path.getsize(path)} bytes, sha256 {h}, mtime {time.strftime('%Y-%m-%d %H:%M:%S', time.gmtime(os.path.getmtime(path)))}")
    srcs = [os.path.join(CSRC, s_) for s_ in SOURCES] + [os.path.join(HOST, s_) for s_ in HOST_SOURCES]
    hs = hashlib.sha256()
    for p_ in sorted(srcs):
        hs.update(open(p_, "rb").read())
    lines.append(f"sources ({len(srcs)} files under csrc/): sha256 of their concatenation in name order {hs.hexdigest()}")
    open(os.path.join(HERE, "build.log"), "w").write("\n".join(lines) + "\n")


if __name__ == "__main__":
    build_host(force="--force" in sys.argv, verbose=True)
    print(build(force="--force" in sys.argv, verbose=True))
    write_build_log()
