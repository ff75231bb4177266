#!/bin/bash
# ThreadSanitizer run of the host pipeline's threads (the thread that drives the steps, the bookkeeping thread, the entropy
# thread and its pool) on the CPU device stand-in: (1) the CLI with every file a window of its own (bookkeeping queued beside the
# next file), (2) two ranks of the multi-GPU mode in one process (tests/devsim/two_ranks_one_process.cpp).  Archives are compared
# with tests/golden/archives.json.  Scratch build under $1 (default /tmp/agc_tsan); libzstd is opened without RTLD_DEEPBIND there
# (the sanitizer runtime refuses it).   usage: scripts/tsan_check.sh [scratch dir]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
T=${1:-/tmp/agc_tsan}
mkdir -p $T/src
python -c "import sys; sys.path.insert(0,'$ROOT'); from tests.devsim import build; build.build()"
cp $ROOT/tests/devsim/_build/libagc_hip.so $T/
cp $ROOT/agc_amd/csrc/host/* $T/src/
sed -i 's/ | RTLD_DEEPBIND//' $T/src/host_support.h $T/src/archive_read.h
sed -i "s#\"../../../include/#\"$ROOT/include/#" $T/src/*.cpp $T/src/*.h
sed "s#\"../../agc_amd/csrc/host/compressor_impl.h\"#\"$T/src/compressor_impl.h\"#; s#\"../../include/agc_hip.h\"#\"$ROOT/include/agc_hip.h\"#" $ROOT/tests/devsim/two_ranks_one_process.cpp > $T/tr.cpp
CXX="g++ -O1 -g -fsanitize=thread -std=c++17 -pthread"
H=$T/src
$CXX -fPIC -shared $H/compressor.cpp $H/compressor_batch.cpp $H/compressor_dist.cpp $H/capi_host.cpp $H/reader.cpp -o $T/libagc_host.so -L$T -lagc_hip -Wl,-rpath,'$ORIGIN' -lz -ldl
$CXX $H/main.cpp -o $T/agc_tsan -L$T -lagc_host -lagc_hip -Wl,-rpath,'$ORIGIN' -lz -ldl
$CXX $T/tr.cpp -o $T/two_ranks_tsan -L$T -lagc_host -lagc_hip -Wl,-rpath,'$ORIGIN' -lz -ldl
cd $ROOT
python - "$T" <<'PY'
import hashlib, json, os, subprocess, sys
sys.path.insert(0, os.getcwd())
from tests import collections as C
T = sys.argv[1]
GOLD = json.load(open("tests/golden/archives.json"))
bad = 0
for name in ["syn_c3_twin", "syn_mixed", "syn_adaptive", "syn_c4_twin", "syn_snp"]:
    args, _ = C.CONFIGS[name]
    opt = {"-k": 31, "-l": 20, "-s": 60000, "-b": 50}
    for i in range(len(args) - 1):
        if args[i] in opt:
            opt[args[i]] = int(args[i + 1])
    files = C.build(name, os.path.join(T, "in_" + name))
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0", AGC_AMD_PAR_MIN="8", AGC_AMD_WINDOW_MAX="1", AGC_AMD_LAPS="1")
    out = os.path.join(T, "cli_%s.agc" % name)
    r = subprocess.run([os.path.join(T, "agc_tsan"), "create"] + args + ["-t", "4", "-o", out] + files, capture_output=True, text=True, env=env)
    same = hashlib.sha256(open(out, "rb").read()).hexdigest() == GOLD[name]["sha256"]
    w = r.stderr.count("WARNING: ThreadSanitizer")
    print(f"cli        {name}: {len(files)} files, bookkeeping queued {r.stderr.count('(queued)')} times, archive identical to the reference's: {same}, ThreadSanitizer warnings: {w}", flush=True)
    bad += (not same) + w
    out = os.path.join(T, "two_%s.agc" % name)
    env.pop("AGC_AMD_LAPS")
    r = subprocess.run([os.path.join(T, "two_ranks_tsan"), out] + [str(opt[x]) for x in ("-k", "-l", "-s", "-b")] + ["1" if "-a" in args else "0"] + files,
                       capture_output=True, text=True, env=env)
    same = r.returncode == 0 and hashlib.sha256(open(out, "rb").read()).hexdigest() == GOLD[name]["sha256"]
    w = r.stderr.count("WARNING: ThreadSanitizer")
    print(f"two ranks  {name}: rc {r.returncode}, archive identical to the reference's: {same}, ThreadSanitizer warnings: {w}", flush=True)
    bad += (not same) + w
sys.exit(1 if bad else 0)
PY
