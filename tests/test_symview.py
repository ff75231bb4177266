"""CPU check of the symbol accessors of the 2-bit layout (agc_amd/csrc/sym_view.h, the header the LZ kernels read samples and
references through): tests/symview_host/sv_host.cpp compares every accessor with plain byte arrays -- both orientations, escaped
blocks (N runs, IUPAC), sequence ends, keys.  The same header is compiled for gfx950 by agc_amd/build.py."""
import os
import subprocess
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))


def test_sym_view_accessors_match_byte_arrays():
    src = os.path.join(HERE, "symview_host", "sv_host.cpp")
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, "sv_host")
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wall", src, "-o", exe])
        for seed in (1, 7, 42):
            out = subprocess.run([exe, str(seed)], capture_output=True, text=True)
            assert out.returncode == 0 and out.stdout.startswith("ok "), out.stdout + out.stderr
