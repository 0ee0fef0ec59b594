"""The C++ host layer (include/qfall_crypto.hpp: the compiled-language mirror of the reference's
trait API above the C ABI) exercised by tests/cpp/test_host.cpp: reference-style round trips and
panics plus bit-exact comparisons with the oracle.  The binary is built by
__graft_entry__.build() (g++, links libqfgpu.so and the oracle) and travels to the GPU box."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "cpp", "test_host")


@pytest.mark.gpu
def test_cpp_host_layer():
    assert os.path.exists(BIN), "tests/cpp/test_host missing: run __graft_entry__.build()"
    p = subprocess.run([BIN], capture_output=True, text=True, timeout=300)
    print(p.stdout[-4000:], p.stderr[-2000:])
    assert p.returncode == 0, p.stdout[-4000:] + p.stderr[-2000:]
    assert "0 failures" in p.stdout


def test_cpp_host_header_compiles():
    """CPU-side: the header is self-contained C++17 and only needs the C ABI header."""
    src = '#include "qfall_crypto.hpp"\nint main() { return qfall::log2_ceil(1073479681) == 30 ? 0 : 1; }\n'
    out = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), "-x", "c++", "-"],
                         input=src, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
