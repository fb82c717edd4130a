"""Runs the C++ test program of the host-side mirror of the reference's decoder interfaces
(parquet-go_amd/host/pq_host.hpp) on the GPU.  The program is built on the CPU beforehand
(`make -C tests/host`, also done by __graft_entry__.build())."""
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
BINARY = os.path.join(HERE, "host", "host_mirror_test")


def test_host_mirror_header_is_self_contained():
    """The mirror header must compile on its own against the public C header (no GPU)."""
    root = os.path.dirname(HERE)
    src = '#include "pq_host.hpp"\nint main() { pqhost::PackedArray a; return a.Reset(3).status; }\n'
    subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Werror", "-x", "c++", "-",
                    "-I", os.path.join(root, "include"), "-I", os.path.join(root, "parquet-go_amd", "host")],
                   input=src.encode(), check=True)


@pytest.mark.gpu
def test_host_mirror_program():
    assert os.path.exists(BINARY), "build it first: make -C tests/host"
    p = subprocess.run([BINARY], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    out = p.stdout.decode(errors="replace")
    print(out)
    assert p.returncode == 0, out[-4000:]
    assert "0 failed" in out
