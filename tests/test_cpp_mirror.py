"""The C++ host-side mirror of the reference's interface (include/impact_voxel.hpp, header-only above the C ABI): it must compile as
plain C++17 with g++ and link against the shared library (every entry point it forwards to exists); on the GPU box the parity program
written against it (tests/cpp/host_mirror_check.cpp: generate, mesh, moments, absorbing sphere, incremental remesh — against the
oracle's C API) must report everything equal."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "cpp", "host_mirror_check")


def build():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "impact_amd", "csrc"), "-j8"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "cpp")], stdout=subprocess.DEVNULL)


def test_mirror_header_compiles_and_links():
    build()
    assert os.path.exists(BIN)
    # the header alone, with warnings as errors
    src = '#include "impact_voxel.hpp"\\nint main() { impact_voxel::SDFGraph g; g.add_node(impact_voxel::SDFNode::new_sphere(1.0f)); return (int)g.nodes().size() - 1; }\\n'
    subprocess.run(["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), "-x", "c++", "-"],
                   input=src.encode().decode("unicode_escape").encode(), check=True)


@pytest.mark.gpu
def test_parity_program_written_against_the_mirror():
    if not os.path.exists(BIN):
        build()
    r = subprocess.run([BIN], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "all equal" in r.stdout, r.stdout[-4000:] + r.stderr[-2000:]
