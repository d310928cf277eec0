"""The drop-in boundary is a C ABI: tests/c_client/client.c is a strict-C99 program (no Python, no torch,
no C++) that includes include/triro_hip.h, links libtriro_hip.so and runs the reference's own test inputs
(test/test.py:6-13, 47-64) through every query family.  CPU suite: it compiles and links with
-pedantic -Wall -Wextra -Werror (the header is clean C, every symbol it uses is exported).  GPU suite: it
runs and checks the hand-derived answers of tests/golden/known_answers.json."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "c_client", "client.c")
OUT = os.path.join(ROOT, "tests", "c_client", "_build", "client")
LIBDIR = os.path.join(ROOT, "trimesh-ray-optix_amd", "lib")
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")


def build_client():
    if not os.path.exists(os.path.join(LIBDIR, "libtriro_hip.so")):
        import __graft_entry__
        __graft_entry__.build()
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    cmd = ["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-D__HIP_PLATFORM_AMD__",
           "-I", os.path.join(ROOT, "include"), "-isystem", os.path.join(ROCM, "include"), SRC, "-o", OUT,
           "-L", LIBDIR, "-ltriro_hip", "-L", os.path.join(ROCM, "lib"), "-lamdhip64", "-lm",
           "-Wl,-rpath," + LIBDIR, "-Wl,-rpath," + os.path.join(ROCM, "lib")]
    p = subprocess.run(cmd, capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    return OUT


def test_c_client_compiles_and_links_as_strict_c99():
    exe = build_client()
    assert os.path.exists(exe)
    # nothing but the C ABI and the HIP runtime: no libtorch, no libpython, no C++ runtime of its own
    needed = subprocess.run(["readelf", "-d", exe], capture_output=True, text=True).stdout
    libs = [ln.split("[")[1].split("]")[0] for ln in needed.splitlines() if "(NEEDED)" in ln]
    assert any(x.startswith("libtriro_hip") for x in libs) and any(x.startswith("libamdhip64") for x in libs)
    assert not any(("torch" in x) or ("python" in x) or ("stdc++" in x) for x in libs), libs


@pytest.mark.gpu
def test_c_client_runs_the_reference_test_inputs_through_the_c_abi():
    exe = build_client()
    p = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "C CLIENT OK" in p.stdout
