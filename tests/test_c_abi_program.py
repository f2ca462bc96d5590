"""include/emei_hip.h is a C header and libemei_hip.so is usable from plain C (no Python, no torch types):
tests/host/c_abi_smoke.c is compiled with gcc -std=c99 here (CPU: header + link check), and run on the GPU."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "host", "c_abi_smoke.c")
LIBDIR = os.path.join(ROOT, "emei_amd")


def _build(tmp_path):
    exe = str(tmp_path / "c_abi_smoke")
    cmd = ["gcc", "-std=c99", "-Wall", "-Werror", "-O1", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROOT, "include"),
           "-I", "/opt/rocm/include", SRC, "-o", exe, "-L", LIBDIR, "-lemei_hip", "-L", "/opt/rocm/lib", "-lamdhip64", "-lm",
           f"-Wl,-rpath,{LIBDIR}", "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.check_call(cmd)
    return exe


def test_header_is_c_and_library_links(tmp_path):
    if not os.path.exists(os.path.join(LIBDIR, "libemei_hip.so")):
        pytest.skip("libemei_hip.so not built")
    exe = _build(tmp_path)
    out = subprocess.run([exe, "--link-only"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "LINK OK" in out.stdout, out.stdout + out.stderr


@pytest.mark.gpu
def test_c_program_runs_the_hot_path(tmp_path):
    exe = _build(tmp_path)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "C ABI OK" in out.stdout, out.stdout + out.stderr
