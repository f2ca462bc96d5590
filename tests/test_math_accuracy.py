"""emei_amd/csrc/emei_math.h compiled for the host (the same header the kernels include) against
long-double libm: the fast sincos must stay within ~1 ulp absolute."""
import os
import re
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fast_sincos_and_division_accuracy():
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, "math_acc")
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-mfma", "-o", exe,
                               os.path.join(ROOT, "tests", "host", "math_accuracy.cpp")])
        out = subprocess.run([exe], capture_output=True, text=True, check=True).stdout
    f64 = [float(x) for x in re.findall(r"f64 .*maxabs ([0-9.e+-]+)", out)]
    tab = [float(x) for x in re.findall(r"tab .*maxabs ([0-9.e+-]+)", out)]
    f32 = [float(x) for x in re.findall(r"f32 .*maxabs ([0-9.e+-]+)", out)]
    div = float(re.search(r"div maxrel ([0-9.e+-]+)", out).group(1))
    assert len(f64) == 7 and max(f64) <= 2.0e-16, out
    assert len(tab) == 6 and max(tab) <= 2.0e-16, out  # the 256-entry table variant the float64 kernels use
    assert len(f32) == 5 and max(f32) <= 1.0e-7, out
    assert div <= 2.3e-16, out
    assert "inf -> nan nan" in out and "nan -> nan nan" in out
