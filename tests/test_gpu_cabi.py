"""The C ABI used the way the reference's maintainers would bind it from native code: a C++
program with no Python and no torch in the process (tests/cabi/cabi_harness.cpp), compiled
against include/asw_hip.h and linked to libasw_hip.so."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cabi_harness_runs_without_python():
    from acousticswarms_speech_amd import native
    native.lib()                       # the prebuilt in-tree library (never rebuilt under a running process)
    pkg = os.path.dirname(native.LIB_PATH)
    exe = os.path.join(ROOT, "tests", "cabi", "cabi_harness")
    subprocess.run(["/opt/rocm/bin/hipcc", "-O2", "-std=c++17", os.path.join(ROOT, "tests", "cabi", "cabi_harness.cpp"),
                    "-I", os.path.join(ROOT, "include"), "-L", pkg, "-lasw_hip", f"-Wl,-rpath,{pkg}", "-o", exe],
                   check=True, capture_output=True, timeout=300)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    print(out.stdout)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "CABI OK" in out.stdout
