"""Sanitizer job for the host code: solver_structure.cpp / solver_incremental.cpp / solver_step.cpp /
graph_coloring.cpp and the host halves of the .hip files, compiled --cuda-host-only with ASan + UBSan against a stand-in HIP
runtime (tests/hostcheck/hip_stub.cpp: device memory is heap memory, kernels never run), driven through random worlds, option
mixes and contact churn (tests/hostcheck/drive.py).  No GPU; a sanitizer report fails the test."""
import glob
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HERE = os.path.join(ROOT, "tests", "hostcheck")


def _asan_runtime():
    hits = sorted(glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so"))
    return hits[-1] if hits else None


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc") or _asan_runtime() is None, reason="needs hipcc and clang's ASan runtime")
def test_host_code_under_asan_and_ubsan():
    subprocess.check_call(["make", "-s", "-j8", "-C", HERE])
    env = dict(os.environ)
    env["LD_PRELOAD"] = _asan_runtime()
    env["ASAN_OPTIONS"] = "detect_leaks=0:abort_on_error=0:exitcode=23"
    env["UBSAN_OPTIONS"] = "print_stacktrace=1:halt_on_error=1:exitcode=24"
    env["S2AMD_LIB"] = os.path.join(HERE, "_build", "libs2amd_hostcheck.so")
    p = subprocess.run([sys.executable, os.path.join(HERE, "drive.py"), "quick"], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    out = p.stdout.decode(errors="replace")
    assert p.returncode == 0 and "HOSTCHECK OK" in out and "AddressSanitizer" not in out and "runtime error" not in out, out[-4000:]
