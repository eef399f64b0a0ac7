"""CPU-only checks of the boundary: the library loads, exports every symbol the header declares,
struct sizes match, and it refuses to run without a GPU (no silent CPU fallback)."""
import ctypes
import os
import re
import subprocess

import pytest

from solver2d_amd import hip, wire

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "solver2d_amd.h")


def declared_functions():
    text = open(HEADER).read()
    return sorted(set(re.findall(r"\b(s2amd_[a-z_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    L = hip.load()
    names = declared_functions()
    assert len(names) >= 15
    for n in names:
        assert hasattr(L, n), n
    assert sorted(names) == sorted(hip.EXPORTS)
    assert L.s2amd_api_version() == wire.API_VERSION


def test_tolerance_mode_library_exports_the_same_abi_and_says_what_it_is():
    """solver2d_amd/libs2amd_fast.so (the same sources, FMA contraction on in the device code): same symbols, same version; the two
    builds tell themselves apart (s2amd_build_flags).  (The device code really differs: wideStepKernel's four main variants hold 512
    v_pk_fma_f32 and 2,000 more v_fma_f32 in the contracted build, none of the former and only the divide / sqrt expansions' of the
    latter in the bit-exact one -- hipcc -S, round 5.)"""
    fast, exact = hip.load(fast=True), hip.load()
    assert fast is not exact
    for n in declared_functions():
        assert hasattr(fast, n), n
    assert fast.s2amd_api_version() == wire.API_VERSION
    assert exact.s2amd_build_flags() == b"fp-contract=off" and fast.s2amd_build_flags() == b"fp-contract=fast"
    # the S2AMD_OPTIONS passthrough names a bad entry instead of raising a bare ValueError
    os.environ["S2AMD_OPTIONS"] = "graph"
    try:
        with pytest.raises(hip.S2AmdError) as e:
            hip.env_options()
        assert "graph" in str(e.value)
        os.environ["S2AMD_OPTIONS"] = "graph=0, strip_patience=2"
        assert hip.env_options() == [("graph", 0), ("strip_patience", 2)]
    finally:
        del os.environ["S2AMD_OPTIONS"]


def test_struct_sizes_match_header(tmp_path):
    src = tmp_path / "sz.c"
    src.write_text('#include "solver2d_amd.h"\n#include <stdio.h>\nint main(){printf("%zu %zu %zu %zu %zu %zu\\n",'
                   'sizeof(s2amdBody),sizeof(s2amdManifoldPoint),sizeof(s2amdContact),sizeof(s2amdJoint),'
                   'sizeof(s2amdStepParams),sizeof(s2amdStepStats));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    sizes = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    assert sizes == [wire.BODY_SIZE, wire.manifold_point_dtype.itemsize, wire.CONTACT_SIZE, wire.JOINT_SIZE,
                     ctypes.sizeof(wire.StepParams), ctypes.sizeof(wire.StepStats)]


def test_solver_enum_is_reference_abi():
    # include/solver2d/types.h:75-88 of the reference
    assert wire.SOLVER_NAMES == ["Jacobi", "PGS", "PGS_NGS", "PGS_NGS_Block", "PGS_Soft", "SoftStep",
                                 "TGS_Sticky", "TGS_Soft", "TGS_NGS", "XPBD"]


def test_fails_loudly_without_gpu():
    if hip.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(hip.S2AmdError) as e:
        hip.Solver(0)
    assert "no CPU path" in str(e.value) or "-3" in str(e.value)


def test_sweep_accounting():
    assert wire.solve_sweeps_per_step("TGS_Soft", 8, 4) == 16
    assert wire.solve_sweeps_per_step("TGS_Soft", 8, 0) == 8
    assert wire.solve_sweeps_per_step("Jacobi", 4, 2) == 6
    assert wire.solve_sweeps_per_step("PGS", 4, 2) == 4


def test_header_is_plain_c_and_cpp(tmp_path):
    """include/solver2d_amd.h is the boundary a C host (the reference is C17) and a C++ host both include: it must compile
    alone, as strict C99 and as C++, with warnings as errors, and pull in nothing but <stdint.h> (no HIP / torch / C++ type can cross it)."""
    import subprocess
    header = os.path.join(ROOT, "include", "solver2d_amd.h")
    includes = [line.strip() for line in open(header) if line.lstrip().startswith("#include")]
    assert includes == ["#include <stdint.h>"], includes
    for compiler, std, name in (("gcc", "-std=c99", "t.c"), ("g++", "-std=c++17", "t.cpp")):
        src = tmp_path / name
        src.write_text('#include "solver2d_amd.h"\nint main(void) { return (int)sizeof(s2amdBody) == 0; }\n')
        subprocess.run([compiler, std, "-pedantic", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-I", os.path.dirname(header), str(src)],
                       check=True, capture_output=True)
