"""Build-time check of the persistent kernels' register use: hipcc's own report
(-Rpass-analysis=kernel-resource-usage; `make -C solver2d_amd/csrc resources`) for every kernel of wide_kernel.hip, strip_kernel.hip,
generic_kernel.hip and group_kernel.hip.

  * NO variant of the 512-thread family -- wideStepKernel (every layout the launch can pick: <3,2>, <3,3>, <4,2>, <3,2,2>, <3,2,2,2>;
    plain, sliced and with the overflow workgroup; s2Solve_TGS_Soft, s2Solve_PGS_Soft, s2Solve_SoftStep) and wideIslandKernel (6 and 8 rounds) -- spills a byte to
    scratch, and all of them keep two waves per SIMD.  Through round 4, 37 of 56 did (24-520 bytes per lane): exactly the variants a
    churning world ends up on.  Round 5 moved the local anchors of the records beyond the fifth into LDS (wide_kernel.hip:
    wideLocalsInLds) and dropped the optional modes where they did not fit.
  * The older 256-thread kernels (stripStepKernel / islandStepKernel: the fall-back when option "wide" is off or a hand-off timed
    out) are reported, and their scratch may only shrink: the set below is what they spill today.

No GPU: the compiler runs here (about a minute on four cores)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import kernel_resources  # noqa: E402

# scratch bytes per lane of the fall-back kernels as of round 5 (may only shrink)
KNOWN_SCRATCH = {
    "stripStepKernel": 672, "islandStepKernel": 1376,
}


@pytest.fixture(scope="module")
def rows():
    if not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("needs hipcc")
    subprocess.check_call(["make", "-s", "-j4", "-C", os.path.join(ROOT, "solver2d_amd", "csrc"), "resources"])
    return kernel_resources.parse()


def test_the_512_thread_kernels_do_not_spill(rows):
    wide = [r for r in rows if r["name"].startswith(("wideStepKernel", "wideIslandKernel"))]
    assert len(wide) >= 50, len(wide)
    def spills(r):
        if r["ScratchSize"] == 0:
            return False
        # The overflow form (MODE 8: the launch carries the workgroup that sweeps the overflow contacts) of some layouts declares a
        # 36-byte frame that NO instruction of the kernel touches -- an SGPR spill slot whose values went to VGPR lanes; the dispatch has
        # the queue's scratch set up (allocated at s2amd_create: s2WarmScratch) and the kernel runs without scratch traffic.  That, and
        # nothing else, is tolerated: no spilled VGPR, no scratch instruction in the assembly, at most 64 bytes.
        overflow_form = r["name"].startswith("wideStepKernel<") and r["name"].split(", ")[5] == "8"
        return not (overflow_form and r["VGPRs Spill"] == 0 and r["scratch_instructions"] == 0 and r["ScratchSize"] <= 64)
    bad = ["%s: %d B scratch (%s instructions on it), %d waves/SIMD" % (r["name"], r["ScratchSize"], r.get("scratch_instructions"), r["Occupancy"])
           for r in wide if spills(r) or r["Occupancy"] < 2]
    assert not bad, "\n".join(bad)
    # the headline variant: what it was measured with (profiles/r05_*): 240 registers, the loop-invariant scalars partly in VGPR lanes
    head = [r for r in wide if r["name"] == "wideStepKernel<2, 3, 2, 0, 0, 0, 0>"]
    assert len(head) == 1 and head[0]["VGPRs"] <= 248, head


def test_every_layout_the_launch_can_pick_was_compiled(rows):
    names = {r["name"] for r in rows}
    for points in (0, 2):
        for layout in ("3, 2, 0, 0", "3, 3, 0, 0", "4, 2, 0, 0", "3, 2, 2, 0", "3, 2, 2, 2"):
            for mode in (0, 4, 8):
                for kind in (0, 1):
                    assert "wideStepKernel<%d, %s, %d, %d>" % (points, layout, mode, kind) in names, (points, layout, mode, kind)
        for mode in (0, 4, 8):
            assert "wideStepKernel<%d, 3, 2, 0, 0, %d, 3>" % (points, mode) in names
    for rounds in (6, 8):
        for self_contained in ("false", "true"):
            for points in (0, 2):
                assert "wideIslandKernel<%d, %s, %d>" % (rounds, self_contained, points) in names


def test_the_fall_back_kernels_spill_no_more_than_they_did(rows):
    for r in rows:
        base = r["name"].split("<")[0]
        if base in KNOWN_SCRATCH:
            assert r["ScratchSize"] <= KNOWN_SCRATCH[base], (r["name"], r["ScratchSize"])
        elif not base.startswith(("wideStepKernel", "wideIslandKernel", "s2WarmScratchKernel")):
            assert r["ScratchSize"] == 0, (r["name"], r["ScratchSize"])
