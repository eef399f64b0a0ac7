"""Table of the compiler's register report (make -C solver2d_amd/csrc resources -> build/resources.txt):
kernel, VGPRs, scratch bytes per lane, occupancy.  tests/test_kernel_resources.py reads the same file.
Usage: python tools/kernel_resources.py [--scratch-only]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REPORT = os.path.join(ROOT, "solver2d_amd", "csrc", "build", "resources.txt")


def parse(path=REPORT):
    rows, cur = [], None
    for line in open(path):
        m = re.search(r"remark:\s+(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|VGPRs Spill|SGPRs Spill|TotalSGPRs): (\S+)", line)
        if not m:
            continue
        key, val = m.group(1), m.group(2)
        if key == "Function Name":
            cur = {"mangled": val}
            rows.append(cur)
        elif cur is not None:
            cur[key.split(" [")[0]] = int(val)
    names = subprocess.run(["c++filt"] + [r["mangled"] for r in rows], capture_output=True, text=True, check=True).stdout.splitlines()
    for r, n in zip(rows, names):
        r["name"] = re.sub(r"\(.*", "", n.replace("void ", ""))
    scratch_instructions(rows, os.path.dirname(path))
    return rows


def scratch_instructions(rows, build_dir):
    """For every kernel that declares a private frame: how many instructions of its body touch it (scratch_load / scratch_store, or the
    buffer forms on the scratch descriptor) -- from the device assembly `make resources` leaves beside the report.  A frame without any is
    a stack object the register allocator never used (an SGPR spill slot that went to VGPR lanes): the kernel runs without scratch
    traffic, the dispatch only has the queue's scratch set up.  None: the assembly is not there."""
    wanted = {r["mangled"]: r for r in rows if r.get("ScratchSize", 0) > 0}
    for r in wanted.values():
        r["scratch_instructions"] = None
    if not wanted:
        return
    touch = re.compile(r"^\s+(scratch_(load|store)|buffer_(load|store)\S*\s.*s\[0:3\])")
    for f in sorted(os.listdir(build_dir)):
        if not (f.startswith("asm_") and f.endswith(".s")):
            continue
        cur = None
        for line in open(os.path.join(build_dir, f), errors="replace"):
            m = re.match(r"^([A-Za-z_.$][\w.$]*):", line)
            if m:
                label = m.group(1)
                if label in wanted:
                    cur = wanted[label]
                    cur["scratch_instructions"] = 0
                elif not label.startswith(".L"):
                    cur = None
                continue
            if cur is not None:
                if ".end_amdhsa_kernel" in line or line.startswith("\t.section") or line.startswith(".Lfunc_end"):
                    cur = None
                elif touch.match(line):
                    cur["scratch_instructions"] += 1


if __name__ == "__main__":
    only = "--scratch-only" in sys.argv
    for r in parse():
        if only and r["ScratchSize"] == 0:
            continue
        touched = "" if r["ScratchSize"] == 0 else ("  scratch instructions: %s" % r.get("scratch_instructions"))
        print("%-60s vgpr %3d  scratch %4d B  occupancy %d  (vgpr spill %d, sgpr spill %d)%s" % (
            r["name"], r["VGPRs"], r["ScratchSize"], r["Occupancy"], r["VGPRs Spill"], r["SGPRs Spill"], touched))
