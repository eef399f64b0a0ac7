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
    return rows


if __name__ == "__main__":
    only = "--scratch-only" in sys.argv
    for r in parse():
        if only and r["ScratchSize"] == 0:
            continue
        print("%-60s vgpr %3d  scratch %4d B  occupancy %d  (vgpr spill %d, sgpr spill %d)" % (
            r["name"], r["VGPRs"], r["ScratchSize"], r["Occupancy"], r["VGPRs Spill"], r["SGPRs Spill"]))
