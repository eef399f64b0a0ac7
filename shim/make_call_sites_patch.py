#!/usr/bin/env python3
"""Writes shim/call_sites.patch: the integration of INTEGRATION.md section 1 as a source patch against a checkout of
erincatto/solver2d -- for a maintainer who prefers three kinds of edited line to the link-time interposition of
shim/Makefile.  The patch only ADDS lines (unified diff with zero context, so it carries no reference text) and is located
by function name, so it is regenerated against whatever revision REF holds:   python3 shim/make_call_sites_patch.py /path/to/solver2d
Apply with `patch -p1 < call_sites.patch` in the checkout, copy shim/s2_amd_binding.{c,h} and include/solver2d_amd.h into
src/, add s2_amd_binding.c to src/CMakeLists.txt and link -ldl."""
import difflib
import os
import re
import sys

SOLVERS = ["Jacobi", "PGS", "PGS_NGS", "PGS_NGS_Block", "PGS_Soft", "SoftStep", "TGS_Sticky", "TGS_Soft", "TGS_NGS", "XPBD"]
INCLUDE = '#include "s2_amd_binding.h"\n'


def after_open_brace(lines, signature_re):
    """index of the line after the `{` that opens the function whose definition matches signature_re"""
    for i, l in enumerate(lines):
        if re.match(signature_re, l) and not l.rstrip().endswith(";"):
            j = i
            while "{" not in lines[j]:
                j += 1
            return j + 1
    raise SystemExit("function not found: " + signature_re)


def last_include(lines):
    return max(i for i, l in enumerate(lines) if l.startswith("#include")) + 1


def main():
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    here = os.path.dirname(os.path.abspath(__file__))
    out = []
    edits = {}
    src = os.path.join(ref, "src")
    for name in sorted(os.listdir(src)):
        if not name.endswith(".c"):
            continue
        text = open(os.path.join(src, name), encoding="latin-1").read().splitlines(keepends=True)
        new = list(text)
        inserts = []  # (index, [lines])
        for s in SOLVERS:
            sig = r"^void s2Solve_%s\(s2World\* world, s2StepContext\* \w+\)" % s
            if any(re.match(sig, l) and not l.rstrip().endswith(";") for l in text):
                ctx = re.search(r"s2StepContext\* (\w+)\)", [l for l in text if re.match(sig, l)][0]).group(1)
                inserts.append((after_open_brace(text, sig), [
                    "\tif (s2amdBinding_IsOpen())\n", "\t{\n",
                    "\t\ts2amdBinding_SolveOrDie(world, %s, s2_solver%s); // (aborts on a device error: never the CPU solver on the same world)\n" % (ctx, s),
                    "\t\treturn;\n", "\t}\n"]))
        if name == "world.c":
            i = after_open_brace(text, r"^void s2World_Step\(")
            # after the line that looks the world up
            while "s2GetWorldFromId" not in text[i]:
                i += 1
            inserts.append((i + 1, [
                "\tif (s2amdBinding_IsOpen())\n", "\t{\n",
                "\t\t// stage 3, the solve and stage 4 on the MI355X (shim/s2_amd_binding.c); stages 1 and 2 are this file's\n",
                "\t\ts2amdBinding_WorldStepOrDie(world, timeStep, velIters, posIters, warmStart, s2UpdateBroadPhasePairs, s2BroadPhase_RebuildTrees);\n",
                "\t\treturn;\n", "\t}\n"]))
            i = after_open_brace(text, r"^void s2DestroyWorld\(")
            while "s2GetWorldFromId" not in text[i]:
                i += 1
            inserts.append((i + 1, ["\ts2amdBinding_DestroyWorld(world);\n"]))
        if not inserts:
            continue
        inserts.append((last_include(text), [INCLUDE]))
        for idx, lines in sorted(inserts, reverse=True):
            new[idx:idx] = lines
        diff = list(difflib.unified_diff(text, new, "a/src/" + name, "b/src/" + name, n=0))
        assert not any(l.startswith("-") and not l.startswith("---") for l in diff), name
        out.extend(diff)
        edits[name] = len(inserts) - 1
    path = os.path.join(here, "call_sites.patch")
    open(path, "w", encoding="latin-1").write("".join(out))
    print("wrote %s: %d files, %s" % (path, len(edits), edits))


if __name__ == "__main__":
    main()
