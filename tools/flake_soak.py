"""Soak of one golden through one path, many times, against the oracle (a check that a parity test cannot pass or fail by timing):
tools/flake_soak.py <golden name without .npz> [--groups 0] [--n 200] [--with-predecessor]"""
import argparse
import glob
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from solver2d_amd import hip  # noqa: E402
from tests import common, golden_util, oraclebind  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("name")
    ap.add_argument("--groups", type=int, default=0)
    ap.add_argument("--n", type=int, default=200)
    ap.add_argument("--with-predecessor", action="store_true", help="solve the golden that precedes it in the test's file list first, every time")
    ap.add_argument("--fresh", action="store_true", help="a new solver object per iteration")
    a = ap.parse_args()
    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "*.npz")))
    files = [f for f in files if not os.path.basename(f).startswith("big_")]
    path = [f for f in files if os.path.basename(f)[:-4] == a.name][0]
    third = files[::3]
    before = third[third.index(path) - 1] if path in third and third.index(path) > 0 else files[files.index(path) - 1]
    params, pre, _ = golden_util.load(path)
    bad = 0
    orders = set()
    s = hip.Solver(0)
    s.set_option("groups", a.groups)
    for i in range(a.n):
        if a.fresh:
            s.close()
            s = hip.Solver(0)
            s.set_option("groups", a.groups)
        if a.with_predecessor:
            p2, pre2, _ = golden_util.load(before)
            g2 = common.copy3(pre2)
            s.solve(p2, *g2)
        got = common.copy3(pre)
        s.solve(params, *got)
        order, offsets = s.contact_order()
        jorder, _ = s.joint_order()
        orders.add(hash(order.tobytes()) ^ hash(offsets.tobytes()))
        want = common.copy3(pre)
        oraclebind.solve(params, *want, contact_order=order, joint_order=jorder)
        try:
            common.compare_exact(got, want, "%s #%d" % (a.name, i))
        except AssertionError as e:
            bad += 1
            st = s.stats()
            print("MISMATCH at", i, str(e)[:300].replace("\n", " | "), {k: st[k] for k in ("groupCount", "launchCount") if k in st}, flush=True)
    s.close()
    print("%s groups=%d: %d iterations, %d mismatches, %d distinct orders" % (a.name, a.groups, a.n, bad, len(orders)))


if __name__ == "__main__":
    main()
