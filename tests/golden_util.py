import glob
import os

import numpy as np

from solver2d_amd import wire

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden_files():
    return sorted(f for f in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")) if not os.path.basename(f).startswith(("bp_", "np_", "world_", "big_")))


def broadphase_files():
    return sorted(glob.glob(os.path.join(GOLDEN_DIR, "bp_*.npz")))


def load(path):
    z = np.load(path)
    pi, pf = z["params"], z["params_f"]
    params = wire.StepParams.make(int(pi[0]), float(pf[0]), int(pi[1]), int(pi[2]), bool(pi[3]), (float(pf[1]), float(pf[2])))
    params.dt = pf[0]
    pre = (z["pre_bodies"].copy(), z["pre_contacts"].copy(), z["pre_joints"].copy())
    post = (z["post_bodies"].copy(), z["post_contacts"].copy(), z["post_joints"].copy())
    assert pre[0].dtype == wire.body_dtype and pre[1].dtype == wire.contact_dtype and pre[2].dtype == wire.joint_dtype
    return params, pre, post


def narrowphase_files():
    return sorted(glob.glob(os.path.join(GOLDEN_DIR, "np_*.npz")))
