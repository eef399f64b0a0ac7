"""CPU statements of the device-side structure stage (solver2d_amd/csrc/structure.hip), test infrastructure.

Islands are DEFINED by solver2d_amd/islands.py (the reference has none); the colouring is the greedy colouring of
the active contacts in descending order of the fixed priority hash fmix32(contact index) -- what Jones-Plassmann
rounds with that priority compute."""
import numpy as np


def fmix32(k):
    k = np.asarray(k, dtype=np.uint64) & 0xffffffff
    k ^= k >> 16
    k = (k * 0x85ebca6b) & 0xffffffff
    k ^= k >> 13
    k = (k * 0xc2b2ae35) & 0xffffffff
    k ^= k >> 16
    return k.astype(np.uint32)


def color_constraints(bodies, contacts):
    """(color_of_contact int32[nc], color_count): sequential greedy in descending fmix32(index) order."""
    nc = len(contacts)
    colour = np.full(nc, -1, dtype=np.int32)
    movable = (bodies["type"] >= 0) & ((bodies["invMass"] != 0) | (bodies["invI"] != 0))
    active = np.flatnonzero(contacts["pointCount"] > 0)
    order = active[np.argsort(fmix32(active), kind="stable")[::-1]]
    used = {}
    a_all, b_all = contacts["bodyA"], contacts["bodyB"]
    for k in order.tolist():
        ends = [int(x) for x in {int(a_all[k]), int(b_all[k])} if movable[x]]
        taken = set()
        for body in ends:
            taken |= used.get(body, set())
        c = 0
        while c in taken:
            c += 1
        colour[k] = c
        for body in ends:
            used.setdefault(body, set()).add(c)
    return colour, (int(colour.max()) + 1 if len(active) else 0)


def check_proper(bodies, contacts, colour):
    """No two contacts of one colour share a movable body; every active contact is coloured, no inactive one is."""
    movable = (bodies["type"] >= 0) & ((bodies["invMass"] != 0) | (bodies["invI"] != 0))
    active = contacts["pointCount"] > 0
    assert (colour[active] >= 0).all() and (colour[~active] == -1).all()
    for c in range(int(colour.max()) + 1 if active.any() else 0):
        ids = np.flatnonzero(colour == c)
        touched = np.concatenate([contacts["bodyA"][ids], contacts["bodyB"][ids]])
        same = contacts["bodyA"][ids] == contacts["bodyB"][ids]
        touched = np.concatenate([touched[: len(ids)], touched[len(ids):][~same]])
        touched = touched[movable[touched]]
        assert len(np.unique(touched)) == len(touched), "colour %d reuses a movable body" % c
