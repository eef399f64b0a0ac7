"""Bisecting tool: tests/test_gpu_world.py's wrecking-ball loop for one seed, every step compared with the oracle chain, stopping at
the FIRST step that differs (with the placement counters of that step).  S2AMD_DEBUG_PLACE=1 shows what was placed where;
S2AMD_OPTIONS=key=value,... switches features off.    python tools/wreck_first_difference.py [solver ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from solver2d_amd import hip, wire
from tests import common, world_chain
from tests.test_gpu_world import _create_contacts
seed = 19
for solver_name in sys.argv[1:] or ["SoftStep"]:
    rng = np.random.default_rng(1000 + seed)
    base = int(rng.integers(30, 75))
    world = world_chain.wreck_world(seed, base)
    vel, pos = common.DEFAULT_ITERS[solver_name]
    params = wire.StepParams.make(solver_name, 1.0 / 60.0, vel, pos, True)
    ref = world_chain.copy_world(world)
    with hip.Solver(0) as s:
        s.set_option("strip_patience", int(rng.integers(0, 4)))
        s.set_option("strip_min_bodies", 0)
        s.set_option("strip_bodies", int(rng.integers(40, 200)))
        s.set_option("max_group_bodies", int(rng.choice([64, 512, 2816])))
        s.world_upload(*[world[k] for k in world_chain.WORLD_KEYS])
        for step in range(40):
            if world_chain.moved_any(ref):
                got = s.world_find_pairs()
                if len(got):
                    print(solver_name, "step", step, "pairs", got[:6].tolist())
                    slots, contacts, pairs = _create_contacts(ref, got)
                    s.world_set_contacts(slots, contacts, pairs)
            info = s.world_step(params)
            order, offs = s.contact_order()
            st = s.stats()
            try:
                world_chain.oracle_world_step(params, ref, contact_order=order)
                out = world_chain.copy_world(world)
                res = s.world_download(*[out[k] for k in world_chain.WORLD_KEYS])
                world_chain.assert_device_equals_oracle(dict(zip(world_chain.WORLD_KEYS, res[:6])), ref, "wreck %d %s step %d" % (seed, solver_name, step))
            except Exception as e:
                print(solver_name, "FIRST DIFFERENCE at step", step, str(e)[:400].replace("\n", " | "))
                print({k: st[k] for k in ("stripCount", "persistent", "pairLanes", "placedContacts", "structureBuilds", "bodiesAdopted", "seamBodiesAdded", "roundsOpened", "kernelLaunches")})
                break
            if st["placedContacts"]:
                print(solver_name, "step", step, "ok", {k: st[k] for k in ("persistent", "pairLanes", "placedContacts", "structureBuilds", "bodiesAdopted", "seamBodiesAdded", "roundsOpened")})
