"""Parity fuzzing on random solver inputs.  CPU part: the oracle must be order-insensitive in the
ways the design relies on (islands solved separately == together).  GPU part: HIP == oracle in the
device's order, bit for bit, for every solver, through both the LDS-group and the global path."""
import numpy as np
import pytest

from solver2d_amd import islands, wire
from tests import common, fuzz_worlds, oraclebind

SEEDS = list(range(12))


@pytest.mark.parametrize("seed", SEEDS[:6])
@pytest.mark.parametrize("solver_name", ["TGS_Soft", "PGS_NGS_Block", "XPBD"])
def test_oracle_is_finite_or_consistent_on_random_worlds(seed, solver_name):
    world = fuzz_worlds.random_world(seed)
    vel, pos = common.DEFAULT_ITERS[solver_name]
    p = wire.StepParams.make(solver_name, 1.0 / 60.0, vel, pos, True)
    a = common.copy3(world)
    b = common.copy3(world)
    oraclebind.solve(p, *a)
    oraclebind.solve(p, *b)
    common.compare_exact(a, b, "determinism")


@pytest.mark.gpu
@pytest.mark.parametrize("groups", [1, 0])
@pytest.mark.parametrize("seed", SEEDS)
def test_gpu_equals_oracle_on_random_worlds(seed, groups):
    from solver2d_amd import hip
    from tests.test_gpu_parity import gpu_vs_oracle_loose

    world = fuzz_worlds.random_world(seed, n_bodies=30 + 7 * seed, n_contacts=60 + 15 * seed, n_joints=8 + seed)
    with hip.Solver(0) as gpu:
        gpu.set_option("groups", groups)
        for solver_name in wire.SOLVER_NAMES:
            vel, pos = common.DEFAULT_ITERS[solver_name]
            for warm in (True, False):
                p = wire.StepParams.make(solver_name, 1.0 / 60.0, vel, pos, warm)
                gpu_vs_oracle_loose(gpu, p, world, "fuzz seed %d %s warm=%d groups=%d" % (seed, solver_name, warm, groups))


@pytest.mark.gpu
@pytest.mark.parametrize("joints", [0, 6])
@pytest.mark.parametrize("seed", SEEDS)
def test_gpu_equals_oracle_on_random_worlds_through_strips(seed, joints):
    """Random contact graphs (arbitrary degrees, one- and two-point contacts, static / kinematic / massless bodies)
    cut into tiny strips: the soft solvers take the persistent strip step when the world has no joints, the strip
    launches or the colour batches otherwise -- every path bit-exact in the reported order."""
    from solver2d_amd import hip
    from tests.test_gpu_parity import gpu_vs_oracle_loose

    world = fuzz_worlds.random_world(seed + 100, n_bodies=60 + 9 * seed, n_contacts=140 + 25 * seed, n_joints=joints)
    with hip.Solver(0) as gpu:
        gpu.set_option("strip_patience", 0)
        gpu.set_option("max_group_bodies", 16)
        gpu.set_option("strip_min_bodies", 0)
        gpu.set_option("strip_bodies", 10)
        for any_solver in (0, 1):
            gpu.set_option("strips_any_solver", any_solver)
            for solver_name in (wire.SOLVER_NAMES if any_solver else ["TGS_Soft", "SoftStep", "PGS_Soft"]):
                vel, pos = common.DEFAULT_ITERS[solver_name]
                for warm in (True, False):
                    p = wire.StepParams.make(solver_name, 1.0 / 60.0, vel, pos, warm)
                    gpu_vs_oracle_loose(gpu, p, world, "fuzz strips seed %d %s warm=%d any=%d" % (seed, solver_name, warm, any_solver))


def perturbed_pyramid(seed, base=36):
    """The pyramid's contact graph (long BFS diameter, degree <= 6: strips and the persistent kernel apply) with
    randomised numbers: velocities, impulses, anchors, separations, one-point contacts, a few inactive contacts,
    kinematic and static bodies in the pile."""
    from solver2d_amd import synthetic
    rng = np.random.default_rng(1000 + seed)
    bodies, contacts, joints = common.copy3(synthetic.pyramid(base))
    nb, nc = len(bodies), len(contacts)
    bodies["linearVelocity"] += rng.normal(0.0, 0.5, (nb, 2)).astype(np.float32)
    bodies["angularVelocity"] += rng.normal(0.0, 0.5, nb).astype(np.float32)
    ang = rng.normal(0.0, 0.05, nb).astype(np.float32)
    bodies["rot"][:, 0], bodies["rot"][:, 1] = np.sin(ang), np.cos(ang)
    bodies["linearDamping"] = rng.uniform(0.0, 0.2, nb).astype(np.float32)
    for i in rng.choice(np.arange(1, nb), size=6, replace=False):
        bodies["type"][i] = wire.BODY_KINEMATIC if i % 2 else wire.BODY_STATIC
        bodies["mass"][i] = bodies["invMass"][i] = bodies["I"][i] = bodies["invI"][i] = 0.0
        if bodies["type"][i] == wire.BODY_STATIC:
            bodies["linearVelocity"][i] = 0.0
            bodies["angularVelocity"][i] = 0.0
    pts = contacts["points"]
    pts["normalImpulse"] = rng.uniform(0.0, 2.0, pts["normalImpulse"].shape).astype(np.float32)
    pts["tangentImpulse"] = rng.normal(0.0, 0.3, pts["tangentImpulse"].shape).astype(np.float32)
    pts["separation"] += rng.normal(0.0, 0.01, pts["separation"].shape).astype(np.float32)
    pts["localAnchorA"] += rng.normal(0.0, 0.02, pts["localAnchorA"].shape).astype(np.float32)
    pts["localAnchorB"] += rng.normal(0.0, 0.02, pts["localAnchorB"].shape).astype(np.float32)
    contacts["friction"] = rng.uniform(0.0, 1.0, nc).astype(np.float32)
    if seed % 2:
        one = rng.random(nc) < 0.3
        contacts["pointCount"][one & (contacts["pointCount"] == 2)] = 1
        contacts["pointCount"][rng.random(nc) < 0.03] = 0
    return bodies, contacts, joints


@pytest.mark.gpu
@pytest.mark.parametrize("solver_name", ["TGS_Soft", "SoftStep", "PGS_Soft"])
def test_persistent_strip_step_on_perturbed_pyramids(solver_name):
    """Even seeds keep every contact two-point (the kernel's POINTS == 2 variant), odd seeds mix one-point and inactive
    contacts (POINTS == 0); both with kinematic / static bodies inside the pile.  Whatever path a world takes must be
    bit-exact; most of them must take the persistent kernel (greedy colouring sometimes needs a 7th colour, which
    sends that world to the multi-launch strip path)."""
    from solver2d_amd import hip
    from tests.test_gpu_parity import gpu_vs_oracle_loose

    vel, pos = common.DEFAULT_ITERS[solver_name]
    persistent, two_point, general = 0, 0, 0
    for seed in SEEDS[:8]:
        world = perturbed_pyramid(seed)
        with hip.Solver(0) as gpu:
            gpu.set_option("strip_patience", 0)
            gpu.set_option("max_group_bodies", 256)  # the 666-body pile fits no group; a strip (>= 2 levels of <= 36 bodies) does
            gpu.set_option("strip_min_bodies", 0)
            gpu.set_option("strip_bodies", 40 + 20 * (seed % 3))
            state = world
            for step in range(2):
                p = wire.StepParams.make(solver_name, 1.0 / 60.0, vel, pos, step == 0 or seed % 3 != 0)
                state = gpu_vs_oracle_loose(gpu, p, state, "perturbed pyramid seed %d %s step %d" % (seed, solver_name, step))
                assert gpu.stats()["stripCount"] > 0
                if gpu.stats()["persistent"]:
                    persistent += 1
                    two_point += seed % 2 == 0
                    general += seed % 2 == 1
    assert persistent >= 8 and two_point >= 2 and general >= 2, (persistent, two_point, general)


@pytest.mark.gpu
@pytest.mark.parametrize("solver_name", ["TGS_Soft", "SoftStep", "PGS_Soft"])
def test_moving_read_only_bodies_shared_between_strips(solver_name):
    """Found by a wider fuzz run (seed 205): kinematic and massless bodies are not written by the sweeps but the body stages
    move them, so every strip that touches one keeps a copy.  With one launch per sweep the copies were re-read from HBM
    while the owner was writing them in the same launch (a race: different bits from run to run).  Such partitions are
    now only run by the one-launch persistent kernel, else the island goes back to colour batches."""
    from solver2d_amd import hip
    from tests.test_gpu_parity import gpu_vs_oracle_loose

    seed = 205
    world = fuzz_worlds.random_world(seed, n_bodies=40 + 11 * (seed % 9), n_contacts=90 + 31 * (seed % 7), n_joints=0)
    vel, pos = common.DEFAULT_ITERS[solver_name]
    for strip_bodies, retry in ((8, 0), (8, 1), (11, 1)):
        with hip.Solver(0) as gpu:
            gpu.set_option("strip_patience", 0)
            gpu.set_option("max_group_bodies", 24)
            gpu.set_option("strip_min_bodies", 0)
            gpu.set_option("strip_bodies", strip_bodies)
            gpu.set_option("strip_retry", retry)
            for rep in range(3):
                p = wire.StepParams.make(solver_name, 1.0 / 60.0, vel, pos, True)
                gpu_vs_oracle_loose(gpu, p, world, "seed 205 %s strip_bodies %d retry %d rep %d" % (solver_name, strip_bodies, retry, rep))
            st = gpu.stats()
            assert st["stripCount"] == 0 or st["persistent"] == 1, st


@pytest.mark.gpu
@pytest.mark.parametrize("strip_bodies", [8, 11, 16])
def test_written_bodies_outside_every_strip(strip_bodies):
    """Found by the same wider fuzz run (seed 259 under XPBD through the test-only `strips_any_solver` option): a static
    body whose rot is not a fixed point of the normalisation is WRITTEN by the position sweeps, but it is not one of the
    loose bodies the strip partition walks, so two strips each believed they owned it.  Such a graph gets no strips."""
    from solver2d_amd import hip
    from tests.test_gpu_parity import gpu_vs_oracle_loose

    seed = 259
    world = fuzz_worlds.random_world(seed, n_bodies=40 + 11 * (seed % 9), n_contacts=90 + 31 * (seed % 7), n_joints=(seed % 5) * 2)
    with hip.Solver(0) as gpu:
        gpu.set_option("strip_patience", 0)
        gpu.set_option("max_group_bodies", 24)
        gpu.set_option("strip_min_bodies", 0)
        gpu.set_option("strip_bodies", strip_bodies)
        gpu.set_option("strips_any_solver", 1)
        for solver_name in ("TGS_Soft", "XPBD", "PGS_NGS", "XPBD"):
            vel, pos = common.DEFAULT_ITERS[solver_name]
            p = wire.StepParams.make(solver_name, 1.0 / 60.0, vel, pos, True)
            gpu_vs_oracle_loose(gpu, p, world, "seed 259 %s strip_bodies %d" % (solver_name, strip_bodies))


@pytest.mark.gpu
def test_pyramids_with_kinematic_and_massless_bodies_inside():
    """Perturbed pyramids in which a few boxes are kinematic or massless (moving, never written by the sweeps) so that
    they touch boxes of several strips: the persistent kernel keeps a consistent copy per workgroup; where it is not
    eligible the island must be back on colour batches (never on the one-launch-per-sweep strip paths)."""
    from solver2d_amd import hip
    from tests.test_gpu_parity import gpu_vs_oracle_loose

    persistent = other = 0
    for seed in range(100, 112):
        world = perturbed_pyramid(seed, base=26 + seed % 15)
        rng = np.random.default_rng(seed)
        b = world[0]
        dyn = np.flatnonzero(b["type"] == wire.BODY_DYNAMIC)
        for i in rng.choice(dyn, size=max(2, len(dyn) // 30), replace=False):
            if rng.random() < 0.6:
                b["type"][i] = wire.BODY_KINEMATIC
            b["mass"][i] = 0
            b["invMass"][i] = 0
            b["I"][i] = 0
            b["invI"][i] = 0
            b["linearVelocity"][i] = rng.uniform(-2, 2, 2)
            b["angularVelocity"][i] = rng.uniform(-1, 1)
        solver_name = ("TGS_Soft", "SoftStep", "PGS_Soft")[seed % 3]
        vel, pos = common.DEFAULT_ITERS[solver_name]
        with hip.Solver(0) as gpu:
            gpu.set_option("strip_patience", 0)
            gpu.set_option("max_group_bodies", 64)
            gpu.set_option("strip_min_bodies", 0)
            gpu.set_option("strip_bodies", 30 + 13 * (seed % 4))
            state = common.copy3(world)
            for step in range(3):
                p = wire.StepParams.make(solver_name, 1.0 / 60.0, vel, pos, True)
                state = gpu_vs_oracle_loose(gpu, p, state, "kinematic pyramid seed %d %s step %d" % (seed, solver_name, step))
            st = gpu.stats()
            assert st["stripCount"] == 0 or st["persistent"] == 1, st
            persistent += st["persistent"]
            other += 1 - st["persistent"]
    assert persistent > 0, (persistent, other)
