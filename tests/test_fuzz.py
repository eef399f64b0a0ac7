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
