"""Multi-GPU driver: one process per GPU, islands sharded across ranks, no data-path collective
inside a step; one all-gather of the owned body records per step so every rank ends the step with
the whole world's body state (what a host-side collision phase, or a future device broad phase,
needs).  Backend "nccl" is RCCL over xGMI on the GPU node; the same code runs over "gloo" on CPU
in tests/test_islands_dist.py with the oracle standing in for the HIP solver.
"""
import numpy as np

from . import islands, wire


class ShardedWorld:
    def __init__(self, bodies, contacts, joints, rank, world_size):
        self.rank, self.world_size = rank, world_size
        self.bodies, self.contacts, self.joints = bodies, contacts, joints
        shards, self.island, self.shard_of_island = islands.shard_world(bodies, contacts, joints, world_size)
        self.shards = shards
        self.mine = shards[rank]
        # fixed-size exchange record: every rank contributes max_owned body records
        self.owned_ids = [sh.body_ids[sh.owned_body] for sh in shards]
        self.max_owned = max(1, max(len(x) for x in self.owned_ids))

    def pack_owned(self):
        """float32[max_owned, 9]: position, rot, linearVelocity, angularVelocity, deltaPosition of the
        bodies this rank owns (padded)."""
        sh = self.mine
        b = sh.bodies[sh.owned_body]
        out = np.zeros((self.max_owned, 9), dtype=np.float32)
        n = len(b)
        out[:n, 0:2] = b["position"]
        out[:n, 2:4] = b["rot"]
        out[:n, 4:6] = b["linearVelocity"]
        out[:n, 6] = b["angularVelocity"]
        out[:n, 7:9] = b["deltaPosition"]
        return out

    def unpack_all(self, gathered):
        """gathered: float32[world_size, max_owned, 9] -> scatter into the full body array."""
        for r in range(self.world_size):
            ids = self.owned_ids[r]
            g = gathered[r, :len(ids)]
            self.bodies["position"][ids] = g[:, 0:2]
            self.bodies["rot"][ids] = g[:, 2:4]
            self.bodies["linearVelocity"][ids] = g[:, 4:6]
            self.bodies["angularVelocity"][ids] = g[:, 6]
            self.bodies["deltaPosition"][ids] = g[:, 7:9]
        return self.bodies


def step_sharded(sw, solve_fn, params, dist=None, torch=None, device="cpu"):
    """One world step: solve my shard with solve_fn(params, bodies, contacts, joints) (in place),
    then all-gather the owned body records.  Returns the full, updated body array."""
    sh = sw.mine
    solve_fn(params, sh.bodies, sh.contacts, sh.joints)
    mine = sw.pack_owned()
    if dist is None or sw.world_size == 1:
        gathered = mine[None]
    else:
        t = torch.from_numpy(mine).to(device)
        out = torch.empty((sw.world_size * t.shape[0], t.shape[1]), dtype=t.dtype, device=device)
        dist.all_gather_into_tensor(out, t)
        gathered = out.cpu().numpy().reshape(sw.world_size, t.shape[0], t.shape[1])
    return sw.unpack_all(gathered)
