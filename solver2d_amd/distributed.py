"""Multi-GPU driver: one process per GPU, islands sharded across ranks, no data-path collective
inside a step; one all-gather of the owned body records per step so every rank ends the step with
the whole world's body state (what a host-side collision phase, or a future device broad phase,
needs).  Backend "nccl" is RCCL over xGMI on the GPU node; the same code runs over "gloo" on CPU
in tests/test_islands_dist.py with the oracle standing in for the HIP solver.
"""
import numpy as np

from . import islands, wire


def _gather_rows(rows, world_size, dist, torch, device):
    """all-gather of one variable-length array of fixed-size records per rank (padded to the longest): list of arrays, one per rank"""
    flat = np.ascontiguousarray(rows).view(np.uint8).reshape(len(rows), -1) if len(rows) else np.zeros((0, rows.dtype.itemsize), dtype=np.uint8)
    count = torch.tensor([len(rows)], dtype=torch.int64, device=device)
    counts = torch.zeros(world_size, dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(counts, count)
    counts = counts.cpu().numpy()
    most = max(int(counts.max()), 1)
    mine = np.zeros((most, flat.shape[1]), dtype=np.uint8)
    mine[: len(flat)] = flat
    t = torch.from_numpy(mine).to(device)
    out = torch.empty((world_size * most, flat.shape[1]), dtype=torch.uint8, device=device)
    dist.all_gather_into_tensor(out, t)
    out = out.cpu().numpy().reshape(world_size, most, flat.shape[1])
    return [np.ascontiguousarray(out[r, : int(counts[r])]).view(rows.dtype).reshape(-1) for r in range(world_size)]


class ShardedWorld:
    def __init__(self, bodies, contacts, joints, rank, world_size, finder=None):
        """finder: the island finder of islands.shard_world (None: the host's; hip.Solver.find_islands: the device's union-find)."""
        self.rank, self.world_size = rank, world_size
        self.bodies, self.contacts, self.joints = bodies, contacts, joints
        self.reshards = 0
        self.finder = finder
        self._partition(None)

    def _partition(self, previous_owner):
        shards, self.island, self.shard_of_island = islands.shard_world(self.bodies, self.contacts, self.joints, self.world_size, previous_owner,
                                                                        finder=self.finder)
        self.shards = shards
        self.mine = shards[self.rank]
        # fixed-size exchange record: every rank contributes max_owned body records
        self.owned_ids = [sh.body_ids[sh.owned_body] for sh in shards]
        self.max_owned = max(1, max(len(x) for x in self.owned_ids))

    def owner_of_body(self):
        """shard that owns each body of the world under the current partition (-1: static or free -- replicated, owned by nobody)"""
        owner = np.full(len(self.bodies), -1, dtype=np.int32)
        for r, ids in enumerate(self.owned_ids):
            owner[ids] = r
        return owner

    def joins_shards(self, body_a, body_b):
        """Would a constraint between these two bodies connect islands that live on different ranks?"""
        owner = self.owner_of_body()
        a, b = int(owner[body_a]), int(owner[body_b])
        return a >= 0 and b >= 0 and a != b

    def reshard(self, contacts=None, joints=None, dist=None, torch=None, device="cpu"):
        """The constraint graph changed -- the collision phase created or destroyed contacts, the same on every rank (it runs on the
        poses every rank holds after the all-gather): islands may have merged or split.  `contacts` / `joints`: the whole world's new
        arrays (None: unchanged), in which the solver state of the constraints that were there before (impulses, TGS_Sticky's friction
        cache) may be stale -- it lives with the owner.  So, once:
          1. every rank contributes the records of the constraints it owns (one all-gather of padded byte records each for contacts
             and joints) and every rank ends with the whole world's solver state;
          2. islands are found again; an island stays on the rank that owned most of its bodies (islands.sticky_partition), so two
             islands that a created contact joined end up on ONE rank -- the one that held the larger --, everything else stays;
          3. every rank extracts its new sub-world (pool order preserved: a sharded solve stays bit-identical to the whole-world solve).
        No body state travels here: the per-step all-gather has already given every rank every body."""
        previous = self.owner_of_body()
        old = self.shards
        if dist is not None and self.world_size > 1:
            for r, rows in enumerate(_gather_rows(self.mine.contacts, self.world_size, dist, torch, device)):
                self._take_contacts(old[r], rows)
            for r, rows in enumerate(_gather_rows(self.mine.joints, self.world_size, dist, torch, device)):
                self._take_joints(old[r], rows)
        else:
            self._take_contacts(self.mine, self.mine.contacts)
            self._take_joints(self.mine, self.mine.joints)
        # bodies: what this rank solved itself (a single process has no exchange to have done it)
        o = self.mine.owned_body
        self.bodies[self.mine.body_ids[o]] = self.mine.bodies[o]
        if contacts is not None:
            # the caller's array decides which slots are live and what their manifolds are; a slot that kept its pair keeps its solver state
            same = (contacts["bodyA"] == self.contacts["bodyA"]) & (contacts["bodyB"] == self.contacts["bodyB"]) & (self.contacts["bodyA"] >= 0)
            merged = contacts.copy()
            for f in ("normalImpulse", "tangentImpulse", "frictionAnchorA", "frictionAnchorB", "frictionNormalA", "frictionNormalB"):
                merged["points"][f][same] = self.contacts["points"][f][same]
            merged["frictionPersisted"][same] = self.contacts["frictionPersisted"][same]
            self.contacts = merged
        if joints is not None:
            self.joints = joints
        self._partition(previous)
        self.reshards += 1
        return self.mine

    def _take_contacts(self, shard, rows):
        if len(rows):
            ids = shard.contact_ids
            keep_a, keep_b, keep_index = self.contacts["bodyA"][ids].copy(), self.contacts["bodyB"][ids].copy(), self.contacts["constraintIndex"][ids].copy()
            self.contacts[ids] = rows
            self.contacts["bodyA"][ids], self.contacts["bodyB"][ids], self.contacts["constraintIndex"][ids] = keep_a, keep_b, keep_index

    def _take_joints(self, shard, rows):
        if len(rows):
            ids = shard.joint_ids
            keep_a, keep_b = self.joints["bodyA"][ids].copy(), self.joints["bodyB"][ids].copy()
            self.joints[ids] = rows
            self.joints["bodyA"][ids], self.joints["bodyB"][ids] = keep_a, keep_b

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


class ResidentShardedWorld:
    """The device-resident form of ShardedWorld (BASELINE.json configs[4]): every rank uploads ITS shard
    once (s2amd_upload), advances it with s2amd_step_resident -- no collective inside a step, islands share no movable
    body -- and contributes one fixed-size record of its body arrays per step -- {position, rot, linearVelocity, angularVelocity}:
    the 28 bytes per body of SURVEY.md 8e, in two 16-byte records -- to ONE all-gather of device tensors
    (torch.distributed backend "nccl" = RCCL over xGMI; "gloo" moves the same records through the host in the tests).
    Nothing but those ever leaves a GPU: constraints and impulses stay in the HBM of their owner.

    The host loop is software-pipelined like bench.py's replica loop: step s+1 is enqueued on the solver's stream BEFORE
    the host waits for the poses of step s and hands them to the collective, through two pose buffers."""

    def __init__(self, sharded, solver, torch, dist=None, backend="nccl", exchange_ranks=None, exchange_rank=0):
        """exchange_ranks: ranks that take part in the per-step all-gather when that is not the partition's own world size -- the
        WEAK-scaling form, where every rank brings its own islands (a partition by construction) and `sharded` is its local world;
        exchange_rank: this rank's place among them (its record's slot in the gathered tensor)."""
        self.sw, self.solver, self.torch, self.dist, self.backend = sharded, solver, torch, dist, backend
        self.exchange_ranks = exchange_ranks if exchange_ranks is not None else sharded.world_size
        self.weak = self.exchange_ranks != sharded.world_size
        # reshard() finds the islands with the device kernel when the solver has one (the CPU tests' stand-in solver has not)
        self.device_islands = hasattr(solver, "find_islands")
        self.exchange_rank = int(exchange_rank) if self.weak else sharded.rank
        if not 0 <= self.exchange_rank < max(self.exchange_ranks, 1):
            raise ValueError("exchange_rank %d outside the %d exchanging ranks" % (self.exchange_rank, self.exchange_ranks))
        sh = sharded.mine
        solver.upload(sh.bodies, sh.contacts, sh.joints)
        self.body_slots = len(sh.bodies)
        # one record per rank, sized for the largest shard (an all-gather wants equal contributions)
        self.record = max(1, max(len(s.bodies) for s in sharded.shards))
        self.raw = dist is None  # a single process: no collective, no torch -- pose records in buffers of the library's own
        if self.raw:
            self.pose_ptr = [solver.device_alloc(self.record * 32) for _ in range(2)]
            self.last = None
        else:
            self.pose = [torch.zeros((self.record, 8), dtype=torch.float32, device="cuda") for _ in range(2)]
            self.pose_ptr = [t.data_ptr() for t in self.pose]
            self.gathered = torch.zeros((self.exchange_ranks * self.record, 8), dtype=torch.float32, device="cuda" if backend == "nccl" else "cpu")
        self.gather_done = [None, None]
        self.enqueued = 0
        self.exchanged = 0

    def enqueue_step(self, params):
        """s2Solve_* of my shard + the export of its poses, enqueued on the solver's stream (option "async")."""
        b = self.enqueued & 1
        if self.gather_done[b] is not None:
            self.gather_done[b].synchronize()  # the collective that last read this buffer (two steps ago)
            self.gather_done[b] = None
        self.solver.step_resident(params)
        self.solver.export_bodies_async(self.pose_ptr[b], self.record, b)
        self.enqueued += 1

    def exchange(self):
        """The step's one exchange: all ranks' pose records into `gathered` (on every rank)."""
        b = self.exchanged & 1
        self.solver.export_wait(b)
        if self.raw:
            self.last = b  # one rank: its record IS the world's
        elif self.backend == "nccl":
            self.dist.all_gather_into_tensor(self.gathered, self.pose[b])
            self.gather_done[b] = self.torch.cuda.Event()
            self.gather_done[b].record()
        else:
            self.dist.all_gather_into_tensor(self.gathered, self.pose[b].cpu())
        self.exchanged += 1

    def run(self, params, steps):
        if steps <= 0:
            return
        self.enqueue_step(params)
        for s in range(steps):
            if s + 1 < steps:
                self.enqueue_step(params)
            self.exchange()

    def world_bodies(self):
        """float32[bodies of the whole world, 8] {position, rot, linearVelocity, angularVelocity, 0} as of the last exchange,
        assembled from the gathered records: every body from the rank that owns it (static bodies from any shard that holds a replica)."""
        if self.raw:
            g = self.solver.device_read(self.pose_ptr[self.last], (1, self.record, 8))
        else:
            self.torch.cuda.synchronize()
            g = self.gathered.cpu().numpy().reshape(self.exchange_ranks, self.record, 8)
            if self.weak:
                # every rank holds a world of its own: the local world's bodies are this rank's record (one shard, slot exchange_rank)
                g = g[self.exchange_rank:self.exchange_rank + 1]
        out = np.zeros((len(self.sw.bodies), 8), dtype=np.float32)
        out[:, 0:2] = self.sw.bodies["position"]
        out[:, 2:4] = self.sw.bodies["rot"]
        out[:, 4:6] = self.sw.bodies["linearVelocity"]
        out[:, 6] = self.sw.bodies["angularVelocity"]
        for r, sh in enumerate(self.sw.shards):
            rows = g[r, : len(sh.bodies)]
            out[sh.body_ids[sh.owned_body]] = rows[sh.owned_body]
        return out

    def world_poses(self):
        """float32[bodies of the whole world, 4] {position, rot}: the first half of world_bodies()."""
        return self.world_bodies()[:, 0:4]

    def reshard(self, contacts=None, joints=None):
        """ShardedWorld.reshard for the resident form: my shard's solver state comes down from the device first (bodies, impulses),
        the whole world's body array is brought up to the last exchange, the constraint state is exchanged and the islands are
        partitioned again (islands that did not change stay on their rank), and my NEW shard goes up (s2amd_upload).  Rare -- a
        created contact that joins islands of two ranks, pairs that separated --, so it may cost a host round trip."""
        if self.weak:
            raise RuntimeError("reshard: the weak-scaling form holds one local world per rank (a partition by construction); "
                               "there are no islands to move between ranks")
        sh = self.sw.mine
        self.solver.synchronize()
        self.solver.download(sh.bodies, sh.contacts, sh.joints)
        if self.exchanged > 0:
            g = self.world_bodies()
            b = self.sw.bodies
            b["position"], b["rot"], b["linearVelocity"], b["angularVelocity"] = g[:, 0:2], g[:, 2:4], g[:, 4:6], g[:, 6]
        if self.device_islands:
            # islands found again on the device (s2amd_find_islands, structure.hip: SURVEY.md 8f row 4); every rank runs the same
            # deterministic labelling on the same arrays, so the ranks agree without exchanging it
            self.sw.finder = self.solver.find_islands
        self.sw.reshard(contacts, joints, dist=self.dist, torch=self.torch, device="cuda" if (self.dist is not None and self.backend == "nccl") else "cpu")
        sh = self.sw.mine
        self.solver.upload(sh.bodies, sh.contacts, sh.joints)
        self.body_slots = len(sh.bodies)
        record = max(1, max(len(s.bodies) for s in self.sw.shards))
        if record != self.record:
            # the pose records follow the largest shard
            self.record = record
            if self.raw:
                for p in self.pose_ptr:
                    self.solver.device_free(p)
                self.pose_ptr = [self.solver.device_alloc(self.record * 32) for _ in range(2)]
            else:
                torch = self.torch
                self.pose = [torch.zeros((self.record, 8), dtype=torch.float32, device="cuda") for _ in range(2)]
                self.pose_ptr = [t.data_ptr() for t in self.pose]
                self.gathered = torch.zeros((self.exchange_ranks * self.record, 8), dtype=torch.float32, device="cuda" if self.backend == "nccl" else "cpu")
        self.gather_done = [None, None]
        self.enqueued = self.exchanged = 0
        self.last = None

    def close(self):
        if self.raw:
            for p in self.pose_ptr:
                self.solver.device_free(p)
            self.pose_ptr = []
