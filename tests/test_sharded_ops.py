"""What one s2amd_sharded_step enqueues, counted by the code that enqueues it (csrc/sharded.hip: s2amd_sharded_count_ops runs the
step's own enqueue functions with every device call switched off): O(shards) stream operations in the two forms n GPUs would use,
where round 5's exchange -- kept as the fall-back -- is quadratic.  Runs without a GPU."""
import ctypes

from solver2d_amd import hip


def _ops(shards, form):
    out = ctypes.c_int32()
    assert hip.load().s2amd_sharded_count_ops(shards, form, ctypes.byref(out)) == 0
    return out.value


def test_stream_operations_per_step_grow_linearly_with_the_shards():
    for form in (0, 1):  # stores (shards of one device), rccl (distinct devices)
        per_shard = [_ops(n, form) / n for n in (1, 2, 4, 8, 16)]
        assert max(per_shard) == min(per_shard), (form, per_shard)
        assert per_shard[0] <= 9
    assert _ops(8, 0) == 8 * 3 and _ops(8, 1) == 8 * 9
    # the peer copies: a wait, a copy and a scatter per ordered pair of shards
    assert _ops(8, 2) == 8 * 9 + 3 * 8 * 7


def test_count_ops_refuses_nonsense():
    out = ctypes.c_int32()
    L = hip.load()
    assert L.s2amd_sharded_count_ops(0, 0, ctypes.byref(out)) != 0
    assert L.s2amd_sharded_count_ops(4, 3, ctypes.byref(out)) != 0
