"""N>1 path on CPU: world_size-2 gloo test of the all-reduce hook exactly as the C library calls it
(through the hmx_allreduce_fn C function pointer), shard bounds, and shard-independent block membership."""
import ctypes
import os
import socket

import numpy as np
import pytest

from harmony_amd import _lib
from harmony_amd.dist import TorchAllReduce, shard_bounds


def test_shard_bounds_cover_and_balance():
    for N, w in [(10, 3), (1000000, 8), (7, 7), (5, 8)]:
        b = shard_bounds(N, w)
        assert b[0][0] == 0 and b[-1][1] == N
        assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
        sizes = [hi - lo for lo, hi in b]
        assert max(sizes) - min(sizes) <= 1


def test_block_membership_is_shard_independent():
    """block(g) depends only on the GLOBAL cell index: shards computing their own cells' blocks reproduce the
    single-process partition, and blocks stay balanced (src/harmony.cpp:280-300)."""
    lib = _lib.load()
    N, nb = 1003, 20
    cpb = int(np.float32(N) * np.float32(0.05))
    whole = np.array([min(lib.hmx_feistel_pos(9, 4, N, g) // cpb, nb - 1) for g in range(N)])
    parts = []
    for lo, hi in shard_bounds(N, 3):
        parts.append(np.array([min(lib.hmx_feistel_pos(9, 4, N, g) // cpb, nb - 1) for g in range(lo, hi)]))
    assert np.array_equal(np.concatenate(parts), whole)
    cnt = np.bincount(whole, minlength=nb)
    assert np.all(cnt[:-1] == cpb) and cnt[-1] == N - cpb * (nb - 1)


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        hook = TorchAllReduce(device=None)
        cfn = _lib.ALLREDUCE_FN(hook)          # the C library holds exactly this kind of pointer
        ok = True
        a = np.arange(1000, dtype=np.int64) * (rank + 1)
        ok &= cfn(None, a.ctypes.data, a.size, 0, None) == 0
        ok &= bool(np.array_equal(a, np.arange(1000, dtype=np.int64) * sum(range(1, world + 1))))
        b = np.full(17, 0.5 + rank, dtype=np.float64)
        ok &= cfn(None, b.ctypes.data, b.size, 1, None) == 0
        ok &= bool(np.allclose(b, sum(0.5 + r for r in range(world))))
        c = np.array([5 + rank, 100 - rank, 7], dtype=np.int64)
        ok &= cfn(None, c.ctypes.data, c.size, 2, None) == 0
        ok &= bool(np.array_equal(c, [5, 100 - (world - 1), 7]))
        ok &= hook.calls == 3 and hook.bytes == 8 * (1000 + 17 + 3)
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_allreduce_hook_gloo_world2():
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=100) for _ in procs)
    for p in procs:
        p.join(30)
    assert res == [(0, True), (1, True)]
