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


def _prep_worker(rank, world, port, q):
    """bench.py's N > 1 host preparation, line by line: this rank's cells from the counter-based generator, GLOBAL level counts by all-reduce,
    argument preparation with those counts and with the level sets fixed (a shard may miss a level)."""
    import sys
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench_data import synth
    from harmony_amd import harmony_options, prepare_setup_args
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n, levels = 60, (3, 40)
        Z, meta, _ = synth(n, d=8, levels=levels, seed=5, shard=rank)
        N_b = []
        for v, L in zip(meta, levels):
            cnt = torch.from_numpy(np.bincount(meta[v], minlength=L).astype(np.int64))
            dist.all_reduce(cnt)
            N_b.append(cnt.numpy().astype(float))
        N_b = np.concatenate(N_b)
        skw, _ = prepare_setup_args(Z, meta, list(meta), nclust=12, N_b=N_b, levels={v: np.arange(L) for v, L in zip(meta, levels)},
                                    options=harmony_options(tau=5))
        q.put((rank, skw["theta"].tolist(), [int(x) for x in skw["B_vec"]], [int(x) for x in skw["Phi"][0]], int(skw["Phi"][3]),
               {k: v.tolist() for k, v in meta.items()}, N_b.tolist()))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_sharded_argument_preparation_gloo_world2():
    """The host side of a sharded run (bench.py, N > 1): every rank prepares hmx_setup's arguments from ITS cells plus all-reduced level
    counts.  Two gloo ranks against one process holding both shards: the same level-expanded theta (tau > 0 makes it depend on the global
    N_b, R/ui.R:254-258), the same B_vec although the second covariate's 40 levels are not all present in a 60-cell shard, and each
    rank's design rows = the single process's rows of its cells."""
    import torch.multiprocessing as mp
    from bench_data import synth
    from harmony_amd import harmony_options, prepare_setup_args
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_prep_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=100) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(30)
    shards = [synth(60, d=8, levels=(3, 40), seed=5, shard=r) for r in range(2)]
    Z = np.concatenate([sh[0] for sh in shards])
    meta = {k: np.concatenate([sh[1][k] for sh in shards]) for k in shards[0][1]}
    whole, _ = prepare_setup_args(Z, meta, list(meta), nclust=12, levels={"cov0": np.arange(3), "cov1": np.arange(40)}, options=harmony_options(tau=5))
    rows = np.asarray(whole["Phi"][0]).reshape(-1, 2)            # C = 2 design rows per cell, cell-major
    assert len(set(np.unique(shards[0][1]["cov1"]))) < 40         # (the case the fixed level sets exist for)
    for rank, theta, B_vec, phi_i, B, m, N_b in res:
        assert np.array_equal(theta, whole["theta"]) and B_vec == [3, 40] and B == 43
        assert np.array_equal(np.asarray(phi_i).reshape(-1, 2), rows[rank * 60:(rank + 1) * 60])
        assert all(np.array_equal(m[k], shards[rank][1][k]) for k in m)
        assert np.array_equal(N_b, np.concatenate([np.bincount(meta[k], minlength=L) for k, L in zip(meta, (3, 40))]))
    assert not np.array_equal(whole["theta"], np.full(43, 2.0))    # tau really scaled it


def test_bench_defaults_by_gpu_count():
    """bench.py's size defaults: one GPU = BASELINE configs[2]; --gpus N > 1 with no size = configs[3] (10M cells, 20 batches, strong
    scaling); an explicit --cells-per-gpu keeps weak scaling; an explicit --total-cells / --batches is respected."""
    import argparse
    import bench
    def ns(**kw):
        a = argparse.Namespace(gpus=1, workload="c3", cells_per_gpu=None, total_cells=0, batches=None)
        a.__dict__.update(kw)
        return bench.apply_size_defaults(a)
    a = ns()
    assert (a.cells_per_gpu, a.total_cells, a.batches, a.default_multi) == (1000000, 0, 10, False)
    for g in (2, 4, 8):
        a = ns(gpus=g)
        assert (a.total_cells, a.batches, a.default_multi) == (10000000, 20, True) and a.total_cells % g == 0
    a = ns(gpus=8, cells_per_gpu=1000000)
    assert (a.total_cells, a.batches, a.default_multi) == (0, 10, False)
    a = ns(gpus=4, total_cells=4000000, batches=10)
    assert (a.total_cells, a.batches, a.default_multi) == (4000000, 10, False)
    a = ns(gpus=8, workload="c5")
    assert (a.total_cells, a.default_multi) == (0, False)
