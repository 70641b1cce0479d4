"""Two PROCESSES, one shard of the cells each, driving the whole sharded RunHarmony (setup -> k-means init -> cluster /
correct to convergence) with the accumulators all-reduced through torch.distributed.  On a single-GPU box both ranks share
GPU 0 and the backend is gloo (RCCL refuses two ranks on one device); on a multi-GPU node run it with --backend nccl.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 tools/dist_two_proc.py

Rank 0 also runs the unsharded problem and checks: identical O tables (integer accumulators => shard-count independent),
identical iteration counts and objective series, Z_corr equal to 1e-6.  Prints DIST2_OK."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import torch.distributed as dist

ap = argparse.ArgumentParser()
ap.add_argument("--backend", default="gloo")
ap.add_argument("--cells", type=int, default=30000)
ap.add_argument("--p2p", action="store_true", help="sum the block contributions inside the persistent chain over the peers' inboxes "
                "(hmx_p2p_*; the handles travel through torch.distributed) instead of one all-reduce per block")
ap.add_argument("--carry", action="store_true", help="force the round-to-round carry of the old contributions (HMX_SOLD_CARRY=1)")
ap.add_argument("--split", type=float, default=0.0, help="two ranks: rank 0 holds this fraction of the cells (unequal shards)")
ap.add_argument("--workload", default="c3", choices=["c3", "c5"], help="c5: BASELINE configs[4]'s shape (K = 200, nested covariates 8 > 64 > 128): the block chain by wave pairs "
                "(k_tile MODE 6; HMX_CHAIN_PAIR=0: one launch + one inbox all-reduce per block step), ridge statistics of Q K (d + 1) > 65536 entries -- the inboxes' reduce-scatter + all-gather path")
ap.add_argument("--chain-max-tpw", default=None, help="HMX_CHAIN_MAX_TPW: tiles per wave up to which a rank would pick the persistent chain")
a = ap.parse_args()
if a.carry:
    os.environ["HMX_SOLD_CARRY"] = "1"
if a.chain_max_tpw:
    os.environ["HMX_CHAIN_MAX_TPW"] = a.chain_max_tpw
if a.p2p and a.backend == "gloo":
    os.environ["HMX_CHAIN_WGS"] = "120"     # both ranks share ONE GPU here: two persistent chains must fit its 256 CUs together
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
ndev = torch.cuda.device_count()
dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")) % ndev)
torch.cuda.set_device(dev)
dist.init_process_group(a.backend, **({"device_id": dev} if a.backend == "nccl" else {}))

from bench_data import synth  # noqa: E402
from harmony_amd import Harmony, prepare_setup_args  # noqa: E402
from harmony_amd.dist import TorchAllReduce, shard_bounds  # noqa: E402

N, K, B = a.cells, 100, 10
levels, nested = (B,), False
if a.workload == "c5":
    K, levels, nested = 200, (8, 64, 128), True
Z, meta, _ = synth(N, d=50, levels=levels, seed=21, nested=nested)          # every rank generates the global problem, keeps its shard
vars_use = list(meta)
bounds = shard_bounds(N, world)
if a.split > 0 and world == 2:
    cut = int(N * a.split)
    bounds = [(0, cut), (cut, N)]
lo, hi = bounds[rank]
N_b = np.concatenate([np.bincount(meta[v], minlength=L).astype(float) for v, L in zip(vars_use, levels)])
skw, _ = prepare_setup_args(Z[lo:hi], {v: meta[v][lo:hi] for v in vars_use}, vars_use, nclust=K, N_b=N_b, levels={v: np.arange(L) for v, L in zip(vars_use, levels)})


def run(obj):
    obj.init_cluster_cpp()
    it = 0
    for it in range(1, 6):
        assert obj.cluster_cpp() == 0
        obj.moe_correct_ridge_cpp()
        if obj.check_convergence(1):
            break
    return it


g = Harmony(device=dev.index, seed=4)
g.set_stream(torch.cuda.current_stream().cuda_stream)
hook = TorchAllReduce(device=dev)
g.set_shard(rank, world, lo, N, hook)
if a.p2p:
    handles = [None] * world
    dist.all_gather_object(handles, g.p2p_export())
    g.p2p_connect(rank, world, handles)
    dist.barrier()
    ok = torch.tensor([1 if g.p2p_selftest() else 0])
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    assert ok.item() == 1, g.p2p_status
    g.p2p_enable(True)
g.setup(**skw)
import time  # noqa: E402
dist.barrier(); t_run = time.perf_counter()
it = run(g)
g.getZcorr(); dist.barrier(); t_run = time.perf_counter() - t_run
Zs = torch.from_numpy(np.ascontiguousarray(g.getZcorr().T))          # [n_local, d]
parts = [torch.empty((h - l, Zs.shape[1]), dtype=Zs.dtype) for l, h in bounds] if rank == 0 else None
if a.backend == "gloo" and a.split > 0:
    objs = [None] * world if rank == 0 else None          # (gloo's gather wants equal shapes)
    dist.gather_object(Zs, objs, dst=0)
    parts = objs
elif a.backend == "gloo":
    dist.gather(Zs, parts, dst=0)
else:
    gl = [torch.empty((h - l, Zs.shape[1]), dtype=Zs.dtype, device=dev) for l, h in bounds]
    dist.all_gather(gl, Zs.to(dev))
    parts = [p.cpu() for p in gl]
O_sh, obj_sh = g.O, g.objective_kmeans
if rank == 0:
    one = Harmony(device=dev.index, seed=4)
    skw1, _ = prepare_setup_args(Z, meta, vars_use, nclust=K)
    one.setup(**skw1)
    it1 = run(one)
    Zall = torch.cat(parts).numpy().T
    assert it == it1, (it, it1)
    # (the tables of ONE clustering pass are integer sums and shard-count independent; after a correction the ridge statistics'
    #  fp32 partial sums group differently per shard, which moves Y -- and through it O -- by a few 1e-7 relative)
    np.testing.assert_allclose(O_sh, one.O, rtol=1e-5, atol=1e-3)
    np.testing.assert_allclose(obj_sh, one.objective_kmeans, rtol=1e-6)
    rel = np.linalg.norm(Zall - one.getZcorr()) / np.linalg.norm(one.getZcorr())
    assert rel < 1e-6, rel
    if a.p2p and not a.chain_max_tpw and a.workload == "c3":
        assert g._scalar("p2p") == 1 and g._scalar("chain") == 1, (g.p2p_status, g._scalar("chain"))
    print("DIST2_OK world=%d backend=%s p2p=%d chain=%d iterations=%d collectives/rank=%d inbox_allreduces/rank=%d big_windows/rank=%d Z_rel=%.1e run=%.1f ms (%s)"
          % (world, a.backend, int(a.p2p), int(g._scalar("chain")), it, hook.calls, int(g._scalar("p2p:allreduce_calls")), int(g._scalar("p2p:allreduce_big_windows")),
             rel, 1e3 * t_run, g.p2p_status), flush=True)
dist.barrier()
dist.destroy_process_group()
