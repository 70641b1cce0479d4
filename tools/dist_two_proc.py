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
a = ap.parse_args()
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
ndev = torch.cuda.device_count()
dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")) % ndev)
torch.cuda.set_device(dev)
dist.init_process_group(a.backend, **({"device_id": dev} if a.backend == "nccl" else {}))

from bench_data import synth  # noqa: E402
from harmony_amd import Harmony, prepare_setup_args  # noqa: E402
from harmony_amd.dist import TorchAllReduce, shard_bounds  # noqa: E402

N, K, B = a.cells, 100, 10
Z, meta, _ = synth(N, d=50, levels=(B,), seed=21)          # every rank generates the global problem, keeps its shard
lo, hi = shard_bounds(N, world)[rank]
N_b = np.bincount(meta["cov0"], minlength=B).astype(float)
skw, _ = prepare_setup_args(Z[lo:hi], {"cov0": meta["cov0"][lo:hi]}, "cov0", nclust=K, N_b=N_b, levels={"cov0": np.arange(B)})


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
g.setup(**skw)
it = run(g)
Zs = torch.from_numpy(np.ascontiguousarray(g.getZcorr().T))          # [n_local, d]
parts = [torch.empty((h - l, Zs.shape[1]), dtype=Zs.dtype) for l, h in shard_bounds(N, world)] if rank == 0 else None
if a.backend == "gloo":
    dist.gather(Zs, parts, dst=0)
else:
    gl = [torch.empty((h - l, Zs.shape[1]), dtype=Zs.dtype, device=dev) for l, h in shard_bounds(N, world)]
    dist.all_gather(gl, Zs.to(dev))
    parts = [p.cpu() for p in gl]
O_sh, obj_sh = g.O, g.objective_kmeans
if rank == 0:
    one = Harmony(device=dev.index, seed=4)
    skw1, _ = prepare_setup_args(Z, meta, "cov0", nclust=K)
    one.setup(**skw1)
    it1 = run(one)
    Zall = torch.cat(parts).numpy().T
    assert it == it1, (it, it1)
    np.testing.assert_allclose(O_sh, one.O, rtol=1e-6, atol=1e-4)
    np.testing.assert_allclose(obj_sh, one.objective_kmeans, rtol=1e-6)
    rel = np.linalg.norm(Zall - one.getZcorr()) / np.linalg.norm(one.getZcorr())
    assert rel < 1e-6, rel
    print("DIST2_OK world=%d backend=%s iterations=%d collectives/rank=%d Z_rel=%.1e" % (world, a.backend, it, hook.calls, rel), flush=True)
dist.barrier()
dist.destroy_process_group()
