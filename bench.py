#!/usr/bin/env python
"""bench.py -- cells/s to convergence of the Harmony clustering+correction loop on MI355X.

Workload (BASELINE.json configs[2], the one the metric is quoted on and that fits one GPU): synthetic
1M cells x 50 PCs, K=100, 10 batches, reference defaults (sigma 0.1, theta 2, lambda auto, block.size 0.05,
max.iter.cluster 4, epsilon 1e-3 / 1e-2, max_iter 10).  With --gpus N every rank holds 1M cells (weak
scaling; 8 ranks ~ configs[3] at 8M cells) and the accumulators are all-reduced over RCCL.

A "step" = one full run from HBM-resident inputs: hmx_restart -> init_cluster_cpp (k-means seeding + 10 Lloyd)
-> {cluster_cpp, moe_correct_ridge_cpp, check_convergence}* until converged.  Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from bench_data import synth  # noqa: E402


def run_to_convergence(obj, max_iter=10):
    obj.restart()
    obj.init_cluster_cpp()
    it = 0
    for it in range(1, max_iter + 1):
        st = obj.cluster_cpp()
        if st != 0:
            raise RuntimeError("cluster_cpp status %d" % st)
        obj.moe_correct_ridge_cpp()
        if obj.check_convergence(1):
            break
    return it


def cpu_baseline(cells, d, K, levels, seed):
    """The oracle (a port: the reference itself cannot be built here) in faithful fp32 mode, GEMM through
    OpenBLAS with 1 thread (the reference's default ncores=1), timed on a bounded sample of the same workload."""
    from harmony_amd import harmony_options, prepare_setup_args
    from oracle.oracle import OracleHarmony, use_openblas
    blas = use_openblas(1)
    Z, meta, _ = synth(cells, d=d, levels=levels, seed=seed)
    skw, _ = prepare_setup_args(Z, meta, list(meta), nclust=K)
    o = OracleHarmony(accurate=False, seed=1)
    o.setup(**skw)
    t0 = time.time()
    o.init_cluster_cpp()
    it = 0
    for it in range(1, 11):
        o.cluster_cpp()
        o.moe_correct_ridge_cpp()
        if o.check_convergence(1):
            break
    dt = time.time() - t0
    return {"value": cells / dt, "unit": "cells/s", "cores": 1, "kind": "port",
            "sample": "oracle (faithful fp32%s), %d cells x %d PCs, K=%d, %d batches, to convergence (%d iterations, %.1f s)"
                      % (", OpenBLAS sgemm 1 thread" if blas else "", cells, d, K, levels[0], it, dt),
            "host_cores_available": os.cpu_count()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--cells-per-gpu", type=int, default=1000000)
    ap.add_argument("--pcs", type=int, default=50)
    ap.add_argument("--clusters", type=int, default=100)
    ap.add_argument("--batches", type=int, default=10)
    ap.add_argument("--cpu-sample", type=int, default=100000, help="cells for the CPU baseline (0 = skip)")
    ap.add_argument("--seed", type=int, default=7)
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1 (nccl == RCCL; gloo only for smoke tests on one GPU)")
    a = ap.parse_args()

    import torch
    from harmony_amd import Harmony, prepare_setup_args

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (a.gpus, world))
    local_rank = local_rank % max(torch.cuda.device_count(), 1)   # (smoke tests may oversubscribe one GPU with gloo)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        # backend "nccl" IS RCCL on ROCm: the accumulators are all-reduced over xGMI through torch.distributed
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(a.backend)
            os.environ["HMX_BENCH_COMM"] = "torch"     # no RCCL communicator without nccl: use the all-reduce hook

    n, d, K, B = a.cells_per_gpu, a.pcs, a.clusters, a.batches
    N = n * world
    Z, meta, _ = synth(n, d=d, levels=(B,), seed=a.seed, shard=rank)
    # harmony_amd is loaded AFTER torch initialised its HIP runtime: both then share ONE runtime in the process,
    # so torch streams, zero-copy tensor views of the library's buffers and RCCL all interoperate.
    obj = Harmony(device=local_rank, seed=1)
    obj.set_stream(torch.cuda.current_stream().cuda_stream)
    N_b = None
    comm_kind = "none"
    if world > 1:
        warm = torch.ones(1, device=dev)
        dist.all_reduce(warm)                         # torch loads + initialises its RCCL; the library binds to the same one
        torch.cuda.synchronize()
        ok = 0
        if os.environ.get("HMX_BENCH_COMM", "rccl") == "rccl":
            # built-in communicator: ncclAllReduce issued by the C library on its own stream -- no Python per collective
            uid = [Harmony.comm_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(uid, src=0, device=dev)
            # ncclCommInitRank runs in a watchdog thread: if the bootstrap of the extra communicator never returns on this
            # node, every rank falls back to the torch.distributed hook instead of hanging the whole job
            import threading
            res = {}

            def _init():
                try:
                    obj.comm_init(rank, world, uid[0])
                    obj.set_shard(rank, world, rank * n, N, None)
                    res["ok"] = 1
                except Exception as e:                # pragma: no cover
                    res["err"] = e

            th = threading.Thread(target=_init, daemon=True)
            th.start()
            th.join(timeout=float(os.environ.get("HMX_COMM_INIT_TIMEOUT", "120")))
            ok = 1 if res.get("ok") else 0
            if not ok:                                # pragma: no cover
                print("rank %d: built-in RCCL communicator failed (%s); using the torch.distributed hook"
                      % (rank, res.get("err", "ncclCommInitRank timed out")), file=sys.stderr)
            flag = torch.tensor([ok], device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            ok = int(flag.item())
        if ok:
            comm_kind = "RCCL over xGMI, ncclAllReduce from the C library (communicator bootstrapped through torch.distributed)"
        else:
            from harmony_amd.dist import TorchAllReduce
            obj = Harmony(device=local_rank, seed=1)
            obj.set_stream(torch.cuda.current_stream().cuda_stream)
            obj.set_shard(rank, world, rank * n, N, TorchAllReduce(device=dev))
            comm_kind = "torch.distributed %s all_reduce hook%s" % (a.backend, " (RCCL over xGMI)" if a.backend == "nccl" else "")
        cnt = torch.from_numpy(np.bincount(meta["cov0"], minlength=B).astype(np.int64)).to(dev)
        dist.all_reduce(cnt)
        N_b = cnt.cpu().numpy().astype(float)
    skw, _ = prepare_setup_args(Z, meta, "cov0", nclust=K, N_b=N_b, levels={"cov0": np.arange(B)})
    obj.setup(**skw)
    del Z

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(a.warmup):
        run_to_convergence(obj)
    obj.set_profile(True)  # HIP events around every launch of the dominant kernel, on the library's stream
    sync()
    t0 = time.perf_counter()
    iters = []
    for _ in range(a.steps):
        iters.append(run_to_convergence(obj))
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = 1e3 * dt / a.steps
    rounds = int(np.sum(obj.kmeans_rounds))

    # roofline of the dominant kernel (k_update: E-step of one block of cells).  Algorithmic bytes per cell
    # per launch: read the cell's normalised embedding row (4d) + write its R row (4K)  [DESIGN.md]
    upd_ms = obj._scalar("prof:update_ms")
    upd_launches = obj._scalar("prof:update_launches")
    upd_cells = obj._scalar("prof:update_cells")  # cells summed over rounds (every round touches every cell once)
    alg_bytes = upd_cells * (4.0 * d + 4.0 * K)
    achieved = alg_bytes / (upd_ms * 1e-3) / 1e9 if upd_ms > 0 else 0.0
    traffic = None  # HBM bytes per launch from the PMC passes (collected separately, see profiles/r1_pmc_summary.json)
    try:
        pm = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic_update_kernel.json")))
        if pm["workload"] == {"cells_per_gpu": n, "pcs": d, "clusters": K, "batches": B}:
            traffic = pm["hbm_bytes_per_launch"]
    except Exception:
        pass
    roofline = {"kernel": "k_tile<NCT,0> (block update of update_R)", "bound": "hbm", "achieved": achieved, "peak": 8000.0,
                "unit": "GB/s", "frac": achieved / 8000.0, "traffic": traffic,
                "avg_launch_us": 1e3 * upd_ms / max(upd_launches, 1), "launches": int(upd_launches),
                "alg_bytes_per_launch": alg_bytes / max(upd_launches, 1),
                "kernel_time_share": upd_ms / (1e3 * dt) if dt > 0 else None}
    out = {
        "metric": "cells_per_sec_to_convergence", "value": N / (ms_per_step * 1e-3), "unit": "cells/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms_per_step,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "synthetic %d cells x %d PCs, K=%d, %d batches%s (BASELINE configs[2] per GPU)"
                               % (N, d, K, B, "" if world == 1 else ", %d cells/GPU cell-sharded" % n),
                   "parallelism": ("cells sharded x%d, all-reduce of O/E/statistics: %s" % (world, comm_kind)) if world > 1 else "single GPU",
                   "harmony_iterations": iters, "kmeans_rounds_last_step": rounds,
                   "s_per_iter": 1e-3 * ms_per_step / max(float(np.mean(iters)), 1.0),
                   "host_phase_ms_per_step": {k: round(obj.timer(k) / (a.steps + a.warmup), 3) for k in
                                              ("init_cluster", "cluster", "update_R", "moe_correct_ridge", "moe_solve_host")}},
        "roofline": roofline,
    }
    if rank == 0 and world == 1 and a.cpu_sample > 0:
        out["cpu_baseline"] = cpu_baseline(a.cpu_sample, d, K, (B,), a.seed)
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
