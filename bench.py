#!/usr/bin/env python
"""bench.py -- cells/s to convergence of the Harmony clustering+correction loop on MI355X.

Default workload (BASELINE.json configs[2], the one the metric is quoted on and that fits one GPU): synthetic
1M cells x 50 PCs, K=100, 10 batches, reference defaults (sigma 0.1, theta 2, lambda auto, block.size 0.05,
max.iter.cluster 4, epsilon 1e-3 / 1e-2, max_iter 10).  `--gpus N` (N > 1) with no size on the command line runs BASELINE
configs[3] -- 10M cells x 50 PCs, K=100, 20 batches IN TOTAL, cell-sharded over the N GPUs ("scaling": "strong": north_star's
">= 6x at 8 GPUs vs 1" is a statement about that workload), accumulators all-reduced over RCCL / exchanged through the peers'
inboxes; the weak-scaling companion (1M cells on every GPU) is the line's `also.weak_scaling_1M_per_gpu`, or the main line when
--cells-per-gpu is given.  `--gpus N` without a launcher (WORLD_SIZE unset) re-launches itself under torch.distributed.run with N
ranks; the line's n_gpus always equals --gpus or the run exits non-zero.

Other driver-reproducible modes:
    --cells-per-gpu 10000000 --batches 20          north_star target: 10M x 50 x K=100 on ONE GPU (configs[3]'s size)
    --workload c5 [--cells-per-gpu N]              configs[4] shape: K=200, 3 nested covariates 8 > 64 > 128 (200 levels)
    --total-cells 10000000 --batches 20 --gpus N   STRONG scaling: configs[3] = 10M cells in total, sharded over the N GPUs
    --also ref,10M,share,c5,pbmc (default at N=1 incl. ref2 = "ref_arith" 2; "none"; "shares": the per-rank shares of the 8-GPU configs at G = 2 / 4 / 8)  extra legs of the same invocation, reported under "also": the reference-arithmetic
                                                   mode on the main workload, 10M cells on this one GPU, the configs[4] shape at 1M,
                                                   configs[1] (pbmc, 30k cells) with its own CPU-oracle timing and parity check

A "step" = one full run from HBM-resident inputs: hmx_restart -> init_cluster_cpp (k-means seeding + 10 Lloyd)
-> {cluster_cpp, moe_correct_ridge_cpp, check_convergence}* until converged.  Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import socket
import subprocess
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


def cpu_baseline(cells, d, K, levels, nested, seed):
    """The oracle (a port: the reference itself cannot be built here) in faithful fp32 mode on a bounded sample of the same
    workload, GEMM through OpenBLAS: 1 thread (the reference's default ncores = 1, R/ui.R:101) = `value`; and all host cores
    (`all_cores`; like the reference's ncores > 1, only the BLAS calls are threaded).  The timed single-thread run doubles as the
    checker of the GPU's reference-arithmetic mode on the same sample (`gpu_reference_arith_vs_this_run`)."""
    from harmony_amd import Harmony, prepare_setup_args
    from oracle.oracle import OracleHarmony, use_openblas
    Z, meta, _ = synth(cells, d=d, levels=levels, seed=seed, nested=nested)
    skw, _ = prepare_setup_args(Z, meta, list(meta), nclust=K)
    out = {}
    ncpu = os.cpu_count() or 1
    for tag, threads in (("one", 1), ("all", ncpu)):
        blas = use_openblas(threads)
        o = OracleHarmony(accurate=False, seed=1)
        o.setup(**skw)
        t0 = time.time()
        o.init_cluster_cpp()
        if tag == "one":
            Y0 = np.array(o.Y, copy=True)          # (normalised centroids after the oracle's own k-means: shared with the GPU check below)
        it = 0
        for it in range(1, 11):
            o.cluster_cpp()
            o.moe_correct_ridge_cpp()
            if o.check_convergence(1):
                break
        dt = time.time() - t0
        out[tag] = (cells / dt, it, dt, blas)
        if tag == "one":
            ref_check = None
            if len(levels) == 1:
                try:       # the GPU in reference arithmetic on the same sample, same centroids, same documented shuffles
                    g = Harmony(seed=1, ref_arith=1)
                    g.setup(**skw)
                    g.init_cluster_cpp(Y0)
                    ig = 0
                    for ig in range(1, 11):
                        g.cluster_cpp()
                        g.moe_correct_ridge_cpp()
                        if g.check_convergence(1):
                            break
                    Zg, Zc, Rg, Rc = g.getZcorr(), o.getZcorr(), g.R, o.R
                    bad = np.where(Rg.argmax(axis=0) != Rc.argmax(axis=0))[0]
                    srt = np.sort(Rc[:, bad], axis=0) if bad.size else np.zeros((2, 0))
                    nob = min(len(g.objective_kmeans), len(o.objective_kmeans))
                    ref_check = {"Z_rel_frobenius": float(np.linalg.norm(Zg - Zc) / np.linalg.norm(Zc)), "hard_assignment_flips": int(bad.size),
                                 "flips_with_margin_ge_1e-5": int(((srt[-1] - srt[-2]) >= 1e-5).sum()) if bad.size else 0,
                                 "objective_rel_max": float(np.max(np.abs(g.objective_kmeans[:nob] - o.objective_kmeans[:nob]) / np.abs(o.objective_kmeans[:nob]))),
                                 "iterations_gpu_cpu": [int(ig), int(it)]}
                    del g
                except Exception as e:                # pragma: no cover
                    ref_check = {"error": repr(e)}
        if ncpu == 1:
            out["all"] = out["one"]
            break
    v1, it, dt, blas = out["one"]
    va, _, dta, _ = out["all"]
    return {"value": v1, "unit": "cells/s", "cores": 1, "kind": "port", "value_measured_on": "%d-cell sample of the workload (--cpu-sample; --cpu-full times the full size, ~1 min at 1M)" % cells,
            "reference_sources": _reference_sources_cpu(d, K, levels, nested, seed),
            "sample": "oracle (faithful fp32%s), %d cells x %d PCs, K=%d, levels %s, to convergence (%d iterations, %.1f s); the "
                      "algorithm is O(N) per iteration, so cells/s at the full size is the same figure up to the iteration count"
                      % (", OpenBLAS sgemm" if blas else "", cells, d, K, "x".join(map(str, levels)), it, dt),
            "all_cores": {"value": va, "cores": ncpu, "seconds": dta,
                          "note": "only the distance GEMM is threaded (as in the reference with ncores > 1, R/ui.R:123-128): the sequential accumulators dominate, more cores do not help"},
            "host_cores_available": ncpu, "full_size": _full_size_cpu(),
            "gpu_reference_arith_vs_this_run": ref_check}


def _reference_sources_cpu(d, K, levels, nested, seed, cells=40000):
    """oracle/_ref/libharmony_ref.so -- the reference's OWN src/harmony.cpp / utils.cpp / timer.cpp, compiled in the build container where
    they lie over oracle/shim's stand-in for the Armadillo / Rcpp headers; the file travels with the tree -- timed on a smaller sample of the
    same workload, one thread.  Reported beside `value`, not instead of it: the shim's Armadillo kernels are eager and BLAS-free, so the
    port with OpenBLAS's sgemm is the faster (= fairer) CPU figure."""
    try:
        from harmony_amd import prepare_setup_args
        from oracle import ref as oref
        if not os.path.exists(oref._SO):
            return None
        Z, meta, _ = synth(cells, d=d, levels=levels, seed=seed, nested=nested)
        skw, _ = prepare_setup_args(Z, meta, list(meta), nclust=K)
        r = oref.RefHarmony(seed=1)
        r.setup(**skw)
        # the reference's sources print progress notes (Rcout -> stdout, e.g. a re-drawn seed in kmeans_centers): this process's stdout is
        # reserved for the ONE JSON line, so file descriptor 1 points at stderr while they run
        sys.stdout.flush()
        keep = os.dup(1)
        os.dup2(2, 1)
        try:
            t0 = time.time()
            r.init_cluster_cpp()
            it = 0
            for it in range(1, 11):
                r.cluster_cpp()
                r.moe_correct_ridge_cpp()
                if r.check_convergence(1):
                    break
            dt = time.time() - t0
        finally:
            try:
                oref.load().ref_flush_stdout()
            except Exception:           # pragma: no cover
                pass
            os.dup2(keep, 1)
            os.close(keep)
        return {"value": cells / dt, "unit": "cells/s", "cores": 1, "kind": "reference",
                "sample": "the reference's own engine sources (oracle/_ref, over oracle/shim's Armadillo stand-in: eager, no BLAS), %d cells x %d PCs, K=%d, "
                          "levels %s, to convergence (%d iterations, %.1f s)" % (cells, d, K, "x".join(map(str, levels)), it, dt)}
    except Exception as e:                # pragma: no cover
        return {"error": repr(e)}


def _full_size_cpu():
    """the same oracle at the FULL configs[2] size, from the committed parity table (profiles/, a builder run: ~1 minute of CPU)"""
    try:
        j = json.load(open(next(f for f in (os.path.join(ROOT, "profiles", "%s_parity_table_1000000.json" % t) for t in ("r6", "r5", "r4", "r3")) if os.path.exists(f))))
        s = j["seconds"]["oracle_faithful"]
        return {"cells": j["workload"]["cells"], "seconds": s, "value": j["workload"]["cells"] / s, "unit": "cells/s",
                "source": "profiles/r*_parity_table_1000000.json of the latest round (tests/test_gpu_parity2.py::test_arithmetic_gap_table[1000000]; 4 BLAS threads, "
                          "shared k-means centres: the init is not in this figure)"}
    except Exception:
        return None


def bench_leg(Harmony, prepare_setup_args, n, d, K, levels, nested, seed, steps, warmup, sync, data=None, **hkw):
    """one extra workload / mode on this GPU: time to convergence from HBM-resident inputs, same step definition as the main line;
    `data` = (Z, meta, label) replaces the synthetic generator (the pbmc 30k leg)"""
    if data is None:
        Z, meta, _ = synth(n, d=d, levels=levels, seed=seed, nested=nested)
        label = "synthetic %d cells x %d PCs, K=%d, levels %s%s" % (n, d, K, "x".join(map(str, levels)), " nested" if nested else "")
    else:
        Z, meta, label = data
    skw, _ = prepare_setup_args(Z, meta, list(meta), nclust=K)
    o = Harmony(seed=1, **hkw)
    o.setup(**skw)
    del Z
    for _ in range(warmup):
        run_to_convergence(o)
    o.set_profile(1)
    sync()
    t0 = time.perf_counter()
    its = [run_to_convergence(o) for _ in range(steps)]
    sync()
    ms = 1e3 * (time.perf_counter() - t0) / steps
    upd_ms, upd_steps = o._scalar("prof:update_ms"), max(o._scalar("prof:update_steps"), 1)
    upd_cells = o._scalar("prof:update_cells")
    kr = np.asarray(o.kmeans_rounds, dtype=np.int64)
    o.set_profile(2)          # the phase table comes from ONE more, untimed run (its ~200 event records would cost the timed ones up to 1 ms each)
    run_to_convergence(o)
    sync()
    ph = {k: round(o._scalar("gputimer:" + k), 3) for k in ("kmeans_centers", "cluster_head", "randomize", "EO_update", "Rcells_update",
                                                             "objective", "ridge_statistics", "arma_inv", "update_Zcorr")}
    ph["Rcells_update"] = round(upd_ms / steps, 3)
    run_bytes = float(n) * float(np.sum(4.0 * d * (4 + kr) + 4.0 * K * (3 + 2 * kr)))
    chain = bool(o._scalar("chain"))
    ach = upd_cells * (4.0 * d + 4.0 * K) / (upd_ms * 1e-3) / 1e9 if upd_ms > 0 else 0.0
    out = {"workload": label,
           "ms_per_step": ms, "cells_per_s": n / (ms * 1e-3), "harmony_iterations": its, "steps": steps, "gpu_phase_ms_per_step": ph, "gpu_phase_from": "one extra untimed run with per-phase events",
           "block_chain": chain and hkw.get("ref_arith") != 1, "avg_block_step_us": 1e3 * upd_ms / upd_steps,
           # the leg's own dominant kernel (the E-step update of update_R, as on the main line): algorithmic bytes (4d + 4K per cell and
           # round) over its HIP-event time on the library's stream
           "roofline": {"kernel": ("k_tile<7,6,...> persistent block chain by wave pairs (two halves of the clusters, several folder workgroups)" if (chain and K > 112 and hkw.get("ref_arith") != 1) else
                                   ("k_tile<%d,4|5,...> persistent block chain" if (chain and hkw.get("ref_arith") != 1) else "k_tile<%d,0,...> one launch per block step") % ((K + 15) // 16)),
                        "bound": "hbm", "achieved": ach, "peak": 8000.0, "unit": "GB/s", "frac": ach / 8000.0,
                        "kernel_time_share": upd_ms / (ms * steps) if ms > 0 else None},
           "roofline_run_frac": run_bytes / (ms * 1e-3) / 8e12}
    if hkw:
        out["mode"] = hkw
        out["seq_residual"] = o._scalar("seq:residual")
    if hkw.get("ref_arith"):
        # The reference-arithmetic leg is not the tile kernel's leg: its time goes into the restarted sequential sums.  Roofline of its
        # own dominant phase, from the per-phase HIP-event brackets of the extra untimed run: algorithmic bytes = what the sums of that phase have to read.
        # (`passes` per group from the object: O/E, objective, ridge, level pairs; "seq:oe_cell_passes" = cells x passes the O / E sums walked since setup.)
        runs_total = warmup + steps + 1
        passes = [float(x) for x in o._get("seq:group_passes")]
        evals = [float(x) for x in o._get("seq:group_runs")]
        oe_cells = float(o._scalar("seq:oe_cell_passes")) / runs_total
        C_ = len(levels)
        phases = {
            "EO_update": {"kernels": "k_seq_oe_pass + k_seq_scan (+ k_oe_fold): a block's (all blocks') O / E sums in the round's shuffled order, src/harmony.cpp:312-313,329-330",
                          "alg_bytes": oe_cells * 4.0 * K, "what": "the R rows of the cells a pass walks (4K per cell and pass)"},
            "objective": {"kernels": "k_obj_terms_mfma + k_seq_obj_fused + k_seq_objf_close: compute_objective's three my_accu chains, src/utils.cpp:67-75",
                          "alg_bytes": evals[1] / runs_total * float(n) * (4.0 * d + 16.0 * K),
                          "what": "per evaluation: embedding + R rows in, R % dist out (terms), then R % dist + R rows in once (the passes run from registers)"},
            "ridge_statistics": {"kernels": "k_seq_ridge_pass_kl + k_seq_scan: Phi* diag(R_k) Z^T as sequential sums, src/harmony.cpp:592-608",
                                 "alg_bytes": passes[2] / runs_total * float(n) * (1.0 + C_) * (4.0 * d + 4.0 * K), "what": "per pass every cell once per chain it is in (all cells + one level per covariate): its embedding row and its R row"},
        }
        dom = max(phases, key=lambda kk: ph.get(kk, 0.0))
        for kk, v in phases.items():
            t_ms = ph.get(kk, 0.0)
            v["gpu_ms_per_step"] = t_ms
            v["achieved"] = v["alg_bytes"] / (t_ms * 1e-3) / 1e9 if t_ms > 0 else 0.0
            v["frac"] = v["achieved"] / 8000.0
            v["share_of_step"] = t_ms / ms if ms > 0 else None
        out["roofline_tile_kernel"] = out["roofline"]          # (the E-step update kernel of this leg: 9-10 % of its time)
        out["roofline"] = {"kernel": phases[dom]["kernels"], "phase": dom, "bound": "hbm (latency-bound in practice: chains of dependent small launches)", "achieved": phases[dom]["achieved"],
                           "peak": 8000.0, "unit": "GB/s", "frac": phases[dom]["frac"], "traffic": None, "kernel_time_share": phases[dom]["share_of_step"],
                           "achieved_is": "algorithmic bytes of the phase (%s) over its GPU time from HIP-event brackets around the phase (one extra untimed run)" % phases[dom]["what"],
                           "by_phase": {kk: {q: v[q] for q in ("gpu_ms_per_step", "achieved", "frac", "share_of_step")} for kk, v in phases.items()},
                           "passes_per_step": {"O/E": passes[0] / runs_total, "objective": passes[1] / runs_total, "ridge": passes[2] / runs_total, "level_pairs": passes[3] / runs_total}}
    del o
    return out


def pbmc30k_leg(Harmony, prepare_setup_args, sync):
    """BASELINE configs[1] at its stated size: ~30k cells x 50 PCs, K = 50, stim / ctrl (bench_data.pbmc30k: the shipped 2 000-cell sample
    resampled to 30k), GPU vs the CPU oracle on the same box -- timed, and compared (the GPU run is checked against this very oracle run)"""
    from bench_data import pbmc30k
    from oracle.oracle import OracleHarmony, use_openblas
    Z, meta = pbmc30k()
    n, d, K = Z.shape[0], Z.shape[1], 50
    leg = bench_leg(Harmony, prepare_setup_args, n, d, K, (2,), False, 0, 5, 2, sync,
                    data=(Z, meta, "pbmc_stim stand-in: %d cells x %d PCs, K=%d, stim/ctrl (BASELINE configs[1])" % (n, d, K)))
    skw, _ = prepare_setup_args(Z, meta, list(meta), nclust=K)
    g = Harmony(seed=1)
    g.setup(**skw)
    Y0 = g.kmeans_centers()
    g.init_cluster_cpp(Y0)
    cpu = {}
    for tag, threads, mask in (("oracle_faithful_1_thread", 1, 0), ("oracle_accurate_4_threads", 4, 15)):
        use_openblas(threads)
        o = OracleHarmony(mask=mask, seed=1)
        o.setup(**skw)
        t0 = time.time()
        o.init_cluster_cpp(Y0)
        it = 0
        for it in range(1, 11):
            o.cluster_cpp(); o.moe_correct_ridge_cpp()
            if o.check_convergence(1):
                break
        dt = time.time() - t0
        cpu[tag] = {"seconds": dt, "cells_per_s": n / dt, "iterations": it, "Z": o.getZcorr()}
    ig = 0
    for ig in range(1, 11):
        g.cluster_cpp(); g.moe_correct_ridge_cpp()
        if g.check_convergence(1):
            break
    Zg = g.getZcorr()
    leg["cpu_oracle"] = {k: {kk: vv for kk, vv in v.items() if kk != "Z"} for k, v in cpu.items()}
    leg["parity_this_run"] = {"Z_rel_vs_oracle_accurate": float(np.linalg.norm(Zg - cpu["oracle_accurate_4_threads"]["Z"]) / np.linalg.norm(cpu["oracle_accurate_4_threads"]["Z"])),
                              "Z_rel_vs_oracle_faithful": float(np.linalg.norm(Zg - cpu["oracle_faithful_1_thread"]["Z"]) / np.linalg.norm(cpu["oracle_faithful_1_thread"]["Z"])),
                              "iterations_gpu": ig, "shared": "k-means centres and block partitions"}
    leg["gpu_vs_cpu_1_thread"] = leg["cells_per_s"] / cpu["oracle_faithful_1_thread"]["cells_per_s"]
    return leg


def headline_parity(n, d, K, levels):
    """where the HEADLINE mode (exact accumulators) stands against both oracle arithmetics at this exact workload, replayed from the committed
    parity table of the round (tests/test_gpu_parity2.py::test_arithmetic_gap_table, a driver-run GPU test) -- so that nobody reads `value` as
    a faithful-arithmetic number: the mode that follows the reference's fp32 arithmetic is `also.reference_arith`"""
    for tag in ("r6", "r5", "r4", "r3"):
        try:
            j = json.load(open(os.path.join(ROOT, "profiles", "%s_parity_table_%d.json" % (tag, n))))
        except Exception:
            continue
        w = j["workload"]
        if (w["cells"], w["pcs"], w["clusters"], w["batches"]) != (n, d, K, levels[0]) or len(levels) != 1:
            continue
        pr = j["pairs"]
        return {"parity_mode": "default: exact accumulators (64-bit fixed point / fp64 in a fixed order); parity target = the oracle with fp64 accumulators",
                "Z_rel_vs_accurate": pr["gpu_vs_oracle_accurate"]["Z_rel"], "Z_rel_vs_faithful": pr["gpu_vs_oracle_faithful"]["Z_rel"],
                "clear_flips_vs_accurate": pr["gpu_vs_oracle_accurate"]["argmax_diff_margin_ge_1e-5"],
                "clear_flips_vs_faithful": pr["gpu_vs_oracle_faithful"]["argmax_diff_margin_ge_1e-5"],
                "reference_arith_mode": {"Z_rel_vs_faithful": pr["gpu_ref_arith_vs_oracle_faithful"]["Z_rel"],
                                         "clear_flips_vs_faithful": pr["gpu_ref_arith_vs_oracle_faithful"]["argmax_diff_margin_ge_1e-5"],
                                         "timed_as": "also.reference_arith"},
                "reference_arith_2_mode": ({"Z_rel_vs_faithful": pr["gpu_ref_arith2_vs_oracle_faithful"]["Z_rel"],
                                            "clear_flips_vs_faithful": pr["gpu_ref_arith2_vs_oracle_faithful"]["argmax_diff_margin_ge_1e-5"],
                                            "flips_at_margin_1e-4_vs_faithful": pr["gpu_ref_arith2_vs_oracle_faithful"]["argmax_diff_margin_ge_1e-4"],
                                            "iterations": pr["gpu_ref_arith2_vs_oracle_faithful"]["iterations"],
                                            "is": "ref_arith = 2: the reference's accumulators for the objective, the ridge statistics and the inverse; exact O / E tables",
                                            "timed_as": "also.reference_arith_2"} if "gpu_ref_arith2_vs_oracle_faithful" in pr else None),
                "the_reference_itself_faithful_vs_accurate": pr["oracle_faithful_vs_oracle_accurate"]["Z_rel"],
                "oracle_pinned_to": "the reference's own src/harmony.cpp / utils.cpp / timer.cpp compiled in place over oracle/shim (a stand-in for the Armadillo / "
                                    "Rcpp headers): the faithful oracle equals them bit for bit (tests/test_oracle_ref.py, tests/golden/ref_sources_*.npz); "
                                    "Armadillo's own kernels stay restated; GPU reference arithmetic vs that library: profiles/r5_gpu_vs_reference_sources.txt.  "
                                    "The accurate oracle (this mode's target) is 2.5e-7 from the same sources built with the reference's own -DHARMONY_SCALAR_DOUBLE, "
                                    "this mode 1.2e-7 (20k cells; profiles/r5_reference_double_precision.json); the reference as it ships is 5e-5 ... 2e-4 from that build",
                "replayed_from": "profiles/%s_parity_table_%d.json (not measured in this run; cpu_baseline.gpu_reference_arith_vs_this_run is a live check on a sample)" % (tag, n)}
    return None


def comm_stats(obj):
    """what travelled how in a sharded run (per rank, since setup): collectives left on the communicator / hook, small all-reduces through the
    peers' inboxes, the state of the in-launch exchange"""
    return {"communicator_or_hook_calls": int(obj._scalar("comm:calls")), "communicator_or_hook_bytes": int(obj._scalar("comm:bytes")),
            "inbox_allreduce_calls": int(obj._scalar("p2p:allreduce_calls")), "p2p_chain": bool(obj._scalar("p2p")),
            "p2p_status": obj.p2p_status, "p2p_selftest_us_per_exchange": obj._scalar("p2p:exchange_us")}


def main_file_bootstrap(a):
    """--bootstrap file: the whole N-rank run WITHOUT torch in the process -- the path a plain C / R host takes.  Ranks are separate processes
    (spawned here if no launcher did: RANK / LOCAL_RANK / WORLD_SIZE in the environment), rank 0 creates the RCCL unique id
    (hmx_comm_unique_id) and passes it through a file, hmx_comm_init builds the communicator and connects + self-tests the peers' inboxes,
    barriers / max-over-ranks / the sanity sums go through hmx_comm_allreduce_host.  Same step, same JSON contract as the default path; the
    single-GPU extras (e2e, `also` legs, CPU baseline) stay with the default invocation."""
    import tempfile
    world = int(os.environ.get("WORLD_SIZE", str(a.gpus)))
    if a.gpus > 1 and "RANK" not in os.environ:
        uid_file = os.path.join(tempfile.gettempdir(), "hmx_bench_uid_%d" % os.getpid())
        procs = []
        for r in range(a.gpus):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), HMX_BENCH_UID_FILE=uid_file)
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
        rc = max(p.wait() for p in procs)
        for f in (uid_file, uid_file + ".tmp"):
            if os.path.exists(f):
                os.remove(f)
        raise SystemExit(rc)
    from harmony_amd import Harmony, prepare_setup_args
    rank, local_rank = int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: refusing to print a line whose n_gpus differs from --gpus" % (a.gpus, world))
    ndev = int(os.environ.get("HMX_BENCH_NDEV", "0"))          # (smoke tests: several ranks on one GPU)
    obj = Harmony(device=(local_rank % ndev) if ndev else local_rank, seed=1)
    if a.workload == "c5":
        levels, nested, K = (8, 64, 128), True, (a.clusters or 200)
    else:
        levels, nested, K = (a.batches,), False, (a.clusters or 100)
    strong = a.total_cells > 0
    n, d = (a.total_cells // world if strong else a.cells_per_gpu), a.pcs
    N = n * world
    if world > 1:
        uid_file = os.environ.get("HMX_BENCH_UID_FILE") or os.path.join(tempfile.gettempdir(), "hmx_bench_uid_%s" % os.environ.get("MASTER_PORT", "0"))
        if rank == 0:
            with open(uid_file + ".tmp", "wb") as fh:
                fh.write(Harmony.comm_unique_id())
            os.replace(uid_file + ".tmp", uid_file)
        t0 = time.time()
        while not os.path.exists(uid_file):
            if time.time() - t0 > 120:
                raise SystemExit("rank %d: no unique id file after 120 s" % rank)
            time.sleep(0.01)
        with open(uid_file, "rb") as fh:
            uid = fh.read()
        obj.comm_init(rank, world, uid)              # communicator + inboxes + transport self-test
        obj.set_shard(rank, world, rank * n, N, None)
    Z, meta, _ = synth(n, d=d, levels=levels, seed=a.seed, shard=rank, nested=nested)
    vars_use = list(meta)
    N_b = None
    if world > 1:
        N_b = np.concatenate([obj.comm_allreduce_host(np.bincount(meta[v], minlength=L).astype(float)) for v, L in zip(vars_use, levels)])
    skw, _ = prepare_setup_args(Z, meta, vars_use, nclust=K, N_b=N_b, levels={v: np.arange(L) for v, L in zip(vars_use, levels)})
    obj.setup(**skw)
    del Z

    def sync():
        if world > 1:
            obj.comm_allreduce_host([0.0])           # device synchronisation + barrier
        else:
            obj._scalar("sync")

    preroll = int(os.environ.get("HMX_BENCH_PREROLL", "16" if n <= 2000000 else "2"))      # (untimed: see main())
    for _ in range(preroll + a.warmup + (1 if world > 1 else 0)):
        run_to_convergence(obj)
    obj.set_profile(1)
    sync()
    t0 = time.perf_counter()
    iters = [run_to_convergence(obj) for _ in range(a.steps)]
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        dt = float(obj.comm_allreduce_host([dt], "max")[0])
    ms_per_step = 1e3 * dt / a.steps
    shard_check = None
    if world > 1:
        O = np.ascontiguousarray(obj.O, dtype=np.float64).ravel()
        lo_, hi_ = obj.comm_allreduce_host(O, "min"), obj.comm_allreduce_host(O, "max")
        shard_check = {"O_identical_on_all_ranks": bool((lo_ == hi_).all()), "sum_O_over_N": float(O.sum()) / float(N)}
        if not shard_check["O_identical_on_all_ranks"] or abs(shard_check["sum_O_over_N"] - float(len(levels))) > 1e-4 * len(levels):
            raise SystemExit("sharded run inconsistent: %r" % (shard_check,))
    upd_ms, upd_launches = obj._scalar("prof:update_ms"), obj._scalar("prof:update_launches")
    upd_cells, upd_steps = obj._scalar("prof:update_cells"), obj._scalar("prof:update_steps")
    kr = np.asarray(obj.kmeans_rounds, dtype=np.int64)
    alg_bytes = upd_cells * (4.0 * d + 4.0 * K)
    achieved = alg_bytes / (upd_ms * 1e-3) / 1e9 if upd_ms > 0 else 0.0
    run_bytes = float(n) * float(np.sum(4.0 * d * (4 + kr) + 4.0 * K * (3 + 2 * kr)))
    on_chain = bool(obj._scalar("chain")) and (world == 1 or bool(obj._scalar("p2p")))
    out = {"metric": "cells_per_sec_to_convergence", "value": N / (ms_per_step * 1e-3), "unit": "cells/s", "n_gpus": world, "steps": a.steps,
           "warmup": a.warmup, "preroll_steps_untimed": preroll, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
           "dtype": "f32", "data": "synthetic",
           "config": {"workload": "synthetic %d cells x %d PCs, K=%d, levels %s%s%s" % (N, d, K, "x".join(map(str, levels)), " nested" if nested else "",
                                                                                          "" if world == 1 else ", %d cells/GPU cell-sharded" % n),
                      "parallelism": "single GPU" if world == 1 else "cells sharded x%d, RCCL over xGMI from the C library" % world,
                      "bootstrap": "file: hmx_comm_unique_id -> file -> hmx_comm_init (communicator, inboxes, self-test); no torch in the process",
                      "comm": comm_stats(obj) if world > 1 else None, "shard_check": shard_check, "harmony_iterations": iters,
                      "kmeans_rounds_last_step": int(kr.sum())},
           "roofline": {"kernel": "k_tile<%d,%s,...>: %s" % ((K + 15) // 16, "4" if on_chain else "0", "persistent block chain, one launch per round" if on_chain else "one launch per block step"),
                        "bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0, "traffic": None,
                        "avg_launch_us": 1e3 * upd_ms / max(upd_launches, 1), "avg_block_step_us": 1e3 * upd_ms / max(upd_steps, 1),
                        "run": {"achieved": run_bytes / (ms_per_step * 1e-3) / 1e9, "frac": run_bytes / (ms_per_step * 1e-3) / 8e12}},
           "cpu_baseline": None}
    if rank == 0:
        print(json.dumps(out), flush=True)


def pmc_refresh(n, d, K, B):
    """--pmc: the dominant kernel's HBM bytes and MFMA-busy cycles per launch, collected NOW with rocprofv3 (when it is on PATH) exactly as
    MI355X_MICROARCH.md's HBM section prescribes -- separate --pmc passes (FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES), each with
    --kernel-trace only, over tools/prof_update.py (setup + init + two clustering calls of this workload); FETCH_SIZE x 2 on gfx950 (it reports
    half the bytes of coalesced reads; calibrated on a plain copy kernel in round 3), KB -> bytes.  Writes profiles/pmc_traffic_update_kernel.json
    -- the file the roofline's `traffic` / `mfma_busy_frac` are read from -- and returns it; None (and the old file stays) when rocprofv3 is missing
    or a pass fails."""
    import collections
    import csv
    import shutil
    import tempfile
    exe = shutil.which("rocprofv3")
    if not exe:
        return None
    tmp = tempfile.mkdtemp(prefix="hmx_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp", HMX_BENCH_PREROLL="0")
    means, durs = {}, {}
    try:
        for tag, counters in (("fetch", ["FETCH_SIZE"]), ("write", ["WRITE_SIZE"]), ("sq", ["SQ_VALU_MFMA_BUSY_CYCLES"])):
            out = os.path.join(tmp, tag)
            cmd = [exe, "--pmc"] + counters + ["--kernel-trace", "--output-format", "csv", "-d", out, "-o", "p", "--", sys.executable,
                   os.path.join(ROOT, "tools", "prof_update.py"), str(n), str(K), str(B)]
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=300, stdin=subprocess.DEVNULL)
            found = [os.path.join(dp, f) for dp, _, fs in os.walk(out) for f in fs if f.endswith("counter_collection.csv")]
            if r.returncode != 0 or not found:
                return None
            agg, cnt = collections.defaultdict(float), collections.Counter()
            for row in csv.DictReader(open(found[0])):
                k = row["Kernel_Name"].split("(")[0].replace("void ", "").strip()
                agg[(k, row["Counter_Name"])] += float(row["Counter_Value"]); cnt[(k, row["Counter_Name"])] += 1
            for key, v in agg.items():
                means[key] = v / cnt[key]
            if tag == "sq":      # durations of the same dispatches (the kernel trace of this pass)
                tr = [os.path.join(dp, f) for dp, _, fs in os.walk(out) for f in fs if f.endswith("kernel_trace.csv")]
                if tr:
                    dsum, dcnt = collections.defaultdict(float), collections.Counter()
                    for row in csv.DictReader(open(tr[0])):
                        k = row["Kernel_Name"].split("(")[0].replace("void ", "").strip()
                        dsum[k] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"]); dcnt[k] += 1
                    durs = {k: dsum[k] / dcnt[k] for k in dsum}
    except Exception:
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    nct = (K + 15) // 16

    def kern(mode):
        for (k, c) in means:
            if k.startswith("hmx::k_tile<%d, %d, 2," % (nct, mode)) and c == "FETCH_SIZE":
                return k
        return None

    k4, k5 = kern(4), kern(5)
    if not k4:
        return None

    def total(k):
        return 2.0 * 1000.0 * means[(k, "FETCH_SIZE")] + 1000.0 * means[(k, "WRITE_SIZE")]

    def busy(k):
        ns = durs.get(k)
        return means.get((k, "SQ_VALU_MFMA_BUSY_CYCLES"), 0.0) / (ns * 1e-9 * 2.4e9 * 1024) if ns else None

    pm = {"workload": {"cells_per_gpu": n, "pcs": d, "clusters": K, "batches": B},
          "kernel": "k_tile<%d,4|5,2,...> (persistent block chain; 5: a round whose R rows nobody reads)" % nct,
          "hbm_bytes_per_launch": total(k4), "mfma_busy_frac": busy(k4),
          "collected": "by this bench.py invocation (--pmc): rocprofv3 --pmc <FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES> --kernel-trace, one pass each, over tools/prof_update.py",
          "source": "FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE, KB -> bytes, mean per dispatch; SQ_VALU_MFMA_BUSY_CYCLES / (duration x 2.4 GHz x 1024 SIMDs)"}
    if k5:
        pm["hbm_bytes_per_launch_without_R_stores"] = total(k5); pm["mfma_busy_frac_without_R_stores"] = busy(k5)
    with open(os.path.join(ROOT, "profiles", "pmc_traffic_update_kernel.json"), "w") as fh:
        json.dump(pm, fh, indent=1)
    return pm


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def apply_size_defaults(a):
    """Defaults by N: one GPU = BASELINE configs[2] (1M cells, 10 batches); N > 1 with no size on the command line = BASELINE configs[3] -- 10M cells
    x 50 PCs, K = 100, 20 batches IN TOTAL, cell-sharded over the N GPUs: north_star's ">= 6x cells/s at 8 GPUs vs 1" is a strong-scaling statement
    on that workload.  Weak scaling (--cells-per-gpu given, or the `also.weak_scaling_1M_per_gpu` leg of the default N > 1 line) keeps 1M cells on
    every GPU."""
    a.default_multi = a.gpus > 1 and a.workload == "c3" and a.cells_per_gpu is None and a.total_cells == 0
    if a.default_multi:
        a.total_cells = 10000000
        if a.batches is None:
            a.batches = 20
    if a.cells_per_gpu is None:
        a.cells_per_gpu = 1000000
    if a.batches is None:
        a.batches = 10
    return a


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="c3", choices=["c3", "c5"], help="c3: one covariate (--batches levels), K=--clusters; "
                    "c5: configs[4] shape, K=200, nested covariates 8 > 64 > 128")
    ap.add_argument("--cells-per-gpu", type=int, default=None, help="weak scaling: this many cells on EVERY GPU (default at --gpus 1: 1000000 = configs[2])")
    ap.add_argument("--total-cells", type=int, default=0, help="strong scaling: this many cells in TOTAL, sharded over --gpus (overrides --cells-per-gpu); "
                    "default at --gpus N > 1 when no size is given: 10000000 with --batches 20 = BASELINE configs[3]")
    ap.add_argument("--also", default=None, help="extra legs at N=1, comma separated: ref (reference arithmetic on the main workload), 10M, share (1.25M cells / 20 batches: one GPU's part of configs[3] on 8), c5, pbmc "
                    "(configs[1] at its stated size, with its own CPU oracle timing); default 'ref,10M,share,c5,pbmc' for the default workload on one GPU, 'none' otherwise")
    ap.add_argument("--pcs", type=int, default=50)
    ap.add_argument("--clusters", type=int, default=None)
    ap.add_argument("--batches", type=int, default=None, help="levels of the one covariate (default 10 = configs[2]; 20 with the configs[3] default of --gpus N > 1)")
    ap.add_argument("--cpu-sample", type=int, default=100000, help="cells for the CPU baseline (0 = skip)")
    ap.add_argument("--cpu-full", action="store_true", help="time the CPU baseline at the FULL workload size in this run (~1 min at 1M cells) instead of replaying the committed figure")
    ap.add_argument("--seed", type=int, default=7)
    ap.add_argument("--no-e2e", action="store_true", help="skip the T_e2e measurement (ingest + egress over PCIe)")
    ap.add_argument("--pmc", action="store_true", help="N = 1: collect the dominant kernel's HBM traffic and MFMA-busy counters with rocprofv3 first (three separate "
                    "--pmc passes, ~1 minute) and refresh profiles/pmc_traffic_update_kernel.json, which the roofline's `traffic` is read from")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1 (nccl == RCCL; gloo only for smoke tests on one GPU)")
    ap.add_argument("--bootstrap", default="torch", choices=["torch", "file"], help="torch: torch.distributed bootstraps the communicator (default); "
                    "file: no torch in the process -- unique id through a file, hmx_comm_init, host reductions through the library")
    a = ap.parse_args()
    apply_size_defaults(a)

    if a.bootstrap == "file":
        return main_file_bootstrap(a)
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher: become one (one process per GPU over RCCL), and make sure the line really is an N-GPU line
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr",
               "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))

    pmc_now = None
    if a.pmc and a.gpus == 1 and a.workload == "c3":      # (before this process touches the GPU: the passes are processes of their own)
        pmc_now = pmc_refresh(a.total_cells or a.cells_per_gpu, a.pcs, a.clusters or 100, a.batches)
        if pmc_now is None:
            print("--pmc: rocprofv3 not found or a counter pass failed; the committed profiles/pmc_traffic_update_kernel.json is replayed", file=sys.stderr)
    import torch
    from harmony_amd import Harmony, prepare_setup_args

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: refusing to print a line whose n_gpus differs from --gpus" % (a.gpus, world))
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

    if a.workload == "c5":
        levels, nested, K = (8, 64, 128), True, (a.clusters or 200)
    else:
        levels, nested, K = (a.batches,), False, (a.clusters or 100)
    strong = a.total_cells > 0
    if strong and a.total_cells % world:
        raise SystemExit("--total-cells must be a multiple of --gpus")
    n, d = (a.total_cells // world if strong else a.cells_per_gpu), a.pcs
    N = n * world
    def build_object(n, N, levels, nested, K):
        """one Harmony object holding this rank's n cells of an N-cell job: communicator (built-in RCCL, else the torch.distributed hook), the peers' inboxes
        with their self-test, global level sizes, setup -- the main workload and, at N > 1, the weak-scaling leg go through the same path"""
        Z, meta, _ = synth(n, d=d, levels=levels, seed=a.seed, shard=rank, nested=nested)
        vars_use = list(meta)
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
                # hmx_comm_init also connects the ranks' inboxes and runs the transport self-test (the path a torch-free C / R host takes);
                # the torch.distributed bootstrap below is only used where that did not switch the peer-to-peer chain on

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
            # Peer-to-peer block chain: the 20 dependent K x B sums of a clustering round happen INSIDE the persistent launch (every
            # GPU writes its table into every peer's inbox over xGMI) instead of 20 launches + 20 all-reduces.  The inbox handles
            # travel through torch.distributed; it is switched on only if the transport self-test passed on EVERY rank.
            p2p_note = obj.p2p_status
            builtin_p2p = bool(ok) and torch.tensor([1 if obj._scalar("p2p") else 0], device=dev)
            if ok:
                dist.all_reduce(builtin_p2p, op=dist.ReduceOp.MIN)
                builtin_p2p = bool(builtin_p2p.item())
            if os.environ.get("HMX_BENCH_P2P", "1") == "0":
                obj.p2p_enable(False)
                p2p_note = "switched off (HMX_BENCH_P2P=0)"
            elif builtin_p2p:
                p2p_note = "bootstrapped by hmx_comm_init: " + obj.p2p_status
            elif world <= 8:
                def agree(flag):
                    t = torch.tensor([1 if flag else 0], device=dev if a.backend == "nccl" else "cpu")
                    dist.all_reduce(t, op=dist.ReduceOp.MIN)
                    return bool(t.item())
                peers_ok = True
                if torch.cuda.device_count() < world:
                    os.environ["HMX_CHAIN_WGS"] = str(max(8, 240 // world))    # smoke tests: the ranks' chains share one GPU
                else:                                     # every GPU of the job must be able to map every other's memory
                    peers_ok = all(g == local_rank or torch.cuda.can_device_access_peer(local_rank, g) for g in range(world))
                handle = None
                try:
                    if peers_ok:
                        handle = obj.p2p_export()
                    else:
                        p2p_note = "no peer access between the GPUs of this job"
                except Exception as e:                    # pragma: no cover
                    handle, p2p_note = None, "export failed: %s" % e
                handles = [None] * world
                dist.all_gather_object(handles, handle)
                good = all(h is not None for h in handles)
                if good:
                    try:
                        obj.p2p_connect(rank, world, handles)
                    except Exception as e:                # pragma: no cover
                        good, p2p_note = False, "connect failed: %s" % e
                good = agree(good)
                if good:
                    dist.barrier()
                    good = agree(obj.p2p_selftest())
                    p2p_note = obj.p2p_status
                if good:
                    obj.p2p_enable(True)
                elif rank == 0:                           # pragma: no cover
                    print("peer-to-peer chain not available (%s): one launch + one all-reduce per block" % p2p_note, file=sys.stderr)
            comm_kind += "; block chain: " + ("peer-to-peer inboxes inside the persistent launch (%s)" % p2p_note
                                              if obj._scalar("p2p") else "one launch + one all-reduce per block (%s)" % p2p_note)
            N_b = []
            for v, L in zip(vars_use, levels):
                cnt = torch.from_numpy(np.bincount(meta[v], minlength=L).astype(np.int64)).to(dev)
                dist.all_reduce(cnt)
                N_b.append(cnt.cpu().numpy().astype(float))
            N_b = np.concatenate(N_b)
        skw, _ = prepare_setup_args(Z, meta, vars_use, nclust=K, N_b=N_b, levels={v: np.arange(L) for v, L in zip(vars_use, levels)})
        t_setup = time.perf_counter()
        obj.setup(**skw)
        t_setup = time.perf_counter() - t_setup
        ingest_ms = obj.timer("ingest_Z")     # H2D of Z (N*d*8 B from pageable host memory) + fp32 conversion, inside setup
        del Z                                 # (skw keeps the d x N matrix the library was given)
        return obj, skw, comm_kind, t_setup, ingest_ms

    obj, skw, comm_kind, t_setup, ingest_ms = build_object(n, N, levels, nested, K)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    if world > 1 and obj._scalar("p2p"):
        # one untimed run with the in-launch exchange before anything is measured: if any rank's chain timed out on its peers
        # (every spin is bounded; the error surfaces at the objective read), EVERY rank goes back to per-block collectives
        try:
            run_to_convergence(obj)
            fine = True
        except Exception as e:                        # pragma: no cover
            fine = False
            print("rank %d: peer-to-peer chain failed (%s)" % (rank, e), file=sys.stderr)
        t = torch.tensor([1 if fine else 0], device=dev if a.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        if not bool(t.item()):                        # pragma: no cover
            obj.p2p_enable(False)
            comm_kind += " -- SWITCHED OFF after a failed trial run: one launch + one all-reduce per block"
    # Pre-roll (untimed, in front of the W warmup steps, same count on every rank): a fresh box's first process measured the same code 7 % slower
    # per chain launch than any later process on that box (profiles/r4_bench_default_first_process.json: 16.05 vs 14.75 us per block step with
    # W = 1) -- clocks and caches of an idle GPU need a few hundred ms of load.  ~0.25 s of the same workload; HMX_BENCH_PREROLL=0 switches it off.
    preroll = int(os.environ.get("HMX_BENCH_PREROLL", "16" if n <= 2000000 else "2"))
    first_run_ms = None
    for i in range(preroll + a.warmup):
        if i == 0:                      # the very first run of this process on this object (cold clocks, cold caches, first launches): reported, never `value`
            sync(); t_first = time.perf_counter()
        run_to_convergence(obj)
        if i == 0:
            sync(); first_run_ms = 1e3 * (time.perf_counter() - t_first)
    # start / stop HIP events attached to every launch of the dominant kernel, on the library's stream (profile level 1; the per-phase event
    # brackets -- level 2, ~200 extra packets per run -- are taken in ONE extra untimed run behind the timed region: HMX_BENCH_NOPROF=1 shows
    # what the events themselves cost)
    obj.set_profile(0 if os.environ.get("HMX_BENCH_NOPROF", "0") == "1" else 1)
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
    prof = {k: obj._scalar("prof:" + k) for k in ("update_ms", "update_launches", "update_cells", "update_steps")}
    kr_timed = np.asarray(obj.kmeans_rounds, dtype=np.int64)
    obj.set_profile(2)
    run_to_convergence(obj)
    sync()
    gpu_phase = {k: round(obj._scalar("gputimer:" + k), 3) for k in
                 ("kmeans_centers", "cluster_head", "randomize", "EO_update", "correct_ridge_loop", "ridge_statistics", "arma_inv", "update_Zcorr")}
    gpu_phase["Rcells_update"] = round(prof["update_ms"] / a.steps, 3)
    gpu_phase["from"] = "one extra untimed run with per-phase event brackets (Rcells_update: the timed runs' own launch events)"
    shard_check = None
    if world > 1:
        # sharded sanity: every rank holds the same global O (integer sums) and it accounts for every cell (once per covariate: sum O = C N)
        O = torch.from_numpy(np.ascontiguousarray(obj.O, dtype=np.float64)).to(dev if a.backend == "nccl" else "cpu")
        lo_, hi_ = O.clone(), O.clone()
        dist.all_reduce(lo_, op=dist.ReduceOp.MIN); dist.all_reduce(hi_, op=dist.ReduceOp.MAX)
        shard_check = {"O_identical_on_all_ranks": bool((lo_ == hi_).all().item()),
                       "sum_O_over_N": float(O.sum().item()) / float(N)}
        if not shard_check["O_identical_on_all_ranks"] or abs(shard_check["sum_O_over_N"] - float(len(levels))) > 1e-4 * len(levels):
            raise SystemExit("sharded run inconsistent: %r" % (shard_check,))
    kr = kr_timed                                           # rounds of every harmony iteration of the LAST timed step
    rounds = int(kr.sum())

    # roofline of the dominant kernel (k_tile<NCT,0>: E-step of one block of cells).  Algorithmic bytes per cell
    # per launch: read the cell's normalised embedding row (4d) + write its R row (4K)  [DESIGN.md]
    upd_ms = prof["update_ms"]
    upd_launches = prof["update_launches"]
    upd_cells = prof["update_cells"]  # cells summed over rounds (every round touches every cell once)
    alg_bytes = upd_cells * (4.0 * d + 4.0 * K)
    achieved = alg_bytes / (upd_ms * 1e-3) / 1e9 if upd_ms > 0 else 0.0
    # ... and what the launches really have to move: rounds whose R rows nobody reads (chain variant 5, Dev::r_store = 0) write no rows --
    # 4d bytes per cell instead of 4d + 4K.  Weighted over the variants of the timed steps; `achieved` stays the nominal figure
    # every round of this project and its verdicts have been priced with.
    cr_, nr__ = float(obj._scalar("chain_rounds") or 0), float(obj._scalar("rounds_without_R") or 0)
    share_noR = min(nr__ / cr_, 1.0) if cr_ > 0 else 0.0
    moved_bytes = upd_cells * (4.0 * d + 4.0 * K * (1.0 - share_noR))
    achieved_moved = moved_bytes / (upd_ms * 1e-3) / 1e9 if upd_ms > 0 else 0.0
    traffic = mfma_util = None  # HBM bytes / MFMA busy per launch from the PMC passes (collected separately, profiles/)
    pm_note = ""
    try:
        pm = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic_update_kernel.json")))
        if pm["workload"] == {"cells_per_gpu": n, "pcs": d, "clusters": K, "batches": levels[0]} and len(levels) == 1:
            traffic = pm["hbm_bytes_per_launch"]
            mfma_util = pm.get("mfma_busy_frac")
            # rounds whose R rows nobody reads run the chain's variant without the stores (Dev::r_store): the launch-weighted mean of the two
            cr, nr_ = float(obj._scalar("chain_rounds") or 0), float(obj._scalar("rounds_without_R") or 0)
            if cr > 0 and "hbm_bytes_per_launch_without_R_stores" in pm:
                w5 = min(nr_ / cr, 1.0)
                traffic = (1 - w5) * pm["hbm_bytes_per_launch"] + w5 * pm["hbm_bytes_per_launch_without_R_stores"]
                mfma_util = (1 - w5) * pm["mfma_busy_frac"] + w5 * pm.get("mfma_busy_frac_without_R_stores", pm["mfma_busy_frac"])
                pm_note = "launch-weighted over the chain's two variants (%.0f%% of the rounds store no R rows); " % (100 * w5)
            pm_note += "collected %s" % pm.get("collected", "in an earlier round")
    except Exception:
        pass
    # SURVEY 8(d) whole-run figure: compulsory bytes per cell per harmony iteration = 4d(4 + I_k) + 4K(3 + 2 I_k)
    run_bytes = float(n) * float(np.sum(4.0 * d * (4 + kr) + 4.0 * K * (3 + 2 * kr)))
    run_gbs = run_bytes / (ms_per_step * 1e-3) / 1e9
    nct = (K + 15) // 16
    on_chain = bool(obj._scalar("chain")) and (world == 1 or bool(obj._scalar("p2p")))
    bf = "true" if obj._scalar("dot_bf") else "false"        # the build of the tile kernels' distance GEMM: split bf16 | fp32 MFMA
    kname = ("k_tile<%d,4|5,2,%s,%s> -- the persistent block chain: ONE launch = one round of update_R (%d block steps, every cell once); variant 5 (no R stores) for rounds whose R rows nobody reads"
             % (nct, "true" if obj._scalar("usig") else "false", bf, int(obj._scalar("n_blocks")))) if on_chain else \
            ("k_tile<%d,0,%d,%s,%s> -- one launch = the block update of one block of update_R" % (nct, int(obj._scalar("upd_wps")), "true" if obj._scalar("usig") else "false", bf))
    roofline = {"kernel": kname, "bound": "hbm", "achieved": achieved_moved, "peak": 8000.0,
                "unit": "GB/s", "frac": achieved_moved / 8000.0, "traffic": traffic, "mfma_busy_frac": mfma_util,
                "achieved_is": "algorithmic bytes the timed launches HAD to move, over the HIP-event time of the launches: 4d per cell and round (read the embedding row) + 4K (write "
                               "the R row) only for the rounds that store their R rows -- %.0f%% of the timed rounds keep them in registers because nobody reads them (DESIGN 4.5)" % (100 * share_noR),
                "share_of_rounds_without_R_stores": share_noR,
                "nominal": {"achieved": achieved, "frac": achieved / 8000.0,
                            "note": "SURVEY 8(d)'s formulation -- every round reads the embedding row and writes the R row, 4d + 4K per cell: the figure rounds 1-5 reported as `frac`"},
                "distance_gemm": "split bf16: 6 x v_mfma_f32_16x16x32_bf16 per 16 x 16 x 32 block on three exact bf16 parts per fp32 operand" if bf == "true" else "v_mfma_f32_16x16x4_f32",
                "traffic_and_mfma_busy_are": (("collected by THIS invocation (--pmc: three separate rocprofv3 --pmc passes over tools/prof_update.py just before the timed run; %s)" % pm_note) if pmc_now
                                              else ("replayed from profiles/pmc_traffic_update_kernel.json (separate rocprofv3 --pmc passes over this kernel, "
                                                    "%s); not collected in this run -- `bench.py --pmc` refreshes it" % pm_note)) if traffic is not None else None,
                "avg_launch_us": 1e3 * upd_ms / max(upd_launches, 1), "launches": int(upd_launches),
                "avg_block_step_us": 1e3 * upd_ms / max(prof["update_steps"], 1),
                "alg_bytes_per_launch": alg_bytes / max(upd_launches, 1),
                "kernel_time_share": upd_ms / (1e3 * dt) if dt > 0 else None,
                "run": {"alg_bytes_per_step_per_gpu": run_bytes, "achieved": run_gbs, "frac": run_gbs / 8000.0,
                        "note": "SURVEY 8(d): N * sum_iters(4d(4+I_k) + 4K(3+2 I_k)) / T_conv, per GPU, vs 8 TB/s"}}
    chain = None
    if obj._scalar("chain"):   # persistent block chain: where its workgroups spent their time (100 MHz ticks -> us per block step)
        dbg = obj._get("chain_dbg")
        steps_ = max(float(dbg[3]) * float(obj._scalar("n_blocks")), 1.0)      # chain launches x block steps per launch
        names = ["folder_wait_arrivals", "folder_fold", "folder_publish", None, "worker_wait_flag", "worker_copy_table",
                 "worker_wait_stores_issued_to_retired", "worker_barrier_arrive", "worker_next_mfma", "worker_tiles_but_last",
                 "worker_last_epilogue", "worker_flush_and_store_issue", "worker_next_mfma_cyclecounter_x100"]
        chain = {nm: round(float(dbg[i]) / 100.0 / steps_, 3) for i, nm in enumerate(names) if nm}
        if len(dbg) >= 48:   # per wave of workgroups 0 and 100: us from "table in LDS" to "own work done" per block step, tiles per step
            for nm, o in (("wg0", 16), ("wg100", 32)):
                chain[nm + "_wave_busy_us"] = [round(float(dbg[o + w]) / 100.0 / steps_, 2) for w in range(8)]
                chain[nm + "_wave_tiles"] = [round(float(dbg[o + 8 + w]) / steps_, 2) for w in range(8)]
        if len(dbg) >= 64:   # the same worker phases for a SIMD's younger wave (wave 4: two tiles, wave 5: one tile of workgroup 0)
            ph = ["wait_flag", "copy_table", "wait_atomics", "barrier_arrive", "next_mfma", "tiles_but_last", "last_epilogue", "flush"]
            for nm, o in (("wg0_wave4", 48), ("wg0_wave5", 56)):
                chain[nm + "_phases_us"] = {p_: round(float(dbg[o + i]) / 100.0 / steps_, 3) for i, p_ in enumerate(ph)}
    # T_e2e: T_conv + H2D of Z (double, the R seam) + D2H of Z_corr (double); PCIe-inclusive, never `value`
    e2e = None
    if not a.no_e2e:
        sync()
        t1 = time.perf_counter()
        zc = obj.getZcorr()
        egress_ms = 1e3 * (time.perf_counter() - t1)
        zc32 = None
        t1 = time.perf_counter()
        zc32 = obj.get_matrix("Z_corr", np.float32)
        egress32_ms = 1e3 * (time.perf_counter() - t1)
        del zc, zc32
        ingest_first_ms = ingest_ms
        if world == 1:
            # the ingest again, into a second object: the first setup of a process also carries the HIP runtime's first-use costs
            # (~20 ms: measured as `ingest_Z_f64_first_call_in_process_ms`), which are not a property of the seam
            o2 = Harmony(device=local_rank, seed=1)
            o2.set_stream(torch.cuda.current_stream().cuda_stream)
            o2.setup(**skw)
            ingest_ms = o2.timer("ingest_Z")
            del o2
        e2e = {"T_conv_ms": ms_per_step, "ingest_Z_f64_ms": ingest_ms, "ingest_Z_f64_first_call_in_process_ms": ingest_first_ms,
               "egress_Zcorr_f64_ms": egress_ms,
               "egress_Zcorr_f32_ms": egress32_ms, "T_e2e_ms": ms_per_step + ingest_ms + egress_ms,
               "cells_per_sec_e2e": n / ((ms_per_step + ingest_ms + egress_ms) * 1e-3), "setup_total_ms": 1e3 * t_setup,
               "note": "per GPU; pageable fp64 matrices on the host side (the R seam), moved through a ring of page-locked slots by 8 host threads while the DMA "
                       "engine works (HMX_XFER=pin: register the caller's matrix instead; HMX_PIN=0: plain pageable copies); the process' one-time "
                       "costs (ring allocation, code-object load) are in setup_total, which also holds Phi -> level codes and the combination sort on the host"}
    out = {
        "metric": "cells_per_sec_to_convergence", "value": N / (ms_per_step * 1e-3), "unit": "cells/s",
        "value_is": "exact-accumulator mode (parity target: the oracle with fp64 accumulators); the reference-arithmetic mode is value_reference_arith",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "preroll_steps_untimed": preroll, "ms_per_step": ms_per_step,
        "first_run_of_this_process_ms": first_run_ms,
        "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "synthetic %d cells x %d PCs, K=%d, levels %s%s%s (BASELINE %s per GPU)"
                               % (N, d, K, "x".join(map(str, levels)), " nested" if nested else "",
                                  "" if world == 1 else ", %d cells/GPU cell-sharded" % n,
                                  "configs[4] shape" if a.workload == "c5" else ("configs[2]" if n == 1000000 and levels == (10,) else "configs[2]/[3] family")),
                   "parallelism": ("cells sharded x%d, all-reduce of O/E/statistics: %s" % (world, comm_kind)) if world > 1 else "single GPU",
                   "comm": comm_stats(obj) if world > 1 else None,
                   "shard_check": shard_check, "harmony_iterations": iters, "kmeans_rounds_last_step": rounds,
                   "s_per_iter": 1e-3 * ms_per_step / max(float(np.mean(iters)), 1.0),
                   "gpu_phase_ms_per_step": gpu_phase, "chain_us_per_block_step": chain, "e2e": e2e},
        "roofline": roofline,
    }
    if world > 1:
        out["config"]["baseline_config"] = ("configs[3]: 10M cells x 50 PCs, K=100, 20 batches, cell-sharded (the default of --gpus N > 1)" if (N, d, K, levels) == (10000000, 50, 100, (20,))
                                            else "configs[4] shape" if a.workload == "c5" else "user-sized")
        out["parity_mode"] = ("default: exact accumulators, identical for every shard count (integer sums); parity target = the reference's own sources built with its "
                              "-DHARMONY_SCALAR_DOUBLE switch (INTEGRATION.md, 'Which result a sharded run returns'); reference arithmetic is a one-GPU mode")
        try:      # DESIGN 5.2's prediction for this very line, made before any run crossed two devices: the hardware run falsifies or confirms it
            pfile = next(f for f in ("r6_scaling_prediction.json", "r5_scaling_prediction.json") if os.path.exists(os.path.join(ROOT, "profiles", f)))
            pj = json.load(open(os.path.join(ROOT, "profiles", pfile)))
            key = "configs3" if (N, d, K, levels) == (10000000, 50, 100, (20,)) else ("configs4" if a.workload == "c5" and N == 5000000 else None)
            if key and str(world) in pj[key]:
                e = pj[key][str(world)]
                out["predicted"] = {"ms_per_step": e["predicted_ms"], "speedup_vs_1_gpu": e["speedup_vs_1"], "one_rank_share_without_exchanges_ms": e["share_ms"],
                                    "source": "profiles/%s (tools/scaling_prediction.py: per-rank shares measured on ONE GPU + measured one-device exchange costs)" % pfile}
        except Exception:
            pass
        if a.default_multi and (a.also or "weak") != "none":
            # the weak-scaling companion of the default N > 1 line: configs[2]'s 1M cells x 10 batches on EVERY GPU
            try:
                del obj
                n2 = 1000000
                o2, _, ck2, _, _ = build_object(n2, n2 * world, (10,), False, 100)
                for _ in range(2):
                    run_to_convergence(o2)
                sync()
                t0 = time.perf_counter()
                its2 = [run_to_convergence(o2) for _ in range(a.steps)]
                sync()
                dt2 = time.perf_counter() - t0
                t = torch.tensor([dt2], dtype=torch.float64, device=dev if a.backend == "nccl" else "cpu")
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dt2 = float(t.item())
                out["also"] = {"weak_scaling_1M_per_gpu": {"workload": "synthetic %d cells x 50 PCs, K=100, 10 batches, %d cells/GPU" % (n2 * world, n2), "scaling": "weak",
                                                           "ms_per_step": 1e3 * dt2 / a.steps, "cells_per_s": n2 * world / (dt2 / a.steps), "harmony_iterations": its2,
                                                           "comm": comm_stats(o2), "parallelism": ck2}}
                del o2
            except Exception as e:     # pragma: no cover  (an extra leg never costs the main line)
                out["also"] = {"weak_scaling_1M_per_gpu": {"error": repr(e)}}
    hp = headline_parity(n, d, K, levels) if world == 1 else None
    if hp:
        out.update({"parity_mode": hp["parity_mode"], "Z_rel_vs_accurate": hp["Z_rel_vs_accurate"], "Z_rel_vs_faithful": hp["Z_rel_vs_faithful"], "parity": hp})
    default_main = world == 1 and a.workload == "c3" and n == 1000000 and levels == (10,) and K == 100 and d == 50
    also = a.also if a.also is not None else ("ref,ref2,10M,share,c5,pbmc" if default_main else "none")
    if world == 1 and also != "none":
        # extra legs of this invocation (never part of `value`): each one is a full run to convergence from HBM-resident inputs
        del obj
        legs = {}
        for leg in also.split(","):
            try:
                if leg == "ref":       # every accumulator group in the reference's fp32 operation order on the main workload
                    legs["reference_arith"] = bench_leg(Harmony, prepare_setup_args, n, d, K, levels, nested, a.seed, 2, 1, sync, ref_arith=1)
                    legs["reference_arith"]["parity"] = "vs the faithful oracle: cpu_baseline.gpu_reference_arith_vs_this_run (live, sample) and profiles/r5_parity_table_*.json (full size: gpu_ref_arith_vs_oracle_faithful)"
                elif leg == "ref2":    # ref_arith = 2 (round 6): every group but the O / E tables -- inside north_star's 1e-4 of the fp32 reference (parity.reference_arith_2_mode), update_R in the block chain
                    legs["reference_arith_2"] = bench_leg(Harmony, prepare_setup_args, n, d, K, levels, nested, a.seed, 2, 1, sync, ref_arith=2)
                    legs["reference_arith_2"]["parity"] = "vs the faithful oracle: profiles/r6_parity_table_*.json (gpu_ref_arith2_vs_oracle_faithful: 1e-4 asserted, same iteration counts)"
                elif leg == "10M":     # north_star's target size on ONE GPU: 10M x 50, K = 100, 20 batches (configs[3]'s total size)
                    legs["10M_one_gpu"] = bench_leg(Harmony, prepare_setup_args, 10000000, 50, 100, (20,), False, a.seed, 2, 1, sync)
                elif leg == "share":   # one GPU's share of configs[3] on an 8-GPU node: 1.25M cells of the 10M, 20 batches
                    legs["configs3_share_1p25M"] = bench_leg(Harmony, prepare_setup_args, 1250000, 50, 100, (20,), False, a.seed, 3, 1, sync)
                    legs["configs3_share_1p25M"]["note"] = "what each rank of `--total-cells 10000000 --batches 20 --gpus 8` computes per step, without the exchanges (no 8-GPU node here)"
                elif leg == "shares":  # (not in the default set) what ONE rank of the 8-GPU configs computes at G = 2 / 4 / 8, without the exchanges: the inputs of DESIGN 5.2's scaling prediction
                    for cells in (5000000, 2500000):
                        legs["configs3_share_%dk" % (cells // 1000)] = bench_leg(Harmony, prepare_setup_args, cells, 50, 100, (20,), False, a.seed, 2, 1, sync)
                    for cells in (2500000, 1250000, 625000, 500000):
                        legs["configs4_share_%dk" % (cells // 1000)] = bench_leg(Harmony, prepare_setup_args, cells, 50, 200, (8, 64, 128), True, a.seed, 2, 1, sync)
                elif leg == "c5":      # configs[4] shape at 1M cells
                    legs["c5_shape_1M"] = bench_leg(Harmony, prepare_setup_args, 1000000, 50, 200, (8, 64, 128), True, a.seed, 2, 1, sync)
                elif leg == "pbmc":    # configs[1] at its stated size, GPU vs CPU oracle
                    legs["pbmc30k"] = pbmc30k_leg(Harmony, prepare_setup_args, sync)
            except Exception as e:     # pragma: no cover  (an extra leg never costs the main line)
                legs[leg] = {"error": repr(e)}
        out["also"] = legs
        ra = legs.get("reference_arith")
        if isinstance(ra, dict) and "cells_per_s" in ra:      # the mode whose arithmetic is the reference's, next to the headline (VERDICT r4 #1c)
            out["value_reference_arith"] = ra["cells_per_s"]
            out["ms_per_step_reference_arith"] = ra["ms_per_step"]
        r2 = legs.get("reference_arith_2")
        if isinstance(r2, dict) and "cells_per_s" in r2:
            out["value_reference_arith_2"] = r2["cells_per_s"]
            out["ms_per_step_reference_arith_2"] = r2["ms_per_step"]
    if rank == 0 and world == 1 and a.cpu_sample > 0:
        out["cpu_baseline"] = cpu_baseline(n if a.cpu_full else a.cpu_sample, d, K, levels, nested, a.seed)
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
