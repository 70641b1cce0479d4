"""The wave-pair block chain (k_tile MODE 6, K = 200) against the launch-per-step path on the same inputs (python tools/gpu_runs/pair_chain_check.py on the GPU box;
PC_CELLS="200000,1000000", PC_K=200): ms per run, phases, the chain's per-step clocks, and how far the two paths' results are from each other (the row sums of
the two paths add the same terms in a different order: R differs in the last bit, nothing is bit-identical)."""
import sys, time, json, os, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
if len(sys.argv) > 1:      # child: one setting, one size
    from harmony_amd import Harmony, prepare_setup_args
    from bench_data import synth
    from bench import run_to_convergence
    n = int(sys.argv[1]); K = int(os.environ.get("PC_K", "200"))
    Z, meta, _ = synth(n, d=50, levels=(8, 64, 128), seed=7, nested=True)
    skw, _ = prepare_setup_args(Z, meta, list(meta), nclust=K)
    o = Harmony(seed=1)
    o.setup(**skw)
    it0 = run_to_convergence(o)
    o._scalar("sync"); t0 = time.perf_counter()
    its = [run_to_convergence(o) for _ in range(2)]
    o._scalar("sync"); ms = 1e3 * (time.perf_counter() - t0) / 2
    zc = o.getZcorr(); obj = np.asarray(o.objective_kmeans); R = o.getR() if n <= 300000 else None
    o.set_profile(2); run_to_convergence(o); o._scalar("sync")
    ph = {k: round(o._scalar("gputimer:" + k), 3) for k in ("kmeans_centers", "cluster_head", "randomize", "EO_update", "Rcells_update", "ridge_statistics", "arma_inv", "update_Zcorr")}
    dbg = o._get("chain_dbg")
    steps_ = max(float(dbg[3]) * float(o._scalar("n_blocks")), 1.0)
    names = ["folder_wait_arrivals", "folder_fold", "folder_publish", None, "worker_wait_flag", "worker_copy_table", "worker_wait_atomics", "worker_barrier_arrive", "worker_next_mfma", "worker_tiles_but_last", "worker_last_epilogue", "worker_flush"]
    chain = {nm: round(float(dbg[i]) / 100.0 / steps_, 3) for i, nm in enumerate(names) if nm}
    chain["wg0_wave_busy_us"] = [round(float(dbg[16 + w]) / 100.0 / steps_, 2) for w in range(8)]
    chain["wg0_wave_tiles"] = [round(float(dbg[24 + w]) / steps_, 2) for w in range(8)]
    chain["wg100_wave_busy_us"] = [round(float(dbg[32 + w]) / 100.0 / steps_, 2) for w in range(8)]
    chain["wg100_wave_tiles"] = [round(float(dbg[40 + w]) / steps_, 2) for w in range(8)]
    print(json.dumps({"cells": n, "pair": o._scalar("chain_pair"), "chain": o._scalar("chain"), "ms": ms, "its": [it0] + its, "rounds": [int(x) for x in o.kmeans_rounds], "phases": ph, "chain_us": chain if o._scalar("chain") else None}), flush=True)
    np.savez(sys.argv[2], zc=zc, obj=obj, **({"R": R} if R is not None else {}))
    sys.exit(0)
for n in [int(x) for x in os.environ.get("PC_CELLS", "200000,1000000").split(",")]:
    outs = []
    for pair in os.environ.get("PC_PAIR", "0,1").split(","):
        f = "/tmp/pc_%s_%d.npz" % (pair, n)
        env = dict(os.environ, HMX_CHAIN_PAIR=pair)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), str(n), f], env=env, capture_output=True, text=True, timeout=900)
        print(r.stdout.strip()[-3000:], flush=True)
        if r.returncode != 0: print("FAILED rc=%d\n%s" % (r.returncode, r.stderr[-3000:]), flush=True)
        outs.append(np.load(f) if os.path.exists(f) else None)
    if len(outs) == 2 and outs[0] is not None and outs[1] is not None:
        a, b = outs
        m = min(len(a["obj"]), len(b["obj"]))
        msg = "n=%d pair vs launch-per-step: Z_corr rel %.3e, objective rel max %.3e (%d vs %d values)" % (n, np.linalg.norm(a["zc"] - b["zc"]) / np.linalg.norm(a["zc"]), np.max(np.abs(a["obj"][:m] - b["obj"][:m]) / np.abs(a["obj"][:m])), len(a["obj"]), len(b["obj"]))
        if "R" in a and "R" in b: msg += ", max|dR| %.3e, argmax differs in %d cells" % (np.max(np.abs(a["R"] - b["R"])), int(np.sum(np.argmax(a["R"], 0) != np.argmax(b["R"], 0))))
        print(msg, flush=True)
