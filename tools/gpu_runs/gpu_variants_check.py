"""configs[4]'s shape (K = 200, 8 > 64 > 128 levels) through two corrections on the GPU with the statistics kernel and the chain switched -- default /
HMX_MOE_STATS=atomic / HMX_CHAIN_PAIR=0 -- and the three variants against each other: Z_corr, O, objective series, subset clusters per correction (python
tools/gpu_runs/gpu_variants_check.py on the GPU box).  The check that cleared the GPU when an oracle run came back different."""
import sys, os, subprocess, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
if len(sys.argv) > 1:
    from harmony_amd import Harmony, prepare_setup_args
    from helpers import synth
    n = int(sys.argv[1])
    Z, meta, _ = synth(n, d=50, levels=(8, 64, 128), seed=11, nested=True)
    skw, _ = prepare_setup_args(Z, meta, list(meta), nclust=200)
    g = Harmony(seed=5); g.setup(**skw)
    Y0 = g.kmeans_centers(); g.init_cluster_cpp(Y0)
    subs = []
    for it in range(2):
        g.cluster_cpp(); g.moe_correct_ridge_cpp(); subs.append(int(g._scalar("subset_clusters")))
    np.savez(sys.argv[2], zc=g.getZcorr(), O=np.array(g.O), obj=np.array(g.objective_kmeans), subs=np.array(subs))
    sys.exit(0)
sys.path.insert(0, "/root/repo/tests")
res = {}
for n in (200000, 1000000):
    for name, env in (("default", {}), ("atomic", {"HMX_MOE_STATS": "atomic"}), ("nopair", {"HMX_CHAIN_PAIR": "0"})):
        f = "/tmp/variant_%s_%d.npz" % (name, n)
        r = subprocess.run([sys.executable, __file__, str(n), f], env=dict(os.environ, PYTHONPATH=os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"), **env), capture_output=True, text=True)
        if r.returncode: print(name, r.stderr[-500:])
        res[(name, n)] = np.load(f)
    a = res[("default", n)]
    for name in ("atomic", "nopair"):
        b = res[(name, n)]
        print(n, name, "Z_rel %.2e" % (np.linalg.norm(a["zc"] - b["zc"]) / np.linalg.norm(b["zc"])), "subs", a["subs"], b["subs"], "O maxdiff %.3g" % np.max(np.abs(a["O"] - b["O"])), "obj rel %.2e" % np.max(np.abs(a["obj"] - b["obj"]) / np.abs(b["obj"])), flush=True)
