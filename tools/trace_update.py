"""Per-wave phase timeline of the block-update kernel k_tile<.,0> (HMX_TRACE=1 diagnostics).

Stamps per wave: 0 wall(10 ns) at entry | 1 clk entry | 2 clk after LDS staging+barrier | 3 clk after first tile's MFMAs
| 4 clk after the overlapped loop | 5 clk after the last epilogue | 6 clk after flush + objective slot | 7 wall at exit.
"""
import os, sys
os.environ["HMX_TRACE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# diagnostics variant of the library (python -m harmony_amd.build --trace); never used by the product path
os.environ.setdefault("HMX_LIB_PATH", os.path.join(ROOT, "harmony_amd", "lib", "libharmony_mi355x_trace.so"))
sys.path.insert(0, ROOT)
import numpy as np
from bench_data import synth
from harmony_amd import Harmony, prepare_setup_args
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 100
B = int(sys.argv[3]) if len(sys.argv) > 3 else 10
Z, meta, _ = synth(N, d=50, levels=(B,), seed=7)
skw, _ = prepare_setup_args(Z, meta, "cov0", nclust=K)
g = Harmony(seed=3)
g.setup(**skw)
g.init_cluster_cpp()
def report():
    t = np.asarray(g._get("trace")).reshape(-1, 16)
    t = t[t[:, 0] > 0]
    w0, w1 = t[:, 0], t[:, 7]
    clk = t[:, 1:7]
    print(f"  {len(t)} active waves; kernel span {(w1.max() - w0.min()) * 0.01:.2f} us; entry skew {(w0.max() - w0.min()) * 0.01:.2f} us; "
          f"per-wave wall median {np.median(w1 - w0) * 0.01:.2f} us, max {(w1 - w0).max() * 0.01:.2f} us")
    names = ["staging+fold+barrier", "first tile MFMAs (incl. row wait)", "overlapped loop", "last epilogue", "flush+objective"]
    d = np.diff(clk, axis=1)
    tot = clk[:, 5] - clk[:, 0]
    for i, nme in enumerate(names):
        print(f"    {nme:36s} median {np.median(d[:, i]):9.0f} clk   p90 {np.percentile(d[:, i], 90):9.0f}   max {d[:, i].max():9.0f}")
    it = t[t[:, 8] > 0]
    if len(it):
        e = np.diff(it[:, 8:12], axis=1)
        for i, nme in enumerate(["last iter: ids/rows issue", "last iter: MFMAs (waits for rows)", "last iter: epilogue (+stores issue)"]):
            print(f"    {nme:36s} median {np.median(e[:, i]):9.0f} clk   p90 {np.percentile(e[:, i], 90):9.0f}   max {e[:, i].max():9.0f}")
    if t[:, 12].max() > 0:
        sg = np.stack([t[:, 12] - t[:, 1], t[:, 13] - t[:, 12], t[:, 14] - t[:, 13], t[:, 15] - t[:, 14], t[:, 2] - t[:, 15]], axis=1)
        for i, nme in enumerate(["stg: entry -> image in LDS", "stg: wait pairs, issue rows", "stg: fold consumed (O in LDS)", "stg: barrier 1", "stg: penalty + barrier 2"]):
            print(f"    {nme:36s} median {np.median(sg[:, i]):9.0f} clk   p90 {np.percentile(sg[:, i], 90):9.0f}   max {sg[:, i].max():9.0f}")
    print(f"    {'total':36s} median {np.median(tot):9.0f} clk   p90 {np.percentile(tot, 90):9.0f}   max {tot.max():9.0f}")


g.cluster_cpp()
print(f"N={N} K={K} B={B}")
for dbg in (0,):
    g._set("upd_debug", dbg)
    g._set("profile", 1)
    g.cluster_cpp()
    ms, nl = g._scalar("prof:update_ms"), g._scalar("prof:update_launches")
    print(f"debug={dbg} (1: no epilogue, 2: no MFMA, 4: no R stores): update kernel {1e3 * ms / max(nl, 1):.2f} us/launch over {int(nl)} launches")
    g._set("profile", 0)
    report()
