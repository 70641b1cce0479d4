#!/usr/bin/env python
"""a falsifiable prediction of the multi-GPU runs this project could never measure (one GPU per box).  Inputs, all measured on ONE
MI355X and committed under profiles/ (tools/final_profiles_r5.sh):
  <tag>_bench_shares.json        time to convergence of ONE rank's share of configs[3] (10M cells / G, 20 batches) and configs[4] (5M / G, K = 200,
                              200 levels) for G = 2, 4, 8, without any exchange (`bench.py --also shares`; G = 1: r5_bench_default.json's also.10M_one_gpu
                              and r5_bench_c5_5M.json)
  <tag>_bench_2ranks_one_gpu_*   two ranks SHARING this GPU with the in-launch exchange / inbox collectives: exchange + collective cost per step on one device
  <tag>_bench_default.json       p2p self-test figure (us per exchange step) and collective counts (config.comm of the 2-rank lines)
Model per rank:  T(G) = T_share(N / G) + n_exchange * t_x + n_small * t_small + n_big * t_big(G)
  configs[3] (K = 100: persistent chain, the per-block sum INSIDE the launch): n_exchange = rounds * (nb + 2) folder exchanges; t_x = the measured one-device
      figure (lower bound) and 5 us (assumed xGMI round trip of a write-through granule + poll); n_small inbox all-reduces at ~12 us (one launch, one trip)
  configs[4] (K = 200: one launch per block step, the K x B table of every step an inbox all-reduce of 40 000 entries): t_step_ar measured on the shared GPU
      as (T_2ranks - 2 T_share(500k)) / block steps; big buffers (ridge statistics 1.3M entries per correction, old sums 800k per round) as reduce-scatter +
      all-gather windows: bytes per link = 2 (G - 1) / G * 16 B * entries / G ... / 50 GB/s effective per direction and link (assumed: a third of the 153 GB/s peak
      for 16-byte granule traffic)
Writes profiles/<tag>_scaling_prediction.json and prints the markdown table."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")
TAG = sys.argv[1] if len(sys.argv) > 1 else "r6"      # python tools/scaling_prediction.py [tag]


def line(fn):
    try:
        return json.loads(open(os.path.join(P, fn)).read().strip().splitlines()[-1])
    except Exception:
        return None


sh, dflt, c5full = line(TAG + "_bench_shares.json"), line(TAG + "_bench_default.json"), line(TAG + "_bench_c5_5M.json")
two_c3, two_c5 = line(TAG + "_bench_2ranks_one_gpu_p2p_chain.json"), line(TAG + "_bench_2ranks_one_gpu_configs4_shape.json")
if not sh or not dflt:
    sys.exit("profiles/%s_bench_shares.json / %s_bench_default.json missing: run tools/final_profiles.sh + tools/pmc_summary.py first" % (TAG, TAG))
legs = sh["also"]
out = {"assumptions": {"t_x_us": [None, 5.0], "t_small_us": 12.0, "link_GBps_effective": 50.0}, "configs3": {}, "configs4": {}}
tx_meas = None
if two_c3 and two_c3["config"].get("comm"):
    tx_meas = two_c3["config"]["comm"].get("p2p_selftest_us_per_exchange")
tx_meas = float(tx_meas) if tx_meas else 2.65
out["assumptions"]["t_x_us"][0] = tx_meas
# ---- configs[3]
T1 = dflt["also"]["10M_one_gpu"]
rows3 = {1: (T1["ms_per_step"], T1["harmony_iterations"][0])}
for G, key in ((2, "configs3_share_5000k"), (4, "configs3_share_2500k")):
    rows3[G] = (legs[key]["ms_per_step"], legs[key]["harmony_iterations"][0])
s8 = dflt["also"]["configs3_share_1p25M"]
rows3[8] = (s8["ms_per_step"], s8["harmony_iterations"][0])
nb = 20
for G, (ms, it) in rows3.items():
    rounds = it * 4
    n_x = rounds * (nb + 2) if G > 1 else 0
    n_small = (it * 6 + 24) if G > 1 else 0          # per iteration: O after the head, ridge statistics, ...; init: Lloyd sums x 10, seeding, protocol
    lo = ms + 1e-3 * (n_x * tx_meas + n_small * 12.0)
    hi = ms + 1e-3 * (n_x * 5.0 + n_small * 12.0)
    out["configs3"][G] = {"share_ms": ms, "iterations": it, "exchanges": n_x, "small_allreduces": n_small, "predicted_ms": [lo, hi],
                          "speedup_vs_1": [rows3[1][0] / hi, rows3[1][0] / lo]}
# ---- configs[4]
rows4 = {}
if c5full:
    rows4[1] = (c5full["ms_per_step"], c5full["config"]["harmony_iterations"][0])
for G, key in ((2, "configs4_share_2500k"), (4, "configs4_share_1250k"), (8, "configs4_share_625k")):
    rows4[G] = (legs[key]["ms_per_step"], legs[key]["harmony_iterations"][0])
t_step_ar = None
if two_c5 and "configs4_share_500k" in legs:
    it2 = two_c5["config"]["harmony_iterations"][0]
    steps = it2 * 4 * nb
    t_step_ar = max(0.0, 1e3 * (two_c5["ms_per_step"] - 2.0 * legs["configs4_share_500k"]["ms_per_step"]) / steps)
out["assumptions"]["configs4_us_per_block_step_allreduce_shared_gpu"] = t_step_ar
for G, (ms, it) in rows4.items():
    rounds = it * 4
    steps = rounds * nb
    key = {2: "configs4_share_2500k", 4: "configs4_share_1250k", 8: "configs4_share_625k"}.get(G)
    chain = bool(key and legs[key].get("block_chain"))          # round 6: the share runs its rounds in the wave-pair chain (k_tile MODE 6): block steps exchanged in the launch
    if chain and G > 1:
        # in-launch exchange per block step (the configs[3] model: nb + 1 exchanges per round of t_x each), no old-sums collective (the folders exchange
        # new(j - 1) - old_local(j)); the ridge statistics stay reduce-scatter + all-gather windows
        n_x = rounds * (nb + 1)
        big_entries = it * 1.3e6
        t_big = 1e3 * (2.0 * (G - 1) / G * 16.0 * big_entries / G) / (50.0e9)
        lo, hi = ms + 1e-3 * n_x * tx_meas + t_big, ms + 1e-3 * n_x * 5.0 + t_big
        out["configs4"][G] = {"share_ms": ms, "iterations": it, "block_chain": True, "in_launch_exchanges": n_x, "block_step_allreduces": 0, "us_per_step_allreduce": 0.0,
                              "big_window_ms": t_big, "predicted_ms": hi, "predicted_ms_range": [lo, hi], "speedup_vs_1": (rows4[1][0] / hi) if 1 in rows4 else None}
        continue
    big_entries = it * 1.3e6 + rounds * 0.8e6                     # ridge statistics per correction + old sums per round
    t_big = 0.0 if G == 1 else 1e3 * (2.0 * (G - 1) / G * 16.0 * big_entries / G) / (50.0e9)      # ms: bytes over one link / effective rate
    # (round 6: the two-rank configs[4] run itself is on the wave-pair chain now and no longer measures a per-step collective: shares above the chain's
    #  envelope are priced with round 5's measurement, 39.5 us per host-launched inbox all-reduce of the 40 000-entry table)
    t_ar = (t_step_ar if (t_step_ar is not None and t_step_ar >= 5.0) else 39.5)
    pred = ms + (1e-3 * steps * t_ar + t_big if G > 1 else 0.0)
    out["configs4"][G] = {"share_ms": ms, "iterations": it, "block_chain": False, "block_step_allreduces": steps if G > 1 else 0, "us_per_step_allreduce": t_ar if G > 1 else 0,
                          "big_window_ms": t_big, "predicted_ms": pred, "speedup_vs_1": (rows4[1][0] / pred) if 1 in rows4 else None}
json.dump(out, open(os.path.join(P, TAG + "_scaling_prediction.json"), "w"), indent=1)
print("| config | G | cells per rank | one rank's share, no exchange (measured) | exchanges / collectives per run | predicted ms per step | predicted speed-up vs 1 GPU |")
print("|---|---|---|---|---|---|---|")
for G in sorted(out["configs3"]):
    r = out["configs3"][G]
    print("| configs[3] 10M x 50, K=100, 20 batches | %d | %.2fM | %.1f ms (%d it.) | %d in-launch exchanges + %d inbox all-reduces | %.1f - %.1f | %.2f - %.2f |"
          % (G, 10.0 / G, r["share_ms"], r["iterations"], r["exchanges"], r["small_allreduces"], r["predicted_ms"][0], r["predicted_ms"][1], r["speedup_vs_1"][0], r["speedup_vs_1"][1]))
for G in sorted(out["configs4"]):
    r = out["configs4"][G]
    how = ("%d in-launch exchanges (wave-pair chain)" % r["in_launch_exchanges"]) if r.get("block_chain") else ("%d per-step inbox all-reduces (%.0f us each)" % (r["block_step_allreduces"], r["us_per_step_allreduce"]))
    print("| configs[4] 5M x 50, K=200, 200 levels | %d | %.3fM | %.1f ms (%d it.) | %s + %.2f ms of reduce-scatter / all-gather windows | %.1f | %s |"
          % (G, 5.0 / G, r["share_ms"], r["iterations"], how, r["big_window_ms"], r["predicted_ms"], ("%.2f" % r["speedup_vs_1"]) if r["speedup_vs_1"] else "-"))
