#!/usr/bin/env python
"""Every kernel of the default bench run against the HBM roofline -- not only the dominant one the bench line prices.

Input: profiles/<tag>_kernel_stats.csv (tag: r6 by default) (rocprofv3 --kernel-trace --stats of `python bench.py` on the final code: 1M cells x 50 PCs, K = 100,
10 batches, fp32; 23 runs to convergence in the trace).  For every kernel that touches N-sized arrays: ALGORITHMIC bytes per launch (SURVEY
8(d)'s per-cell figures x the cells one launch processes -- compulsory traffic only: what has to be read or written once, side inputs of a
few bytes per cell counted where they are the kernel's whole job), average launch duration, achieved GB/s, fraction of the 8 TB/s peak,
share of the run.  Output: profiles/<tag>_roofline_all_kernels.json + a markdown table on stdout.  No GPU needed.
"""
import csv
import json
import os
import re
import sys

TAG = sys.argv[1] if len(sys.argv) > 1 else "r6"      # python tools/roofline_all_kernels.py [tag]

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N, d, K, B = 1000000, 50, 100, 10
PEAK = 8000.0   # GB/s, /opt/skills/guides/MI355X_MICROARCH.md

# kernel name pattern -> (bytes per cell, what the figure counts, reference lines the kernel stands for)
PER_CELL = [
    (r"k_tile<7, 5,", 4 * d, "one round of update_R without R stores: read the embedding row (4d); R rows stay in registers", "src/harmony.cpp:293-332"),
    (r"k_tile<7, 4,", 4 * d + 4 * K, "one round of update_R: read the embedding row (4d), write the R row (4K)", "src/harmony.cpp:293-332"),
    (r"k_tile<7, 1,", 4 * d + 4 * d, "cluster_cpp head: read Z_corr (4d), write it normalised (4d); the R rows it computes are not stored inside cluster_cpp "
                                      "(the first round recomputes them: SURVEY's 4K write does not happen)", "src/harmony.cpp:220-227"),
    (r"k_tile<7, 2,", 4 * d, "one Lloyd iteration: read the embedding row (4d)", "src/utils.cpp:53-64"),
    (r"k_tile<7, 3,", 4 * d, "seeding: all K exponential races in one pass over the embedding (4d)", "src/utils.cpp:10-49"),
    (r"k_moe_stats_q<", 4 * d + 4 * K, "ridge statistics: read Z_orig (4d) and R (4K)", "src/harmony.cpp:592-608"),
    (r"k_moe_apply_mfma<", 4 * d + 4 * K + 4 * d, "correction: read Z_orig (4d) and R (4K), write Z_corr (4d)", "src/harmony.cpp:615"),
    (r"k_normalize4<", 4 * d + 4 * d, "setup: cosine-normalise the embedding (read 4d, write 4d)", "src/harmony.cpp:42"),
    (r"k_convert_in<double>", 8 * d + 4 * d, "ingest: fp64 slab in (8d), fp32 out (4d)", "src/harmony.cpp:41"),
    (r"k_shuf_count", 4 * 4, "shuffle of four rounds: one 32-bit position per cell and round (4 x 4 B written as counts / read back)", "src/harmony.cpp:272-300"),
    (r"k_shuf_place", 4 * 4 + 4 * 4, "shuffle of four rounds: cell ids placed in their padded block order (4 x (4 + 4) B)", "src/harmony.cpp:272-300"),
]


def main():
    rows, total_ns = [], 0.0
    with open(os.path.join(ROOT, "profiles", TAG + "_kernel_stats.csv")) as fh:
        stats = list(csv.DictReader(fh))
    total_ns = sum(float(r["TotalDurationNs"]) for r in stats)
    for r in stats:
        name = r["Name"].replace("void ", "").replace("hmx::", "")
        name = re.sub(r"\(.*$", "", name)
        avg_us = float(r["AverageNs"]) / 1e3
        share = float(r["TotalDurationNs"]) / total_ns
        hit = next((p for p in PER_CELL if re.search(p[0], name)), None)
        row = {"kernel": name, "calls": int(r["Calls"]), "avg_us": round(avg_us, 1), "share_of_gpu_time": round(share, 4)}
        if hit:
            cells = N
            if "k_convert_in" in name or "k_normalize4" in name:       # slab kernels: N cells over `calls` / (runs of setup) launches
                cells = None
            if cells:
                by = hit[1] * cells
                row.update(alg_bytes_per_cell=hit[1], alg_bytes_per_launch=by, achieved_GBps=round(by / (avg_us * 1e-6) / 1e9, 1),
                           frac_of_8TBps=round(by / (avg_us * 1e-6) / 1e9 / PEAK, 3), counts=hit[2], reference=hit[3])
            else:
                row.update(alg_bytes_per_cell=hit[1], counts=hit[2] + " (slab launches: not priced per launch)", reference=hit[3])
        else:
            row["counts"] = "K x B / K x d tables, scans, solves: latency-bound small launches (no N-sized traffic)"
        rows.append(row)
    pmc = json.load(open(os.path.join(ROOT, "profiles", TAG + "_pmc_summary.json")))["kernels"]
    for r in rows:                      # measured HBM bytes per launch (rocprofv3 --pmc, separate passes, gfx950 FETCH_SIZE correction: <tag>_pmc_summary.json)
        m = pmc.get("hmx::" + r["kernel"])
        if m:
            r["hbm_bytes_measured"] = m["hbm_total_bytes"]
            r["hbm_GBps_measured"] = round(m["hbm_total_bytes"] / (r["avg_us"] * 1e-6) / 1e9, 1)
            r["hbm_frac_measured"] = round(r["hbm_GBps_measured"] / PEAK, 3)
            if "alg_bytes_per_launch" in r:
                r["traffic_over_algorithmic"] = round(m["hbm_total_bytes"] / r["alg_bytes_per_launch"], 2)
    priced = [r for r in rows if "frac_of_8TBps" in r]
    out = {"workload": "1M cells x 50 PCs, K = 100, 10 batches, fp32 (bench.py default, profiles/%s_kernel_stats.csv)" % TAG, "peak_GBps": PEAK,
           "kernels": rows,
           "time_weighted_frac_of_priced_kernels": round(sum(r["frac_of_8TBps"] * r["share_of_gpu_time"] for r in priced) / sum(r["share_of_gpu_time"] for r in priced), 3),
           "share_of_gpu_time_priced": round(sum(r["share_of_gpu_time"] for r in priced), 3)}
    json.dump(out, open(os.path.join(ROOT, "profiles", TAG + "_roofline_all_kernels.json"), "w"), indent=1)
    print("| kernel | what one launch has to move | B / cell | avg µs | algorithmic GB/s | **of 8 TB/s** | measured HBM MB (PMC) | measured / algorithmic | measured of 8 TB/s | share of GPU time |")
    print("|---|---|---|---|---|---|---|---|---|---|")
    for r in rows:
        if "frac_of_8TBps" in r:
            print("| `%s` | %s | %d | %.0f | %.0f | **%.2f** | %s | %s | %s | %.1f %% |" % (
                r["kernel"], r["counts"], r["alg_bytes_per_cell"], r["avg_us"], r["achieved_GBps"], r["frac_of_8TBps"],
                "%.0f" % (r["hbm_bytes_measured"] / 1e6) if "hbm_bytes_measured" in r else "—", "%.2f" % r["traffic_over_algorithmic"] if "traffic_over_algorithmic" in r else "—",
                "%.2f" % r["hbm_frac_measured"] if "hbm_frac_measured" in r else "—", 100 * r["share_of_gpu_time"]))
    rest = [r for r in rows if "frac_of_8TBps" not in r]
    print("| %d other kernels | tables, scans, solves, slab conversions, memsets | — | %.1f (mean) | — | — | %.1f %% |"
          % (len(rest), sum(r["avg_us"] * r["calls"] for r in rest) / sum(r["calls"] for r in rest), 100 * sum(r["share_of_gpu_time"] for r in rest)))
    print("time-weighted fraction over the priced kernels: %.3f (they are %.1f %% of the GPU time)" % (out["time_weighted_frac_of_priced_kernels"], 100 * out["share_of_gpu_time_priced"]))


if __name__ == "__main__":
    main()
