"""gpurun_out/<tag>fin (tools/final_profiles.sh) -> profiles/<tag>_* (python tools/pmc_summary.py [tag], default r6): PMC summary per kernel (HBM bytes with the gfx950 FETCH_SIZE
correction, MFMA busy fraction), kernel stats, bench lines, and the per-launch traffic file bench.py reads."""
import ast, csv, json, os, re, shutil, sys
TAG = sys.argv[1] if len(sys.argv) > 1 else "r6"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = os.path.join(ROOT, "gpurun_out", TAG + "fin"); P = os.path.join(ROOT, "profiles")
def parse(fn):
    out = {}
    for line in open(os.path.join(R, fn)):
        m = re.match(r"^(.*?) (\{.*\}) dispatches: (\d+)", line.strip())
        if m: out[m.group(1).strip()] = (ast.literal_eval(m.group(2)), int(m.group(3)))
    return out
fe, wr, sq = parse("pmc_fetch.txt"), parse("pmc_write.txt"), parse("pmc_sq.txt"); sq2 = {}
stats = {r["Name"].split("(")[0].replace("void ", "").strip(): float(r["AverageNs"]) / 1e3 for r in csv.DictReader(open(os.path.join(R, "kernel_stats.csv")))}
summ = {"source": "rocprofv3 --pmc <counters> --kernel-trace -- python tools/prof_update.py 1000000 (separate passes per counter group, tools/final_profiles.sh); durations from rocprofv3 --kernel-trace --stats of bench.py (profiles/" + TAG + "_kernel_stats.csv)",
        "notes": ["FETCH_SIZE / WRITE_SIZE are KB per dispatch (mean)",
                  "gfx950: FETCH_SIZE reports 1/2 of the bytes of coalesced reads (MI355X_MICROARCH.md, HBM section): x2, calibrated in round 3 on k_copy (reads 208.0 MB, FETCH_SIZE 101.6 MB: profiles/r3_pmc_summary.json)",
                  "SQ_VALU_MFMA_BUSY_CYCLES counts cycles (32 per v_mfma_f32_16x16x4_f32, ~16 per v_mfma_f32_16x16x32_bf16); mfma_busy_frac = MFMA_BUSY_CYCLES / (duration x 2.4 GHz x 1024 SIMDs)",
                  "k_tile<7,4,2,true,true> / k_tile<7,5,2,true,true> are the persistent block chain (split-bf16 build) with / without the R stores (Dev::r_store): one dispatch = one clustering round = 20 block steps"], "kernels": {}}
for k in sorted(set(fe) | set(wr) | set(sq)):
    name = k.replace("void ", "").strip(); e = {}
    if k in fe: e["FETCH_SIZE_KB"] = fe[k][0]["FETCH_SIZE"]; e["hbm_read_bytes_corrected"] = 2 * 1000 * fe[k][0]["FETCH_SIZE"]
    if k in wr: e["WRITE_SIZE_KB"] = wr[k][0]["WRITE_SIZE"]; e["hbm_write_bytes"] = 1000 * wr[k][0]["WRITE_SIZE"]
    if "hbm_read_bytes_corrected" in e and "hbm_write_bytes" in e: e["hbm_total_bytes"] = e["hbm_read_bytes_corrected"] + e["hbm_write_bytes"]
    if k in sq: e.update(sq[k][0])
    if k in sq2: e.update(sq2[k][0])
    dur = stats.get(name)
    if dur:
        e["avg_duration_us"] = dur
        if "hbm_total_bytes" in e: e["hbm_GBps"] = e["hbm_total_bytes"] / dur / 1e3
        if e.get("SQ_VALU_MFMA_BUSY_CYCLES"): e["mfma_busy_frac"] = e["SQ_VALU_MFMA_BUSY_CYCLES"] / (dur * 1e-6 * 2.4e9 * 1024)
    summ["kernels"][name] = e
json.dump(summ, open(os.path.join(P, TAG + "_pmc_summary.json"), "w"), indent=1)
c4 = summ["kernels"]["hmx::k_tile<7, 4, 2, true, true>"]; c5 = summ["kernels"]["hmx::k_tile<7, 5, 2, true, true>"]
json.dump({"workload": {"cells_per_gpu": 1000000, "pcs": 50, "clusters": 100, "batches": 10},
           "kernel": "k_tile<7,4|5,2,true,true> (persistent block chain, split-bf16 build, one launch = 20 block steps; 5: a round whose R rows nobody reads, Dev::r_store = 0)",
           "hbm_bytes_per_launch": c4["hbm_total_bytes"], "hbm_bytes_per_launch_without_R_stores": c5["hbm_total_bytes"],
           "mfma_busy_frac": c4["mfma_busy_frac"], "mfma_busy_frac_without_R_stores": c5["mfma_busy_frac"], "collected": "round " + TAG[1:] + ", on the final code (tools/final_profiles.sh)",
           "source": "profiles/" + TAG + "_pmc_summary.json (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE; SQ_VALU_MFMA_BUSY_CYCLES / (duration x 2.4 GHz x 1024 SIMDs); separate --pmc passes)"},
          open(os.path.join(P, "pmc_traffic_update_kernel.json"), "w"), indent=1)
for src, dst in [("kernel_stats.csv", TAG + "_kernel_stats.csv"), ("bench_rocprof.json", TAG + "_bench_rocprof.json"), ("bench_default.json", TAG + "_bench_default.json"),
                 ("bench_10M.json", TAG + "_bench_10M_one_gpu.json"), ("bench_c5_1M.json", TAG + "_bench_c5_1M.json"), ("bench_c5_5M.json", TAG + "_bench_c5_5M.json"),
                 ("bench_2ranks_p2p.json", TAG + "_bench_2ranks_one_gpu_p2p_chain.json"), ("bench_2ranks_allreduce.json", TAG + "_bench_2ranks_one_gpu_allreduce_per_block.json"),
                 ("ref_kernel_stats.csv", TAG + "_ref_arith_kernel_stats.csv"), ("ref_profile.json", TAG + "_ref_arith_profile.json"), ("ref_profile_c5.json", TAG + "_ref_arith_profile_c5_shape.json"), ("round_timeline.txt", TAG + "_round_timeline.txt"),
                 ("bench_strong_1gpu_10M.json", TAG + "_bench_total_cells_10M_one_gpu.json"), ("bench_shares.json", TAG + "_bench_shares.json"),
                 ("bench_pmc_selfcollected.json", TAG + "_bench_pmc_selfcollected.json"), ("bench_2ranks_c5.json", TAG + "_bench_2ranks_one_gpu_configs4_shape.json"),
                 ("../" + TAG + "_parity_c5_5M.json", TAG + "_parity_c5_5M.json"), ("../" + TAG + "_parity_c4_10M.json", TAG + "_parity_c4_10M.json")]:
    if os.path.exists(os.path.join(R, src)): shutil.copy(os.path.join(R, src), os.path.join(P, dst))
for k, v in summ["kernels"].items():
    print(k, {x: (round(y, 3) if isinstance(y, float) else y) for x, y in v.items() if x in ("avg_duration_us", "hbm_GBps", "mfma_busy_frac", "hbm_total_bytes")})
