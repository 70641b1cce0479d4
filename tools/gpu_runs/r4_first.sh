#!/bin/bash
# round 4, first look: reference-arithmetic rewrite (packed ridge pass, MFMA objective terms, fused O/E fold, several covariates),
# 128-byte rows A/B, the new tests
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4a; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_seq.py -x -q -m gpu 2>&1 | tail -15
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "stand_alone or pbmc or cell_lines_small_full or two_covariates or synthetic_shapes or 100k" 2>&1 | tail -5
for pad in 1 0; do
HMX_ZS_PAD=$pad timeout 300 python bench.py --steps 10 --warmup 2 --cpu-sample 0 --no-e2e --also none > $O/bench_pad$pad.json 2> $O/bench_pad$pad.err; echo "bench pad=$pad rc=$?"
python - <<PY
import json
j = json.loads(open("$O/bench_pad$pad.json").read().strip().splitlines()[-1])
print("pad=$pad ms_per_step", j["ms_per_step"], "iters", j["config"]["harmony_iterations"], "step_us", j["roofline"].get("avg_block_step_us"), "frac", j["roofline"]["frac"])
print("   phases", j["config"]["gpu_phase_ms_per_step"])
print("   chain", j["config"].get("chain_us_per_block_step"))
PY
done
timeout 600 python bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-e2e --also ref,pbmc > $O/bench_ref.json 2> $O/bench_ref.err; echo "bench ref rc=$?"
python - <<PY
import json
j = json.loads(open("$O/bench_ref.json").read().strip().splitlines()[-1])
for k, v in j["also"].items():
    print(k, {kk: vv for kk, vv in v.items() if kk in ("ms_per_step", "harmony_iterations", "gpu_phase_ms_per_step", "seq_residual", "error", "cpu_oracle", "parity_this_run", "gpu_vs_cpu_1_thread", "roofline")})
PY
tail -3 $O/*.err
