#!/bin/bash
# round 3: split-bf16 build of the tile kernels -- its own tests, the parity subsets that exercise every tile mode, then the bench A/B
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3bf; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_dots.py -x -q -s 2>&1 | tail -30 | tee $O/dots.log
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity2.py -x -q -k "not 1M and not 10M and not config5 and not two_processes and not gap_table and not full_size" 2>&1 | tail -8 | tee $O/parity.log
for dot in bf16 f32; do
  HMX_DOT=$dot timeout 300 python bench.py --steps 10 --warmup 2 --cpu-sample 0 --no-e2e --also none > $O/bench_$dot.json 2> $O/bench_$dot.err; echo "bench $dot rc=$?"; tail -2 $O/bench_$dot.err
done
python - <<'PY'
import json
for dot in ("bf16", "f32"):
    try:
        j = json.loads(open("gpurun_out/r3bf/bench_%s.json" % dot).read().strip().splitlines()[-1])
    except Exception as e:
        print(dot, "no line", e); continue
    print(dot, "ms_per_step", j["ms_per_step"], "iters", j["config"]["harmony_iterations"], "step_us", j["roofline"].get("avg_block_step_us"), "frac", j["roofline"]["frac"])
    print("   phases", j["config"]["gpu_phase_ms_per_step"])
    print("   chain", j["config"].get("chain_us_per_block_step"))
PY
