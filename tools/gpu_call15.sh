#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 600 python bench.py --gpus 2 --backend gloo --cells-per-gpu 500000 --steps 3 --warmup 1 --no-e2e > gpurun_out/c15_b2.json 2> gpurun_out/c15_b2.err; echo rc=$?
HMX_BENCH_P2P=0 timeout 600 python bench.py --gpus 2 --backend gloo --cells-per-gpu 500000 --steps 3 --warmup 1 --no-e2e > gpurun_out/c15_b2_nop2p.json 2> gpurun_out/c15_b2_nop2p.err; echo rc=$?
timeout 600 python -m pytest tests/test_gpu_parity2.py -x -q -m gpu -k "two_processes" 2>&1 | tail -5
tail -3 gpurun_out/c15_b2.err
python - <<'PY'
import json
for f in ("c15_b2", "c15_b2_nop2p"):
    try:
        j = json.loads(open("gpurun_out/%s.json" % f).read().strip().splitlines()[-1])
        print(f, j["ms_per_step"], j["config"]["parallelism"], j["config"]["shard_check"], j["roofline"]["avg_block_step_us"])
    except Exception as e:
        print(f, "ERR", e)
PY
