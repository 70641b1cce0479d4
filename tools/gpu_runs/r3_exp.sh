#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3; mkdir -p $O
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29577 tools/dist_two_proc.py --p2p --split 0.9 --chain-max-tpw 0.02 > $O/dbg_unequal.out 2> $O/dbg_unequal.err
echo rc=$?; tail -2 $O/dbg_unequal.out; grep -v "^\[Gloo\]\|^W0\|^E0\|^$\|amdgpu.ids\|socket.cpp\|^\*\*\*\|OMP_NUM" $O/dbg_unequal.err | head -12 | cut -c1-300
timeout 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "single_cluster or stand_alone" 2>&1 | tail -15 | cut -c1-300
for w in 256 252 248 240; do
  HMX_CHAIN_WGS=$w timeout 300 python bench.py --steps 5 --warmup 2 --cpu-sample 0 --no-e2e --also none > $O/bench_wgs$w.json 2> $O/bench_wgs$w.err
  python - $w <<'PY'
import json, sys
j = json.loads(open("gpurun_out/r3/bench_wgs%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
print("chain_wgs", sys.argv[1], "ms_per_step %.3f" % j["ms_per_step"], "block step %.2f" % j["roofline"]["avg_block_step_us"], j["config"]["gpu_phase_ms_per_step"])
PY
done
for pin in 1 0; do
  HMX_PIN=$pin timeout 300 python bench.py --steps 3 --warmup 1 --cpu-sample 0 --also none > $O/bench_pin$pin.json 2> $O/bench_pin$pin.err
  python - $pin <<'PY'
import json, sys
j = json.loads(open("gpurun_out/r3/bench_pin%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
print("HMX_PIN", sys.argv[1], j["config"]["e2e"])
PY
done
