#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3x; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity2.py -x -q -k "ingest or f32_and_device or shard or fixture or cell_lines or tiny" 2>&1 | tail -3
for v in "ring:HMX_X=1" "pin:HMX_XFER=pin"; do
  name=${v%%:*}; kv=${v#*:}
  env $kv timeout 300 python bench.py --steps 3 --warmup 1 --cpu-sample 0 --also none > $O/bench2_$name.json 2> $O/bench2_$name.err; echo "$name rc=$?"
  python - <<PY
import json
j = json.loads(open("gpurun_out/r3x/bench2_$name.json").read().strip().splitlines()[-1])
e = j["config"]["e2e"]
print("$name", {k: round(v, 2) for k, v in e.items() if isinstance(v, float)})
PY
done
