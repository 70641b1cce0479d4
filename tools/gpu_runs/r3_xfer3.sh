#!/bin/bash
# round 3: first transfer of a process against the steady state -- transfer workers bound to the caller's NUMA node (default), unbound, none
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3x; mkdir -p $O
for n in /sys/devices/system/node/node*; do echo "$(basename $n): $(cat $n/cpulist)"; done
cat /proc/sys/kernel/numa_balancing 2>/dev/null
for v in "bound:HMX_X=1" "unbound:HMX_XFER_AFFINITY=0" "nothreads:HMX_XFER_THREADS=0"; do
  name=${v%%:*}; kv=${v#*:}
  env $kv timeout 300 python bench.py --steps 2 --warmup 1 --cpu-sample 0 --also none > $O/bench3_$name.json 2> $O/bench3_$name.err; echo "$name rc=$?"
  python - <<PY
import json
j = json.loads(open("gpurun_out/r3x/bench3_$name.json").read().strip().splitlines()[-1])
e = j["config"]["e2e"]
print("$name", {k: round(v, 2) for k, v in e.items() if isinstance(v, float)})
PY
done
