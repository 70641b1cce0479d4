#!/bin/bash
# round 3: bench.py with two ranks sharing this box's one GPU (gloo for the remaining collectives): in-launch exchange of the block chain
# (the library's own inbox bootstrap) and one all-reduce per block
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3two; mkdir -p $O
timeout 600 python bench.py --gpus 2 --backend gloo --cells-per-gpu 500000 --steps 3 --warmup 1 --no-e2e > $O/bench_2ranks_p2p.json 2> $O/bench_2ranks_p2p.err; echo rc=$?; tail -3 $O/bench_2ranks_p2p.err; tail -1 $O/bench_2ranks_p2p.json | cut -c1-400
HMX_BENCH_P2P=0 timeout 600 python bench.py --gpus 2 --backend gloo --cells-per-gpu 500000 --steps 3 --warmup 1 --no-e2e > $O/bench_2ranks_allreduce.json 2> $O/bench_2ranks_allreduce.err; echo rc=$?; tail -1 $O/bench_2ranks_allreduce.json | cut -c1-300
python - <<'PY'
import json
for n in ("p2p", "allreduce"):
    try:
        j = json.loads(open("gpurun_out/r3two/bench_2ranks_%s.json" % n).read().strip().splitlines()[-1])
        print(n, j["ms_per_step"], j["value"], j["n_gpus"], j["config"].get("parallelism"), j["config"].get("shard_check"), j["roofline"]["kernel"][:30])
    except Exception as e:
        print(n, "no line", e)
PY
