#!/bin/bash
# round 5, last GPU seconds: the checker libraries (oracle, oracle/_ref) were rebuilt after the LAPACK / BLAS liberties went in -- the GPU tests that
# lean on them hardest, the smoke test and one bench line with the new cpu_baseline.reference_sources leg (product library unchanged: md5 in the log)
exec </dev/null
R=$GRAFT_REPO_ROOT; cd $R || exit 1
mkdir -p gpurun_out/r5z
md5sum harmony_amd/lib/libharmony_mi355x.so oracle/libharmony_oracle.so oracle/_ref/*.so > gpurun_out/r5z/md5.txt
(timeout 40 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2) > gpurun_out/r5z/smoke.log &
timeout 60 python -m pytest tests/test_gpu_seq.py -q -m gpu -x 2>&1 | tail -3 > gpurun_out/r5z/seq_tests.log
wait
timeout 70 python bench.py --also none --steps 3 --warmup 1 > gpurun_out/r5z/bench.json 2> gpurun_out/r5z/bench.err; echo rc=$?
cat gpurun_out/r5z/smoke.log gpurun_out/r5z/seq_tests.log
python - <<'PY'
import json
j = json.loads(open("gpurun_out/r5z/bench.json").read().strip().splitlines()[-1])
print(j["ms_per_step"], j["cpu_baseline"]["value"], j["cpu_baseline"].get("reference_sources"), len(open("gpurun_out/r5z/bench.json").read().strip().splitlines()))
PY
timeout 60 python tools/openblas_point_probe.py > gpurun_out/r5z/probe.log 2>&1; echo probe rc=$?; tail -c 1500 gpurun_out/r5z/probe.log
