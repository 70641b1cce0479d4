#!/bin/bash
# kernel table of the reference-arithmetic mode (rocprofv3 --kernel-trace --stats) at BASELINE configs[2]; LEG_CELLS=n, LEG_SET="seq_fused=0" (the round-5 kernels) ...
# -> gpurun_out/ref_arith_kernel_table.txt, ref_arith_kernel_stats.csv
exec </dev/null
R=$GRAFT_REPO_ROOT; cd $R || exit 1
mkdir -p gpurun_out
cat > /tmp/leg2.py <<'PY'
import sys, time, json, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import numpy as np
from harmony_amd import Harmony, prepare_setup_args
from bench_data import synth
from bench import run_to_convergence
n = int(os.environ.get("LEG_CELLS", "1000000"))
Z, meta, _ = synth(n, d=50, levels=(10,), seed=7)
skw, _ = prepare_setup_args(Z, meta, list(meta), nclust=100)
o = Harmony(seed=1, ref_arith=1)
for kv in os.environ.get("LEG_SET", "").split(","):
    if kv:
        k, v = kv.split("="); o._set(k, int(v))
o.setup(**skw)
run_to_convergence(o)
o._scalar("sync"); t0 = time.perf_counter()
its = [run_to_convergence(o) for _ in range(2)]
o._scalar("sync"); ms = 1e3 * (time.perf_counter() - t0) / 2
print(json.dumps({"ms": ms, "its": its, "passes": o._get("seq:group_passes").tolist(), "runs": o._get("seq:group_runs").tolist()}))
PY
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_d
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_d -o p -- python /tmp/leg2.py > /tmp/leg2.out 2>&1
grep '^{' /tmp/leg2.out > $R/gpurun_out/ref_arith_kernel_table.txt
python - <<'PY' >> $R/gpurun_out/ref_arith_kernel_table.txt
import csv, glob
f = (glob.glob("/tmp/prof_d/**/*kernel_stats.csv", recursive=True) + glob.glob("/tmp/prof_d/*kernel_stats.csv"))[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:22]:
    print("   %-58s calls %6s avg %9.1f us (min %.1f max %.1f)  total %8.2f ms  %5.1f%%" % (r["Name"][:58], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, 100 * float(r["TotalDurationNs"]) / tot))
PY
cp $(find /tmp/prof_d -name "*kernel_stats.csv" | head -1) $R/gpurun_out/ref_arith_kernel_stats.csv
cat $R/gpurun_out/ref_arith_kernel_table.txt
