#!/bin/bash
# round 6: kernel table of the reference-arithmetic mode at configs[4]'s shape (K = 200, 200 levels in three nested covariates), 1M cells
exec </dev/null
R=$GRAFT_REPO_ROOT; cd $R || exit 1
mkdir -p gpurun_out
cat > /tmp/leg3.py <<'PY'
import sys, time, json, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import numpy as np
from harmony_amd import Harmony, prepare_setup_args
from bench_data import synth
from bench import run_to_convergence
n = int(os.environ.get("LEG_CELLS", "1000000"))
Z, meta, _ = synth(n, d=50, levels=(8, 64, 128), seed=7, nested=True)
skw, _ = prepare_setup_args(Z, meta, list(meta), nclust=200)
o = Harmony(seed=1, ref_arith=1)
o.setup(**skw)
o._scalar("sync"); t0 = time.perf_counter()
its = [run_to_convergence(o) for _ in range(1)]
o._scalar("sync"); ms = 1e3 * (time.perf_counter() - t0) / 1
o.set_profile(2); run_to_convergence(o); o._scalar("sync")
ph = {k: round(o._scalar("gputimer:" + k), 3) for k in ("kmeans_centers", "cluster_head", "randomize", "EO_update", "Rcells_update", "objective", "ridge_statistics", "arma_inv", "update_Zcorr")}
print(json.dumps({"ms": ms, "its": its, "phases": ph, "passes": o._get("seq:group_passes").tolist(), "runs": o._get("seq:group_runs").tolist()}))
PY
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_f
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_f -o p -- python /tmp/leg3.py > /tmp/leg3.out 2>&1
grep '^{' /tmp/leg3.out > $R/gpurun_out/r6_f.txt || tail -5 /tmp/leg3.out > $R/gpurun_out/r6_f.txt
python - <<'PY' >> $R/gpurun_out/r6_f.txt
import csv, glob
f = (glob.glob("/tmp/prof_f/**/*kernel_stats.csv", recursive=True) + glob.glob("/tmp/prof_f/*kernel_stats.csv"))[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:18]:
    print("   %-58s calls %6s avg %9.1f us (min %.1f max %.1f)  total %8.2f ms  %5.1f%%" % (r["Name"][:58], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, 100 * float(r["TotalDurationNs"]) / tot))
PY
cat $R/gpurun_out/r6_f.txt
