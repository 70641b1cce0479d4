#!/bin/bash
# round 6: kernel-level timing of the objective's fused launch (rocprofv3 --kernel-trace --stats over a probe of 100M terms per chain)
exec </dev/null
R=$GRAFT_REPO_ROOT; cd $R || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_seq.py -q -x -k 'term_arrays' 2>&1 | tail -3 > gpurun_out/r6_c.txt
cat > /tmp/probe.py <<'PY'
import sys, os, ctypes as C
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import numpy as np
from harmony_amd import _lib
n = int(sys.argv[1]); passes = int(sys.argv[2]); seg = int(sys.argv[3])
rng = np.random.default_rng(1)
T = (rng.random((3, n), dtype=np.float32) * 1e-3).astype(np.float32)
tot = np.empty(3, np.float32); mm, res = C.c_int64(-1), C.c_double(-1)
fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
for _ in range(3):
    st = _lib.load().hmx_debug_seq_arr(fp(T), n, 3, seg, passes, fp(tot), C.byref(mm), C.byref(res))
print(st, tot, mm.value, res.value)
PY
cd /tmp && export TMPDIR=/tmp
for cfg in "100000000 2 0 0" "100000000 4 0 0" "100000000 2 0 8" "100000000 2 0 9"; do
  set -- $cfg; export HMX_OBJF_DBG=$4
  rm -rf /tmp/prof_c
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c -o p -- python /tmp/probe.py $1 $2 $3 > /tmp/probe.out 2>&1
  echo "== n=$1 passes=$2 seg=$3 dbg=$4" >> $R/gpurun_out/r6_c.txt
  grep -v 'simple_timer\|Opened' /tmp/probe.out | tail -5 >> $R/gpurun_out/r6_c.txt; find /tmp/prof_c -type f | head -5 >> $R/gpurun_out/r6_c.txt
  python - <<'PY' >> $R/gpurun_out/r6_c.txt
import csv, glob
for f in glob.glob("/tmp/prof_c/**/*kernel_stats.csv", recursive=True) + glob.glob("/tmp/prof_c/*kernel_stats.csv"):
    for r in list(csv.DictReader(open(f)))[:6]:
        print("   %-60s calls %5s avg %10.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
    break
PY
done
cat $R/gpurun_out/r6_c.txt
