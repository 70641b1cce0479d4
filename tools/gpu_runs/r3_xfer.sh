#!/bin/bash
# round 3: the R seam through the page-locked ring + host threads (default) against registering the caller's matrix (HMX_XFER=pin)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3x; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity2.py -x -q -k "ingest or f32_and_device or shard or fixture or cell_lines or tiny" 2>&1 | tail -3
python - <<'PY'
import os, sys, time, numpy as np
sys.path.insert(0, ".")
from bench_data import synth
from harmony_amd import Harmony, prepare_setup_args
Z, meta, _ = synth(1000000, d=50, levels=(10,), seed=7)
skw, _ = prepare_setup_args(Z, meta, "cov0", nclust=100)
for thr in ("8", "pin"):
    if thr == "pin": os.environ["HMX_XFER"] = "pin"
    else: os.environ["HMX_XFER_THREADS"] = thr
    for rep in range(2):
        g = Harmony(seed=1); g.setup(**skw)
        t0 = time.perf_counter(); zc = g.getZcorr(); eg = 1e3 * (time.perf_counter() - t0)
        print(thr, rep, "ingest %.1f ms , egress %.1f ms" % (g.timer("ingest_Z"), eg), flush=True)

        del g, zc
PY
