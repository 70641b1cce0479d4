#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for EXTRA in "" "--p2p"; do for cells in 400000 2000000; do
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 tools/dist_two_proc.py --cells $cells $EXTRA 2>&1 | grep -E "DIST2_OK|Mismatch|Max abs|Max rel|AssertionError|it, it1|\(.*\)$" | head -12
done; done | tee gpurun_out/c14_p2p.log
