#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3; mkdir -p $O
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29577 tools/dist_two_proc.py --p2p --split 0.9 --chain-max-tpw 0.02 > $O/dbg_unequal.out 2> $O/dbg_unequal.err
echo rc=$?; tail -5 $O/dbg_unequal.out; grep -v "^\[Gloo\]\|^W0\|^E0\|^$" $O/dbg_unequal.err | head -40
