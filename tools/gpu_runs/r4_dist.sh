#!/bin/bash
# round 4: the sharded protocol paths on one GPU (two processes), the torch-free bench line
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 1200 python -m pytest tests/test_gpu_parity2.py -x -q -m gpu -s -k "two_processes or bench_bootstraps or torch_free_c_host" 2>&1 | tail -25
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "virtual_shards or rccl or nccl_hook" 2>&1 | tail -4
