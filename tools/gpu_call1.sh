#!/bin/bash
# round-2 GPU call 1: refactor validation (old suite + new fast tests), default bench line, torch-free RCCL probe
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -x -q -k "not arithmetic_gap_table and not config5" 2>&1 | tail -25 ) > gpurun_out/c1_tests.log 2>&1
( timeout 300 python bench.py --steps 10 --warmup 2 ) > gpurun_out/c1_bench.json 2> gpurun_out/c1_bench.err
gcc -std=c11 -Iinclude examples/comm_example.c -Lharmony_amd/lib -lharmony_mi355x -Wl,-rpath,$PWD/harmony_amd/lib -lm -o /tmp/comm_example
( NCCL_DEBUG=INFO NCCL_DEBUG_SUBSYS=INIT,BOOTSTRAP,NET timeout 90 /tmp/comm_example 0 1 /tmp/hmx_uid_a ) > gpurun_out/c1_comm_w1.log 2>&1
echo "exit $?" >> gpurun_out/c1_comm_w1.log
( NCCL_SOCKET_IFNAME=lo NCCL_DEBUG=INFO NCCL_DEBUG_SUBSYS=INIT,BOOTSTRAP,NET timeout 90 /tmp/comm_example 0 1 /tmp/hmx_uid_b ) > gpurun_out/c1_comm_w1_lo.log 2>&1
echo "exit $?" >> gpurun_out/c1_comm_w1_lo.log
ip addr > gpurun_out/c1_ip.log 2>&1; hostname >> gpurun_out/c1_ip.log 2>&1
tail -5 gpurun_out/c1_tests.log; head -c 600 gpurun_out/c1_bench.json; tail -5 gpurun_out/c1_comm_w1.log
