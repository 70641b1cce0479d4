#!/bin/bash
# round 3: every test that reaches the device solve with several covariates / subset masks / fixed lambda, after the assembly + sparse Schur change
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity2.py tests/test_gpu_seq.py -x -q -k "two_covariates or fixed_lambda or synthetic_shapes or envelope_shapes or virtual_shards or fallback_paths or pbmc or config5_shape_200k or full_size_named or reference_arithmetic_fixture or needs_one_gpu or carried or cell_lines" 2>&1 | tail -4
