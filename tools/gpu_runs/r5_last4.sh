#!/bin/bash
# round 5: everything of the GPU suite but its three long oracle tables (1M: r5_last3.sh; 2M and the configs[4] shape at 1M: the evidence run), on the final tree
exec </dev/null
R=$GRAFT_REPO_ROOT; cd $R || exit 1
timeout 900 python -m pytest tests -q -m gpu -k "not (gap_table and 1000000) and not (gap_table and 2000000) and not config5_shape_1M" 2>&1 | tail -4
