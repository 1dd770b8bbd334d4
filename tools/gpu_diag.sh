#!/bin/bash
# timing-build diagnostics on the GPU box: in-kernel section times + per-launch role time stamps
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-diag}; mkdir -p $O; cd $R
timeout -k 10 500 python tools/gpu_timing.py 1000000 64 - 1000 150 > $O/timing.log 2>&1; tail -6 $O/timing.log
timeout -k 10 500 python tools/gpu_roles.py 70 150 64 > $O/roles_warm.log 2>&1; tail -1 $O/roles_warm.log
