#!/bin/bash
# timing-build diagnostics on the GPU box: in-kernel section times + per-launch role time stamps
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-diag}; mkdir -p $O; cd $R
if [ "$2" != "noroles" ]; then timeout -k 10 500 python tools/gpu_roles.py 256 150 64 > $O/roles.log 2>&1; tail -2 $O/roles.log; fi
if [ "$3" != "notiming" ]; then timeout -k 10 500 python tools/gpu_timing.py 1000000 64 - 1000 150 > $O/timing.log 2>&1; tail -6 $O/timing.log;
  timeout -k 10 500 python tools/gpu_timing.py 1000000 64 - 1000 150 steady > $O/timing_steady.log 2>&1; tail -6 $O/timing_steady.log;
  timeout -k 10 500 python tools/gpu_timing.py 1000000 64 - 1000 150 warm > $O/timing_warm.log 2>&1; tail -5 $O/timing_warm.log; fi
