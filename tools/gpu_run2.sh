#!/bin/bash
# quick parity + timing diagnostics + the c18 full-size test
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-run2}; mkdir -p $O; cd $R
bash tools/gpu_quick.sh $1 || exit 1
bash tools/gpu_diag.sh $1
if [ "$2" = "c18" ]; then timeout -k 10 900 python -m pytest tests/test_gpu_fullsize.py -k c18 -x -q --durations=3 > $O/c18.log 2>&1; tail -6 $O/c18.log; fi
