#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-c19}; mkdir -p $O; cd $R
timeout -k 10 1100 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "c19 and t64" -s --durations=3 > $O/c19.log 2>&1; tail -8 $O/c19.log | cut -c1-600
