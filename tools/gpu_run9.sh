#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-run9}; mkdir -p $O; cd $R
timeout -k 10 500 python tools/gpu_timing_dec.py 300000 > $O/dec_timing.log 2>&1; tail -3 $O/dec_timing.log | cut -c1-900
timeout -k 10 900 python bench.py --reads 10000000 --genome 300000000 --gs 300 --steps 1 --warmup 0 --no-cpu-baseline --no-pcie --no-t255 --no-rows --concurrent 0 > $O/bench_10M.log 2>&1; tail -1 $O/bench_10M.log | cut -c1-2500
