#!/bin/bash
# quick parity + protocol tests + A/B + the rows (rooflines, reference legs)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-run6}; mkdir -p $O; cd $R
bash tools/gpu_quick.sh $1 || exit 1
timeout -k 10 600 python -m pytest tests/test_gpu_protocol.py -x -q -m gpu > $O/proto.log 2>&1; prc=$?; tail -3 $O/proto.log
[ $prc -eq 0 ] || exit 1
tag=$1; shift
timeout -k 10 500 python tools/ab_bench.py --reps 2 fqsqueezer_amd/libfqsx.so "$@" > $O/ab.log 2>&1; tail -4 $O/ab.log | cut -c1-330
timeout -k 10 900 python tools/bench_rows.py 300000 > $O/rows.log 2>&1; tail -1 $O/rows.log | cut -c1-3000
