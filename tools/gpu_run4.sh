#!/bin/bash
# quick parity + large-k geometry tests + A/B against baseline libraries + default geometry
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-run4}; mkdir -p $O; cd $R
bash tools/gpu_quick.sh $1 || exit 1
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "large_k or default_geometry" --durations=4 > $O/pytest_k.log 2>&1; tail -6 $O/pytest_k.log
shift
timeout -k 10 500 python tools/ab_bench.py --reps 2 fqsqueezer_amd/libfqsx.so "$@" > $O/ab.log 2>&1; tail -4 $O/ab.log | cut -c1-330
timeout -k 10 300 python tools/gpu_default_geometry.py 64 > $O/geom.log 2>&1; tail -1 $O/geom.log | cut -c1-900
