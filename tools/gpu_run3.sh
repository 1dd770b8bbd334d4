#!/bin/bash
# quick parity + A/B against a baseline library + timing diagnostics + default geometry + concurrent files
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-run3}; mkdir -p $O; cd $R
bash tools/gpu_quick.sh $1 || exit 1
if [ -n "$2" ]; then timeout -k 10 400 python tools/ab_bench.py --reps 2 fqsqueezer_amd/libfqsx.so $2 > $O/ab.log 2>&1; tail -3 $O/ab.log | cut -c1-400; fi
timeout -k 10 300 python tools/gpu_default_geometry.py 64 > $O/geom.log 2>&1; tail -1 $O/geom.log | cut -c1-900
timeout -k 10 300 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-pcie --no-t255 --no-rows --concurrent 4 > $O/conc.log 2>&1; tail -1 $O/conc.log | python -c "import sys,json; print(json.dumps(json.loads(sys.stdin.read())['concurrent_files']))"
[ "$3" = "nodiag" ] || bash tools/gpu_diag.sh $1
