#!/bin/bash
# K bench processes at once on one GPU: does aggregate throughput scale with independent files?
K=${1:-4}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/multi; mkdir -p $O
cd $R
for k in $(seq 1 $K); do
  python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pcie --no-t255 --concurrent 0 > $O/p$k.log 2>&1 &
done
wait
for k in $(seq 1 $K); do grep -o '"value": [0-9.]*' $O/p$k.log | head -1; done
