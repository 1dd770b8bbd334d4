#!/bin/bash
# HBM traffic of the rows' kernels, one kernel's rows at a time: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes of
# `tools/bench_rows.py 300000 <rows>` (counters in KiB; tools/pmc_summary.py sums them per kernel).  -qm o and -qm 8 in passes of their own.
# usage: tools/gpu_prof_rows.sh <tag>
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-rows_pmc}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for rows in decode sorted original_order pe_sorted quality_o quality_8; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 400 rocprofv3 --pmc $c --output-format csv -d $O/p_${rows}_$c -- python3 $R/tools/bench_rows.py 300000 $rows > $O/${rows}_$c.log 2>&1
    python3 $R/tools/pmc_summary.py $O/p_${rows}_$c > $O/${rows}_$c.txt 2>&1; rm -rf $O/p_${rows}_$c
  done
  echo "== $rows"; cat $O/${rows}_FETCH_SIZE.txt $O/${rows}_WRITE_SIZE.txt | grep -E "k_encode|k_decode|k_qual" | cut -c1-200
done
