#!/bin/bash
# the rows beside the headline path: rates (tools/bench_rows.py) and rocprofv3 kernel statistics of the same pass
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-rows}; mkdir -p $O; cd $R
timeout -k 10 600 python tools/bench_rows.py 300000 > $O/rows.log 2>&1; tail -1 $O/rows.log | cut -c1-1200
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/tools/bench_rows.py 300000 > $O/stats.log 2>&1
find $O/stats -name "*kernel_stats.csv" -exec cp {} $O/rows_kernel_stats.csv \; ; rm -rf $O/stats
head -8 $O/rows_kernel_stats.csv
