#!/bin/bash
# Round-end measurement on the GPU box: parity tests, default bench, rocprofv3 kernel stats, PMC traffic passes.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final
mkdir -p $O
cd $R
timeout -k 10 1200 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
timeout -k 10 900 python bench.py > $O/bench.log 2>&1; echo "bench rc=$?" >> $O/bench.log
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-pcie --no-t255 --concurrent 0 > $O/stats.log 2>&1
timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-pcie --no-t255 --concurrent 0 > $O/pmc_fetch.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-pcie --no-t255 --concurrent 0 > $O/pmc_write.log 2>&1
cd $R
python tools/pmc_summary.py $O/pmc_fetch > $O/pmc_fetch.txt 2>&1
python tools/pmc_summary.py $O/pmc_write > $O/pmc_write.txt 2>&1
# keep only the small summaries
find $O/stats -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
rm -rf $O/stats $O/pmc_fetch $O/pmc_write
tail -3 $O/pytest.log; tail -2 $O/bench.log; head -5 $O/kernel_stats.csv; cat $O/pmc_fetch.txt $O/pmc_write.txt | head
