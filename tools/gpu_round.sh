#!/bin/bash
# One measurement call on the GPU box: smoke, GPU parity tests, default bench (the metric's 150 bp workload), rocprofv3
# kernel stats, FETCH/WRITE traffic passes and the instruction-cache / issue counters of the encode kernel.
# usage: tools/gpu_round.sh <tag> [notest|profonly] [nopmc]   (profonly: only the rocprofv3 passes)
TAG=${1:-run}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
if [ "$2" != "profonly" ]; then
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 || { tail -5 $O/smoke.log; exit 1; }
tail -1 $O/smoke.log | cut -c1-200
fi
if [ "$2" != "notest" ] && [ "$2" != "profonly" ]; then
  timeout -k 10 2400 python -m pytest tests -x -q -m gpu --durations=6 > $O/pytest.log 2>&1; prc=$?; echo "pytest rc=$prc" >> $O/pytest.log
  tail -12 $O/pytest.log
  [ $prc -eq 0 ] || exit 1
fi
if [ "$2" != "profonly" ]; then
timeout -k 10 1500 python bench.py > $O/bench.log 2>&1; echo "bench rc=$?" >> $O/bench.log
tail -2 $O/bench.log | cut -c1-400
fi
[ "$3" = "nopmc" ] && exit 0
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-pcie --no-t255 --no-rows --no-large --concurrent 0"
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $B > $O/stats.log 2>&1
find $O/stats -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \; ; rm -rf $O/stats
timeout -k 10 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- $B > $O/pmc_fetch.log 2>&1
timeout -k 10 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- $B > $O/pmc_write.log 2>&1
timeout -k 10 900 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE --output-format csv -d $O/pmc_ic -- $B > $O/pmc_ic.log 2>&1
timeout -k 10 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_IFETCH --output-format csv -d $O/pmc_sq -- $B > $O/pmc_sq.log 2>&1
cd $R
for k in fetch write ic sq; do python tools/pmc_summary.py $O/pmc_$k > $O/pmc_$k.txt 2>&1; rm -rf $O/pmc_$k; done
head -6 $O/kernel_stats.csv; cat $O/pmc_fetch.txt $O/pmc_write.txt $O/pmc_ic.txt $O/pmc_sq.txt | grep -E "encode|block" | cut -c1-600
