#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/icache; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE --output-format csv -d $O/p1 -- python3 $R/bench.py --reads 300000 --steps 1 --warmup 0 --no-cpu-baseline --no-pcie --no-t255 > $O/p1.log 2>&1
timeout 600 rocprofv3 --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $O/p2 -- python3 $R/bench.py --reads 300000 --steps 1 --warmup 0 --no-cpu-baseline --no-pcie --no-t255 > $O/p2.log 2>&1
cd $R
python tools/pmc_summary.py $O/p1 | grep -E "encode|insert"
python tools/pmc_summary.py $O/p2 | grep -E "encode|insert"
tail -2 $O/p1.log | cut -c1-300
rm -rf $O/p1 $O/p2
