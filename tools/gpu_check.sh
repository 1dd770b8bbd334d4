#!/bin/bash
# quick GPU check: parity tests + short bench + (optional) full-workload reference CPU timing
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/check
mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -x -q -m gpu --durations=8 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
timeout 600 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-pcie --no-t255 > $O/bench.log 2>&1; echo "bench rc=$?" >> $O/bench.log
if [ "$1" = "ref" ]; then timeout 1500 python tools/ref_cpu_full.py 64 8 > $O/ref_cpu.log 2>&1; echo "ref rc=$?" >> $O/ref_cpu.log; fi
tail -15 $O/pytest.log; tail -2 $O/bench.log; tail -4 $O/ref_cpu.log 2>/dev/null
