#!/bin/bash
# development iteration on the GPU box: parity tests, short bench, in-kernel timing breakdown
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/iter
mkdir -p $O
cd $R
timeout -k 10 240 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1 || { tail -5 $O/smoke.log; exit 1; }
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; prc=$?; echo "pytest rc=$prc" >> $O/pytest.log
tail -4 $O/pytest.log
[ $prc -eq 0 ] || exit 1
timeout -k 10 600 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-pcie --no-t255 --concurrent 0 > $O/bench.log 2>&1; echo "bench rc=$?" >> $O/bench.log
tail -2 $O/bench.log
if [ -f tools/libfqsx_timing.so ]; then timeout -k 10 300 python tools/gpu_timing.py 1000000 64 > $O/timing.log 2>&1; tail -5 $O/timing.log; fi

timeout -k 10 400 python tools/gpu_blocks.py > $O/blocks.log 2>&1; tail -1 $O/blocks.log
exit 0
