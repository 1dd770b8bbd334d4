#!/bin/bash
# full GPU test suite (without the full-size tests) + sharded world-1 bench lines (native RCCL driver)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-run7}; mkdir -p $O; cd $R
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 || { tail -5 $O/smoke.log; exit 1; }
timeout -k 10 900 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_fullsize.py --durations=6 > $O/pytest.log 2>&1; prc=$?; tail -12 $O/pytest.log
[ $prc -eq 0 ] || exit 1
timeout -k 10 400 python bench.py --sharded --steps 1 --warmup 0 > $O/sharded.log 2>&1; tail -1 $O/sharded.log | cut -c1-900
