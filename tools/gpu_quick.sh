#!/bin/bash
# quick check of a risky change on the GPU box: smoke, the small parity tests, one short bench pass -- each under a short timeout
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-quick}; mkdir -p $O; cd $R
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 || { tail -5 $O/smoke.log; exit 1; }
tail -1 $O/smoke.log | cut -c1-120
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "10k or ragged or short_and_long or test_hip_matches_reference_150bp or pipeline or tiny_blocks" --durations=5 > $O/pytest.log 2>&1; prc=$?
tail -9 $O/pytest.log
[ $prc -eq 0 ] || exit 1
timeout -k 10 400 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-pcie --no-t255 --no-rows --concurrent 0 > $O/bench.log 2>&1; echo "bench rc=$?" >> $O/bench.log
tail -2 $O/bench.log | cut -c1-900
