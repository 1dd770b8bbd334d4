#!/bin/bash
# protocol fall-back tests (tests/test_gpu_protocol.py) + one short bench pass
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-proto}; mkdir -p $O; cd $R
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 || { tail -5 $O/smoke.log; exit 1; }
tail -1 $O/smoke.log | cut -c1-120
timeout -k 10 900 python -m pytest tests/test_gpu_protocol.py -q -m gpu --durations=8 > $O/pytest.log 2>&1; prc=$?
tail -25 $O/pytest.log
timeout -k 10 400 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-pcie --no-t255 --no-rows --concurrent 0 > $O/bench.log 2>&1; echo "bench rc=$?" >> $O/bench.log
tail -2 $O/bench.log | cut -c1-900
exit $prc
