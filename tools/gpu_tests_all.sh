#!/bin/bash
# the complete GPU test suite as the driver runs it (full-size c18 / c19 runs included)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-tests}; mkdir -p $O; cd $R
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 || { tail -5 $O/smoke.log; exit 1; }
tail -1 $O/smoke.log | cut -c1-100
timeout -k 10 1050 python -m pytest tests -x -q -m gpu --durations=8 > $O/pytest.log 2>&1; prc=$?; tail -14 $O/pytest.log
exit $prc
