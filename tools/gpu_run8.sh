#!/bin/bash
# decoder: parity (decode tests, round trip) + rows
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-run8}; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "decodes or round_trip" > $O/pytest.log 2>&1; prc=$?; tail -3 $O/pytest.log
[ $prc -eq 0 ] || exit 1
timeout -k 10 900 python tools/bench_rows.py 300000 > $O/rows.log 2>&1; tail -1 $O/rows.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps(d['decode']))"
