#!/bin/bash
# all GPU tests, verbose, the whole log kept (a crash's first lines matter)
out=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4_pytest}
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 1800 python -X faulthandler -m pytest tests -v -m gpu -x $PYTEST_ARGS > $out/pytest_full.txt 2>&1
echo "rc $?"
grep -n "PASSED\|FAILED\|ERROR" $out/pytest_full.txt | tail -8
grep -n -B2 -A25 "Fatal Python error\|terminate called" $out/pytest_full.txt | head -80
tail -5 $out/pytest_full.txt
