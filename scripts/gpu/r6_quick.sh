#!/bin/bash
# r6: the GPU tests that exercise the fragment pipeline, then the kernels alone
cd $GRAFT_REPO_ROOT
timeout 1500 python -X faulthandler -m pytest tests -q -m gpu -x > gpurun_out/${1:-r6_quick}_pytest.txt 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/${1:-r6_quick}_pytest.txt
bash scripts/gpu/r6_alone.sh ${1:-r6_quick}
