#!/bin/bash
# r6: the lockstep fragment extension kernel against the plain-layout one - GPU tests, then an interleaved A/B of the default bench
cd $GRAFT_REPO_ROOT
bash scripts/gpu/check.sh ${1:-r6_first} --no-cpu-baseline
AB_ENV_slab="GC_EXTEND_SLAB=1" bash scripts/gpu/ab.sh ${1:-r6_first}_ab 2 prod prod:slab
