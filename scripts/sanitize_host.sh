#!/bin/bash
# CPU-only sanitizer runs of the host code (GPU AddressSanitizer is not available on the pool):
#   ASan + UBSan: index cache write / check / load / rebuild, then checks of 600 files with a mutated payload and a recomputed checksum
#   TSan:         the threaded start-up builders (MPC index per component, minimizer scan per node chunk) with 8 threads
#   ASan + UBSan: the oracle itself (whole pipeline, stitching, encoders) under its golden and unit tests
#   ASan + UBSan: the encoders' split-node cursor over every letter of a graph
# Usage (repo root): bash scripts/sanitize_host.sh
set -eu
root=$(cd "$(dirname "$0")/.." && pwd)
H=$root/graphchainer_amd/csrc/host
work=/tmp/gc_sanitize
mkdir -p $work
g++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-omit-frame-pointer -I$H $root/scripts/sanitize/cache_fuzz.cpp $H/gc_index_cache.cpp $H/gc_graph.cpp $H/gc_minimizer.cpp -o $work/cache_fuzz -lpthread
$work/cache_fuzz $root/tests/golden/syn20k.gfa 600
(cd $root && python3 -c "from graphchainer_amd.synth import SynthGraph; SynthGraph(600_000, seed=5).write_gfa('$work/g600k.gfa')")
g++ -O1 -g -std=c++17 -fsanitize=thread -I$H $root/scripts/sanitize/threaded_build.cpp $H/gc_graph.cpp $H/gc_minimizer.cpp -o $work/threaded_build -lpthread
GC_BUILD_THREADS=8 $work/threaded_build $work/g600k.gfa
g++ -O1 -g -std=c++17 -fPIC -fsanitize=address,undefined -fno-omit-frame-pointer -Wno-sign-compare -shared -o $work/liboracle_asan.so $root/oracle/oracle_capi.cpp $H/gc_graph.cpp $H/gc_minimizer.cpp
(cd $root && GC_ORACLE_LIBRARY=$work/liboracle_asan.so ASAN_OPTIONS=detect_leaks=0 LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) python3 -m pytest tests/test_oracle_golden.py tests/test_oracle_units.py -x -q)
g++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-omit-frame-pointer -I$H $root/tests/output_host/letters_test.cpp $H/gc_output.cpp $H/gc_graph.cpp -o $work/letters_test -lpthread -lz
ASAN_OPTIONS=detect_leaks=0 $work/letters_test $root/tests/golden/syn20k.gfa
echo "sanitizers: clean"
