#!/bin/bash
# CPU-only sanitizer runs of the host code (GPU AddressSanitizer is not available on the pool):
#   ASan + UBSan: index cache write / check / load / rebuild, then checks of 600 files with a mutated payload and a recomputed checksum
#   TSan:         the threaded start-up builders (MPC index per component, minimizer scan per node chunk) with 8 threads
#   ASan + UBSan: the oracle itself (whole pipeline, stitching, encoders) under its golden and unit tests
#   ASan + UBSan: the encoders' split-node cursor over every letter of a graph
#   TSan, ASan + UBSan (r4): the flat first build of a graph on 8 threads against the literal builder; the hash-order replay against the real containers
#   ASan + UBSan (r5): the libstdc++ sort replays; the whole product library's host side behind ctypes (non-GPU tests)
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
# r4: the flat, threaded first build (gc_graph_fast.cpp) against the literal one - TSan with 8 threads over a synthetic genome, ASan + UBSan over the same and the golden graphs; gc::HashOrder against the real containers
(cd $root && python3 -c "from graphchainer_amd.synth import SynthGenome; SynthGenome(6, 120_000, seed=9, multi_allelic=0.1, nested=0.1, minus_links=0.3, repeats=2, repeat_len=1200).write_gfa('$work/genome.gfa')")
g++ -O1 -g -std=c++17 -fsanitize=thread -I$H $root/tests/graph_build/build_compare.cpp $H/gc_graph.cpp $H/gc_graph_fast.cpp -o $work/build_compare_tsan -lpthread
GC_BUILD_THREADS=8 $work/build_compare_tsan $work/genome.gfa
g++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-omit-frame-pointer -I$H $root/tests/graph_build/build_compare.cpp $H/gc_graph.cpp $H/gc_graph_fast.cpp -o $work/build_compare_asan -lpthread
for f in $work/genome.gfa $root/tests/golden/syn20k.gfa $root/tests/golden/ref_test_graph.gfa; do GC_BUILD_THREADS=8 ASAN_OPTIONS=detect_leaks=0 $work/build_compare_asan $f; done
g++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-omit-frame-pointer -I$H $root/tests/hashorder/hashorder_test.cpp -o $work/hashorder_asan
ASAN_OPTIONS=detect_leaks=0 $work/hashorder_asan
# r5: the libstdc++ sort replays (one lane; the independent-steps form the wave kernel runs) against std::sort under ASan + UBSan
g++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-omit-frame-pointer -I$root/graphchainer_amd/csrc/hip $root/tests/stdsort/stdsort_test.cpp -o $work/stdsort_asan
ASAN_OPTIONS=detect_leaks=0 $work/stdsort_asan
# r5: the whole product library's host side (hipcc, -fno-gpu-sanitize) behind ctypes: the non-GPU tests that load it - both graph builders, the index cache and its mutated files, the ABI tests
make -C $root/graphchainer_amd/csrc variant NAME=asan FLAGS="-fsanitize=address,undefined -fsanitize-recover=address -fno-gpu-sanitize -fno-omit-frame-pointer -g -shared-libsan" > /dev/null
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
set +e
(cd $root && ASAN_OPTIONS=detect_leaks=0:halt_on_error=0:verify_asan_link_order=0 UBSAN_OPTIONS=print_stacktrace=1 LD_PRELOAD=$RT GC_LIBRARY=$root/graphchainer_amd/libgraphchainer_amd_asan.so \
  python3 -m pytest tests/test_index_cache.py tests/test_graph_build.py tests/test_library_exports.py tests/test_host_logic.py -q -s -m "not gpu" > $work/library_asan.log 2>&1)
library_rc=$?     # (set -e is off for this one command: its exit code and its report count decide below - ADVICE r5: "|| true" and an unconditional "clean" hid both)
set -e
tail -1 $work/library_asan.log
reports=$(grep -c 'runtime error\|AddressSanitizer' $work/library_asan.log || true)
echo "sanitizer reports in the library run: $reports (pytest exit code $library_rc)"
rm -f $root/graphchainer_amd/libgraphchainer_amd_asan.so
if [ "$library_rc" -ne 0 ] || [ "$reports" -ne 0 ]; then
  echo "sanitizers: FAILED - see $work/library_asan.log"
  grep -n 'runtime error\|AddressSanitizer' $work/library_asan.log | head -20
  exit 1
fi
echo "sanitizers: clean"
