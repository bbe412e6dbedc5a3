#!/bin/bash
# The library's host-only paths (graph builders, index cache, host logic) under AddressSanitizer + UBSan, in this container (no GPU needed; GPU sanitizers are not available on the pool):
#   bash scripts/asan_host.sh        -> builds graphchainer_amd/libgraphchainer_amd_asan.so and runs the non-GPU tests that load the library against it
set -e
cd "$(dirname "$0")/.."
make -C graphchainer_amd/csrc variant NAME=asan FLAGS="-fsanitize=address,undefined -fno-gpu-sanitize -fno-omit-frame-pointer -g -shared-libsan"
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
export ASAN_OPTIONS=detect_leaks=0:halt_on_error=0:verify_asan_link_order=0 UBSAN_OPTIONS=print_stacktrace=1
LD_PRELOAD=$RT GC_LIBRARY=$PWD/graphchainer_amd/libgraphchainer_amd_asan.so python -m pytest tests/test_index_cache.py tests/test_graph_build.py tests/test_library_exports.py tests/test_host_logic.py -q -s -m "not gpu" > /tmp/gc_asan.log 2>&1 || true
tail -2 /tmp/gc_asan.log
echo "sanitizer reports: $(grep -c 'runtime error\|AddressSanitizer' /tmp/gc_asan.log)"
