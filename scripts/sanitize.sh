#!/bin/bash
# CPU-build sanitizer runs (GPU AddressSanitizer is not available on this pool; everything here runs without a GPU).
#   1. oracle under AddressSanitizer + UndefinedBehaviorSanitizer: tests/test_oracle_*.py, tests/test_golden.py
#   2. the shared-memory exchange (csrc/ecc_exchange.cpp) under ThreadSanitizer: 4 ranks as threads, 2000 generations
#   3. the group's worker hand-off (csrc/ecc_worker_pool.h) under ThreadSanitizer: 8 ranks, 20 000 jobs, sleeps, failures
#   3b. the pose batch's host-side comparison (csrc/ecc_pose_diff.h) under ThreadSanitizer: 1 and 8 threads, the same result
#   4. host code of libecc_hip.so under UndefinedBehaviorSanitizer: the no-GPU ABI / host-function tests
# usage: scripts/sanitize.sh [logfile]     (default profiles/r03_sanitize.log)
set -u
cd "$(dirname "$0")/.."
LOG=${1:-profiles/r03_sanitize.log}
TMP=$(mktemp -d /tmp/ecc_san.XXXXXX)
fail=0
{
echo "== sanitize.sh $(date -u +%Y-%m-%dT%H:%MZ)  gcc $(gcc -dumpversion)  $(/opt/rocm/bin/hipcc --version 2>/dev/null | grep -m1 -i 'hip version')"

echo; echo "== 1. oracle: -fsanitize=address,undefined (make -C oracle asan)"
make -s -C oracle asan || fail=1
ASAN_RT=$(gcc -print-file-name=libasan.so)
LD_PRELOAD=$ASAN_RT ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
  ECC_ORACLE_LIB=$PWD/oracle/libecc_oracle_asan.so OMP_NUM_THREADS=4 \
  python -m pytest tests/test_oracle_pins.py tests/test_oracle_properties.py tests/test_oracle_independent.py tests/test_oracle_radon_contract.py tests/test_golden.py -q -m "not gpu" -p no:cacheprovider 2>&1 | tail -4
[ ${PIPESTATUS[0]} -eq 0 ] || fail=1

echo; echo "== 2. ecc_exchange.cpp: -fsanitize=thread (tests/c/tsan_exchange.cpp)"
g++ -std=c++17 -O1 -g -fsanitize=thread -fno-omit-frame-pointer -Iinclude tests/c/tsan_exchange.cpp epipolarconsistency_amd/csrc/ecc_exchange.cpp \
  -lrt -lpthread -o $TMP/tsan_exchange && TSAN_OPTIONS=halt_on_error=1 $TMP/tsan_exchange || fail=1

echo; echo "== 3. ecc_worker_pool.h: -fsanitize=thread (tests/c/tsan_worker_pool.cpp)"
g++ -std=c++17 -O1 -g -fsanitize=thread -fno-omit-frame-pointer tests/c/tsan_worker_pool.cpp -lpthread -o $TMP/tsan_worker_pool \
  && TSAN_OPTIONS=halt_on_error=1 $TMP/tsan_worker_pool || fail=1

echo; echo "== 3b. ecc_pose_diff.h: -fsanitize=thread (tests/c/tsan_pose_diff.cpp)"
g++ -std=c++17 -O1 -g -fsanitize=thread -fno-omit-frame-pointer tests/c/tsan_pose_diff.cpp -lpthread -o $TMP/tsan_pose_diff \
  && TSAN_OPTIONS=halt_on_error=1 $TMP/tsan_pose_diff || fail=1

echo; echo "== 4. libecc_hip.so host code: -Xarch_host -fsanitize=undefined (ECC_HIP_LIB), no-GPU ABI and host-function tests"
python - <<PY || fail=1
import os, subprocess, sys
sys.path.insert(0, os.getcwd())
from epipolarconsistency_amd import build
objs = []
for s in build.SOURCES:
    o = os.path.join("$TMP", os.path.splitext(s)[0] + ".o")
    cmd = ["/opt/rocm/bin/hipcc"] + build.FLAGS + build.PER_SOURCE_FLAGS.get(s, []) + ["-Xarch_host", "-fsanitize=undefined", "-Xarch_host", "-fno-sanitize=vptr,function",
           "-Xarch_host", "-fno-sanitize-recover=undefined", "-c", os.path.join(build.CSRC, s), "-o", o]
    subprocess.run(cmd, check=True)
    objs.append(o)
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-fsanitize=undefined"] + objs + ["-lrt", "-lpthread", "-ldl", "-o", "$TMP/libecc_hip_ubsan.so"], check=True)
print("built $TMP/libecc_hip_ubsan.so")
PY
# a shared library is linked without the sanitizer runtime (the executable is expected to bring it): preload clang's
UBSAN_RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.ubsan_standalone-x86_64.so | head -1)
LD_PRELOAD=$UBSAN_RT UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 ECC_HIP_LIB=$TMP/libecc_hip_ubsan.so \
  python -m pytest tests/test_abi_and_host.py tests/test_bench_helpers.py -q -m "not gpu" -p no:cacheprovider 2>&1 | tail -4
[ ${PIPESTATUS[0]} -eq 0 ] || fail=1

echo; echo "== result: $([ $fail -eq 0 ] && echo CLEAN || echo FAILED)"
} 2>&1 | tee "$LOG"
rm -rf "$TMP"
grep -q "== result: CLEAN" "$LOG"
