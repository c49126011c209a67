#!/bin/bash
# Sanitizer runs of the HOST side of libstarkperp (SURVEY section 5 "Race detection /
# sanitizers").  Run on the GPU box:   bash tools/run_sanitizers.sh [out_dir]
#   1. ASan + UBSan:  tests/cabi/cabi_smoke.c and tests/cabi/cabi_threads.cpp (8 host threads on every stateful
#      entry point, one and two contexts) as native programs against csrc/build_asan/libstarkperp_asan.so, then the Python GPU
#      suites that exercise host concurrency with the runtime preloaded (STARKPERP_LIB points the ctypes layer at
#      the instrumented library);
#   2. TSan: the two native programs against csrc/build_tsan/libstarkperp_tsan.so.
# The libraries are built by `make -C stark-perpetual_amd/csrc SAN=address|thread` (host code instrumented, device
# code untouched).  Output: one log per run + summary.txt under out_dir (default gpurun_out/sanitizers).
set -u
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="${1:-$ROOT/gpurun_out/sanitizers}"
mkdir -p "$OUT"
CSRC="$ROOT/stark-perpetual_amd/csrc"
CLANG=/opt/rocm/lib/llvm/bin/clang
RT="$($CLANG -print-resource-dir)/lib/linux"
export LD_LIBRARY_PATH="$RT:$CSRC/build_asan:$CSRC/build_tsan:${LD_LIBRARY_PATH:-}"
SUMMARY="$OUT/summary.txt"
: > "$SUMMARY"
note() { echo "$*" | tee -a "$SUMMARY"; }
note "sanitizer runs $(date -u +%Y-%m-%dT%H:%M:%SZ) on $(hostname); clang resource dir $RT"
for san in asan tsan; do
  [ -f "$CSRC/build_$san/libstarkperp_$san.so" ] || make -C "$ROOT/stark-perpetual_amd/csrc" SAN=$([ $san = asan ] && echo address || echo thread) -j 16 > "$OUT/build_$san.log" 2>&1
done

# protect_shadow_gap=0: the HIP runtime maps device memory into ASan's shadow gap; detect_leaks=0 for the Python
# runs only (the interpreter leaks by design); halt_on_error so that the first finding fails the run
ASAN_NATIVE="protect_shadow_gap=0:detect_leaks=1:halt_on_error=1:abort_on_error=0:detect_stack_use_after_return=1"
ASAN_PY="protect_shadow_gap=0:detect_leaks=0:halt_on_error=1"
export UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1"
export LSAN_OPTIONS="suppressions=$ROOT/tools/lsan.supp:print_suppressions=0"
export TSAN_OPTIONS="halt_on_error=0:second_deadlock_stack=1:suppressions=$ROOT/tools/tsan.supp:history_size=4"

run() {  # name, command...
  local name="$1"; shift
  local file="${name%% *}"
  local t0=$SECONDS
  timeout 1500 "$@" > "$OUT/$file.log" 2>&1
  local rc=$?
  local findings
  findings=$(grep -c -E "ERROR: AddressSanitizer|ERROR: LeakSanitizer|runtime error:|WARNING: ThreadSanitizer" "$OUT/$file.log")
  local verdict
  verdict=$(grep -h -E "cabi_threads ok|cabi_smoke ok|passed|failed|selftest-.* done" "$OUT/$file.log" | tail -n 1 | cut -c1-110)
  note "$(printf '%-48s rc=%-3d findings=%-3d %4ds  %s' "$name" $rc "$findings" $((SECONDS - t0)) "$verdict")"
}

# ---- ASan + UBSan, native -------------------------------------------------------------------------------
$CLANG -fsanitize=address,undefined -shared-libsan -g -O1 -I"$ROOT/include" "$ROOT/tests/cabi/cabi_smoke.c" \
  -o "$OUT/cabi_smoke_asan" -L"$CSRC/build_asan" -lstarkperp_asan -Wl,-rpath,"$CSRC/build_asan" -Wl,-rpath,"$RT" 2> "$OUT/build_cabi_smoke_asan.log"
${CLANG}++ -std=c++17 -fsanitize=address,undefined -shared-libsan -g -O1 -pthread -I"$ROOT/include" \
  "$ROOT/tests/cabi/cabi_threads.cpp" -o "$OUT/cabi_threads_asan" -L"$CSRC/build_asan" -lstarkperp_asan \
  -Wl,-rpath,"$CSRC/build_asan" -Wl,-rpath,"$RT" 2> "$OUT/build_cabi_threads_asan.log"
ASAN_OPTIONS=$ASAN_NATIVE run "asan_selftest_overflow (1 finding expected)" "$OUT/cabi_threads_asan" selftest-overflow
ASAN_OPTIONS=$ASAN_NATIVE run asan_cabi_smoke "$OUT/cabi_smoke_asan"
# CABI_FAST_EXIT: leak check, then _exit - the HSA runtime's own exit-time destructors can trip a CHECK inside AMD's
# ASan device allocator (no frame of ours), which would hide the program's verdict
ASAN_OPTIONS=$ASAN_NATIVE CABI_FAST_EXIT=1 run asan_cabi_threads_1ctx "$OUT/cabi_threads_asan" 1 3
ASAN_OPTIONS=$ASAN_NATIVE CABI_FAST_EXIT=1 run asan_cabi_threads_2ctx "$OUT/cabi_threads_asan" 2 3
ASAN_OPTIONS=$ASAN_NATIVE CABI_FAST_EXIT=1 run asan_cabi_threads_1ctx_long "$OUT/cabi_threads_asan" 1 12

# ---- ASan + UBSan under the Python suites that exercise host concurrency ---------------------------------
# ASan's dlopen interceptor makes the sanitizer runtime the caller of every dlopen, so the RUNPATH of torch's own
# libraries no longer finds their siblings (libcaffe2_nvrtc.so ...): name torch/lib explicitly
TORCH_LIB="$(python -c 'import os, importlib.util as u; print(os.path.join(list(u.find_spec("torch").submodule_search_locations)[0], "lib"))')"
export LD_LIBRARY_PATH="$LD_LIBRARY_PATH:$TORCH_LIB"
PYSAN="env LD_PRELOAD=$RT/libclang_rt.asan-x86_64.so ASAN_OPTIONS=$ASAN_PY STARKPERP_LIB=$CSRC/build_asan/libstarkperp_asan.so"
cd "$ROOT"
# pytest.main + os._exit: the interpreter's exit would run the HSA runtime's static destructors under ASan, whose own
# device allocator then calls back into the half-torn-down runtime (SEGV / CHECK inside libhsa-runtime64 <- operator
# delete <- __cxa_finalize: no frame of this library; seen in two of three runs) and would bury the suite's verdict
PYTEST_MAIN='import os, sys, pytest; rc = pytest.main(sys.argv[1:]); sys.stdout.flush(); sys.stderr.flush(); os._exit(int(rc))'
run asan_pytest_cabi        $PYSAN python -c "$PYTEST_MAIN" -x -q -m gpu tests/test_gpu_cabi.py -k "not c_consumer and not native_threads" -p no:cacheprovider
run asan_pytest_keyed       $PYSAN python -c "$PYTEST_MAIN" -x -q -m gpu tests/test_gpu_keyed_verify.py -p no:cacheprovider
run asan_pytest_state       $PYSAN python -c "$PYTEST_MAIN" -x -q -m gpu tests/test_gpu_state.py -p no:cacheprovider
run asan_pytest_multidevice $PYSAN python -c "$PYTEST_MAIN" -x -q -m gpu tests/test_gpu_multi_device.py -p no:cacheprovider

# ---- TSan, native ---------------------------------------------------------------------------------------
$CLANG -fsanitize=thread -shared-libsan -g -O1 -I"$ROOT/include" "$ROOT/tests/cabi/cabi_smoke.c" \
  -o "$OUT/cabi_smoke_tsan" -L"$CSRC/build_tsan" -lstarkperp_tsan -Wl,-rpath,"$CSRC/build_tsan" -Wl,-rpath,"$RT" 2> "$OUT/build_cabi_smoke_tsan.log"
${CLANG}++ -std=c++17 -fsanitize=thread -shared-libsan -g -O1 -pthread -I"$ROOT/include" \
  "$ROOT/tests/cabi/cabi_threads.cpp" -o "$OUT/cabi_threads_tsan" -L"$CSRC/build_tsan" -lstarkperp_tsan \
  -Wl,-rpath,"$CSRC/build_tsan" -Wl,-rpath,"$RT" 2> "$OUT/build_cabi_threads_tsan.log"
run "tsan_selftest_race (1 finding expected)" "$OUT/cabi_threads_tsan" selftest-race
run tsan_cabi_smoke "$OUT/cabi_smoke_tsan"
run tsan_cabi_threads_1ctx "$OUT/cabi_threads_tsan" 1 3
run tsan_cabi_threads_2ctx "$OUT/cabi_threads_tsan" 2 3
run tsan_cabi_threads_1ctx_long "$OUT/cabi_threads_tsan" 1 12

rm -f "$OUT"/cabi_*_asan "$OUT"/cabi_*_tsan
note "---- findings (first lines) ----"
grep -h -E "ERROR: AddressSanitizer|ERROR: LeakSanitizer|runtime error:|WARNING: ThreadSanitizer|SUMMARY:" "$OUT"/*.log | sort | uniq -c | sort -rn | head -60 >> "$SUMMARY"
cat "$SUMMARY"
