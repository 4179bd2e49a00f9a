#!/bin/bash
# tests/sanitize_cpu.sh -- the host side of libcxlspeckv.so (engine bookkeeping on the fake device, slab pool, coherence
# directory, legacy address space, C ABI) under AddressSanitizer + UBSan, and the threaded C-ABI test under
# ThreadSanitizer.  CPU only: GPU ASan / XNACK are not available on the MI355X pool; the device code is compiled as usual
# (clang ignores -fsanitize for gfx950).  Builds into tests/_build/{asan,tsan}; any report makes the run fail.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
RT=$(dirname "$(/opt/rocm/lib/llvm/bin/clang++ -print-file-name=libclang_rt.asan-x86_64.so)")
make -s -j8 -C "$ROOT/cxl-speckv_amd/csrc" OUT="$ROOT/tests/_build/asan" EXTRA="-fsanitize=address,undefined -fno-omit-frame-pointer -g" 2>&1 | grep -v "option-ignored\|^$" || true
make -s -j8 -C "$ROOT/cxl-speckv_amd/csrc" OUT="$ROOT/tests/_build/tsan" EXTRA="-fsanitize=thread -fno-omit-frame-pointer -g" 2>&1 | grep -v "option-ignored\|^$" || true
cd "$ROOT"
log=$(mktemp)
SPECKV_LIB_PATH="$ROOT/tests/_build/asan/libcxlspeckv.so" LD_PRELOAD="$RT/libclang_rt.asan-x86_64.so" \
  ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
  python -m pytest tests -q -s -m "not gpu" -p no:cacheprovider > "$log" 2>&1 || { tail -40 "$log"; exit 1; }
if grep -q "runtime error\|AddressSanitizer" "$log"; then grep -B2 -A20 "runtime error\|AddressSanitizer" "$log" | head -80; exit 1; fi
echo "asan+ubsan: $(tail -1 "$log")"
SPECKV_LIB_PATH="$ROOT/tests/_build/tsan/libcxlspeckv.so" LD_PRELOAD="$RT/libclang_rt.tsan-x86_64.so" \
  TSAN_OPTIONS=halt_on_error=0:report_signal_unsafe=0 \
  python -m pytest tests/test_cabi_boundary.py -q -s -k many_threads -p no:cacheprovider > "$log" 2>&1 || { tail -40 "$log"; exit 1; }
if grep -q "WARNING: ThreadSanitizer" "$log"; then grep -A30 "WARNING: ThreadSanitizer" "$log" | head -80; exit 1; fi
echo "tsan: $(tail -1 "$log")"
rm -rf "$ROOT/tests/_build/asan" "$ROOT/tests/_build/tsan"      # 20 MB that would otherwise travel with every gpurun snapshot
