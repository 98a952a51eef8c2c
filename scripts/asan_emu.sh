#!/bin/bash
# The kernel sources under AddressSanitizer on the CPU SIMT emulator (test infrastructure only): device
# memory is host malloc there, so an out-of-bounds access of a kernel is an ASan report.  Builds
# build/asan/libg16_emu.so (the scheduler itself stays uninstrumented: it switches ucontext stacks) and
# runs the emulator-backed tests against it.  ~25 min on 8 cores.  Usage: scripts/asan_emu.sh [pytest -k expr]
set -e
cd "$(dirname "$0")/.."; ROOT=$PWD
mkdir -p build/asan
make -C circom_compat_amd/csrc -j8 emu EMU_BUILD=../../build/emu_asan EMU_OUT=$ROOT/build/asan/libg16_emu.so \
  EMU_FLAGS="-O1 -g -std=c++17 -fPIC -DG16_EMU -include $ROOT/tests/emu/emu_hip.h -fsanitize=address -fno-omit-frame-pointer -Wno-unknown-pragmas" > build/asan_build.log 2>&1
# libstdc++ is preloaded as well: python does not link it, and ASan's __cxa_throw interceptor aborts the
# process at the first C++ exception inside the library when it cannot find the real one
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libstdc++.so.6)" ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0 \
  G16_EMU_LIB=$ROOT/build/asan/libg16_emu.so python -m pytest tests/test_kernels.py tests/test_verify.py tests/test_emu_arith.py tests/test_loaders_abi.py \
  -x -q -m "not gpu" ${1:+-k "$1"}
