#!/bin/bash
# hipemu -- TEST INFRASTRUCTURE ONLY: compile the product's pure-HIP sources for the host CPU against the
# tests/emu shim (EMU_FLAGS=-g for a debuggable build: the debug info of the unrolled kernels doubles the compile time) (see tests/emu/hip/hip_runtime.h).  The product never loads this library.
set -e
cd "$(dirname "$0")/../.."
CXX=${EMU_CXX:-/opt/rocm/lib/llvm/bin/clang++}
# (the two RD_BUILD_* options: 4 "compute units", fp16 persistent kernel in its production forms only -- k_conv3.h conv_num_cus)
$CXX -x c++ -std=c++17 -O2 -fPIC -shared ${EMU_FLAGS} -DRD_BUILD_NUM_CUS=4 -DRD_BUILD_F16_PRODUCTION_FORMS_ONLY -Itests/emu -Iinclude rangedet_amd/csrc/rd_api.hip -o tests/emu/librangedet_emu.so
