#!/bin/bash
# tools/libaugcl_tuning.so = the library built with -DAUGCL_TUNING (environment overrides of the strip tables, in-kernel
# phase stamps of the x3 conv kernels); objects go to a scratch copy of csrc so the product objects stay untouched.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
T=/tmp/augcl_tuning_build/augcl_amd/csrc
mkdir -p $T /tmp/augcl_tuning_build/include
cp $ROOT/include/augcl.h /tmp/augcl_tuning_build/include/
cp $ROOT/augcl_amd/csrc/*.hip $ROOT/augcl_amd/csrc/*.cpp $ROOT/augcl_amd/csrc/*.h $ROOT/augcl_amd/csrc/Makefile $T/
make -j8 -C $T EXTRA_FLAGS="-DAUGCL_TUNING $AUGCL_ABL" OUT=$ROOT/tools/libaugcl_tuning.so 2>&1 | grep -E "error|Error" || true
ls -la $ROOT/tools/libaugcl_tuning.so
