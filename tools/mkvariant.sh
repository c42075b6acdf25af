#!/bin/bash
# Variant build of libplonk_hip.so for tools/ab.sh: tools/mkvariant.sh <name> "<extra hipcc flags>" [sources to recompile, default msm.hip]
# (plonk_amd/lib_<name>/ receives the variant library and a copy of the host library; the other objects are taken from the tree's build)
name=$1; extra=$2; shift; shift
srcs="${@:-msm.hip}"
cd "$(dirname "$0")/../plonk_amd/csrc" || exit 1
mkdir -p ../build_$name ../lib_$name
for o in msm ntt poly api; do cp -p ../build/$o.o ../build_$name/$o.o; done
touch ../build_$name/*.o
for s in $srcs; do rm -f ../build_$name/${s%.hip}.o; done
make OBJDIR=../build_$name LIBDIR=../lib_$name EXTRA="$extra" -j4 2>&1 | grep -E "error|Error" 
cp -p ../lib/libplonk_host.so ../lib_$name/
ls -la ../lib_$name/
