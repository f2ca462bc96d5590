#!/bin/bash
# tools/build_cycle_variant.sh <body tag, e.g. ch_f64 | hp_f64>  ->  gpurun_abl_cyc_<tag>.so for tools/cycle_profile.py:
# ONE body translation unit compiled with -DEMEI_CYCLE_PROFILE (as emei_amd/csrc/Makefile compiles it otherwise), every other
# object from the normal build (hipcc fails on some units with the marks: "Operand has incorrect register class").
set -e
TAG=$1
cd "$(dirname "$0")/../emei_amd/csrc"
make -s -j8 all > /dev/null
rm -rf build_cyc && mkdir build_cyc && cp build/*.o build_cyc/
make -s -B OBJDIR=$PWD/build_cyc EXTRA=-DEMEI_CYCLE_PROFILE $PWD/build_cyc/body_tu_$TAG.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../gpurun_abl_cyc_$TAG.so build_cyc/*.o
rm -rf build_cyc
echo "built gpurun_abl_cyc_$TAG.so"
python3 ../../tools/isa_scan.py ../../gpurun_abl_cyc_$TAG.so body_rollout
