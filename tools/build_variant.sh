#!/bin/bash
# Build a variant library for A/B runs: tools/build_variant.sh <name> [extra hipcc flags, e.g. -DEMEI_X=1]
# -> gpurun_abl_<name>.so (git-ignored, travels to the GPU box)
set -e
NAME=$1; shift
cd "$(dirname "$0")/../emei_amd/csrc"
make -s -j8 OBJDIR=$PWD/build_$NAME OUT=$PWD/../../gpurun_abl_$NAME.so EXTRA="$*" all
echo "built gpurun_abl_$NAME.so"
