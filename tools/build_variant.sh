#!/bin/bash
# Build a variant library for A/B runs: tools/build_variant.sh <name> [extra hipcc flags, e.g. -DEMEI_X=1]
# -> gpurun_abl_<name>.so (git-ignored, travels to the GPU box)
set -e
NAME=$1; shift
cd "$(dirname "$0")/../emei_amd/csrc"
EXTRA="$*"
# statistics builds: no atomic optimizer (emei_device.h: EMEI_STAT_*: it wraps every counter in an s_and_saveexec region)
case "$EXTRA" in *EMEI_NEWTON_STATS*) EXTRA="$EXTRA -mllvm -amdgpu-atomic-optimizer-strategy=None";; esac
make -s -j8 OBJDIR=$PWD/build_$NAME OUT=$PWD/../../gpurun_abl_$NAME.so EXTRA="$EXTRA" all
echo "built gpurun_abl_$NAME.so"
# every variant build is audited like the shipped library (tests/test_isa_guards.py): numbers taken from a miscompiled
# variant are worthless (round 2's Hopper RK4 statistics)
python3 ../../tools/isa_scan.py ../../gpurun_abl_$NAME.so
