#!/bin/bash
# Runs on the GPU box (via gpurun): collects the rocprofv3 stats + counter passes of every bench workload and condenses them
# into profiles/<round>_<workload>_* and profiles/traffic.json (gpurun merges gpurun_out/ back; profiles_out/ is copied by hand).
# Usage: tools/profile_all.sh <round> [workload ...]
set -e
RND=${1:-r02}; shift || true
WL=${@:-cartpole_swingup cartpole_balancing invpend invpend_balancing dpend cheetah cheetah_sweep1 hopper hopper_sweep1}
mkdir -p gpurun_out/profiles_$RND
for w in $WL; do
  case $w in
    cartpole_swingup)   ARGS="--workload cartpole_swingup";   PAT=pend_rollout_staged; N=65536;  T=1000; B=22;  ;;
    cartpole_balancing) ARGS="--workload cartpole_balancing"; PAT=pend_rollout_staged; N=65536;  T=500;  B=22;  ;;
    invpend)            ARGS="--workload invpend";            PAT=pend_rollout_staged; N=262144; T=250;  B=25;  ;;
    invpend_balancing)  ARGS="--workload invpend_balancing";  PAT=pend_rollout_staged; N=262144; T=250;  B=25;  ;;
    dpend)              ARGS="--workload dpend";              PAT=body_rollout;        N=262144; T=100;  B=33;  ;;
    cheetah)            ARGS="--workload cheetah";            PAT=body_rollout;        N=131072; T=100;  B=101; ;;
    cheetah_sweep1)     ARGS="--workload cheetah --solver sweep1"; PAT=body_rollout;   N=131072; T=100;  B=101; ;;
    hopper)             ARGS="--workload hopper";             PAT=body_rollout;        N=131072; T=100;  B=65;  ;;
    hopper_sweep1)      ARGS="--workload hopper --solver sweep1"; PAT=body_rollout;    N=131072; T=100;  B=65;  ;;
  esac
  echo "=== $w"
  bash tools/collect_profiles.sh ${RND}_$w $ARGS > gpurun_out/collect_${RND}_$w.log 2>&1
  python3 tools/summarize_profiles.py gpurun_out/prof_${RND}_$w $RND $w $PAT $N $T $B $((N / 64)) > gpurun_out/profiles_$RND/$w.summary.txt 2>&1 || tail -5 gpurun_out/profiles_$RND/$w.summary.txt
  tail -4 gpurun_out/profiles_$RND/$w.summary.txt
done
cp profiles/${RND}_* profiles/traffic.json gpurun_out/profiles_$RND/ 2>/dev/null || true
