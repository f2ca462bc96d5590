#!/bin/bash
# A/B two (or more) builds of libemei_hip.so ON ONE BOX, interleaved, reporting the rollout kernel time
# of bench.py's default workload per round (box-to-box clock differences are ~10 %, larger than most
# kernel edits).  Usage (inside gpurun): tools/ab.sh <rounds> libA.so libB.so [...]   [-- extra bench args]
ROUNDS=$1; shift
LIBS=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do LIBS+=("$1"); shift; done
[ "$1" == "--" ] && shift
for r in $(seq 1 $ROUNDS); do
  for l in "${LIBS[@]}"; do
    ms=$(EMEI_HIP_LIB=$PWD/$l python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python3 -c "import sys,json; print('%.4f' % json.loads(sys.stdin.read())['roofline']['kernel_ms'])")
    echo "round $r $l $ms"
  done
done | tee gpurun_out/ab.log
python3 - <<'PY'
import collections
d=collections.defaultdict(list)
for l in open('gpurun_out/ab.log'):
    _,r,lib,ms=l.split(); d[lib].append(float(ms))
for lib,v in d.items():
    v=sorted(v); print(f"{lib:32s} min {v[0]:.4f}  median {v[len(v)//2]:.4f}  max {v[-1]:.4f}")
PY
