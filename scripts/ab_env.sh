#!/bin/bash
# A/B of environment switches on ONE GPU box (boxes differ by +-5 %): bench runs alternate between the settings, the
# stage timers of every run are printed.   usage: scripts/ab_env.sh <reps> "VAR=1" "VAR=0" ... [-- bench args]
cd "$(dirname "$0")/.."
REPS=$1; shift
SETS=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do SETS+=("$1"); shift; done
[ "$1" == "--" ] && shift
mkdir -p gpurun_out/ab
FILES=()
for r in $(seq 1 $REPS); do
  i=0
  for s in "${SETS[@]}"; do
    f=gpurun_out/ab/env_${i}_$r.json
    env $s python bench.py --no-cpu-baseline "$@" > $f 2>/dev/null
    FILES+=($f); i=$((i+1))
  done
done
i=0
for s in "${SETS[@]}"; do echo "set $i: $s"; i=$((i+1)); done
python scripts/stage_ms.py "${FILES[@]}"
