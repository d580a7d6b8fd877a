#!/bin/bash
# A/B of library variants on the GPU box: one bench run per variant (and the regular build first), stage timers side by
# side.   usage: scripts/ab.sh [bench args ...] -- <variant name> ...
cd "$(dirname "$0")/.."
ARGS=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do ARGS+=("$1"); shift; done
shift
mkdir -p gpurun_out/ab
ulimit -c 0   # a variant with wrong results may abort: no core files on the box
python bench.py --no-cpu-baseline "${ARGS[@]}" > gpurun_out/ab/base.json 2>/dev/null
FILES=(gpurun_out/ab/base.json)
for v in "$@"; do
  RDG_LIB_PATH=$PWD/rodygs_amd/csrc/variants/$v.so python bench.py --no-cpu-baseline "${ARGS[@]}" > gpurun_out/ab/$v.json 2>/dev/null
  FILES+=(gpurun_out/ab/$v.json)
done
python bench.py --no-cpu-baseline "${ARGS[@]}" > gpurun_out/ab/base_again.json 2>/dev/null
python scripts/stage_ms.py "${FILES[@]}" gpurun_out/ab/base_again.json
