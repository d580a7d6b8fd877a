#!/bin/bash
# Ablation / tuning builds of the compositing kernels on the GPU box: rebuilds rdg_render.o with the given macro sets
# (one argument = one variant) and prints the bench's stage timers.  Results of RDG_ABL_* builds are WRONG by
# construction; only the timing is read.   usage: scripts/render_ablation.sh ["-DA" "-DB -DC" ...]
cd "$(dirname "$0")/.."
if [ $# -eq 0 ]; then set -- "-DRDG_ABL_NOATOMIC" "-DRDG_ABL_NOATOMIC -DRDG_ABL_NOPARK" "-DRDG_ABL_NODPP" "-DRDG_ABL_PREONLY"; fi
for v in "" "$@"; do
  rm -f rodygs_amd/csrc/rdg_render.o
  make -C rodygs_amd/csrc EXTRA="$v" > /dev/null 2>&1 || { echo "build failed: $v"; continue; }
  python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('variant [$v]', round(d['value'],1), {k:round(x,4) for k,x in d['stage_ms'].items() if 'render' in k})"
done
rm -f rodygs_amd/csrc/rdg_render.o; make -C rodygs_amd/csrc > /dev/null 2>&1
