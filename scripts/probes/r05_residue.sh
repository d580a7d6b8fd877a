#!/bin/bash
# the FAIL / or32 cases of profiles/r04_parity_sweep.txt (aniso, radix + deterministic), one by one, both backward modes
for c in 58 153 223 9 65 275; do
  echo "== case $c radix+det"; RDG_SWEEP_PROFILE=aniso RDG_BIN_MODE=radix RDG_DETERMINISTIC=1 python3 scripts/parity_sweep.py 1 410000 $c 2>&1 | grep -v "^[0-9]* of" | cut -c1-900
  echo "== case $c default";   RDG_SWEEP_PROFILE=aniso python3 scripts/parity_sweep.py 1 410000 $c 2>&1 | grep -v "^[0-9]* of" | cut -c1-900
done
