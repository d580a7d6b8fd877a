for v in "" loss_rotate "" loss_rotate; do
  if [ -z "$v" ]; then python scripts/probes/loss_probe.py | sed "s/^/regular      /"; else RDG_LIB_PATH=$PWD/rodygs_amd/csrc/variants/$v.so python scripts/probes/loss_probe.py | sed "s/^/$v /"; fi
done
