for v in "" loss_nt1 loss_nt3 loss_nt6 "" loss_nt1; do
  if [ -z "$v" ]; then python scripts/probes/loss_probe.py | sed "s/^/regular (nt2) /"; else RDG_LIB_PATH=$PWD/rodygs_amd/csrc/variants/$v.so python scripts/probes/loss_probe.py | sed "s/^/$v /"; fi
done
python -m pytest tests -m gpu -x -q -k "fused_photometric_loss" 2>&1 | tail -3
RDG_LIB_PATH=$PWD/rodygs_amd/csrc/variants/loss_nt3.so python -m pytest tests -m gpu -x -q -k "fused_photometric_loss" 2>&1 | tail -2
RDG_LIB_PATH=$PWD/rodygs_amd/csrc/variants/loss_nt6.so python -m pytest tests -m gpu -x -q -k "fused_photometric_loss" 2>&1 | tail -2
