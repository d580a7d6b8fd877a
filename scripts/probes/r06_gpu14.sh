python scripts/probes/loss_probe.py
python scripts/probes/loss_probe.py
python -m pytest tests -m gpu -x -q -k "loss or ssim or photometric or psnr" 2>&1 | tail -3
