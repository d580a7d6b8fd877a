"""Adam kernel timing at the bench size for the variant selected by RDG_ADAM_VAR / RDG_ADAM_BLOCKS."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from rodygs_amd.dp import FlatParams
from rodygs_amd.trainstep import fused_adam_
P = 1_000_000
spec = {"xyz": ((P, 3), 1e-3), "features": ((P, 16, 3), 2e-3), "scaling": ((P, 3), 1e-3), "rotation": ((P, 4), 1e-3),
        "opacity": ((P, 1), 1e-2), "motion_coeff": ((P, 1, 16), 1e-4)}
fp = FlatParams(spec, "cuda")
fp.flat_grad.normal_()
for _ in range(5): fused_adam_(fp, row_lr={"features": (48, 3, 1e-4)})
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ev[0].record()
for _ in range(50): fused_adam_(fp, row_lr={"features": (48, 3, 1e-4)})
ev[1].record(); torch.cuda.synchronize()
ms = ev[0].elapsed_time(ev[1]) / 50
print(f"VAR={os.environ.get('RDG_ADAM_VAR','0')} BLOCKS={os.environ.get('RDG_ADAM_BLOCKS','2048')}: {ms*1e3:.1f} us  {28*75*P/ms/1e9:.2f} TB/s")
