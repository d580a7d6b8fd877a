import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, numpy as np
from test_oracle_golden import run_rigidity_case
from oracle import knn_oracle as KO
for name in ["all"]:
    l1, g1, gold = run_rigidity_case(name, "cuda")
    l2, g2, _ = run_rigidity_case(name, "cuda", KO.knn_points_batched, KO.knn_gather)
    l3, g3, _ = run_rigidity_case(name, "cpu", KO.knn_points_batched, KO.knn_gather)
    print("loss hip", float(l1), "gpu-oracle-ops", float(l2), "cpu", float(l3), "gold", float(gold[name + ".loss"]))
    for k in g1:
        if g1[k] is None: continue
        w = torch.from_numpy(gold[f"{name}.d_{k}"])
        s = w.abs().max()
        print(k, "hip-vs-gold", float((g1[k].cpu() - w).abs().max() / s), "gpuoracle-vs-gold", float((g2[k].cpu() - w).abs().max() / s),
              "hip-vs-gpuoracle", float((g1[k] - g2[k]).abs().max().cpu() / s))
