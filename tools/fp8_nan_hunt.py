"""First non-finite step of an fp8 run at the benchmark shape: per-step losses, the fp8 scale slots, and (re-running the failing step
with anomaly checks) the first layer whose output is non-finite.  Usage: python tools/fp8_nan_hunt.py [steps] [batch]"""
import copy, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import upsparts_amd  # noqa
from upsparts_amd import configs, ops
from upsparts_amd.model import TrainModel, Trainer
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 80
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
cfg = copy.deepcopy(configs.cub_config(n_parts=10, batch_size=B)); cfg["precision"] = "fp8"
dev = torch.device("cuda:0")
model = TrainModel(cfg, device=dev, seed=0); tr = Trainer(cfg, None, model)
g = torch.Generator().manual_seed(0)
views = {k: (torch.rand(B, 128, 128, 3, generator=g) * 2 - 1).to(dev) for k in ("view0", "view1", "view0_target")}
for s in range(steps):
    losses = tr.train_step(views)
    vals = {k: float(v) for k, v in losses.items()}
    F = ops.Fp8
    sc = F.scale[:F.count]
    bad_sc = int((~torch.isfinite(sc)).sum()) + int((sc == 0).sum())
    pbad = [n for n, p in model.variables.items() if not bool(torch.isfinite(p).all())]
    fin = all(v == v and abs(v) != float("inf") for v in vals.values())
    if s < 3 or not fin or bad_sc or pbad or s % 10 == 0:
        print("step", s, {k: round(v, 4) for k, v in vals.items()}, "scales min/max %.3g %.3g" % (float(sc.min()), float(sc.max())),
              "bad scales", bad_sc, "non-finite params", pbad[:4], flush=True)
    if not fin or pbad:
        break
