"""Per-step GPU time of the first N steps of the headline workload (events on the launching stream, read at the end): how long the
step takes to reach its steady state (allocator growth, lazy conversions, clocks).   usage: python3 tools/probes/step_times.py [N]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import upsparts_amd  # noqa
from upsparts_amd import configs
from upsparts_amd.model import TrainModel, Trainer
n = int(sys.argv[1]) if len(sys.argv) > 1 else 80
dev = torch.device("cuda:0")
cfg = configs.cub_config(n_parts=10, batch_size=64); cfg["precision"] = os.environ.get("PRECISION", "bf16")
model = TrainModel(cfg, device=dev, seed=0); tr = Trainer(cfg, None, model)
g = torch.Generator().manual_seed(1)
batch = {k: (torch.rand(64, 128, 128, 3, generator=g) * 2 - 1).to(dev) for k in ("view0", "view1", "view0_target")}
ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
ev[0].record()
for i in range(n):
    tr.train_step(batch)
    ev[i + 1].record()
torch.cuda.synchronize()
ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(n)]
for a in range(0, n, 10):
    print("steps %3d..%3d: " % (a, min(n, a + 10) - 1) + " ".join("%6.2f" % v for v in ms[a:a + 10]))
print("reserved GB", torch.cuda.memory_reserved() / 2**30)
