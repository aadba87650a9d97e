"""Soak run: N training steps at the benchmark shape, memory and step time sampled along the way (allocator growth, leaked side-stream
references, drifting step time).  Usage: python tools/soak.py [steps] [precision]"""
import copy, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import upsparts_amd  # noqa
from upsparts_amd import configs
from upsparts_amd.model import TrainModel, Trainer
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
cfg = copy.deepcopy(configs.cub_config(n_parts=10, batch_size=64)); cfg["precision"] = sys.argv[2] if len(sys.argv) > 2 else "bf16"
dev = torch.device("cuda:0")
model = TrainModel(cfg, device=dev, seed=0); tr = Trainer(cfg, None, model)
g = torch.Generator().manual_seed(0)
views = {k: (torch.rand(64, 128, 128, 3, generator=g) * 2 - 1).to(dev) for k in ("view0", "view1", "view0_target")}
t0 = time.time(); last = t0
for s in range(steps):
    losses = tr.train_step(views)
    if (s + 1) % 50 == 0:
        torch.cuda.synchronize()
        now = time.time()
        bad = [k for k, v in losses.items() if not torch.isfinite(torch.as_tensor(float(v)))]
        print("step %4d  %.1f ms/step  allocated %.2f GB  reserved %.2f GB  max %.2f GB  %s" % (
            s + 1, (now - last) / 50 * 1e3, torch.cuda.memory_allocated() / 2**30, torch.cuda.memory_reserved() / 2**30,
            torch.cuda.max_memory_allocated() / 2**30, "NON-FINITE " + ",".join(bad) if bad else "finite"), flush=True)
        last = now
