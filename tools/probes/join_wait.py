"""How long the launching stream waits at the join with the critics' streams (UPS_JOIN_TIMING=1 makes the step record events around
that join).  Usage: UPS_JOIN_TIMING=1 [UPS_CRITIC_STREAMS=0] python3 tools/probes/join_wait.py"""
import os
import sys
import time

os.environ["UPS_JOIN_TIMING"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
import upsparts_amd  # noqa: E402,F401
from upsparts_amd import configs  # noqa: E402
from upsparts_amd.model import TrainModel, Trainer  # noqa: E402

dev = torch.device("cuda:0")
cfg = configs.cub_config(n_parts=10, batch_size=64)
cfg["precision"] = os.environ.get("PRECISION", "bf16")
model = TrainModel(cfg, device=dev, seed=0)
tr = Trainer(cfg, None, model)
g = torch.Generator().manual_seed(1234)
batch = {k: (torch.rand(64, 128, 128, 3, generator=g) * 2 - 1).to(dev) for k in model.inputs}
for _ in range(8):
    tr.train_step(batch)
torch.cuda.synchronize()
tr._join_events = []
tr._tail_events = []
t0 = time.perf_counter()
for _ in range(20):
    tr.train_step(batch)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 20
w = [a.elapsed_time(b) for a, b in tr._join_events]
t = [a.elapsed_time(b) for a, b in tr._tail_events]
print("{} streams={} step {:.2f} ms; wait at the critics' join: mean {:.3f} ms, max {:.3f} ms; wait for the weight-gradient streams at the end of "
      "the backward pass: mean {:.3f} ms, max {:.3f} ms".format(cfg["precision"], os.environ.get("UPS_CRITIC_STREAMS", "1"), dt * 1e3, sum(w) / len(w), max(w),
                                                                sum(t) / len(t), max(t)))
