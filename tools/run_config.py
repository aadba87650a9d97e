"""Run a few training steps of a named full-size config on one GPU and report ms/step (smoke + timing aid).
Usage: python tools/run_config.py cub|pennaction|deepfashion [size] [parts] [batch] [steps]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import upsparts_amd  # noqa
from upsparts_amd import configs
from upsparts_amd.model import TrainModel, Trainer
name = sys.argv[1]
S = int(sys.argv[2]) if len(sys.argv) > 2 else 128
P = int(sys.argv[3]) if len(sys.argv) > 3 else 10
B = int(sys.argv[4]) if len(sys.argv) > 4 else 64
steps = int(sys.argv[5]) if len(sys.argv) > 5 else 4
fn = {"cub": configs.cub_config, "pennaction": configs.pennaction_config, "deepfashion": configs.deepfashion_config}[name]
cfg = fn(P, B, S)
if name != "deepfashion" and S == 256:      # 4 -> 256 takes six doublings
    cfg["dv"]["config"] = [16] + cfg["dv"]["config"]; cfg["dv"]["upsample_config"] = ["linear"] * 6
    cfg["patch_size"] = 64
cfg["precision"] = "bf16"
cfg["hip_graph"] = os.environ.get("UPS_GRAPH", "0") == "1"
dev = torch.device("cuda:0")
model = TrainModel(cfg, device=dev, seed=0); tr = Trainer(cfg, None, model)
g = torch.Generator().manual_seed(1)
batch = {k: (torch.rand(B, S, S, 3, generator=g) * 2 - 1).to(dev) for k in ("view0", "view1", "view0_target")}
for _ in range(4):
    tr.train_step(batch)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(steps):
    losses = tr.train_step(batch)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
print("{} S={} P={} B={}: {:.2f} ms/step = {:.1f} img/s; peak mem {:.1f} GB; losses {}".format(
    name, S, P, B, dt * 1e3, B / dt, torch.cuda.max_memory_allocated() / 2**30,
    {k: round(float(v), 3) for k, v in losses.items()}))
