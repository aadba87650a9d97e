"""Which input-gradient launches of one training step take act' from sign bits (ops.SignBits), which still re-read the forward input?
Usage: python tools/probes/sign_bits_coverage.py [precision]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import upsparts_amd  # noqa
from upsparts_amd import configs, ops
from upsparts_amd.model import TrainModel, Trainer
dev = torch.device("cuda:0")
cfg = configs.cub_config(n_parts=10, batch_size=64)
cfg["precision"] = sys.argv[1] if len(sys.argv) > 1 else "bf16"
model = TrainModel(cfg, device=dev, seed=0)
tr = Trainer(cfg, None, model)
g = torch.Generator().manual_seed(1)
batch = {k: (torch.rand(64, 128, 128, 3, generator=g) * 2 - 1).to(dev) for k in model.inputs}
for _ in range(2):
    tr.train_step(batch)
ops.SignBits.stats = {}
tr.train_step(batch)
torch.cuda.synchronize()
rows = sorted(ops.SignBits.stats.items(), key=lambda kv: -(kv[0][1] * kv[0][2] * kv[0][3] * kv[0][4]))
print("{:44s} {:>5s} {:>9s} {:>5s} {:>3s} {:>6s} {:>8s}  bits".format("layer", "n", "h x w", "ld", "k", "stride", "MB of x"))
tot = [0.0, 0.0]
for (name, n, h, w, ld, k, st, has), cnt in rows:
    mb = n * h * w * ld * 2 / 1e6 * cnt
    tot[1 if has else 0] += mb
    print("{:44s} {:5d} {:4d}x{:<4d} {:5d} {:3d} {:6d} {:8.1f}  {}".format(name, n, h, w, ld, k, st, mb, "yes" if has else "NO") + (" (x{})".format(cnt) if cnt > 1 else ""))
print("forward-input bytes re-read for act': {:.0f} MB; replaced by sign bytes: {:.0f} MB".format(tot[0], tot[1]))
