"""Stability check: train the full-size CUB config (bf16) for N steps on one fixed batch of smooth synthetic views and
print the loss trajectory (the reconstruction loss must fall, nothing may become non-finite)."""
import os, sys, math
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import upsparts_amd  # noqa
from upsparts_amd import configs
from upsparts_amd.model import TrainModel, Trainer
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
B, S = 16, 128
cfg = configs.cub_config(10, B, S)
cfg["precision"] = sys.argv[2] if len(sys.argv) > 2 else "bf16"
dev = torch.device("cuda:0")
model = TrainModel(cfg, device=dev, seed=0); tr = Trainer(cfg, None, model)
g = torch.Generator().manual_seed(5)
def smooth():
    x = torch.randn(B, 3, 16, 16, generator=g)
    return torch.tanh(1.5 * F.interpolate(x, size=(S, S), mode="bilinear", align_corners=True)).permute(0, 2, 3, 1).contiguous()
v0 = smooth(); v1 = smooth()
batch = {"view0": v0.to(dev), "view1": v1.to(dev), "view0_target": v0.to(dev)}
for s in range(steps):
    losses = tr.train_step(batch)
    if s % max(1, steps // 10) == 0 or s == steps - 1:
        vals = {k: float(v) for k, v in losses.items()}
        assert all(math.isfinite(v) for v in vals.values()), (s, vals)
        logs = tr.fetch_logs()
        print("step {:4d} rec {:9.3f} dv {:9.3f} mi0 {:.3f} loa {:.3f} lor {:.3f} kl {:.4f} var {:.3f}".format(
            s, vals["decoder_delta"], vals["decoder_visualize"], vals["mi0_discriminator"], logs["loa"], logs["lor"],
            logs["mask0_kl"], logs["variance_loss"]))
