"""fp8 hand-off census of a few training steps: how many convolution launches ran on fp8 operands, how many read / wrote an fp8
copy (ops.Fp8.stats), and the step time.  Usage: UPS_F8_PRODUCER=1 python tools/f8_stats.py [batch]"""
import copy, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import upsparts_amd  # noqa
from upsparts_amd import configs, ops
from upsparts_amd.model import TrainModel, Trainer
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
cfg = copy.deepcopy(configs.cub_config(n_parts=10, batch_size=B)); cfg["precision"] = os.environ.get("UPS_PREC", "fp8")
dev = torch.device("cuda:0")
model = TrainModel(cfg, device=dev, seed=0); tr = Trainer(cfg, None, model)
g = torch.Generator().manual_seed(0)
views = {k: (torch.rand(B, 128, 128, 3, generator=g) * 2 - 1) for k in ("view0", "view1", "view0_target")}
for s in range(8):
    for k in ops.Fp8.stats: ops.Fp8.stats[k] = 0
    torch.cuda.synchronize(); t0 = time.time()
    losses = tr.train_step(views)
    torch.cuda.synchronize()
    print(s, "%.1f ms" % ((time.time() - t0) * 1e3), dict(ops.Fp8.stats), {k: round(float(v), 3) for k, v in list(losses.items())[:3]})
