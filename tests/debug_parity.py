"""Per-variable gradient / loss parity of one fp32 step against the CPU oracle (debug aid).
Usage: python tests/debug_parity.py [cub|pennaction|deepfashion] [tiny|small]"""
import copy, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import upsparts_amd  # noqa
from upsparts_amd.model import TrainModel, Trainer
from oracle import configs, ref_model as R
variant = sys.argv[1] if len(sys.argv) > 1 else "cub"
size = sys.argv[2] if len(sys.argv) > 2 else "tiny"
cfg = copy.deepcopy(configs.tiny_config(variant=variant) if size == "tiny" else configs.small_config(variant=variant))
cfg["precision"] = "fp32"; cfg["vgg_widths"] = (8, 8, 16, 16, 16)
dev = torch.device("cuda:0")
params = R.init_params(cfg, 0); vp = R.vgg_params(7, widths=cfg["vgg_widths"])
model = TrainModel(cfg, device=dev, seed=0); tr = Trainer(cfg, None, model)
views, noise = R.synthetic_views(cfg), R.synthetic_noise(cfg)
o, Lo, log, _, grads = R.gradients(params, cfg, views, noise, R.initial_state(cfg), 0, vp, dtype=torch.float64)
losses = tr.train_step(views, noise)
for k in Lo:
    print("loss {:20s} oracle {:12.6f} hip {:12.6f}".format(k, float(Lo[k]), float(losses[k])))
logs = tr.fetch_logs()
for k in logs:
    if k in log and not k.startswith("loss_"):
        a, b = float(log[k]), logs[k]
        if abs(a - b) > 1e-3 * max(1e-6, abs(a)):
            print("LOG MISMATCH", k, a, b)
worst = {}
for n, g in grads.items():
    h = model.bank.grads[n].cpu().double()
    e = float((h - g).abs().max() / max(float(g.abs().max()), 1e-12))
    key = n.split("/")[0]
    if e > worst.get(key, ("", 0))[1]:
        worst[key] = (n, e)
for k, (n, e) in worst.items():
    print("grad {:20s} worst {:34s} rel {:.3e}".format(k, n, e))
