"""How far do the state lines of the one training log the reference ships (cub/train/log.txt:204-680, P = 25, B = 8, 128x128, random
init) constrain the restated optimiser (edflow's Adam betas, update order)?  Runs the trainer for 128 steps over several seeds
with beta = (0.5, 0.9) (the restatement's reading of edflow's defaults) and with TensorFlow's defaults (0.9, 0.999) and prints, per
logged step, the reference value beside mean / min / max over the seeds.  (GPU box; data: synthetic smooth views, stand-in VGG.)
Usage: python tools/pin_log.py [seeds] [steps]"""
import copy
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import upsparts_amd  # noqa: E402,F401
from upsparts_amd import configs  # noqa: E402
from upsparts_amd.model import TrainModel, Trainer  # noqa: E402

STEPS = (0, 2, 4, 8, 16, 32, 64, 128)
REF = {   # cub/train/log.txt, LoggingHook blocks at global_step 0, 2, 4, 8, 16, 32, 64, 128
    "lor": [0.0, -0.029, 0.00093, -0.00722, -0.06427, -0.12097, -0.28761, -0.59405],
    "loa": [0.0, 0.0, 0.04859, 2.36224, 0.50684, 0.0, 5.15629, 0.0],
    "avg_loss_dis0": [1.0, 0.99684, 0.99069, 0.98079, 0.96387, 0.93222, 0.87944, 0.81457],
    "avg_loss_dis1": [1.0, 0.99756, 0.99267, 0.98206, 0.9651, 0.93269, 0.88384, 0.81444],
    "bottleneck_loss": [2.17496, 1.70786, 1.50157, 1.51438, 1.0195, 1.00262, 3.13894, 2.5898],
    "mask0_kl": [0.91682, 0.9717, 1.08057, 0.96673, 0.92913, 1.0403, 2.05073, 3.20033],
    "prior_gmrf": [136.26453, 184.19302, 235.29639, 147.57761, 114.99298, 343.88037, 1379.20105, 2379.23486],
    "variance_loss": [16.7656, 16.73101, 16.631, 16.77249, 16.73885, 16.41346, 14.47222, 12.81866],
    "patch_loss": [15294.75, 15275.75, 15286.0, 15273.5, 15263.5, 15101.125, 14278.875, 12935.625],
    "weakly_superv_loss_p": [2.75952, 2.73195, 2.6807, 2.73414, 2.75526, 2.69636, 2.22222, 1.61847],
    "loss_mi0_discriminator": [0.6605, 0.73373, 0.75796, 0.81907, 0.78083, 0.62451, 0.74687, 1.09582],
}


def smooth_views(B, S, seed):
    g = torch.Generator().manual_seed(seed)
    out = {}
    for k in ("view0", "view1", "view0_target"):
        x = torch.randn(B, 3, S // 8, S // 8, generator=g)
        x = torch.tanh(1.5 * torch.nn.functional.interpolate(x, size=(S, S), mode="bilinear", align_corners=True))
        out[k] = x.permute(0, 2, 3, 1).contiguous()
    return out


def run(betas, seed, steps, precision="bf16"):
    cfg = copy.deepcopy(configs.cub_config(n_parts=25, batch_size=8))
    cfg.update({"precision": precision, "beta1": betas[0], "beta2": betas[1], "noise_seed": 4321 + seed})
    dev = torch.device("cuda:0")
    model = TrainModel(cfg, device=dev, seed=seed)
    tr = Trainer(cfg, None, model)
    logs = {}
    for s in range(steps + 1):
        batch = {k: v.to(dev) for k, v in smooth_views(8, 128, 1000 * seed + s).items()}
        tr.train_step(batch)
        if s in STEPS:
            lg = tr.fetch_logs()
            logs[s] = {k: lg[k] for k in REF}
    return logs


def main():
    seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    res = {}
    for betas in ((0.5, 0.9), (0.9, 0.999)):
        res[betas] = [run(betas, sd, steps) for sd in range(seeds)]
    for k in REF:
        print("== " + k)
        for i, s in enumerate(STEPS):
            if s > steps:
                continue
            row = "  step {:4d}  ref {:12.5f}".format(s, REF[k][i])
            for betas in res:
                v = torch.tensor([r[s][k] for r in res[betas]], dtype=torch.float64)
                row += "   b={}: mean {:11.5f} [{:11.5f}, {:11.5f}]".format(betas, float(v.mean()), float(v.min()), float(v.max()))
            print(row)


if __name__ == "__main__":
    main()
