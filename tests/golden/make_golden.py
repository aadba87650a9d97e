"""Generates tests/golden/tiny_step.npz (CUB yaml) and tiny_step_pennaction.npz (PennAction yaml: encoder1 with
CoordConv, mi_target 1.5) / tiny_step_deepfashion.npz (SB_model48c variant) from the CPU oracle (fp64) on the tiny config:
inputs (views, every noise tensor), forward outputs, all loss scalars, per-key gradients
(full tensors for small variables, norms for all), parameters after 1 and 2 TF-Adam steps and the
Lagrangian / EMA state.  The reference itself cannot be imported here (TensorFlow 1.14 / edflow absent),
so these vectors pin the RESTATEMENT (oracle/ref_model.py), not the reference binary.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import configs, ref_model as R  # noqa: E402

VGG_W = (8, 8, 16, 16, 16)
FULL = ("encoder_0/conv2d_0/V", "encoder_0/conv2d_3/b", "encoder_1/conv2d_0/V", "decoder_visualize/conv2d_1/V",
        "decoder_visualize/conv2d_6/V", "decoder_delta/conv2d_5/V", "decoder_delta/conv2d_7/V",
        "mi0_discriminator/conv2d_0/V", "mi_estimator/conv2d_11/b")


def main(variant="cub"):
    cfg = configs.tiny_config(variant=variant)
    params = R.init_params(cfg, 0)
    vp = R.vgg_params(7, widths=VGG_W)
    views, noise = R.synthetic_views(cfg), R.synthetic_noise(cfg)
    state, adam = R.initial_state(cfg), R.init_adam(params)
    out = {"in_" + k: v.numpy() for k, v in views.items()}
    out.update({"in_" + k: v.numpy() for k, v in noise.items()})
    p = params
    for step in range(2):
        p, adam, state, o, L, log, grads = R.train_step(p, adam, cfg, views, noise, state, step, vp, dtype=torch.float64)
        pre = "s{}_".format(step)
        for k in ("l0_mean", "l1_mean", "m0", "m1", "generated", "local_app_features1", "out_parts_soft"):
            out[pre + k] = o[k].detach().float().numpy()
        out[pre + "hard0"] = R.hard_max(o["m0"]).numpy().astype(np.uint8)
        out[pre + "hard1"] = R.hard_max(o["m1"]).numpy().astype(np.uint8)
        if o["px0"] is not None:
            out[pre + "px0"] = o["px0"].numpy().astype(np.int32)
            out[pre + "px1"] = o["px1"].numpy().astype(np.int32)
        out[pre + "out_parts_hard"] = o["out_parts_hard"].numpy().astype(np.int32)
        for k, v in L.items():
            out[pre + "loss_" + k] = np.float64(float(v))
        for k, v in log.items():
            if not k.startswith("_"):
                out[pre + "log_" + k] = np.float64(float(v))
        names = sorted(grads)
        out[pre + "grad_names"] = np.array(names)
        out[pre + "grad_norms"] = np.array([float(grads[n].norm()) for n in names])
        out[pre + "param_norms"] = np.array([float(p[n].double().norm()) for n in names])
        for n in FULL + (("d_single/conv2d_0/V", "d_alpha/conv2d_3/V", "d_pi/conv2d_6/V") if variant == "deepfashion" else ()):
            out[pre + "grad/" + n] = grads[n].float().numpy()
            out[pre + "param/" + n] = p[n].float().numpy()
        for k, v in state.items():
            out[pre + "state_" + k] = np.float64(v)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tiny_step.npz" if variant == "cub" else "tiny_step_{}.npz".format(variant))
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    for v in (sys.argv[1:] or ["cub", "pennaction", "deepfashion"]):
        main(v)
