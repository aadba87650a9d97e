"""Generates tests/golden/full_cub128_step.npz: ONE training step of the FULL-WIDTH CUB yaml (128x128, n_parts 10, batch 2,
widths of train_cub_subset_tps.yaml:147-182, VGG19-topology trunk at its real widths with the seeded stand-in weights) from
the fp64 CPU oracle -- scalars, per-variable gradient norms, a few small summaries and the hard-mask argmax maps only (< 1 MB).
With `resize256` as argument: the same step with `perceptual_input: resize256` (edflow's original_scale reading) ->
full_cub128_step_resize256.npz; `resize256_crop224`: the third reading (one random 224x224 window of the resized images, its
corner an explicit noise input `crop_yx`) -> full_cub128_step_resize256_crop224.npz.  Inputs are regenerated from seeds by
the test (R.synthetic_views / R.synthetic_noise).

With `confident`: forward only, with the last convolution of the mask decoder scaled by CONFIDENT_SCALE so that the masks
are as confident as the reference's after training: its log reports mask0_kl 6.18 at step ~71k (cub/train/log.txt:20385-20442;
two maps, P = 25 -> 0.13 nats of entropy per pixel); scale 100 gives 0.124 nats here (random init: 2.2 of ln 10 = 2.30, with
the noise-free argmax decided by 1e-2 logit gaps) -> full_cub128_confident.npz.

With a config name of CONFIGS (the other BASELINE.json configs at their full widths, batch 1-2): the same one-step fixture for
    pennaction128  PennAction yaml (encoder1.coords True), 128x128, 10 parts, batch 2          (BASELINE config #4's model)
    deepfashion256 DeepFashion SB_model48c yaml, 256x256, 16 parts, batch 1                    (config #3's model and size)
    cub256p20      CUB yaml at 256x256, 20 parts, batch 1, one more decoder level, patch 64    (config #5's model and size)

    python tests/golden/make_golden_full.py [native|resize256|resize256_crop224|confident|pennaction128|deepfashion256|cub256p20]
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import configs, ref_model as R  # noqa: E402


N_PROJ = 16                       # seeded random projections <g, r_k> per variable (round 4: gradient DIRECTION, not only norms)


def projection_vectors(name, shape, k=N_PROJ):
    """r_0..r_{k-1} ~ N(0, 1) for one variable, fp32, seeded by the variable's name (the GPU test regenerates them)."""
    import zlib
    g = torch.Generator().manual_seed(zlib.crc32(("proj/" + name).encode()))
    return torch.randn((k,) + tuple(shape), generator=g, dtype=torch.float32)


def gradient_projections(grads, names):
    out = np.zeros((len(names), N_PROJ))
    for i, n in enumerate(names):
        r = projection_vectors(n, grads[n].shape).double()
        out[i] = (r.reshape(N_PROJ, -1) @ grads[n].double().reshape(-1)).numpy()
    return out


CONFIDENT_SCALE = 100.0
CONFIDENT_LAYER = "decoder_visualize/conv2d_9"


def confident_params(cfg, seed=0):
    params = R.init_params(cfg, seed)
    for suf in ("/V", "/b"):
        params[CONFIDENT_LAYER + suf] = params[CONFIDENT_LAYER + suf] * CONFIDENT_SCALE
    return params


def confident():
    cfg = configs.cub_config(n_parts=10, batch_size=2)
    params = confident_params(cfg)
    views, noise = R.synthetic_views(cfg), R.synthetic_noise(cfg)
    with torch.no_grad():
        o = R.forward(params, cfg, views, noise, dtype=torch.float64)
    out = {"hard0_argmax": R.hard_max(o["m0"]).argmax(-1).numpy().astype(np.uint8),
           "hard1_argmax": R.hard_max(o["m1"]).argmax(-1).numpy().astype(np.uint8),
           "out_parts_hard": o["out_parts_hard"].numpy().astype(np.uint8),
           "l0_mean_std": np.float64(float(o["l0_mean"].std())),
           "top2_gap_median": np.float64(float((o["l0_mean"].topk(2, dim=-1).values[..., 0] - o["l0_mean"].topk(2, dim=-1).values[..., 1]).median())),
           "generated_8x8": torch.nn.functional.avg_pool2d(o["generated"].permute(0, 3, 1, 2), 16).permute(0, 2, 3, 1).float().numpy()}
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "full_cub128_confident.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes; l0_mean std", out["l0_mean_std"], "median top-2 gap", out["top2_gap_median"])


def cub256_config(n_parts=20, batch_size=1):
    """The CUB yaml at 256x256: the mask decoder needs one more (16-wide) level to reach 256 from the 4x4 code, the rectangle
    patch scales with the image."""
    c = configs.cub_config(n_parts=n_parts, batch_size=batch_size, spatial_size=256)
    c["dv"] = dict(c["dv"], config=[16] + list(c["dv"]["config"]), upsample_config=["linear"] * 6)
    c["patch_size"] = 64
    return c


CONFIGS = {
    "pennaction128": lambda: configs.pennaction_config(n_parts=10, batch_size=2, spatial_size=128),
    "deepfashion256": lambda: configs.deepfashion_config(n_parts=16, batch_size=1, spatial_size=256),
    "cub256p20": lambda: cub256_config(20, 1),
}


def main(mode="native"):
    if mode == "confident":
        return confident()
    cfg = CONFIGS[mode]() if mode in CONFIGS else configs.cub_config(n_parts=10, batch_size=2)
    S = cfg["spatial_size"]
    params = R.init_params(cfg, 0)
    vp = R.vgg_params(7)
    views, noise = R.synthetic_views(cfg), R.synthetic_noise(cfg)
    t0 = time.time()
    o, L, log, new_state, grads = R.gradients(params, cfg, views, noise, R.initial_state(cfg), 0, vp, dtype=torch.float64,
                                              perceptual_mode="native" if mode in CONFIGS else mode, scheme="per_key")
    print("oracle step: {:.1f} s".format(time.time() - t0))
    out = {}
    for k, v in L.items():
        out["loss_" + k] = np.float64(float(v))
    for k, v in log.items():
        if not k.startswith("_"):
            out["log_" + k] = np.float64(float(v))
    names = sorted(grads)
    out["grad_names"] = np.array(names)
    out["grad_norms"] = np.array([float(grads[n].norm()) for n in names])
    out["grad_sums"] = np.array([float(grads[n].sum()) for n in names])
    out["grad_proj"] = gradient_projections(grads, names)
    out["hard0_argmax"] = R.hard_max(o["m0"]).argmax(-1).numpy().astype(np.uint8)
    out["hard1_argmax"] = R.hard_max(o["m1"]).argmax(-1).numpy().astype(np.uint8)
    out["out_parts_hard"] = o["out_parts_hard"].numpy().astype(np.uint8)
    if o.get("px0") is not None:        # (SB_model48c has no rectangle patches)
        out["px0"] = o["px0"].numpy().astype(np.int32)
        out["px1"] = o["px1"].numpy().astype(np.int32)
    gen = o["generated"].detach()
    out["generated_8x8"] = torch.nn.functional.avg_pool2d(gen.permute(0, 3, 1, 2), S // 8).permute(0, 2, 3, 1).float().numpy()
    out["generated_absmean"] = np.float64(float(gen.abs().mean()))
    out["l0_mean_16x16"] = torch.nn.functional.avg_pool2d(o["l0_mean"].detach().permute(0, 3, 1, 2), S // 16).permute(0, 2, 3, 1).float().numpy()
    out["l0_mean_norm"] = np.float64(float(o["l0_mean"].norm()))
    out["l1_mean_norm"] = np.float64(float(o["l1_mean"].norm()))
    out["feat_norm"] = np.float64(float(o["local_app_features1"].norm()))
    for k, v in new_state.items():
        out["state_" + k] = np.float64(v)
    out["crop_yx"] = noise["crop_yx"].numpy().astype(np.int32)
    fname = "full_{}_step.npz".format(mode) if mode in CONFIGS else (
        "full_cub128_step.npz" if mode == "native" else "full_cub128_step_{}.npz".format(mode))
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), fname)
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main(*(sys.argv[1:2]))
