"""Configs of the product path (the yaml surface as dicts).

``cub_config`` restates cub/code/SB_model48i/train_cub_subset_tps.yaml (Y:19-194) as a dict,
with n_parts / batch_size / use_tps overridable (BASELINE configs use P=10, use_tps False);
``pennaction_config`` / ``deepfashion_config`` do the same for the other two shipped yamls.
(The reduced-width parity configs of the test suite live with the test infrastructure, not here.)
"""
import copy


def _stair(start, start_value, step_size, stair_factor, cmin, cmax):
    return {"var_type": "staircase", "options": {"start": start, "start_value": start_value,
            "step_size": step_size, "stair_factor": stair_factor, "clip_min": cmin, "clip_max": cmax}}


def cub_config(n_parts=10, batch_size=8, spatial_size=128, use_tps=False):
    c = {
        "model": "nips19.SB_model48i.model.TrainModel",
        "iterator": "nips19.SB_model48i.model.Trainer",
        "batch_size": batch_size, "spatial_size": spatial_size, "patch_size": 32,
        "lr": 2.0e-4, "lr_decay_begin": 1000000, "lr_decay_end": 1000001,
        "log_freq": 250, "ckpt_freq": 10000, "num_steps": 1000000,
        "kl_weight": {"start": 0, "end": 1, "start_value": 1.0, "end_value": 1.0},
        "mumford_sha_alpha": _stair(65000, 1.0, 10000, 10, 1.0e-2, 1.0e-2),
        "MI": {"mi_target": 0.1, "mi_slack": 0.05, "loa_init": 0.0, "loa_lr": 4.0, "loa_adaptive": True,
               "lor_init": 0.0, "lor_lr": 0.05, "lor_min": -3.0, "lor_max": 7.5, "lor_adaptive": True},
        "mumford_sha_lambda": _stair(65000, 1.0, 5000, 10, 1.0, 1.0),
        "variance_weight": _stair(31000, 50, 30000, 2, 1, 1),
        "prior_gmrf_weight": _stair(100000, 1.0e-2, 20000, 3.14, 1.0e-3, 1.0e-3),
        "patch_loss_weight": _stair(100000, 1.0e-2, 20000, 3.14, 1.0e-4, 1.0e-4),
        "prior_mumford_sha_weight": _stair(81000, 1.0, 20000, 10, 1.0e-5, 1.0e-5),
        "weakly_superv_loss_weight_p": {"var_type": "linear", "options": {
            "start": 60000, "end": 80000, "start_value": 1, "end_value": 1.0e3,
            "clip_min": 1.0, "clip_max": 1.0}},
        "n_parts": n_parts, "gamma": 10, "restore_exclude": [],
        "z0_size": 256, "local_app_size": 64,
        "test_mode": False, "adversarial_regularization": True, "variational_regularization": True,
        "entropy_func": "entropy",
        "encoder0": {"config": [32, 64, 128, 128, 256, 256], "extra_resnets": 4,
                     "activation": "leaky_relu", "coords": True},
        "final_hour": {"config": [32, 64], "extra_resnets": 0, "upsample_method": "linear",
                       "activation": "leaky_relu", "coords": False},
        "encoder1": {"config": [32, 64, 128, 128, 256, 256], "extra_resnets": 4,
                     "activation": "leaky_relu", "coords": False},
        "dv": {"config": [16, 32, 32, 128, 128, 256], "upsample_config": ["linear"] * 5,
               "activation": "leaky_relu", "coords": True},
        "discriminator": {"activation": "leaky_relu", "coords": False},
        "use_tps": use_tps,
        "tps_parameters": {"scal": 0.8, "tps_scal": 0.15, "rot_scal": 0.2, "off_scal": 0.2,
                           "scal_var": 0.1, "augm_scal": 1.0},
    }
    return c


def pennaction_config(n_parts=10, batch_size=8, spatial_size=128):
    """pennaction/code/SB_model48i/train_pennaction.yaml: the CUB yaml with ``encoder1.coords: True`` (line 163),
    ``MI.mi_target: 1.5`` (line 50) and ``use_tps: False`` (line 186); the model file differs from CUB's only in
    ``make_tps`` (unused with use_tps False)."""
    c = cub_config(n_parts, batch_size, spatial_size, use_tps=False)
    c["encoder1"]["coords"] = True
    c["MI"]["mi_target"] = 1.5
    return c


def deepfashion_config(n_parts=16, batch_size=8, spatial_size=128):
    """deepfashion/code/SB_model48c/train_deepfashion.yaml as a dict (n_parts / batch / size overridable; the yaml ships
    25 parts at 128x128, BASELINE config #3 asks for 16 parts at 256x256).  The model is the SB_model48c variant: two
    inputs, no rectangles / patch loss, Mumford-Shah prior on the logits, squared renormalised variance and the three
    single-sample decoders d_single / d_alpha / d_pi.  At 256x256 the mask decoders need one more upsampling level."""
    c = cub_config(n_parts, batch_size, spatial_size, use_tps=False)
    c.update({"model": "nips19.SB_model48c.model.TrainModel", "iterator": "nips19.SB_model48c.model.Trainer",
              "lr_decay_begin": 500000, "lr_decay_end": 500001, "ckpt_freq": 20000, "num_steps": 500000})
    c["MI"].update({"mi_target": 2.0, "mi_slack": 0.5, "lor_min": 0.0})
    c["variance_weight"] = _stair(41000, 1, 40000, 10, 1.0, 1.0e5)
    c["prior_mumford_sha_weight"] = _stair(100000, 1.0e-2, 20000, 3.14, 1.0e-6, 1.0e-6)
    c["weakly_superv_loss_weight_p"] = _stair(41000, 1, 1, 1.0e3, 1.0, 1.0e3)
    for k in ("patch_loss_weight", "gamma", "patch_size", "entropy_func", "use_tps", "tps_parameters",
              "adversarial_regularization", "variational_regularization"):
        c.pop(k, None)
    levels = 5 + max(0, (spatial_size // 128).bit_length() - 1)       # 4 -> 128 takes 5 doublings, 4 -> 256 six
    dvc = [16, 32, 32, 128, 128, 256]
    dvc = [16] * (levels + 1 - len(dvc)) + dvc
    c["dv"] = {"config": dvc, "upsample_config": ["linear"] * levels, "activation": "leaky_relu", "coords": True}
    c["d_single"] = copy.deepcopy(c["dv"])
    return c


def cub256_config(n_parts=20, batch_size=8):
    """BASELINE config #5's model: the CUB yaml at 256x256 -- the mask decoder needs one more level to reach 256 from its 4x4
    code (only the LENGTH of `dv.config` matters: linear up-sampling ignores the widths, nn.py:834-847) and the rectangle
    patch scales with the image."""
    c = cub_config(n_parts, batch_size, 256)
    c["dv"] = dict(c["dv"], config=[16] + list(c["dv"]["config"]), upsample_config=["linear"] * 6)
    c["patch_size"] = 64
    return c


# The BASELINE.json configurations as bench.py runs them: name -> (config builder, image size, parts, per-GPU batch, precision,
# algorithmic training GFLOP per image, note).  Config #2 is the headline.
BENCH_CONFIGS = {
    "cub128p10": (lambda b: cub_config(10, b), 128, 10, 64, "bf16", 268.9, "BASELINE config #2 (headline)"),
    "deepfashion256p16": (lambda b: deepfashion_config(16, b, 256), 256, 16, 32, "bf16", 1202.6,
                          "BASELINE config #3: DeepFashion SB_model48c yaml at 256x256, 16 parts, global batch 256 = 32 per GPU on 8 GPUs"),
    "pennaction128": (lambda b: pennaction_config(10, b, 128), 128, 10, 32, "bf16", 271.2,
                      "BASELINE config #4: PennAction yaml (encoder1.coords), 128x128, global batch 128 = 32 per GPU on 4 GPUs; the "
                      "'equivariance-via-flow loss' is not a reference feature (SURVEY 0.5): timed without it"),
    "cub256p20": (lambda b: cub256_config(20, b), 256, 20, 16, "fp8", 1287.8,
                  "BASELINE config #5: CUB yaml at 256x256, 20 parts, fp8 MFMA conv path; BASELINE names no batch: 16 per GPU"),
}
