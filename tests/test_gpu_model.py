"""GPU parity of the whole training step (forward, 7 per-key losses, per-key gradients, TF-Adam, state)
against the CPU oracle on the tiny golden config, through TrainModel / Trainer (the drop-in surface)."""
import copy

import pytest
import torch

from util import assert_close, rel_err

pytestmark = pytest.mark.gpu

VGG_W = (8, 8, 16, 16, 16)


def _setup(precision, dev, variant="cub", size="tiny"):
    import upsparts_amd  # noqa: F401
    from upsparts_amd.model import TrainModel, Trainer
    from oracle import ref_model as R, configs
    if size == "tiny_det":      # stochastic_l: False -> the mask logits are used without noise (model.py:420-421)
        cfg = configs.tiny_config(variant=variant)
        cfg["stochastic_l"] = False
    elif size == "tiny_tps":      # the shipped CUB yaml's default: in-graph TPS augmentation of all three views
        cfg = configs.tiny_config(variant=variant)
        cfg["use_tps"] = True
        cfg.setdefault("tps_parameters", {"scal": 0.8, "tps_scal": 0.15, "rot_scal": 0.2, "off_scal": 0.2, "scal_var": 0.1,
                                          "augm_scal": 1.0})
    elif size == "tiny_subpixel":   # nn.upsample(method="subpixel") -- the reference signatures' default (M:135, M:87): conv to 4 C + depth_to_space
        cfg = configs.tiny_config(variant=variant)
        cfg["dv"]["upsample_config"] = ["subpixel", "subpixel"]
        cfg["final_hour"]["upsample_method"] = "subpixel"
    elif size == "tiny_nearest":    # nearest-neighbour up-sampling in the hourglass and one level of the mask decoder
        cfg = configs.tiny_config(variant=variant)
        cfg["dv"]["upsample_config"] = ["nearest_neighbor", "linear"]
        cfg["final_hour"]["upsample_method"] = "nearest_neighbor"
    elif size == "tiny_elu":        # `activation: elu` (nn.py:747-758) in every sub-network: materialised activation (nets.Scope.conv2d)
        cfg = configs.tiny_config(variant=variant)
        for key in ("encoder0", "encoder1", "dv", "final_hour", "discriminator"):
            cfg[key] = dict(cfg[key], activation="elu")
    elif size == "tiny25":        # 25 parts (the shipped yamls' part count), odd batch: ragged lane groups / fallback conv paths
        cfg = configs.tiny_config(n_parts=25, batch_size=3, variant=variant)
    else:
        cfg = configs.tiny_config(variant=variant) if size == "tiny" else configs.small_config(variant=variant)
    cfg = copy.deepcopy(cfg)
    cfg["precision"] = precision
    cfg["vgg_widths"] = VGG_W
    params = R.init_params(cfg, 0)
    vp = R.vgg_params(7, widths=VGG_W)
    model = TrainModel(cfg, device=dev, seed=0)
    trainer = Trainer(cfg, None, model)
    # same initial weights by construction (same per-name seeding); verify instead of loading
    for n, p in params.items():
        assert torch.equal(model.variables[n].detach().cpu(), p), n
    for blk in trainer.vgg.layers:
        for lay in blk:
            assert torch.equal(lay.V.cpu(), vp[lay.name + "/V"]), lay.name
    views = R.synthetic_views(cfg)
    noise = R.synthetic_noise(cfg)
    return cfg, R, params, vp, model, trainer, views, noise


CASES = [("cub", "tiny"), ("pennaction", "tiny"), ("deepfashion", "tiny"), ("cub", "small"), ("deepfashion", "small"),
         ("cub", "tiny25"), ("deepfashion", "tiny25"), ("cub", "tiny_tps"), ("cub", "tiny_det"), ("cub", "tiny_subpixel"),
         ("cub", "tiny_nearest"), ("cub", "tiny_elu")]


@pytest.mark.parametrize("variant,size", CASES)
def test_train_step_fp32_matches_oracle(dev, variant, size):
    cfg, R, params, vp, model, trainer, views, noise = _setup("fp32", dev, variant, size)
    state = R.initial_state(cfg)
    adam = R.init_adam(params)
    p = params
    for step in range(2):
        p_new, adam, state_new, o, Lo, log, grads = R.train_step(p, adam, cfg, views, noise, state, step, vp,
                                                                 dtype=torch.float64, scheme="per_key")
        losses = trainer.train_step(views, noise)
        dbg = trainer._debug
        B = cfg["batch_size"]
        assert_close(dbg["l_mean"][:B], o["l0_mean"].float(), 1e-3, "l0_mean step {}".format(step))
        assert_close(dbg["l_mean"][B:], o["l1_mean"].float(), 1e-3, "l1_mean")
        assert_close(dbg["m"][:B], o["m0"].float(), 1e-3, "m0")
        hard_o = torch.cat([R.hard_max(o["m0"]), R.hard_max(o["m1"])], 0).float()
        agree = float((dbg["hard"].cpu() == hard_o).float().mean())
        assert agree >= 0.999, "hard masks differ: agreement {}".format(agree)
        if variant != "deepfashion":
            assert torch.equal(dbg["px"].cpu().long(), torch.cat([o["px0"], o["px1"]], 0)), "rectangle centres"
        assert_close(dbg["generated"][..., :3].float(), o["generated"].float(), 1e-3, "generated")
        for k in Lo:
            lo = float(Lo[k]); lh = float(losses[k])
            assert abs(lo - lh) <= 1e-3 * max(1.0, abs(lo)), "loss {} step {}: oracle {} hip {}".format(k, step, lo, lh)
        logs = trainer.fetch_logs()
        log_keys = ("prior_gmrf", "mask0_kl", "weakly_superv_loss_p", "variance_loss", "bottleneck_loss", "mi_constraint",
                    "independent_mi_constraint", "perceptual")
        log_keys += (("prior_mumford_sha", "sigma1_00", "sigma2_01") if variant == "deepfashion" else
                     ("patch_loss", "zr_mumford_sha", "z_area_cost", "z_mumford_sha_smoothness_cost",
                      "z_mumford_sha_contour_cost"))
        for k in log_keys:
            lo = float(log[k])
            assert abs(lo - logs[k]) <= 1e-3 * max(1e-6, abs(lo)) + 1e-9, "log {}: oracle {} hip {}".format(k, lo, logs[k])
        # three bars each: max-norm, RMS and element-wise on the channel's own scale (util.py).  The element-wise bar holds for the
        # FIRST step (identical weights on both sides); after an Adam update the two sides' weights differ by up to +-lr on elements
        # whose gradient is ~0 (Adam normalises it: see the parameter check below), which the second step's gradients inherit
        # element by element -- there the max-norm and RMS bars stand
        for n, g in grads.items():
            assert_close(model.bank.grads[n], g.float(), 2e-3, "gradient {} (step {})".format(n, step), elementwise=step == 0)
        # Adam normalises the gradient (m / (sqrt(v) + eps)), so an element whose gradient is ~0 can move by up to
        # +-lr_t with either sign: compare updates in units of the learning rate, not relative to |param|.
        lr = cfg["lr"]
        bad = tot = 0
        for n in p_new:
            d = (model.variables[n].detach().cpu() - p_new[n].float()).abs()
            assert float(d.max()) <= 2.1 * lr * (step + 1), "param {} after step {}: |delta| {}".format(n, step, float(d.max()))
            bad += int((d > 0.05 * lr).sum()); tot += d.numel()
        assert bad <= 0.002 * tot, "{} of {} parameters differ by more than 5% of lr after step {}".format(bad, tot, step)
        for k in ("loa", "lor", "avg_mim", "avg_independent_mim", "avg_acc0", "avg_loss_dis1"):
            assert abs(float(trainer.state[k]) - state_new[k]) <= 1e-3 * max(1e-3, abs(state_new[k])), k
        p, state = p_new, state_new


@pytest.mark.parametrize("variant,size", CASES)
def test_train_step_bf16_close_to_oracle(dev, variant, size):
    cfg, R, params, vp, model, trainer, views, noise = _setup("bf16", dev, variant, size)
    state = R.initial_state(cfg)
    o, Lo, log, _, grads = R.gradients(params, cfg, views, noise, state, 0, vp, dtype=torch.float64)
    losses = trainer.train_step(views, noise)
    dbg = trainer._debug
    hard_o = torch.cat([R.hard_max(o["m0"]), R.hard_max(o["m1"])], 0).float()
    inter = ((dbg["hard"].cpu() > 0) & (hard_o > 0)).sum(dim=(1, 2)).float()
    union = ((dbg["hard"].cpu() > 0) | (hard_o > 0)).sum(dim=(1, 2)).float().clamp(min=1)
    assert float((inter / union).mean()) >= 0.99, "part-mask IoU vs oracle"
    for k in Lo:
        lo = float(Lo[k]); lh = float(losses[k])
        assert abs(lo - lh) <= 5e-2 * max(1.0, abs(lo)), "loss {}: oracle {} hip(bf16) {}".format(k, lo, lh)


@pytest.mark.parametrize("mask_decoder", ["fp16", "same"])
def test_fp8_step_tracks_bf16(dev, mask_decoder):
    """precision: fp8 (BASELINE config #5): wide 3x3 / stride-1 convolutions on e4m3 / e5m2 MFMA operands with delayed scaling.
    Three training steps from the same initial weights as a bf16 run: every loss stays within 10 % and the sampled part masks
    agree on >= 95 % of the pixels; the fp8 layers are re-quantised after each optimizer step.  "fp16" = the mode's default since
    round 4 (the mask decoder's FORWARD stays fp16, its two input-gradient passes run on e5m2 copies: masks as in bf16 mode);
    "same" = the round-3 form (that forward on e4m3 copies too), which keeps the forward hand-off covered."""
    import upsparts_amd  # noqa: F401
    from upsparts_amd import ops
    from upsparts_amd.model import TrainModel, Trainer
    from oracle import ref_model as R, configs
    cfg = copy.deepcopy(configs.small_config())
    cfg["vgg_widths"] = (64, 64, 64, 64, 64)
    cfg["dv"].update({"config": [64, 64, 64, 64]})
    cfg["encoder0"].update({"config": [64, 64, 64, 64]})
    views, noise = R.synthetic_views(cfg), R.synthetic_noise(cfg)
    runs = {}
    for prec in ("bf16", "fp8"):
        c = copy.deepcopy(cfg)
        c["precision"] = prec
        c["mask_decoder_dtype"] = mask_decoder
        model = TrainModel(c, device=dev, seed=0)
        trainer = Trainer(c, None, model)
        out = []
        for _ in range(3):
            losses = trainer.train_step(views, noise)
            out.append({k: float(v) for k, v in losses.items()})
        runs[prec] = (out, trainer._debug["hard"].cpu().clone())
        if prec == "fp8":
            F = model.fp8
            assert F.count >= 4, "fp8 layers used: {}".format(F.count)
            assert F.stats["dgrad_copy_in"] > 0 and F.stats["dgrad_f8"] > 0, F.stats                  # gradient copies were handed on
            if mask_decoder == "same":
                assert F.stats["fwd_copy_in"] > 0, F.stats                                            # ... and forward copies
            scales = F.scale[:F.count].cpu()
            assert bool(torch.isfinite(scales).all()) and float(scales.min()) > 0
            assert all(l._cache[k]["version"] == ops.WeightVersion.value for l in F.layers for k in ("f8", "f8g") if k in l._cache)
    for step in range(3):
        for k, vb in runs["bf16"][0][step].items():
            vf = runs["fp8"][0][step][k]
            assert abs(vb - vf) <= 0.1 * max(1.0, abs(vb)), "step {} loss {}: bf16 {} fp8 {}".format(step, k, vb, vf)
    agree = float(((runs["bf16"][1] > 0) == (runs["fp8"][1] > 0)).float().mean())
    assert agree >= (0.99 if mask_decoder == "fp16" else 0.95), agree


def test_inference_outputs(dev):
    cfg, R, params, vp, model, trainer, views, noise = _setup("fp32", dev)
    out = model.forward(views, noise)
    o = R.forward(params, cfg, views, noise, dtype=torch.float64)
    assert set(out) >= {"generated", "m0_sample", "out_parts_hard", "out_parts_soft", "view0_mask00_rgb"}
    assert_close(out["out_parts_soft"], o["out_parts_soft"].float(), 1e-3, "out_parts_soft")
    assert float((out["out_parts_hard"].cpu() == o["out_parts_hard"]).float().mean()) >= 0.999
    assert_close(out["generated"], o["generated"].float(), 1e-3, "generated")


def test_runner_end_to_end_with_csv_dataset_and_checkpoint(dev, tmp_path):
    """`python -m upsparts_amd.runner -t <yaml>` (the `edflow -t` work-alike): reference import paths in the yaml, the csv pair
    dataset, LoggingHook lines at steps 0, 2, 4 (the reference's cadence and alphabetical key order), a checkpoint every ckpt_freq steps and a lazy restore from it."""
    import numpy as np
    import yaml
    from PIL import Image
    import upsparts_amd  # noqa: F401
    from upsparts_amd import runner
    from oracle import configs
    rng = np.random.RandomState(0)
    rows = ["character_id,relative_file_path_,foo,category"]
    for i in range(8):
        Image.fromarray(rng.randint(0, 255, (24, 24, 3), dtype=np.uint8)).save(str(tmp_path / "im{}.png".format(i)))
        rows.append("{},im{}.png,x,bird".format(i // 2, i))
    (tmp_path / "train.csv").write_text("\n".join(rows) + "\n")
    cfg = copy.deepcopy(configs.tiny_config())
    cfg.update({"dataset": "src.data.data.AugmentedPair2", "data_root": str(tmp_path), "data_csv": str(tmp_path / "train.csv"),
                "data_csv_columns": ["character_id", "relative_file_path_", "foo", "category"], "data_csv_has_header": True,
                "data_avoid_identity": False, "precision": "bf16", "vgg_widths": list(VGG_W), "use_tps": True,
                "ckpt_freq": 2, "log_freq": 250, "num_steps": 5})
    assert cfg["model"] == "nips19.SB_model48i.model.TrainModel"
    ypath = tmp_path / "train.yaml"
    ypath.write_text(yaml.safe_dump(cfg))
    root = tmp_path / "run"
    it = runner.main(["-t", str(ypath), "-p", str(root), "--strict-dataset"])
    assert it.global_step == 5
    log = (root / "train" / "log.txt").read_text()
    for s in (0, 2, 4):
        assert "[INFO] [LoggingHook]: global_step: {}\n".format(s) in log
    for s in (1, 3):                # cub/train/log.txt:221,279,337: 0, 2, 4, 8, ...
        assert "[INFO] [LoggingHook]: global_step: {}\n".format(s) not in log
    assert "loss_decoder_visualize" in log
    first = [ln.split("]: ")[1].split(":")[0] for ln in log.splitlines() if "[LoggingHook]" in ln][:12]
    assert first[0] == "adversarial_constraint" and first == sorted(first), first      # log.txt:204-215 (alphabetical)
    assert "[INFO] [LoggingHook]: project root: " in log
    ck = root / "train" / "checkpoints" / "model.ckpt-4"
    assert ck.exists() and (root / "train" / "checkpoints" / "model.ckpt-2").exists()
    assert (root / "train" / "checkpoints" / "model.ckpt-5").exists()          # final state at loop exit
    it2 = runner.main(["-t", str(ypath), "-p", str(tmp_path / "run2"), "-c", str(ck), "--num_steps", "6", "--strict-dataset"])
    assert it2.global_step == 6
    # restored weights: the step-4 checkpoint was written after the update of step 4, training resumed at global step 4
    saved = torch.load(str(ck), map_location="cpu")
    assert saved["global_step"] == 4 and saved["adam"]["encoder_0"]["t"] == 4      # file name == stored step == Adam t
    # `-e`: test-mode forward over one epoch of the csv dataset from the checkpoint, outputs pickled
    import pickle
    data = runner.main(["-e", str(ypath), "-p", str(tmp_path / "ev"), "-c", str(ck), "--strict-dataset"])
    with open(str(tmp_path / "ev" / "eval" / "4" / "model_outputs.p"), "rb") as f:
        disk = pickle.load(f)
    assert disk["outputs"]["out_parts_hard"].shape == (8, 16, 16) and disk["outputs"]["generated"].shape == (8, 16, 16, 3)
    assert np.array_equal(disk["outputs"]["out_parts_hard"], data["outputs"]["out_parts_hard"])
    assert disk["outputs"]["out_parts_hard"].max() < cfg["n_parts"]


def test_vgg_weights_key_loads_keras_style_kernels(dev, tmp_path):
    """`vgg_weights`: the perceptual trunk takes real (here: random, Keras-named) HWIO kernels instead of the stand-ins."""
    import numpy as np
    import upsparts_amd  # noqa: F401
    from upsparts_amd.model import TrainModel, Trainer
    from oracle import ref_model as R, configs
    cfg = copy.deepcopy(configs.tiny_config())
    cfg.update(precision="fp32", vgg_widths=VGG_W)
    vp = R.vgg_params(123, widths=VGG_W)                      # a different seed than the stand-ins (7)
    keras = {}
    for k, v in vp.items():
        blk = k.split("/")[1]
        keras[blk + ("/kernel:0" if k.endswith("/V") else "/bias:0")] = v.numpy()
    np.savez(str(tmp_path / "vgg.npz"), **keras)
    cfg["vgg_weights"] = str(tmp_path / "vgg.npz")
    model = TrainModel(cfg, device=dev, seed=0)
    tr = Trainer(cfg, None, model)
    assert tr.vgg_pretrained
    views, noise = R.synthetic_views(cfg), R.synthetic_noise(cfg)
    losses = tr.train_step(views, noise)
    _, Lo, _, _, _ = R.gradients(R.init_params(cfg, 0), cfg, views, noise, R.initial_state(cfg), 0, vp, dtype=torch.float64)
    for k in Lo:
        assert abs(float(Lo[k]) - float(losses[k])) <= 1e-3 * max(1.0, abs(float(Lo[k]))), k
    with pytest.raises(KeyError):
        np.savez(str(tmp_path / "bad.npz"), **{k: v for k, v in keras.items() if "block3" not in k})
        Trainer(dict(cfg, vgg_weights=str(tmp_path / "bad.npz")), None, model)


def test_tensorflow_checkpoint_import_export(dev, tmp_path):
    """`-c model.ckpt-<step>` with a TensorFlow-1.x tensor bundle (the reference's checkpoint format, model.py:592-602):
    variables matched by name, missing ones ignored, restore_exclude honoured, Adam slots and the step restored."""
    import upsparts_amd  # noqa: F401
    from upsparts_amd.model import TrainModel, Trainer
    from upsparts_amd import tfckpt
    from oracle import ref_model as R, configs
    cfg = copy.deepcopy(configs.tiny_config())
    cfg.update(precision="fp32", vgg_widths=VGG_W)
    model = TrainModel(cfg, device=dev, seed=0)
    tr = Trainer(cfg, None, model)
    views, noise = R.synthetic_views(cfg), R.synthetic_noise(cfg)
    for _ in range(3):
        tr.train_step(views, noise)
    prefix = str(tmp_path / "model.ckpt-3")
    tr.export_tf_checkpoint(prefix)
    hdr, entries = tfckpt.read_index(prefix + ".index")
    assert "encoder_0/conv2d_0/V" in entries and "encoder_0/conv2d_0/V/Adam_1" in entries and "global_step" in entries
    cfg2 = copy.deepcopy(cfg)
    cfg2["restore_exclude"] = ["mi_estimator"]
    model2 = TrainModel(cfg2, device=dev, seed=5)                    # different initial weights
    tr2 = Trainer(cfg2, None, model2)
    tr2.initialize(prefix)
    assert tr2.global_step == 3
    for n, p in model.variables.items():
        same = torch.equal(p.detach(), model2.variables[n].detach())
        assert same == ("mi_estimator" not in n), n
    assert torch.equal(model.bank.adam_v["decoder_delta/conv2d_2/V"], model2.bank.adam_v["decoder_delta/conv2d_2/V"])
    assert model2.bank.groups["encoder_0"]["t"] == 3 and model2.bank.groups["mi_estimator"]["t"] == 0
    # the restored trainer continues exactly like the original (same Adam state, same step)
    tr.train_step(views, noise)
    cfg3 = copy.deepcopy(cfg)
    model3 = TrainModel(cfg3, device=dev, seed=9)
    tr3 = Trainer(cfg3, None, model3)
    tr3.initialize(prefix)
    # the Lagrangian scalars travel as the reference's unnamed tf.Variables (`Variable`, `Variable_1`, ... by creation order)
    assert tr3.state_from_tf["lon"] == "Variable" and tr3.state_from_tf["lor"] == "Variable_9" and not tr3.not_restored_from_tf
    bundle = tfckpt.read_bundle(prefix)
    assert abs(float(bundle["Variable_9"]) - float(tr3.state["lor"])) < 1e-7 and float(bundle["Variable_4"]) != 1.0     # avg_loss_dis0 moved
    tr3.train_step(views, noise)
    for n in model.variables:              # every key continues exactly, the ones whose losses involve the Lagrangian state included
        assert torch.allclose(model.variables[n], model3.variables[n], atol=1e-7), n
    # a bundle without (all of) them: reported, not guessed
    t2 = {k: v for k, v in bundle.items() if k != "Variable_3"}
    tfckpt.write_bundle(str(tmp_path / "part.ckpt-3"), t2)
    tr4 = Trainer(cfg3, None, TrainModel(cfg3, device=dev, seed=9))
    tr4.initialize(str(tmp_path / "part.ckpt-3"))
    assert not tr4.state_from_tf and "Variable_9" in tr4.not_restored_from_tf


def test_fix_weights_and_pretrain_keys(dev):
    """yaml keys `fix_weights` (model.py:1062-1067: the listed optimizer keys are dropped) and `pretrain` (model.py:785-797:
    decoder_visualize is trained on the reconstruction loss only)."""
    import upsparts_amd  # noqa: F401
    from upsparts_amd.model import TrainModel, Trainer
    from oracle import ref_model as R, configs
    cfg = copy.deepcopy(configs.tiny_config())
    cfg.update(precision="fp32", vgg_widths=VGG_W, fix_weights=["encoder_1", "mi_estimator"], pretrain=True)
    model = TrainModel(cfg, device=dev, seed=0)
    tr = Trainer(cfg, None, model)
    before = {k: g["flat"]["p"].clone() for k, g in model.bank.groups.items()}
    views, noise = R.synthetic_views(cfg), R.synthetic_noise(cfg)
    losses = tr.train_step(views, noise)
    assert set(losses) == set(R.SUBMODULES) - {"encoder_1", "mi_estimator"}
    for k, g in model.bank.groups.items():
        changed = not torch.equal(before[k], g["flat"]["p"])
        assert changed == (k not in ("encoder_1", "mi_estimator")), k
    assert abs(float(losses["decoder_visualize"]) - float(losses["decoder_delta"])) < 1e-6      # pretrain: no prior terms
    params, vp = R.init_params(cfg, 0), R.vgg_params(7, widths=VGG_W)
    _, Lo, _, _, grads = R.gradients(params, cfg, views, noise, R.initial_state(cfg), 0, vp, dtype=torch.float64)
    assert set(Lo) == set(losses)
    for n, g in grads.items():
        if "decoder_visualize" in n:
            assert rel_err(model.bank.grads[n], g.float()) <= 2e-3, n


def test_hip_graph_replay_matches_eager(dev):
    """`hip_graph: True`: the captured step (three streams, ~2 000 launches) replayed with new inputs / noise / Adam step
    size must reproduce the eager trainer bit for bit, including a re-capture when a schedule constant changes."""
    import upsparts_amd  # noqa: F401
    from upsparts_amd.model import TrainModel, Trainer
    from oracle import ref_model as R, configs
    cfg = copy.deepcopy(configs.tiny_config())
    cfg.update(precision="bf16", vgg_widths=VGG_W)
    cfg["variance_weight"]["options"].update(start=4, step_size=1, stair_factor=2.0, clip_min=1.0, clip_max=64.0)   # moves at step 5
    runs = {}
    for mode in ("eager", "graph"):
        c = copy.deepcopy(cfg)
        c["hip_graph"] = mode == "graph"
        model = TrainModel(c, device=dev, seed=0)
        tr = Trainer(c, None, model)
        hist = []
        for step in range(7):
            views = R.synthetic_views(c, seed=100 + step)
            noise = R.synthetic_noise(c, seed=200 + step)
            losses = tr.train_step(views, noise)
            hist.append({k: float(v) for k, v in losses.items()})
        runs[mode] = (hist, {k: g["flat"]["p"].detach().cpu().clone() for k, g in model.bank.groups.items()},
                      {k: float(v) for k, v in tr.state.items()}, tr.global_step, tr._g)
    assert runs["graph"][4] is not None and runs["graph"][4]["graph"] is not None, "the graph was never captured"
    assert runs["eager"][3] == runs["graph"][3] == 7
    for a, b in zip(runs["eager"][0], runs["graph"][0]):
        assert a == b, (a, b)
    for k in runs["eager"][1]:
        assert torch.equal(runs["eager"][1][k], runs["graph"][1][k]), k
    assert runs["eager"][2] == runs["graph"][2]


@pytest.mark.parametrize("variant", ["cub", "deepfashion"])
def test_side_stream_schedules_are_bit_identical(variant, dev, monkeypatch):
    """The critics on three side streams (with SB_model48c's single-sample decoders behind them), the appearance encoder enqueued
    beside the pose encoder and the in-place single-split weight gradient change WHERE and WHEN launches run, not what they
    compute: four steps with every switch on must equal four steps on the one-stream schedule bit for bit (losses, every
    parameter, the state scalars)."""
    import upsparts_amd  # noqa: F401
    from upsparts_amd import model as M, ops
    from oracle import ref_model as R, configs
    cfg = copy.deepcopy(configs.tiny_config(variant=variant))
    cfg.update(precision="bf16", vgg_widths=VGG_W)
    runs = {}
    for mode in ("one_stream", "side_streams"):
        on = mode == "side_streams"
        monkeypatch.setattr(M, "CRITIC_STREAMS", on)
        monkeypatch.setattr(M, "EARLY_ALPHA", on)
        monkeypatch.setattr(ops.Streams, "enabled", on)
        c = copy.deepcopy(cfg)
        model = M.TrainModel(c, device=dev, seed=0)
        tr = M.Trainer(c, None, model)
        hist = []
        for step in range(4):
            views = R.synthetic_views(c, seed=100 + step)
            noise = R.synthetic_noise(c, seed=200 + step)
            losses = tr.train_step(views, noise)
            hist.append({k: float(v) for k, v in losses.items()})
        torch.cuda.synchronize()
        runs[mode] = (hist, {k: g["flat"]["p"].detach().cpu().clone() for k, g in model.bank.groups.items()},
                      {k: float(v) for k, v in tr.state.items()})
    for a, b in zip(runs["one_stream"][0], runs["side_streams"][0]):
        assert a == b, (a, b)
    for k in runs["one_stream"][1]:
        assert torch.equal(runs["one_stream"][1][k], runs["side_streams"][1][k]), k
    assert runs["one_stream"][2] == runs["side_streams"][2]


def test_grouped_critics_match_the_generic_path(dev, monkeypatch):
    """The critics' towers as grouped launches (ops.TowersFn, the default where the latent widths allow it) against the generic
    convolution path (UPS_TOWERS=0) inside the whole step: the yaml's latent widths (256 / 64) at tiny spatial sizes, one step from
    the same weights, views and noise.  The storage points are the same, so the two differ by fp32 summation order only: critic
    losses, the adversarial term's effect (encoder_0's gradient) and every critic gradient agree (cosine >= 0.9999, norms within
    1 %), the state update agrees; and the one-launch state
    update leave a grouped run BIT-identical."""
    import upsparts_amd  # noqa: F401
    from upsparts_amd import model as M, ops
    from oracle import ref_model as R, configs
    cfg = copy.deepcopy(configs.tiny_config(variant="cub"))
    cfg.update(precision="bf16", vgg_widths=VGG_W, z0_size=256, local_app_size=64)

    def run(steps=2):
        c = copy.deepcopy(cfg)
        model = M.TrainModel(c, device=dev, seed=0)
        tr = M.Trainer(c, None, model)
        out = []
        for step in range(steps):
            losses = tr.train_step(R.synthetic_views(c, seed=100 + step), R.synthetic_noise(c, seed=200 + step))
            torch.cuda.synchronize()
            out.append(({k: float(v) for k, v in losses.items()},
                        {k: g["flat"]["g"].detach().cpu().clone() for k, g in model.bank.groups.items()},
                        {k: g["flat"]["p"].detach().cpu().clone() for k, g in model.bank.groups.items()},
                        {k: float(v) for k, v in tr.state.items()}))
        return out

    calls = {"n": 0}
    orig = ops.TowersFn.forward

    def counted(ctx, towers, *tensors):
        calls["n"] += 1
        return orig(ctx, towers, *tensors)
    monkeypatch.setattr(ops.TowersFn, "forward", staticmethod(counted))
    grouped = run()
    assert calls["n"] == 2, "the grouped path did not run"
    monkeypatch.setattr(ops, "TOWERS", False)
    generic = run()
    assert calls["n"] == 2, "UPS_TOWERS=0 still took the grouped path"
    (la, ga, _, sa), (lb, gb, _, sb) = grouped[0], generic[0]
    for k in la:
        assert abs(la[k] - lb[k]) <= 2e-3 * max(1.0, abs(lb[k])), (k, la[k], lb[k])
    for k in ga:
        a, b = ga[k].double(), gb[k].double()
        cos = float((a * b).sum() / (a.norm() * b.norm() + 1e-30))
        bar = 0.9999 if "discriminator" in k or "estimator" in k or k == "encoder_0" else 0.999
        assert cos >= bar and abs(float(a.norm() / (b.norm() + 1e-30)) - 1.0) <= 1e-2, (k, cos, float(a.norm()), float(b.norm()))
    for k in sa:
        assert abs(sa[k] - sb[k]) <= 1e-4 * max(1.0, abs(sb[k])), (k, sa[k], sb[k])
    # host-order / stream-order variants and the torch form of the state update: the same launches, so the same bits
    monkeypatch.setattr(ops, "TOWERS", True)
    for name, val in (("STATE_KERNEL", False),):
        monkeypatch.setattr(M, name, val)
        other = run()
        monkeypatch.setattr(M, name, not val)
        for (l0, g0, p0, s0), (l1, g1, p1, s1) in zip(grouped, other):
            assert l0 == l1 and s0 == s1, name
            for k in p0:
                assert torch.equal(p0[k], p1[k]), (name, k)


def test_forty_steps_bf16_and_fp8_track_fp32(dev):
    """A short training run on a fixed synthetic batch (mid-size config with 64-wide decoders, 32x32): 40 optimizer steps in
    fp32, bf16 and fp8-forward mode from the same initial weights.  Every loss stays finite, the reconstruction loss falls by
    more than half, and the low-precision runs stay in the band of the fp32 trajectory (within 30 % at every fifth step): rounding
    does not blow up through Adam's state, the Lagrangian multipliers or the delayed fp8 scales."""
    import math
    import upsparts_amd  # noqa: F401
    from upsparts_amd import ops
    from upsparts_amd.model import TrainModel, Trainer
    from oracle import ref_model as R, configs
    cfg = copy.deepcopy(configs.small_config())
    cfg["vgg_widths"] = (64, 64, 64, 64, 64)
    cfg["dv"].update({"config": [64, 64, 64, 64]})
    cfg["encoder0"].update({"config": [64, 64, 64, 64]})
    cfg["lr"] = 1.0e-3
    views, noise = R.synthetic_views(cfg), R.synthetic_noise(cfg)
    hist = {}
    for prec in ("fp32", "bf16", "fp8"):
        c = copy.deepcopy(cfg)
        c["precision"] = prec
        model = TrainModel(c, device=dev, seed=0)
        trainer = Trainer(c, None, model)
        h = []
        slots = None
        for step in range(40):
            losses = trainer.train_step(views, noise)
            if step % 5 == 0 or step == 39:
                h.append({k: float(v) for k, v in losses.items()})
            if step == 3:
                slots = model.fp8.count
        assert model.fp8.count == slots, "fp8 scale slots keep being allocated: {} -> {}".format(slots, model.fp8.count)
        hist[prec] = h
    print("decoder_delta loss, steps 0, 5, ..., 35, 39:", {p: [round(r["decoder_delta"], 2) for r in h] for p, h in hist.items()})
    for prec, h in hist.items():
        for row in h:
            assert all(math.isfinite(v) for v in row.values()), (prec, row)
        assert h[-1]["decoder_delta"] < 0.5 * h[0]["decoder_delta"], (prec, h[0]["decoder_delta"], h[-1]["decoder_delta"])
    # (the dynamics at this step size amplify rounding differences from step to step: the runs share a band, not a curve --
    # measured: 366.7 -> 161.1 (fp32), -> 147.6 (bf16), -> 149.6 (fp8), largest excursion 19 % at step 10)
    for prec in ("bf16", "fp8"):
        for a, b in zip(hist["fp32"], hist[prec]):
            assert abs(a["decoder_delta"] - b["decoder_delta"]) <= 0.3 * abs(a["decoder_delta"]), (prec, a["decoder_delta"], b["decoder_delta"])


def test_reference_log_state_trajectory(dev):
    """What the ONE training log the reference ships (cub/train/log.txt:204-560; P = 25, B = 8, 128x128, random init) pins of the
    restated optimiser / state machinery, and what it does not (tools/pin_log.py prints the full table).

    Pinned (critics, EMAs and the two Lagrangian multipliers depend on the latent codes only, not on the data set or the
    perceptual trunk's weights): over 4 seeds of the restated trainer with edflow's betas (0.5, 0.9)
      * avg_loss_dis0 / avg_loss_dis1 (EMA 0.99 of the critic losses, model.py:28-35, 829-834) at global steps 8, 16, 32 bracket
        the logged 0.98079 / 0.96387 / 0.93222 and 0.98206 / 0.96510 / 0.93269 (window = the seeds' range widened by 0.006);
      * lor (model.py:921-930: lor += 0.05 * (independent_mim - mi_target), clipped) at steps 16 and 32 brackets -0.0643 / -0.1210;
      * every seed leaves loa >= 0 and the EMAs inside (0, 1].
    NOT pinned: TensorFlow's default betas (0.9, 0.999) produce the same windows (measured: tools/pin_log.py) -- this log does
    not discriminate the Adam betas; and the mask statistics (mask0_kl, weakly_superv_loss_p, prior_gmrf) stay at their
    random-init values through step 32 in the reference while they move within the first steps here (synthetic views, stand-in
    VGG weights) -- an open difference that is reported, not asserted."""
    import upsparts_amd  # noqa: F401
    from upsparts_amd import configs
    from upsparts_amd.model import TrainModel, Trainer
    ref = {"avg_loss_dis0": {8: 0.98079, 16: 0.96387, 32: 0.93222}, "avg_loss_dis1": {8: 0.98206, 16: 0.96510, 32: 0.93269},
           "lor": {16: -0.06427, 32: -0.12097}}
    got = {k: {s: [] for s in v} for k, v in ref.items()}
    for seed in range(4):
        cfg = copy.deepcopy(configs.cub_config(n_parts=25, batch_size=8))
        cfg.update({"precision": "bf16", "noise_seed": 4321 + seed})
        model = TrainModel(cfg, device=dev, seed=seed)
        tr = Trainer(cfg, None, model)
        assert (tr.beta1, tr.beta2) == (0.5, 0.9)
        for s in range(33):
            g = torch.Generator().manual_seed(1000 * seed + s)
            x = {}
            for k in ("view0", "view1", "view0_target"):      # smooth synthetic views (as the oracle's synthetic_views)
                t = torch.randn(8, 3, 16, 16, generator=g)
                x[k] = torch.tanh(1.5 * torch.nn.functional.interpolate(t, size=(128, 128), mode="bilinear", align_corners=True)
                                  ).permute(0, 2, 3, 1).contiguous().to(dev)
            tr.train_step(x)
            if s in (8, 16, 32):
                lg = tr.fetch_logs()
                for k in got:
                    if s in got[k]:
                        got[k][s].append(lg[k])
                assert lg["loa"] >= 0.0 and 0.0 < lg["avg_loss_dis0"] <= 1.0 and 0.0 < lg["avg_dis0_accuracy"] <= 1.0
    for k, per_step in ref.items():
        for s, want in per_step.items():
            lo, hi = min(got[k][s]), max(got[k][s])
            pad = 0.006 if k != "lor" else 0.06
            print("{} @ step {}: reference {:.5f}, restatement [{:.5f}, {:.5f}]".format(k, s, want, lo, hi))
            assert lo - pad <= want <= hi + pad, "{} at step {}: reference {} outside [{}, {}] +- {}".format(k, s, want, lo, hi, pad)


def test_reference_log_mask_statistics_conditional_pin(dev):
    """Round 4 (tools/pin_log.py, profiles/round4_pin_log_*.txt).  The reference's log holds the mask
    statistics at their random-init values through global step 32; the restated trainer moves them within two steps.  The sweep
    over what the log does not record -- one image for all three views (the logged csv setting) vs independent views, smooth /
    i.i.d. / 1-over-f textures, TPS on / off, the g vs g - 1 alignment of the logged state -- closes NONE of it (`mask0_kl` >= 2.5
    at steps 2-4 in all 24 cells), and neither does removing the reconstruction term from decoder_visualize's gradient (the priors
    alone drive it).  ONE thing does: decoder_visualize stepping 10-30x slower than Adam(lr) moves it.  This test states that
    finding as an executable fact -- with the mask decoder's step scaled by 0.03 on the logged run's data setting, EVERY quantity
    the log prints at global steps 2, 4, 8, 16, 32 lies in a window around the logged value:
      mask0_kl, weakly_superv_loss_p, patch_loss, variance_loss: inside the logged run's own range over steps 0..32 (+- its spread);
      prior_gmrf: inside [100, 360] (logged 115 ... 344);
      bottleneck_loss (encoder_0's KL: an independent check of the encoder's optimizer wiring): the seeds' MEAN within 20 % of the log
      at steps 2-16, every seed within 45 % (single seeds spread +-20 ... 35 % at steps 8 / 16).
    It is a CONDITIONAL pin: it says the rest of the restated graph / optimizer reproduces the log once the mask decoder is slow,
    not why the reference's was (gradient-magnitude intermittency on real images under Adam's second-moment estimate, or an
    edflow optimizer detail: the source is absent).  The product path keeps plain Adam(lr) on every key."""
    import importlib.util
    import os
    import upsparts_amd  # noqa: F401
    from upsparts_amd import configs, ops
    from upsparts_amd.model import TrainModel, Trainer
    spec = importlib.util.spec_from_file_location("pin_log", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                                          "tools", "pin_log.py"))
    pin = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(pin)
    steps = (2, 4, 8, 16, 32)
    idx = {s: pin.STEPS.index(s) for s in steps}
    windows = {"mask0_kl": (0.85, 1.15), "weakly_superv_loss_p": (2.60, 2.80), "patch_loss": (15050.0, 15350.0),
               "variance_loss": (16.3, 17.0), "prior_gmrf": (100.0, 360.0)}
    for k, (lo, hi) in windows.items():          # the windows contain the reference's own values at these steps
        assert all(lo <= pin.REF[k][idx[s]] <= hi for s in steps), k
    bott = {}
    for seed in range(3):
        cfg = copy.deepcopy(configs.cub_config(n_parts=25, batch_size=8, use_tps=True))
        # the trainer's diagnostic hook (Trainer.probe): decoder_visualize alone steps with 0.03 * lr
        cfg.update({"precision": "bf16", "noise_seed": 4321 + seed, "probe": {"lr_scale": {"decoder_visualize": 0.03}}})
        model = TrainModel(cfg, device=dev, seed=seed)
        tr = Trainer(cfg, None, model)
        for s in range(32):
            batch = {k: v.to(dev) for k, v in pin.make_views("same", "pink", 8, 128, 1000 * seed + s).items()}
            tr.train_step(batch)
            g_step = s + 1                       # the logged state at global step g carries g - 1 updates
            if g_step in steps:
                lg = tr.fetch_logs()
                for k, (lo, hi) in windows.items():
                    assert lo <= lg[k] <= hi, "seed {} global step {}: {} = {} outside [{}, {}] (logged {})".format(
                        seed, g_step, k, lg[k], lo, hi, pin.REF[k][idx[g_step]])
                if g_step <= 16:
                    bott.setdefault(g_step, []).append(lg["bottleneck_loss"])
    for g_step, vals in bott.items():            # encoder_0's KL: the seeds' mean within 20 % of the log, every seed within 45 %
        want = pin.REF["bottleneck_loss"][idx[g_step]]
        mean = sum(vals) / len(vals)
        print("bottleneck_loss @ global step {}: logged {:.4f}, restatement mean {:.4f} {}".format(g_step, want, mean, [round(v, 3) for v in vals]))
        assert abs(mean - want) <= 0.20 * want, (g_step, mean, want)
        assert all(abs(v - want) <= 0.45 * want for v in vals), (g_step, vals, want)


def test_resumed_run_continues_the_noise_stream(dev, tmp_path):
    """Round-4 advisor: a checkpoint carries the sampling-noise stream's offset (and the TPS / crop generator's state), so a run
    restored from it draws the noise the uninterrupted run draws next -- it used to restart at offset 0 and replay steps 0..N's
    eps_pi / eps_l.  Two steps, checkpoint, one more step; a fresh trainer restored from the file takes that third step with the
    SAME device-drawn noise: identical losses and parameters (bf16, TPS on: both generators matter)."""
    import upsparts_amd  # noqa: F401
    from upsparts_amd.model import TrainModel, Trainer
    from oracle import ref_model as R, configs
    cfg = copy.deepcopy(configs.tiny_config())
    cfg.update(precision="bf16", vgg_widths=VGG_W, use_tps=True)
    cfg.setdefault("tps_parameters", {"scal": 0.8, "tps_scal": 0.15, "rot_scal": 0.2, "off_scal": 0.2, "scal_var": 0.1, "augm_scal": 1.0})
    views = R.synthetic_views(cfg)
    model = TrainModel(cfg, device=dev, seed=0)
    tr = Trainer(cfg, None, model)
    for _ in range(2):
        tr.train_step(views)                          # noise drawn on the device by the trainer
    path = str(tmp_path / "model.ckpt-2")
    tr.save_checkpoint(path)
    assert tr._noise.offset > 0
    want = {k: float(v) for k, v in tr.train_step(views).items()}
    model2 = TrainModel(cfg, device=dev, seed=3)      # different initial weights: everything must come from the file
    tr2 = Trainer(cfg, None, model2)
    tr2.initialize(path)
    assert tr2.global_step == 2 and tr2._noise.offset > 0
    got = {k: float(v) for k, v in tr2.train_step(views).items()}
    for k in want:
        assert got[k] == want[k], "loss {} of the resumed step: {} vs {} (noise stream not continued?)".format(k, got[k], want[k])
    for n, p in model.variables.items():
        assert torch.equal(p.detach(), model2.variables[n].detach()), n


def test_step_that_fails_after_an_early_adam_poisons_the_trainer(dev, tmp_path):
    """Round-4 advisor (medium): with the per-key Adam queued behind each segment's weight gradients, a step that raises later in
    the backward pass leaves encoder_1 / decoder_delta one optimizer step ahead of the other keys.  The trainer must not carry on
    as if nothing happened: it joins its streams, re-converts the weight copies and refuses further steps and checkpoints."""
    import upsparts_amd  # noqa: F401
    from upsparts_amd import model as M
    from upsparts_amd.model import TrainModel, Trainer
    from oracle import ref_model as R, configs
    if not (M.EARLY_ADAM and M.LATE_JOIN):
        pytest.skip("early per-key Adam is switched off in this environment")
    cfg = copy.deepcopy(configs.tiny_config())
    cfg.update(precision="bf16", vgg_widths=VGG_W)
    views = R.synthetic_views(cfg)
    model = TrainModel(cfg, device=dev, seed=0)
    tr = Trainer(cfg, None, model)
    tr.train_step(views)
    t_before = {k: g["t"] for k, g in model.bank.groups.items()}

    def boom(c):
        raise RuntimeError("injected failure in the pose encoder's backward")
    tr._bwd_pose = boom
    with pytest.raises(RuntimeError, match="injected failure"):
        tr.train_step(views)
    ahead = [k for k, g in model.bank.groups.items() if g["t"] != t_before[k]]
    assert ahead, "the failing step was expected to have stepped some keys early (EARLY_ADAM)"
    assert tr._poisoned and all(k in tr._poisoned for k in ahead)
    torch.cuda.synchronize()
    with pytest.raises(RuntimeError, match="inconsistent"):
        tr.train_step(views)
    with pytest.raises(RuntimeError, match="refusing to write a checkpoint"):
        tr.save_checkpoint(str(tmp_path / "bad.ckpt-1"))
