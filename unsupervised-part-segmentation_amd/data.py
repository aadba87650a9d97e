"""Data feeding for the runner: work-alikes of the reference's dataset classes (outside the hot path).

  * ``StochasticPairs``  -- eddata.stochastic_pair.StochasticPairs (external; used by the PennAction / DeepFashion yamls):
    a csv with at least ``character_id`` and ``relative_file_path_`` columns; example i = (image i, a random image j of
    the same character_id), both resized to ``spatial_size`` and scaled to [-1, 1]; ``data_avoid_identity`` excludes
    j = i when the character has more than one image; ``data_flip_h`` / ``data_flip_v`` flip BOTH views together.
  * ``AugmentedPair2``   -- cub/code/data/data.py:52-175 on top of it: adds ``view0_target`` (a copy of view0).  The
    appearance / shape augmentations (data_augment_appearance / data_augment_shape, both False in the shipped yaml) are the
    numpy restatement of the albumentations pipelines in ``augment.py`` (same transforms, defaults and sync rules; the
    sample stream is not albumentations').

The exact resize filter / cropping of eddata's ``preprocess_image`` is not visible in the reference tree: bilinear resize
of the whole image is used (UNVERIFIED).  ``batches`` turns a dataset into the ``{"view0", "view1"[, "view0_target"]}``
float32 NHWC batches ``Trainer.iterate`` consumes, with a small thread pool for decoding.
"""
import concurrent.futures as cf
import os

import numpy as np
import torch


def add_choices(character_ids):
    """cub/code/data/data.py:31-50: for every row the indices of all rows with the same character_id."""
    cid = np.asarray(character_ids)
    by_cid = {c: np.nonzero(cid == c)[0] for c in np.unique(cid)}
    return [by_cid[c] for c in cid]


class StochasticPairs(object):
    n_images = 2

    def __init__(self, config):
        import pandas as pd
        self.config = config
        self.size = config.get("spatial_size", 256)
        self.root = config["data_root"]
        cols = config.get("data_csv_columns", ["character_id", "relative_file_path_"])
        header = 0 if config.get("data_csv_has_header", False) else None
        df = pd.read_csv(config["data_csv"], header=header, names=cols if header is None else None)
        if header == 0:
            df.columns = list(cols)[:len(df.columns)]
        self.labels = {c: df[c].tolist() for c in df.columns}
        self.labels["file_path_"] = [os.path.join(self.root, p) for p in self.labels["relative_file_path_"]]
        self.labels["choices"] = add_choices(self.labels["character_id"])
        self.avoid_identity = config.get("data_avoid_identity", True)
        self.flip_h, self.flip_v = config.get("data_flip_h", False), config.get("data_flip_v", False)
        self.seed = int(config.get("data_seed", 1))
        self.prng = np.random.RandomState(self.seed)          # (kept for callers that draw from the dataset-wide stream)
        self.draws = {}                                        # index -> number of examples drawn so far
        # optional ground-truth label maps for evaluation (eval_01.py:229-383 reads them as batch["gt_segmentation"]):
        # a csv column with the relative path of a label image (nearest-neighbour resized like denseposelib.resize_labels)
        self.gt_column = config.get("data_gt_segmentation_column")

    def __len__(self):
        return len(self.labels["character_id"])

    def preprocess_image(self, path):
        from PIL import Image
        img = Image.open(path).convert("RGB").resize((self.size, self.size), Image.BILINEAR)
        return np.asarray(img, dtype=np.float32) / 127.5 - 1.0

    def _rng(self, i):
        """Per-index, per-draw generator: examples are decoded by a thread pool, so partner and flip decisions must not
        depend on which thread reaches a shared stream first (reproducible pairings)."""
        k = self.draws.get(int(i), 0)
        self.draws[int(i)] = k + 1
        return np.random.RandomState([self.seed & 0x7fffffff, int(i), k])

    def pick_partner(self, i, rng=None):
        rng = rng if rng is not None else self._rng(i)
        choices = self.labels["choices"][i]
        if self.avoid_identity and len(choices) > 1:
            choices = [c for c in choices if c != i]
        return int(rng.choice(choices))

    def preprocess_labels(self, path):
        from PIL import Image
        return np.asarray(Image.open(path).resize((self.size, self.size), Image.NEAREST), dtype=np.int64)

    def get_example(self, i):
        rng = self._rng(i)
        j = self.pick_partner(i, rng)
        view0 = self.preprocess_image(self.labels["file_path_"][i])
        view1 = self.preprocess_image(self.labels["file_path_"][j])
        flip_h = self.flip_h and rng.rand() < 0.5
        flip_v = self.flip_v and rng.rand() < 0.5
        if flip_h:
            view0, view1 = view0[:, ::-1].copy(), view1[:, ::-1].copy()
        if flip_v:
            view0, view1 = view0[::-1].copy(), view1[::-1].copy()
        ex = {"view0": view0, "view1": view1}
        if self.gt_column and self.gt_column in self.labels:
            gt = self.preprocess_labels(os.path.join(self.root, self.labels[self.gt_column][i]))
            if flip_h:
                gt = gt[:, ::-1].copy()
            if flip_v:
                gt = gt[::-1].copy()
            ex["gt_segmentation"] = gt
        return ex


class AugmentedPair2(StochasticPairs):
    n_images = 3

    def __init__(self, config):
        super(AugmentedPair2, self).__init__(config)
        self.use_appearance_augmentation = config.get("data_augment_appearance", False)       # data.py:56-57
        self.use_shape_augmentation = config.get("data_augment_shape", False)

    def get_example(self, i):
        from . import augment
        ex = super(AugmentedPair2, self).get_example(i)
        view0, view1 = ex["view0"], ex["view1"]
        target = view0.copy()                           # data.py:164
        if self.use_appearance_augmentation or self.use_shape_augmentation:
            rng = np.random.RandomState([self.seed & 0x7fffffff, int(i), self.draws[int(i)], 7])
            if self.use_appearance_augmentation:        # data.py:167-169: view1 and the target share one realisation
                view0, = augment.stochastic_appearance_augmentation(rng, view0)
                view1, target = augment.stochastic_appearance_augmentation(rng, view1, target)
            if self.use_shape_augmentation:             # data.py:171-173: view0 and the target share one realisation
                view0, target = augment.stochastic_shape_augmentation(rng, view0, target)
                view1, = augment.stochastic_shape_augmentation(rng, view1)
        ex.update(view0=view0, view1=view1, view0_target=target)
        return ex


def batches(dataset, batch_size, shuffle=True, workers=8, seed=0, epochs=None, pad_last=False):
    """Endless (or ``epochs``-bounded) iterator of float32 NHWC torch batches.  The model's batch size is static
    (model.py:320): in training the ragged last batch of an epoch is dropped; with ``pad_last`` (evaluation) it is filled up
    by repeating its last example and carries ``"valid": n`` so that the caller keeps only the first n rows."""
    rng = np.random.RandomState(seed)
    pool = cf.ThreadPoolExecutor(max_workers=workers)
    ep = 0
    while epochs is None or ep < epochs:
        order = rng.permutation(len(dataset)) if shuffle else np.arange(len(dataset))
        nb = -(-len(order) // batch_size) if pad_last else len(order) // batch_size
        for b in range(nb):
            idx = list(order[b * batch_size:(b + 1) * batch_size])
            valid = len(idx)
            idx += [idx[-1]] * (batch_size - valid)
            exs = [dataset.get_example(i) for i in idx] if workers <= 1 else list(pool.map(dataset.get_example, idx))
            out = {k: torch.from_numpy(np.stack([e[k] for e in exs])) for k in exs[0]}
            if pad_last:
                out["valid"] = valid
            yield out
        ep += 1
