"""Data feeding for the runner: work-alikes of the reference's dataset classes (outside the hot path).

  * ``StochasticPairs``  -- eddata.stochastic_pair.StochasticPairs (external; used by the PennAction / DeepFashion yamls):
    a csv with at least ``character_id`` and ``relative_file_path_`` columns; example i = (image i, a random image j of
    the same character_id), both resized to ``spatial_size`` and scaled to [-1, 1]; ``data_avoid_identity`` excludes
    j = i when the character has more than one image; ``data_flip_h`` / ``data_flip_v`` flip BOTH views together.
  * ``AugmentedPair2``   -- cub/code/data/data.py:52-175 on top of it: adds ``view0_target`` (a copy of view0).  The
    albumentations appearance / shape augmentations (data_augment_appearance / data_augment_shape, both False in the
    shipped yaml) are not reproduced and raise when requested.

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
        self.prng = np.random.RandomState(config.get("data_seed", 1))

    def __len__(self):
        return len(self.labels["character_id"])

    def preprocess_image(self, path):
        from PIL import Image
        img = Image.open(path).convert("RGB").resize((self.size, self.size), Image.BILINEAR)
        return np.asarray(img, dtype=np.float32) / 127.5 - 1.0

    def pick_partner(self, i):
        choices = self.labels["choices"][i]
        if self.avoid_identity and len(choices) > 1:
            choices = [c for c in choices if c != i]
        return int(self.prng.choice(choices))

    def get_example(self, i):
        j = self.pick_partner(i)
        view0 = self.preprocess_image(self.labels["file_path_"][i])
        view1 = self.preprocess_image(self.labels["file_path_"][j])
        if self.flip_h and self.prng.rand() < 0.5:
            view0, view1 = view0[:, ::-1].copy(), view1[:, ::-1].copy()
        if self.flip_v and self.prng.rand() < 0.5:
            view0, view1 = view0[::-1].copy(), view1[::-1].copy()
        return {"view0": view0, "view1": view1}


class AugmentedPair2(StochasticPairs):
    n_images = 3

    def __init__(self, config):
        super(AugmentedPair2, self).__init__(config)
        if config.get("data_augment_appearance", False) or config.get("data_augment_shape", False):
            raise NotImplementedError("albumentations appearance / shape augmentation (cub/code/data/data.py:57-117) is not "
                                      "reproduced; the shipped yaml keeps both off")

    def get_example(self, i):
        ex = super(AugmentedPair2, self).get_example(i)
        ex["view0_target"] = ex["view0"].copy()         # data.py:164
        return ex


def batches(dataset, batch_size, shuffle=True, workers=8, seed=0, epochs=None):
    """Endless (or ``epochs``-bounded) iterator of float32 NHWC torch batches; the last ragged batch of an epoch is dropped
    (the model's batch size is static, model.py:320)."""
    rng = np.random.RandomState(seed)
    pool = cf.ThreadPoolExecutor(max_workers=workers)
    ep = 0
    while epochs is None or ep < epochs:
        order = rng.permutation(len(dataset)) if shuffle else np.arange(len(dataset))
        for b in range(len(order) // batch_size):
            idx = order[b * batch_size:(b + 1) * batch_size]
            exs = list(pool.map(dataset.get_example, idx))
            yield {k: torch.from_numpy(np.stack([e[k] for e in exs])) for k in exs[0]}
        ep += 1
