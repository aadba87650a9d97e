"""Part-segmentation evaluation helpers: the IoU protocol of cub/code/eval/eval_iclr_01/eval_01.py:229-383.

The reference takes these from the un-vendored ``denseposelib`` (``compute_best_iou_remapping``, ``remap_parts``,
``compute_iou``); semantics re-derived from their call sites: every inferred part id is mapped to the ground-truth label it
overlaps best (IoU over the whole evaluation set), the remapped prediction is scored per label, and the reported "overall"
number averages the labels except ``background`` (cub_semantic_ours.ipynb:615).  NumPy only -- outside the hot path.
"""
import numpy as np


def compute_best_iou_remapping(inferred, gt):
    """inferred, gt: integer label maps of equal shape [N,H,W] -> {inferred id: gt label with the highest IoU}."""
    inferred, gt = np.asarray(inferred), np.asarray(gt)
    mapping = {}
    gl = np.unique(gt)
    for p in np.unique(inferred):
        m = inferred == p
        best, best_iou = int(gl[0]), -1.0
        for g in gl:
            t = gt == g
            union = np.logical_or(m, t).sum()
            iou = np.logical_and(m, t).sum() / union if union else 0.0
            if iou > best_iou:
                best, best_iou = int(g), float(iou)
        mapping[int(p)] = best
    return mapping


def remap_parts(labels, mapping):
    labels = np.asarray(labels)
    out = np.zeros_like(labels)
    for k, v in mapping.items():
        out[labels == k] = v
    return out


def compute_iou(pred, gt):
    """-> (iou per label, labels) over the labels present in gt."""
    pred, gt = np.asarray(pred), np.asarray(gt)
    labels = np.unique(gt)
    ious = []
    for g in labels:
        a, b = pred == g, gt == g
        union = np.logical_or(a, b).sum()
        ious.append(np.logical_and(a, b).sum() / union if union else 0.0)
    return np.asarray(ious, dtype=np.float64), labels


def evaluate_parts(out_parts_hard, gt_segmentation, background_label=0):
    """out_parts_hard [N,H,W] (TrainModel.outputs["out_parts_hard"]), gt [N,H,W] -> per-label IoU and the overall number.

    Protocol of eval_01.py:229-383: ONE remapping for the whole set (best IoU over all images), then the IoU of every label
    PER IMAGE (a row of part_ious.csv; labels absent from an image's ground truth are left out of that row), the per-label
    means over the images (mean_part_ios.csv) and their mean without the background = "overall".  ``pooled`` additionally
    reports the IoU over all pixels of the set at once."""
    out_parts_hard, gt_segmentation = np.asarray(out_parts_hard), np.asarray(gt_segmentation)
    mapping = compute_best_iou_remapping(out_parts_hard, gt_segmentation)
    pred = remap_parts(out_parts_hard, mapping)
    labels_all = np.unique(gt_segmentation)
    rows = []
    for i in range(len(pred)):
        ious, labels = compute_iou(pred[i], gt_segmentation[i])
        rows.append(dict(zip(labels.tolist(), ious.tolist())))
    per_label = {int(g): float(np.mean([r[g] for r in rows if g in r])) for g in labels_all.tolist()}
    fg = [v for g, v in per_label.items() if g != background_label]
    pooled, pl = compute_iou(pred, gt_segmentation)
    return {"mapping": mapping, "iou": per_label, "per_image": rows,
            "overall": float(np.mean(fg)) if fg else float("nan"),
            "pooled": dict(zip(pl.tolist(), pooled.tolist()))}


def write_eval_tables(res, root, global_step, part_names=None, background_label=0):
    """The files eval_01.py:229-383 leaves in the evaluation directory, from ``evaluate_parts``' result:
    ``part_ious.csv``     one row per image: global_step, batch_idx, then one IoU column per ground-truth part (-1.0 where the
                          image's ground truth lacks the part, eval_01.py:299-309, 371)
    ``mean_part_ios.csv`` (the reference's spelling) the column means over the rows with the -1 entries left out plus
                          ``overall`` = the mean of the part columns without ``background``, as a psql table (eval_01.py:373-383)
    ``best_remapping.yml`` inferred part id -> ground-truth label (eval_01.py:254-258).
    part_names: {ground-truth label: column name} (the sorted keys of the yaml's dp_semantic_remap_dict in the reference);
    default "background" for `background_label`, "part_<label>" otherwise."""
    import os
    import pandas as pd
    import yaml
    from tabulate import tabulate
    labels = sorted(res["iou"])
    names = {g: ("background" if g == background_label else "part_{}".format(g)) for g in labels}
    names.update(part_names or {})
    cols = [names[g] for g in labels]
    rows = []
    for i, r in enumerate(res["per_image"]):
        row = {"global_step": global_step, "batch_idx": i}
        row.update({names[g]: float(r.get(g, -1.0)) for g in labels})
        rows.append(row)
    df = pd.DataFrame(rows, columns=["global_step", "batch_idx"] + cols)
    os.makedirs(root, exist_ok=True)
    df.to_csv(os.path.join(root, "part_ious.csv"), index=False, header=True)
    df_mean = df[df != -1].mean().to_frame().transpose()
    df_mean["overall"] = df_mean[[c for c in cols if c != "background"]].mean(axis=1)
    with open(os.path.join(root, "mean_part_ios.csv"), "w") as f:
        print(tabulate(df_mean, headers="keys", tablefmt="psql", showindex="never"), file=f)
    with open(os.path.join(root, "best_remapping.yml"), "w") as f:
        yaml.dump({"best_remapping": {int(k): int(v) for k, v in res["mapping"].items()}}, f, default_flow_style=False)
    return df, df_mean
