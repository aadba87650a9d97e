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
