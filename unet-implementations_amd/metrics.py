"""Drop-in `SegmentationMetrics` (reference: Our_UNet/utils/metrics.py:7-235).

Same accumulators (`intersections`, `unions`, `true_positives`, `false_positives`,
`false_negatives`, `total_pixels`, `correct_pixels`) and `compute_*` methods as the reference
class, for the 3-class pet masks.  The reference moves every prediction / target to the host
and loops over classes in numpy (`_update_single`, :59-91); here one launch of
`unet_argmax_dice_counts` produces the nine integer counts per batch
{intersection, predicted, labelled} x class (ignore pixels masked out) and they are summed in
an int64 device tensor - no host sync until a `compute_*` method (or an accumulator) is read.

`update_from_logits(logits, target)` takes the network output itself (argmax inside the
kernel); `update(pred, target)` takes class maps like the reference's method (device tensors;
routed through the same kernel as one-hot scores).  Targets hold {0, 1, 2, ignore_index}
(Our_UNet/src/train.py:300: everything else was mapped to 0 by the dataset).
"""
import numpy as np
import torch

from . import ops


class SegmentationMetrics:
    def __init__(self, num_classes: int = 3, ignore_index: int = 255):
        if num_classes != 3:
            raise NotImplementedError("the HIP count kernel handles exactly 3 classes")
        self.num_classes = num_classes
        self.ignore_index = ignore_index
        self._counts = None      # int64 [3, 3] on the device: {inter, predicted, labelled} per class
        self.reset()

    # reference: utils/metrics.py:25-34
    def reset(self):
        if self._counts is not None:
            self._counts.zero_()

    def _add(self, counts):
        if self._counts is None:
            self._counts = counts.clone()
        else:
            self._counts += counts

    def update_from_logits(self, logits, target):
        """logits fp32 [B, 3, H, W], target int64 [B, H, W], both on the device."""
        if not logits.is_cuda:
            raise RuntimeError("unet-implementations_amd.SegmentationMetrics runs on MI355X only "
                               "(no CPU fallback exists)")
        if target.dtype != torch.int64:
            target = target.long()
        _, counts = ops.argmax_dice_counts(logits.float(), target, self.ignore_index,
                                           want_preds=False)
        self._add(counts)

    # reference: utils/metrics.py:36-57
    def update(self, pred, target):
        """pred / target: class maps [H, W] or [B, H, W] (torch tensors on the device)."""
        if not torch.is_tensor(pred) or not torch.is_tensor(target) or not pred.is_cuda:
            raise RuntimeError("unet-implementations_amd.SegmentationMetrics takes device tensors "
                               "(no CPU fallback exists)")
        if pred.dim() == 2:
            pred, target = pred[None], target[None]
        p = pred.long()
        scores = torch.stack([(p == c).float() for c in range(self.num_classes)], dim=1)
        self.update_from_logits(scores.contiguous(), target)

    # ---- accumulators with the reference's names (float64 numpy arrays / integers) -------------
    def _host(self):
        if self._counts is None:
            return np.zeros((3, 3), dtype=np.int64)
        return self._counts.cpu().numpy()

    @property
    def intersections(self):
        return self._host()[:, 0].astype(np.float64)

    @property
    def unions(self):
        c = self._host()
        return (c[:, 1] + c[:, 2] - c[:, 0]).astype(np.float64)

    @property
    def true_positives(self):
        return self.intersections

    @property
    def false_positives(self):
        c = self._host()
        return (c[:, 1] - c[:, 0]).astype(np.float64)

    @property
    def false_negatives(self):
        c = self._host()
        return (c[:, 2] - c[:, 0]).astype(np.float64)

    @property
    def total_pixels(self):
        return int(self._host()[:, 2].sum())       # every valid pixel is labelled 0, 1 or 2

    @property
    def correct_pixels(self):
        return int(self._host()[:, 0].sum())

    # ---- reference: utils/metrics.py:93-235 ----------------------------------------------------
    def compute_pixel_accuracy(self) -> float:
        total = self.total_pixels
        return float(self.correct_pixels / total) if total > 0 else float("nan")

    def compute_iou(self, cls: int) -> float:
        u = self.unions[cls]
        return float(self.intersections[cls] / u) if u > 0 else float("nan")

    def _mean_valid(self, fn):
        vals = [fn(c) for c in range(self.num_classes)]
        vals = [v for v in vals if not np.isnan(v)]
        return float(sum(vals) / len(vals)) if vals else float("nan")

    def compute_mean_iou(self) -> float:
        return self._mean_valid(self.compute_iou)

    def compute_dice(self, cls: int) -> float:
        tp, fp, fn = self.true_positives[cls], self.false_positives[cls], self.false_negatives[cls]
        den = 2 * tp + fp + fn
        return float(2 * tp / den) if den > 0 else float("nan")

    def compute_mean_dice(self) -> float:
        return self._mean_valid(self.compute_dice)

    def compute_precision(self, cls: int) -> float:
        tp, fp = self.true_positives[cls], self.false_positives[cls]
        return float(tp / (tp + fp)) if (tp + fp) > 0 else float("nan")

    def compute_recall(self, cls: int) -> float:
        tp, fn = self.true_positives[cls], self.false_negatives[cls]
        return float(tp / (tp + fn)) if (tp + fn) > 0 else float("nan")

    def compute_f1_score(self, cls: int) -> float:
        return self.compute_dice(cls)

    def get_all_metrics(self):
        results = {"pixel_accuracy": self.compute_pixel_accuracy(),
                   "mean_iou": self.compute_mean_iou(),
                   "mean_dice": self.compute_mean_dice(), "class_metrics": {}}
        for cls in range(self.num_classes):
            results["class_metrics"][f"class_{cls}"] = {
                "iou": self.compute_iou(cls), "dice": self.compute_dice(cls),
                "precision": self.compute_precision(cls), "recall": self.compute_recall(cls),
                "f1_score": self.compute_f1_score(cls)}
        return results
