"""Mirror of the reference's `utils/metric_tool.py` (ConfuseMatrixMeter / cm2score /
get_confuse_matrix, reference utils/metric_tool.py:49-128) with the per-iteration binarise +
confusion matrix done on the GPU (`c3d_confusion2`), so the training loop copies 4 integers
to the host instead of two full label maps (reference scripts/train_BCD.py:204-225)."""
import numpy as np
import torch

from .. import ops


def cm2F1(cm):
    tp, fn, fp = cm[1, 1], cm[1, 0], cm[0, 1]
    e = np.finfo(np.float32).eps
    recall = tp / (tp + fn + e)
    precision = tp / (tp + fp + e)
    return 2 * recall * precision / (recall + precision + e)


def cm2score(cm):
    tp, fn, fp, tn = cm[1, 1], cm[1, 0], cm[0, 1], cm[0, 0]
    e = np.finfo(np.float32).eps
    oa = (tp + tn) / (tp + fn + fp + tn + e)
    recall = tp / (tp + fn + e)
    precision = tp / (tp + fp + e)
    f1 = 2 * recall * precision / (recall + precision + e)
    iou = tp / (tp + fp + fn + e)
    pre = ((tp + fn) * (tp + fp) + (tn + fp) * (tn + fn)) / (tp + fp + tn + fn) ** 2
    kappa = (oa - pre) / (1 - pre)
    return {"Kappa": kappa, "IoU": iou, "F1": f1, "OA": oa, "recall": recall, "precision": precision, "Pre": pre}


def get_confuse_matrix(num_classes, label_gts, label_preds):
    """Host (numpy) version, same semantics as the reference."""
    cm = np.zeros((num_classes, num_classes))
    for gt, pr in zip(label_gts, label_preds):
        gt, pr = gt.flatten(), pr.flatten()
        mask = (gt >= 0) & (gt < num_classes)
        cm += np.bincount(num_classes * gt[mask].astype(int) + pr[mask], minlength=num_classes ** 2).reshape(
            num_classes, num_classes)
    return cm


class ConfuseMatrixMeter:
    def __init__(self, n_class=2):
        assert n_class == 2
        self.n_class = n_class
        self.sum = np.zeros((2, 2))
        self._dev = None

    def update_cm_device(self, prob, target):
        """prob: model output in [0,1]; target in {0,1}.  Threshold is strict `> 0.5`.
        Accumulates on the device; call `sync()` (or get_scores) to fold into `self.sum`."""
        if self._dev is None:
            self._dev = torch.zeros(4, dtype=torch.int64, device=prob.device)
        ops.confusion2(prob.detach().contiguous().float(), target.detach().contiguous().float(), self._dev)

    def sync(self):
        if self._dev is not None:
            self.sum += self._dev.cpu().numpy().astype(np.float64).reshape(2, 2)
            self._dev.zero_()

    def update_cm(self, pr, gt, weight=1):
        val = get_confuse_matrix(self.n_class, gt, pr)
        self.sum += val * weight
        return cm2F1(val)

    def get_scores(self):
        self.sync()
        return cm2score(self.sum)
