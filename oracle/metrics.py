"""ORACLE (test infrastructure; only tests/ may import this): CPU restatement of the reference's SCD / CC
validation metrics in plain numpy / torch-CPU, each function citing the reference lines it follows.

Pinned by tests/golden/scd_metrics.npz (produced by oracle/gen_golden.py::run_scd_metrics from the REAL
/root/reference/model/utils.py functions) and, when the reference tree is present, by
tests/test_cpu.py::test_metric_oracle_equals_imported_reference."""
import math

import numpy as np


class AverageMeter:
    """reference model/utils.py:278-310: `avg = sum / count` where `sum` accumulates val * WEIGHT and `count` the COUNT
    argument (the first update stores count / val*weight as given)."""

    def __init__(self):
        self.initialized, self.val, self.avg, self.sum, self.count = False, None, None, None, None

    def update(self, val, count=1, weight=1):
        if not self.initialized:
            self.val, self.avg, self.count, self.sum, self.initialized = val, val, count, val * weight, True
        else:
            self.val = val
            self.count += count
            self.sum += val * weight
            self.avg = self.sum / self.count

    def value(self):
        return self.val

    def average(self):
        return self.avg


def accuracy(pred, label, ignore_zero=False):
    """reference model/utils.py:313-319."""
    valid = (label > 0) if ignore_zero else (label >= 0)
    acc_sum = (valid * (pred == label)).sum()
    valid_sum = valid.sum()
    return float(acc_sum) / (valid_sum + 1e-10), valid_sum


def fast_hist(a, b, n):
    """reference model/utils.py:321-323."""
    k = (a >= 0) & (a < n)
    return np.bincount(n * a[k].astype(int) + b[k], minlength=n ** 2).reshape(n, n)


def cal_kappa(hist):
    """reference model/utils.py:330-342."""
    if hist.sum() == 0:
        return 0
    po = np.diag(hist).sum() / hist.sum()
    pe = np.matmul(hist.sum(1), hist.sum(0).T) / hist.sum() ** 2
    return 0 if pe == 1 else (po - pe) / (1 - pe)


def scores_from_hist(hist):
    """reference model/utils.py:356-378 (everything after the histogram loop of SCDD_eval_all)."""
    hist = np.asarray(hist, dtype=np.float64)
    hist_fg = hist[1:, 1:]
    c2 = np.zeros((2, 2))
    c2[0][0] = hist[0][0]
    c2[0][1] = hist.sum(1)[0] - hist[0][0]
    c2[1][0] = hist.sum(0)[0] - hist[0][0]
    c2[1][1] = hist_fg.sum()
    hist_n0 = hist.copy()
    hist_n0[0][0] = 0
    kappa_n0 = cal_kappa(hist_n0)
    iu = np.diag(c2) / (c2.sum(1) + c2.sum(0) - np.diag(c2))
    iou_fg, iou_mean = iu[1], (iu[0] + iu[1]) / 2
    sek = (kappa_n0 * math.exp(iou_fg)) / math.e
    pixel_sum = hist.sum()
    change_pred_sum = pixel_sum - hist.sum(1)[0].sum()
    change_label_sum = pixel_sum - hist.sum(0)[0].sum()
    sc_tp = np.diag(hist[1:, 1:]).sum()
    precision, recall = sc_tp / change_pred_sum, sc_tp / change_label_sum
    # scipy.stats.hmean of two numbers (reference: stats.hmean([SC_Precision, SC_Recall]))
    fscd = 2.0 / (1.0 / precision + 1.0 / recall) if precision > 0 and recall > 0 else 0.0
    return fscd, iou_mean, sek


def SCDD_eval_all(preds, labels, num_class):
    """reference model/utils.py:345-378."""
    hist = np.zeros((num_class, num_class))
    for pred, label in zip(preds, labels):
        p, l = np.array(pred), np.array(label)
        assert set(np.unique(p)).issubset({0, 1, 2, 3, 4, 5, 6}), "unrecognized label number"
        assert p.shape == l.shape, "The size of prediction and target must be the same"
        hist += fast_hist(p.flatten(), l.flatten(), num_class)
    return scores_from_hist(hist)


def caption_accuracy(scores, targets, k):
    """reference model/utils.py:493-507: top-k accuracy in percent of the packed (rows, vocab) scores."""
    batch_size = targets.size(0)
    _, ind = scores.topk(k, 1, True, True)
    correct = ind.eq(targets.view(-1, 1).expand_as(ind))
    return correct.view(-1).float().sum().item() * (100.0 / batch_size)
