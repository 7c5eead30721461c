"""Segmentation metrics with the reference's arithmetic (reference metrics.py:60-85, 122-126).
The per-image confusion histogram itself is computed on the device (ops.argmax_hist); these are
the host-side finishers on the 19x19 int64 result."""
import numpy as np


def fast_hist(label_pred, label_true, num_classes):
    """reference metrics.py:122-126 (host version, for inputs that already live on the host)."""
    mask = (label_true >= 0) & (label_true < num_classes)
    return np.bincount(num_classes * label_true[mask].astype(int) + label_pred[mask],
                       minlength=num_classes ** 2).reshape(num_classes, num_classes)


def per_class_iou(hist):
    hist = np.asarray(hist, dtype=np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.diag(hist) / (hist.sum(axis=1) + hist.sum(axis=0) - np.diag(hist))


def miou_from_hist(hist):
    return float(np.nanmean(per_class_iou(hist)))


def evaluate_eval(hist, epoch=0, dataset_name=None, dataset=None):
    """reference metrics.py:60-85: prints and returns the summary numbers."""
    if hist is None:
        return {"mean_iu": 0}
    hist = np.asarray(hist, dtype=np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        acc = np.diag(hist).sum() / hist.sum()
        acc_cls = np.nanmean(np.diag(hist) / hist.sum(axis=1))
    iu = per_class_iou(hist)
    freq = hist.sum(axis=1) / hist.sum()
    mean_iu = float(np.nanmean(iu))
    fwavacc = float((freq[freq > 0] * iu[freq > 0]).sum())
    print("Dataset name: {}".format(dataset_name))
    for idx, v in enumerate(iu):
        print("{:2d}    {:5.1f}".format(idx, v * 100))
    print("mean {}".format(mean_iu))
    return {"acc": float(acc), "acc_cls": float(acc_cls), "mean_iu": mean_iu, "fwavacc": fwavacc, "iu": iu}
