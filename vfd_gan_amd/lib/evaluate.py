"""Scores of the evaluation sweep.  Same surface as the reference's lib/evaluate.py:14-91 —
``evaluate(labels, scores, best, iter, saveto, metric)`` with metric in {'roc', 'auprc', 'pr', 'f1_score'} plus the three
helpers it dispatches to — and the same artefacts when a score beats `best`: a two-column curve file ``ROC_%03d`` /
``PR_%03d`` in `saveto` and, if matplotlib can be imported, the matching PNG.  Written table-driven here: one curve
description per metric, one routine that stores a curve.

Host code, as in the reference (its test sweeps flatten every batch to numpy first, models/mygannet.py:439-440): the
sweep's tensors are reduced to two flat arrays once per sweep; nothing here is on the training hot path.
"""
import csv
import os

from scipy.interpolate import interp1d
from scipy.optimize import brentq
from sklearn import metrics as skm

F1_THRESHOLD = 0.20      # reference :24

# how each curve is drawn when it is kept: axis names, title, legend corner
_CURVES = {
    "ROC": dict(x="False Positive Rate", y="True Positive Rate", title="Receiver operating characteristic", legend="lower right",
                style=dict(color="darkorange", lw=2)),
    "PR": dict(x="Recall", y="Precision", title="Precision-Recall Curve", legend="lower right", style={}),
}


def _pyplot():
    """matplotlib is optional (the curve file is always written); headless backend."""
    try:
        import matplotlib
        matplotlib.use("Agg")
        from matplotlib import pyplot
        return pyplot
    except Exception:  # noqa: BLE001
        return None


def _keep_curve(kind, xs, ys, area, index, folder, marker=None):
    """Store curve `kind` number `index` under `folder`: the (x, y) pairs as CSV rows and, when possible, the figure with
    the anti-diagonal, the unit axes and an optional marked point (x, y, legend suffix)."""
    stem = os.path.join(folder, "%s_%03d" % (kind, index))
    look = _CURVES[kind]
    plt = _pyplot()
    if plt is not None:
        label = "(AUC = %0.2f)" % area if marker is None else "(AUC = %0.2f, EER = %0.2f)" % (area, marker[0])
        fig = plt.figure()
        plt.plot(xs, ys, label=label, **look["style"])
        if marker is not None:
            plt.plot([marker[0]], [marker[1]], marker="o", markersize=5, color="navy")
        plt.plot([0, 1], [1, 0], color="navy", lw=1, linestyle=":")
        plt.xlim([0.0, 1.0])
        plt.ylim([0.0, 1.05])
        plt.xlabel(look["x"])
        plt.ylabel(look["y"])
        plt.title(look["title"])
        plt.legend(loc=look["legend"])
        plt.savefig(stem + ".png")
        plt.close(fig)
    with open(stem, "w", newline="") as fh:
        csv.writer(fh).writerows(zip(xs, ys))


def _beats(value, best, folder):
    return best is not None and folder is not None and value > best


def roc(labels, scores, best, iter, saveto=None):
    """Area under the ROC curve; a new best is kept with its equal-error-rate point (reference :31-65)."""
    fpr, tpr, _ = skm.roc_curve(labels, scores)
    area = skm.auc(fpr, tpr)
    if _beats(area, best, saveto):
        eer = brentq(lambda t: 1.0 - t - interp1d(fpr, tpr)(t), 0.0, 1.0)      # where FPR == 1 - TPR
        _keep_curve("ROC", fpr, tpr, area, iter, saveto, marker=(eer, 1.0 - eer))
    return area


def auprc(labels, scores):
    """Average precision (reference :67-69)."""
    return skm.average_precision_score(labels, scores)


def pr(labels, scores, best, iter, saveto=None):
    """Area under the precision-recall curve; a new best is kept (reference :72-91)."""
    precision, recall, _ = skm.precision_recall_curve(labels, scores)
    area = skm.auc(recall, precision)
    if _beats(area, best, saveto):
        _keep_curve("PR", recall, precision, area, iter, saveto)
    return area


def _f1_at_threshold(labels, scores):
    # binarises `scores` IN PLACE, as the reference does (:24-27): callers ask for this metric last
    hot = scores >= F1_THRESHOLD
    scores[hot] = 1
    scores[~hot] = 0
    return skm.f1_score(labels, scores)


def evaluate(labels, scores, best=None, iter=None, saveto=None, metric=None):
    if metric == "roc":
        return roc(labels, scores, best, iter, saveto)
    if metric == "pr":
        return pr(labels, scores, best, iter, saveto)
    if metric == "auprc":
        return auprc(labels, scores)
    if metric == "f1_score":
        return _f1_at_threshold(labels, scores)
    raise NotImplementedError("Check the evaluation metric.")
