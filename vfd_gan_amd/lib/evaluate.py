"""Scores of the evaluation sweep: mirror of the reference's lib/evaluate.py:14-91 — ``evaluate(labels, scores, best, iter,
saveto, metric)`` with metric in {'roc', 'auprc', 'pr', 'f1_score'}, the same scikit-learn calls, the same curve files
(``ROC_%03d`` / ``PR_%03d`` CSVs and, when matplotlib is importable, the PNGs) written when a score beats `best`.

Host code, as in the reference (it flattens every test batch to numpy first, models/mygannet.py:439-440): the sweep's
tensors are reduced to two flat arrays once per sweep; nothing here is on the training hot path.
"""
from __future__ import print_function

import csv
import os

from scipy.interpolate import interp1d
from scipy.optimize import brentq
from sklearn.metrics import auc, average_precision_score, f1_score, precision_recall_curve, roc_curve


def _plt():
    try:
        import matplotlib
        matplotlib.use("Agg")
        import matplotlib.pyplot as plt
        return plt
    except Exception:  # noqa: BLE001  (plots are optional; the CSV curves are always written)
        return None


def evaluate(labels, scores, best=None, iter=None, saveto=None, metric=None):
    if metric == 'roc':
        return roc(labels, scores, best, iter, saveto)
    elif metric == 'auprc':
        return auprc(labels, scores)
    elif metric == 'pr':
        return pr(labels, scores, best, iter, saveto)
    elif metric == 'f1_score':
        threshold = 0.20                      # reference :24-27 (in place, as there: call it last)
        scores[scores >= threshold] = 1
        scores[scores < threshold] = 0
        return f1_score(labels, scores)
    else:
        raise NotImplementedError("Check the evaluation metric.")


def roc(labels, scores, best, iter, saveto=None):
    """ROC curve, its area and (for the plot) the equal error rate: reference :31-65."""
    fpr, tpr, _ = roc_curve(labels, scores)
    roc_auc = auc(fpr, tpr)
    if best is not None and roc_auc > best and saveto is not None:
        eer = brentq(lambda x: 1. - x - interp1d(fpr, tpr)(x), 0., 1.)
        plt = _plt()
        if plt is not None:
            plt.figure()
            plt.plot(fpr, tpr, color='darkorange', lw=2, label='(AUC = %0.2f, EER = %0.2f)' % (roc_auc, eer))
            plt.plot([eer], [1 - eer], marker='o', markersize=5, color="navy")
            plt.plot([0, 1], [1, 0], color='navy', lw=1, linestyle=':')
            plt.xlim([0.0, 1.0])
            plt.ylim([0.0, 1.05])
            plt.xlabel('False Positive Rate')
            plt.ylabel('True Positive Rate')
            plt.title('Receiver operating characteristic')
            plt.legend(loc="lower right")
            plt.savefig(os.path.join(saveto, "ROC_%03d.png" % (iter)))
            plt.close()
        with open(os.path.join(saveto, 'ROC_%03d' % (iter)), 'w', newline='') as f:
            writer = csv.writer(f)
            for data in zip(fpr, tpr):
                writer.writerow(data)
    return roc_auc


def auprc(labels, scores):
    return average_precision_score(labels, scores)


def pr(labels, scores, best, iter, saveto=None):
    """Precision-recall curve and its area: reference :72-91."""
    precision, recall, _ = precision_recall_curve(labels, scores)
    pr_auc = auc(recall, precision)
    if best is not None and pr_auc > best and saveto is not None:
        plt = _plt()
        if plt is not None:
            plt.figure()
            plt.plot(recall, precision, label='(AUC = %0.2f)' % (pr_auc))
            plt.plot([0, 1], [1, 0], color='navy', lw=1, linestyle=':')
            plt.xlim([0.0, 1.0])
            plt.ylim([0.0, 1.05])
            plt.xlabel('Recall')
            plt.ylabel('Precision')
            plt.title('Precision-Recall Curve')
            plt.legend(loc='lower right')
            plt.savefig(os.path.join(saveto, 'PR_%03d.png' % (iter)))
            plt.close()
        with open(os.path.join(saveto, 'PR_%03d' % (iter)), 'w', newline='') as f:
            writer = csv.writer(f)
            for data in zip(recall, precision):
                writer.writerow(data)
    return pr_auc
