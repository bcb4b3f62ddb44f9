"""Helpers to compare tensors with the golden summaries written by tests/golden/make_fixtures.py."""
import json
import os

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def load_golden():
    with open(os.path.join(HERE, "golden", "reference_vectors.json")) as f:
        js = json.load(f)
    npz = dict(np.load(os.path.join(HERE, "golden", "reference_vectors.npz")))
    return js, npz


def check_summary(t, ref, rtol, what=""):
    """`t` against a summarize() record: element count, strided samples, sum / abs-sum / square-sum."""
    t = t.detach().double().cpu().reshape(-1)
    assert t.numel() == ref["n"], (what, t.numel(), ref["n"])
    scale = max(ref["abssum"] / ref["n"], 1e-12)          # mean |x|: the natural unit for absolute errors
    got = t[torch.tensor(ref["idx"])]
    exp = torch.tensor(ref["samples"], dtype=torch.float64)
    assert float((got - exp).abs().max()) <= rtol * max(float(exp.abs().max()), scale) * 4, (what, "samples", got, exp)
    assert abs(float(t.abs().sum()) - ref["abssum"]) <= rtol * ref["abssum"] + 1e-12, (what, "abssum", float(t.abs().sum()), ref["abssum"])
    assert abs(float((t * t).sum()) - ref["sqsum"]) <= 2 * rtol * ref["sqsum"] + 1e-12, (what, "sqsum")
    assert abs(float(t.sum()) - ref["sum"]) <= rtol * ref["abssum"] + 1e-12, (what, "sum", float(t.sum()), ref["sum"])


def check_errs(got, ref, rtol, what=""):
    for k, v in ref.items():
        assert abs(got[k] - v) <= rtol * max(abs(v), 1e-3), (what, k, got[k], v)
