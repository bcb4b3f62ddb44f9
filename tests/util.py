import torch


def relerr(a, b):
    """max |a-b| / (max |b| + tiny) on float64 CPU copies."""
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


# stated tolerances (max-norm relative error per tensor)
TOL = {torch.float32: 2e-5, torch.bfloat16: 2.5e-2}


def relrms(a, b):
    """rms(a-b) / rms(b): the metric stated for bf16 whole-network outputs (max-norm is dominated by a few
    worst-case roundings after ~10 stacked bf16 layers)."""
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    return float(((a - b).pow(2).mean() / (b.pow(2).mean() + 1e-30)).sqrt())
