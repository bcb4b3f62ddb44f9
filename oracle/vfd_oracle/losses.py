"""Losses and initialisers of the reference hot path, restated (reference lib/utils.py)."""
import torch
import torch.nn as nn


def l2_loss(input, target, size_average=True):
    """lib/utils.py:59-63."""
    if size_average:
        return torch.mean(torch.pow((input - target), 2))
    return torch.pow((input - target), 2)


def weighted_bce(input, target, pos_weight=2):
    """lib/utils.py:65-71 — note the weight sits on the NEGATIVE class and 1-1e-8 == 1.0 in float32."""
    input = torch.clamp(input, min=1e-8, max=1 - 1e-8)
    if pos_weight is not None:
        loss = (target * torch.log(input)) + pos_weight * (1 - target) * torch.log(1 - input)
    else:
        loss = target * torch.log(input) + (1 - target) * torch.log(1 - input)
    return torch.neg(torch.mean(loss))


def gray2rgb(video):
    """lib/utils.py:91-92."""
    return torch.cat([video, video, video], dim=1)


def weights_init(m):
    """lib/utils.py:51-56: only Conv3d and BatchNorm3d instances are touched."""
    if isinstance(m, nn.Conv3d):
        m.weight.data.normal_(0.0, 0.02)
    elif isinstance(m, nn.BatchNorm3d):
        m.weight.data.normal_(1.0, 0.02)
        m.bias.data.fill_(0)
