"""Host-side mirror of the hot-path helpers of the reference's lib/utils.py (weights_init :51-56, l2_loss :59-63,
weighted_bce :65-71, gray2rgb :91-92, threshold :149-152, fix_model_state_dict :15-22).  The cv2 / tensorboard
helpers of that file (video_to_flow, morphology_proc, update_summary ...) are out of scope."""
from collections import OrderedDict

import torch
import torch.nn as nn

from .. import functional as F
from ..functional import ClTensor


def fix_model_state_dict(state_dict):
    """Strip the ``module.`` prefix DataParallel checkpoints carry (reference lib/utils.py:15-22; the reference
    forgets to import OrderedDict there)."""
    new_state_dict = OrderedDict()
    for k, v in state_dict.items():
        name = k[7:] if k.startswith("module.") else k
        new_state_dict[name] = v
    return new_state_dict


def weights_init(m):
    """Reference lib/utils.py:51-56: Conv3d.weight ~ N(0, .02); BatchNorm3d.weight ~ N(1, .02), bias = 0.
    ConvTranspose3d, Linear, BatchNorm1d and all 2-D layers are NOT touched (isinstance semantics)."""
    if isinstance(m, nn.Conv3d):
        m.weight.data.normal_(0.0, 0.02)
        F.invalidate_weight_cache()
    elif isinstance(m, nn.BatchNorm3d):
        m.weight.data.normal_(1.0, 0.02)
        m.bias.data.fill_(0)


def weights_init_dcgan(m):
    """Upstream GANomaly initialiser (the reference imports it from a lib.networks module that is not in its tree,
    models/ganomaly.py:18): Conv* ~ N(0, .02); BatchNorm* weight ~ N(1, .02), bias 0.  Build-declared default."""
    classname = m.__class__.__name__
    if classname.find("Conv") != -1:
        m.weight.data.normal_(0.0, 0.02)
        F.invalidate_weight_cache()
    elif classname.find("BatchNorm") != -1:
        m.weight.data.normal_(1.0, 0.02)
        m.bias.data.fill_(0)


def _cl(x):
    return x if isinstance(x, ClTensor) else F.to_cl(x)


def l2_loss(input, target, size_average=True):
    """Reference lib/utils.py:59-63.  Returns a 0-d float32 device tensor with an autograd edge."""
    if not size_average:
        raise NotImplementedError("size_average=False is never used on the hot path")
    return F.l2_loss(_cl(input), _cl(target))


def weighted_bce(input, target, pos_weight=2):
    """Reference lib/utils.py:65-71 (pos_weight weighs the NEGATIVE class; the clamp upper bound is 1.0 in fp32)."""
    return F.weighted_bce(_cl(input), _cl(target), pos_weight)


def gray2rgb(video):
    """Reference lib/utils.py:91-92."""
    return F.gray2rgb(_cl(video))


def threshold(data):
    """Reference lib/utils.py:149-152: (data > 0.5) as float, reference layout."""
    t = data.to_torch() if isinstance(data, ClTensor) else data
    return (t > 0.5).float()


def morphology_proc(video, literal=False):
    """Reference lib/utils.py:139-147: 5 x 5 morphological opening of a (N,1,T,H,W) mask — there on the CPU through cv2 with a
    device -> host -> device round trip, here one HIP launch pair on the device (vfd_morph_open5x5; cv2's default border:
    pixels outside the plane do not take part).  Returns a float32 device tensor of the input's shape.

    WHICH plane is opened is a deliberate deviation ("parity unpinned"): the reference's loop
    ``for v in video: [cv2.morphologyEx(i, ...) for i in v]`` hands cv2 the whole (T,H,W) block of a clip, which cv2 reads as
    rows = T, cols = H, channels = W — the code as executed opens the (T,H) plane of every W column.  The default here opens
    every (H,W) FRAME, what the function's name, its caller (a per-frame foreground mask, models/mygannet.py:404-407) and
    the kernel size say was meant.  ``literal=True`` reproduces the executed behaviour (planes = N*W over the (T,H) axes).
    Either way cv2 is absent from this image: the opening is pinned by scipy.ndimage.grey_opening, not by cv2 output, and the
    scores test() derives from it are not claimed to match the reference's."""
    from .._lib import check, load, require_device, stream
    t = video.to_torch() if isinstance(video, ClTensor) else video
    require_device(t)
    t = t.contiguous().float()
    if t.dim() < 3:
        raise RuntimeError("morphology_proc expects (..., H, W) frames")
    if literal:
        if t.dim() != 5:
            raise RuntimeError("morphology_proc(literal=True) expects the reference's (N,1,T,H,W) block")
        t = t.permute(0, 1, 4, 2, 3).contiguous()         # (N,1,W,T,H): planes of (T,H)
    H, W = int(t.shape[-2]), int(t.shape[-1])
    planes = t.numel() // (H * W)
    tmp, out = torch.empty_like(t), torch.empty_like(t)
    check(load().vfd_morph_open5x5(t.data_ptr(), tmp.data_ptr(), out.data_ptr(), planes, H, W, 0.0, 0, stream()), "morph_open5x5")
    if literal:
        out = out.permute(0, 1, 3, 4, 2).contiguous()
    return out


def normalize(tensor):
    """Reference lib/utils.py:81-89: shift to the range (0, 1)."""
    mn, mx = float(tensor.min()), float(tensor.max())
    return (tensor.clamp(min=mn, max=mx) - mn) / (mx - mn + 1e-5)


def rgb_to_gray(video):
    """cv2.COLOR_RGB2GRAY of reference lib/utils.py:131-136 on a (N,3,T,H,W) device tensor: 0.299 R + 0.587 G + 0.114 B."""
    return (0.299 * video[:, 0:1] + 0.587 * video[:, 1:2] + 0.114 * video[:, 2:3])


def predict_forg(gout, input):
    """Reference models/anogan.py:24-37: |gout - input|, every FRAME (all clips, all channels of time step t) shifted to
    (0,1) on its own, then grey — on the device, (N,1,T,H,W) out."""
    diff = torch.abs(gout - input)
    frames = [normalize(diff[:, :, t]) for t in range(diff.shape[2])]       # reference normalises per time step over (B,C,H,W)
    return rgb_to_gray(torch.stack(frames, dim=2))
