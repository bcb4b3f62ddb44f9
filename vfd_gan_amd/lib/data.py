"""Synthetic clip source with the 4-tuple contract of the reference's dataset (lib/data.py:78):
``(input, real, gt, lb)`` = forged clip, original clip, tamper-edge mask, per-frame label.

Video decode (cv2) is out of scope; the clips below follow SURVEY.md section 8(d):
``real`` ~ U(-1,1) low-pass filtered along T, ``input`` = ``real`` with one noised rectangle per clip,
``gt`` = that rectangle's edge map in {0,1}, ``lb`` = 1.  Generated on the host from a seeded generator so that
the CPU oracle and the HIP path see identical bytes.
"""
import torch


def synthetic_batch(batchsize, nfr=16, isize=128, ich=3, seed=1234):
    g = torch.Generator(device="cpu").manual_seed(int(seed))
    B, T, S = batchsize, nfr, isize
    base = torch.rand(B, ich, 1, S, S, generator=g) * 2 - 1
    drift = (torch.rand(B, ich, T, S, S, generator=g) * 2 - 1) * 0.15
    real = (base + torch.cumsum(drift, dim=2) / max(T ** 0.5, 1.0)).clamp_(-1, 1)
    inp = real.clone()
    gt = torch.zeros(B, 1, T, S, S)
    for b in range(B):
        h = int(torch.randint(S // 8, S // 3 + 1, (1,), generator=g))
        w = int(torch.randint(S // 8, S // 3 + 1, (1,), generator=g))
        y0 = int(torch.randint(1, S - h - 1, (1,), generator=g))
        x0 = int(torch.randint(1, S - w - 1, (1,), generator=g))
        inp[b, :, :, y0:y0 + h, x0:x0 + w] = torch.rand(ich, T, h, w, generator=g) * 2 - 1
        gt[b, 0, :, y0, x0:x0 + w] = 1
        gt[b, 0, :, y0 + h - 1, x0:x0 + w] = 1
        gt[b, 0, :, y0:y0 + h, x0] = 1
        gt[b, 0, :, y0:y0 + h, x0 + w - 1] = 1
    lb = torch.ones(B, T)
    return inp, real, gt, lb


def synthetic_flow(batchsize, nfr=16, isize=128, seed=4321):
    """Stand-in for lib/utils.py:94-129 (CPU Farneback optical flow, out of scope): U(-1,1) 3-channel video."""
    g = torch.Generator(device="cpu").manual_seed(int(seed))
    return torch.rand(batchsize, 3, nfr, isize, isize, generator=g) * 2 - 1


class SyntheticClips:
    """Iterable with ``len()`` yielding the 4-tuple; stands where the reference's torch DataLoader does."""

    def __init__(self, args, steps, seed=1234, pin=True):
        self.args, self.steps, self.seed, self.pin = args, steps, seed, pin

    def __len__(self):
        return self.steps

    def __iter__(self):
        for i in range(self.steps):
            batch = synthetic_batch(self.args.batchsize, self.args.nfr, self.args.isize, self.args.ich,
                                    self.seed + i)
            if self.pin and torch.cuda.is_available():
                batch = tuple(t.pin_memory() for t in batch)
            yield batch


class DataLoader:
    """Same entry points as the reference's lib/data.py:114-161: ``DataLoader(args).load_data()`` returns
    ``{'train': ..., 'test': ...}``."""

    def __init__(self, args, rank=0):
        self.args, self.rank = args, rank

    def load_data(self):
        steps = getattr(self.args, "steps_per_epoch", 8)
        return {"train": SyntheticClips(self.args, steps, seed=1234 + 1000 * self.rank),
                "test": SyntheticClips(self.args, max(1, steps // 4), seed=99991 + 1000 * self.rank)}
