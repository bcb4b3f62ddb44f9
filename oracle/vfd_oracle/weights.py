"""Deterministic, RNG-version-independent parameter fill shared by the fixture generator and the tests.

Fixtures must not depend on torch's init RNG stream (it differs between torch versions), so
every tensor of a ``state_dict`` is regenerated from numpy's PCG64 keyed by (seed, crc32(key name))."""
import zlib

import numpy as np
import torch


def fill_module(module, seed, running_stats=True):
    """Overwrite every parameter / running statistic of `module` in place; returns the module."""
    sd = module.state_dict()
    new = {}
    for k, v in sd.items():
        rng = np.random.Generator(np.random.PCG64([int(seed), zlib.crc32(k.encode())]))
        shape = tuple(v.shape)
        if k.endswith("num_batches_tracked"):
            new[k] = v.clone()
        elif k.endswith("running_mean"):
            new[k] = torch.from_numpy((rng.standard_normal(shape) * 0.05).astype(np.float32)) if running_stats else v.clone()
        elif k.endswith("running_var"):
            new[k] = torch.from_numpy((1.0 + 0.1 * rng.random(shape)).astype(np.float32)) if running_stats else v.clone()
        elif v.dim() >= 2:      # conv / conv-transpose / linear weights: zero-mean, fan-in scaled (floor .02 as weights_init)
            fan_in = int(np.prod(shape[1:]))
            std = max(0.02, 1.0 / np.sqrt(fan_in))
            new[k] = torch.from_numpy((rng.standard_normal(shape) * std).astype(np.float32))
        elif k.endswith("weight"):  # norm scale
            new[k] = torch.from_numpy((1.0 + 0.1 * rng.standard_normal(shape)).astype(np.float32))
        else:                   # biases
            new[k] = torch.from_numpy((0.05 * rng.standard_normal(shape)).astype(np.float32))
    module.load_state_dict(new)
    return module


def seeded_tensor(shape, seed, lo=-1.0, hi=1.0):
    rng = np.random.Generator(np.random.PCG64(int(seed)))
    return torch.from_numpy((lo + (hi - lo) * rng.random(tuple(shape))).astype(np.float32))


def seeded_normal(shape, seed):
    rng = np.random.Generator(np.random.PCG64(int(seed)))
    return torch.from_numpy(rng.standard_normal(tuple(shape)).astype(np.float32))


def summarize(t):
    """Compact signature of a tensor for fixtures of large outputs: shape, sum, abs-sum, 16 strided samples."""
    t = t.detach().double().reshape(-1)
    n = t.numel()
    k = min(16, n)
    idx = (torch.arange(k, dtype=torch.int64) * (n - 1)) // max(k - 1, 1)
    return {"n": n, "sum": float(t.sum()), "abssum": float(t.abs().sum()), "sqsum": float((t * t).sum()),
            "samples": t[idx].tolist(), "idx": idx.tolist()}
