"""Deterministic, implementation-independent model states and inputs for parity tests.

A `state_dict` is filled key by key from a numpy generator seeded with
crc32(key) ^ seed, so the reference (in `tests/golden/make_golden.py`), the CPU
oracle and the HIP product all receive bit-identical weights without any of
them depending on another's construction order or RNG consumption.
"""
import zlib

import numpy as np
import torch


def _rng(key, seed):
    return np.random.RandomState((zlib.crc32(key.encode()) ^ (seed * 2654435761)) & 0x7FFFFFFF)


def seeded_state(model, seed=0):
    """Return a state_dict for `model` with reproducible values (fp32, logical OIHW shapes)."""
    out = {}
    for key, ref in model.state_dict().items():
        shape = tuple(ref.shape)
        r = _rng(key, seed)
        leaf = key.rsplit(".", 1)[-1]
        if leaf == "num_batches_tracked":
            val = np.zeros(shape, np.int64)
        elif leaf == "running_mean":
            val = r.standard_normal(shape) * 0.1
        elif leaf == "running_var":
            val = r.uniform(0.5, 1.5, shape)
        elif len(shape) == 4:  # conv weight
            fan_in = shape[1] * shape[2] * shape[3]
            val = r.standard_normal(shape) / np.sqrt(fan_in)
        elif leaf == "weight" and shape == (1,):  # PReLU slope
            val = np.full(shape, 0.25)
        elif leaf == "weight":  # BatchNorm gamma
            val = r.uniform(0.5, 1.5, shape)
        else:  # conv / BatchNorm bias
            val = r.standard_normal(shape) * 0.1
        out[key] = torch.from_numpy(np.asarray(val)).to(ref.dtype)
    return out


def seeded_inputs(n, h, w, seed=0, n_classes=13, ignore_frac=0.0, ignore_index=13):
    """RGB in [-1,1] (clamped normal), IR uniform in [-1,1], int64 labels."""
    r = np.random.RandomState(1234 + seed)
    rgb = np.clip(r.standard_normal((n, 3, h, w)) * 0.5, -1, 1).astype(np.float32)
    ir = (r.uniform(-1, 1, (n, 1, h, w))).astype(np.float32)
    lab = r.randint(0, n_classes, (n, h, w)).astype(np.int64)
    if ignore_frac > 0:
        lab[r.uniform(size=lab.shape) < ignore_frac] = ignore_index
    return torch.from_numpy(rgb), torch.from_numpy(ir), torch.from_numpy(lab)


def set_dropout(model, p):
    """Parity runs pin Dropout2d (reference pspnet.py:49,55) to p=0 so train-mode outputs are RNG-free."""
    for m in model.modules():
        if m.__class__.__name__.endswith("Dropout2d"):
            m.p = p


def tensor_digest(t, stride=1):
    """Small, order-sensitive summary of a tensor for golden files: (sum, abs-sum, strided sample)."""
    t = t.detach().double().cpu()
    flat = t.reshape(-1)
    return np.array([float(flat.sum()), float(flat.abs().sum()), float((flat * flat).sum())]), \
        flat[::stride].float().numpy().copy()
