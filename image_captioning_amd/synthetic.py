"""Seeded weights and synthetic inputs shared by the tests, the benchmark and the golden-vector generator.

Pretrained weights are not available offline, so parity is checked on weights drawn from a fixed
recipe: every tensor of a ``state_dict`` is filled from a ``torch.Generator`` seeded by the global
seed and the tensor's key, so the same key always gets the same values regardless of model or
creation order, on any machine with this torch build (CPU generator).
"""
from __future__ import annotations

import hashlib
import math

import torch


def _gen(seed: int, key: str) -> torch.Generator:
    h = hashlib.sha256(f"{seed}:{key}".encode()).digest()
    g = torch.Generator(device="cpu")
    g.manual_seed(int.from_bytes(h[:8], "little") & 0x7FFFFFFFFFFFFFFF)
    return g


def seeded_state_dict(module: torch.nn.Module, seed: int = 0, gain: float = 1.0) -> dict:
    """A state_dict for ``module`` with non-trivial values everywhere:
    weights ~ U(-a, a) with a = gain * sqrt(3 / fan_in); biases ~ U(-0.1, 0.1); norm scales ~ U(0.5, 1.5);
    BatchNorm running_mean ~ N(0, 0.25), running_var ~ U(0.5, 2)."""
    out = {}
    for key, ref in module.state_dict().items():
        g = _gen(seed, key)
        shape = tuple(ref.shape)
        leaf = key.rsplit(".", 1)[-1]
        if leaf == "num_batches_tracked":
            t = torch.tensor(100, dtype=ref.dtype)
        elif leaf == "running_mean":
            t = torch.randn(shape, generator=g) * 0.5
        elif leaf == "running_var":
            t = torch.rand(shape, generator=g) * 1.5 + 0.5
        elif ref.dim() <= 1:
            is_norm_scale = leaf == "weight"
            t = torch.rand(shape, generator=g) + 0.5 if is_norm_scale else (torch.rand(shape, generator=g) - 0.5) * 0.2
        else:
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            if "embed" in key or "embedding" in key:
                t = torch.randn(shape, generator=g)
            else:
                a = gain * math.sqrt(3.0 / max(fan_in, 1))
                t = (torch.rand(shape, generator=g) * 2 - 1) * a
        out[key] = t.to(ref.dtype)
    return out


def synthetic_images(n: int, seed: int = 1234, size: int = 224) -> torch.Tensor:
    """uint8 uniform pixels -> /255 -> ImageNet normalisation, fp32 NCHW [n, 3, size, size] (CPU)."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    u8 = torch.randint(0, 256, (n, size, size, 3), generator=g, dtype=torch.uint8)
    return normalize_u8(u8)


def normalize_u8(u8_nhwc: torch.Tensor) -> torch.Tensor:
    mean = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
    x = u8_nhwc.permute(0, 3, 1, 2).float() / 255.0
    return ((x - mean) / std).contiguous()


def synthetic_captions(n: int, T: int, vocab: int, start: int, end: int, pad: int, seed: int = 1234) -> torch.Tensor:
    """int64 [n, T]: col 0 = start, middle uniform in [4, vocab), last = end; a quarter of the rows get a pad tail."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed + 1)
    cap = torch.randint(4, vocab, (n, T), generator=g, dtype=torch.int64)
    cap[:, 0] = start
    cap[:, -1] = end
    for r in range(0, n, 4):
        cut = int(torch.randint(2, T - 1, (1,), generator=g))
        cap[r, cut] = end
        cap[r, cut + 1:] = pad
    return cap
