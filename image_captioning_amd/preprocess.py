"""Deterministic image preprocessing.

The reference evaluates with ``data.image_transforms`` (data.py:166-173), which applies a random flip,
rotation and contrast jitter even at test time (TransformerTest.py:348), so its inputs are not
reproducible.  This is the deterministic tail of that pipeline: RGB -> 224x224 bilinear resize ->
[0,1] -> ImageNet mean/std, NCHW fp32.
"""
from __future__ import annotations

import numpy as np
import torch

from .synthetic import normalize_u8


def load_image_u8(path: str, size: int = 224) -> np.ndarray:
    from PIL import Image
    img = Image.open(path).convert("RGB").resize((size, size), Image.BILINEAR)
    return np.asarray(img, dtype=np.uint8)


def to_model_input(u8_nhwc) -> torch.Tensor:
    """uint8 [N, H, W, 3] (numpy or torch) -> normalised fp32 [N, 3, H, W]."""
    t = torch.as_tensor(np.asarray(u8_nhwc))
    if t.dim() == 3:
        t = t.unsqueeze(0)
    return normalize_u8(t)
