"""CPU restatement (numpy) of the resize the reference's image transform performs (data.py:170, datachar.py:129:
``transforms.Resize((224, 224))`` on a PIL image = ``Image.resize(size, BILINEAR)``) and of ToTensor + Normalize
(data.py:171-172).  Test infrastructure only: nothing in the product imports this file.

The arithmetic lives in Pillow (third party; 12.2.0 in this image), not under the reference tree; Pillow is importable
here, so this restatement is pinned by running Pillow itself on the same pixels (tests/test_oracle_golden.py) and by the
committed fixture tests/golden/pil_resize.npz (written by oracle/make_resize_fixture.py).

Pillow's 8-bit resampler, as restated here: per axis, output sample i averages the input samples inside
[c - s, c + s) around c = (i + 0.5) * in/out with s = max(in/out, 1), with triangle weights normalised to sum 1 and
rounded to 22-bit fixed point; sums start at 2^21 (round half up), are shifted right by 22 and clipped to [0, 255];
the horizontal pass runs first and its 8-bit result feeds the vertical pass.
"""
import numpy as np

BITS = 22


def axis_taps(n_in, n_out):
    """-> list of (first index, int64 tap array) per output sample."""
    scale = n_in / n_out
    fs = scale if scale > 1.0 else 1.0
    out = []
    for i in range(n_out):
        c = (i + 0.5) * scale
        lo = int(c - fs + 0.5)
        lo = lo if lo > 0 else 0
        hi = int(c + fs + 0.5)
        hi = hi if hi < n_in else n_in
        ws = []
        acc = 0.0
        for x in range(lo, hi):
            a = (x - c + 0.5) * (1.0 / fs)
            a = -a if a < 0 else a
            w = 1.0 - a if a < 1.0 else 0.0
            ws.append(w)
            acc += w
        if acc != 0.0:
            ws = [w / acc for w in ws]
        out.append((lo, np.array([int(0.5 + w * (1 << BITS)) for w in ws], dtype=np.int64)))
    return out


def _pass(img, taps, axis):
    """img uint8 [..]: filter along `axis` (0 = rows / vertical, 1 = columns / horizontal)."""
    img = np.moveaxis(img, axis, 0).astype(np.int64)
    res = np.empty((len(taps),) + img.shape[1:], dtype=np.uint8)
    for i, (lo, k) in enumerate(taps):
        acc = np.tensordot(k, img[lo:lo + len(k)], axes=(0, 0)) + (1 << (BITS - 1))
        res[i] = np.clip(acc >> BITS, 0, 255).astype(np.uint8)
    return np.moveaxis(res, 0, axis)


def resize_bilinear_u8(img, size=224):
    """uint8 [H, W, 3] -> uint8 [size, size, 3]."""
    h, w, _ = img.shape
    tmp = _pass(img, axis_taps(w, size), 1)
    return _pass(tmp, axis_taps(h, size), 0)


def to_tensor_normalize(u8_hwc, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225)):
    """uint8 [H, W, 3] -> float32 [3, H, W] with torchvision's order of operations: x / 255, then (x - mean) / std, all fp32."""
    x = u8_hwc.astype(np.float32).transpose(2, 0, 1) / np.float32(255.0)
    m = np.array(mean, dtype=np.float32).reshape(3, 1, 1)
    s = np.array(std, dtype=np.float32).reshape(3, 1, 1)
    return ((x - m) / s).astype(np.float32)
