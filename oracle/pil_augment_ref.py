"""CPU restatement (numpy) of the random augmentations of the reference's image transform for GIVEN parameters
(data.py:167-169, datachar.py:125-129): RandomHorizontalFlip -> RandomRotation(20) -> ColorJitter(contrast=0.3) on a PIL
image.  Test infrastructure only: nothing in the product imports this file.

torchvision (absent here) hands the work to Pillow: ``img.transpose(FLIP_LEFT_RIGHT)``, ``img.rotate(angle,
NEAREST, expand=False, fillcolor=0)`` and ``ImageEnhance.Contrast(img).enhance(factor)``.  Pillow 12.2.0 is importable
here, so this restatement is pinned by running Pillow itself on the same pixels and parameters
(tests/test_oracle_golden.py) and by the committed fixture tests/golden/pil_augment.npz
(oracle/make_augment_fixture.py).  The random DRAWS (which flip, angle, factor) are torchvision's use of torch's global
generator and stay parity-unpinned.

Pillow's arithmetic as restated:
* rotate: angle taken modulo 360; 0 returns the image unchanged, 180 is an exact transpose (as are 90 / 270 for square
  images); otherwise the inverse affine map about the centre (w/2, h/2) with cos / sin rounded to 15 decimals, evaluated
  for NEAREST in 16.16 fixed point: coefficient FIX(v) = floor(v * 65536 + 0.5), the constant terms take the half-pixel
  offset (c + a/2 + b/2), the source index is the running sum >> 16, and pixels whose source lies outside stay 0.
* contrast: mean = int(mean of the L channel + 0.5) with L = (19595 R + 38470 G + 7471 B + 32768) >> 16; every channel
  becomes mean + factor * (v - mean) in single precision, truncated toward zero for 0 <= factor <= 1, clipped to
  [0, 255] and truncated otherwise.
"""
import math

import numpy as np


def hflip(img):
    return img[:, ::-1].copy()


def _fix(v):
    return int(math.floor(v * 65536.0 + 0.5))


def rotate_coeffs(w, h, angle):
    """The six 16.16 fixed-point coefficients (a0, a1, a2, a3, a4, a5) of Pillow's NEAREST affine pass for
    Image.rotate(angle): source x = (a2 + x*a0 + y*a1) >> 16, source y = (a5 + x*a3 + y*a4) >> 16.  None when Pillow takes
    one of its exact shortcuts instead (angle % 360 in {0, 180}, or {90, 270} for a square image)."""
    angle = angle % 360.0
    if angle == 0 or angle == 180 or (angle in (90, 270) and w == h):
        return None
    cx, cy = w / 2.0, h / 2.0
    a = -math.radians(angle)
    m = [round(math.cos(a), 15), round(math.sin(a), 15), 0.0, round(-math.sin(a), 15), round(math.cos(a), 15), 0.0]
    m[2] = m[0] * -cx + m[1] * -cy + m[2] + cx
    m[5] = m[3] * -cx + m[4] * -cy + m[5] + cy
    return (_fix(m[0]), _fix(m[1]), _fix(m[2] + m[0] * 0.5 + m[1] * 0.5), _fix(m[3]), _fix(m[4]), _fix(m[5] + m[3] * 0.5 + m[4] * 0.5))


def rotate_nearest(img, angle):
    """uint8 [H, W, C] -> same shape: Image.rotate(angle) with NEAREST resampling, no expansion, zero fill."""
    h, w = img.shape[:2]
    co = rotate_coeffs(w, h, angle)
    if co is None:
        a = angle % 360.0
        return img.copy() if a == 0 else np.rot90(img, int(a // 90)).copy()
    a0, a1, a2, a3, a4, a5 = co
    ys, xs = np.mgrid[0:h, 0:w].astype(np.int64)
    sx = (a2 + xs * a0 + ys * a1) >> 16
    sy = (a5 + xs * a3 + ys * a4) >> 16
    ok = (sx >= 0) & (sx < w) & (sy >= 0) & (sy < h)
    out = np.zeros_like(img)
    out[ok] = img[sy[ok], sx[ok]]
    return out


def luma_mean(img):
    l = (img[..., 0].astype(np.int64) * 19595 + img[..., 1].astype(np.int64) * 38470 + img[..., 2].astype(np.int64) * 7471 + 0x8000) >> 16
    return int(l.sum() / l.size + 0.5)


def adjust_contrast(img, factor):
    mean = np.float32(luma_mean(img))
    f = np.float32(factor)
    v = mean + f * (img.astype(np.float32) - mean)
    if 0.0 <= factor <= 1.0:
        return v.astype(np.int32).astype(np.uint8)          # C cast: truncation toward zero (never negative here)
    return np.where(v <= 0, 0, np.where(v >= 255, 255, v.astype(np.int32))).astype(np.uint8)


def augment(img, flip, angle, contrast):
    """uint8 [H, W, 3] -> uint8 [H, W, 3]: the three augmentations in the reference's order."""
    if flip:
        img = hflip(img)
    img = rotate_nearest(img, angle)
    return adjust_contrast(img, contrast)
