"""Writes tests/golden/pil_augment.npz: Pillow's own output (the library torchvision delegates to for PIL images) of
flip -> rotate(NEAREST, zero fill) -> ImageEnhance.Contrast for fixed parameters, so that the numpy restatement in
oracle/pil_augment_ref.py and the GPU kernel stay pinned on a box where Pillow differs or is absent.

    python oracle/make_augment_fixture.py
"""
import os

import numpy as np
from PIL import Image, ImageEnhance

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = [  # h, w, flip, angle, contrast
    (37, 53, False, 13.7, 1.21), (53, 37, True, -19.99, 0.71), (24, 24, True, 90.0, 1.0), (24, 24, False, 270.0, 1.3),
    (31, 45, False, 180.0, 0.7), (45, 31, True, 0.0, 1.15), (64, 48, False, 90.0, 0.9), (169, 274, True, 7.25, 1.05),
    (5, 7, False, -20.0, 1.3), (1, 9, True, 3.0, 0.85),
]


def main():
    rng = np.random.default_rng(5)
    out = {"cases": np.array([[c[0], c[1], int(c[2])] for c in CASES]), "angles": np.array([c[3] for c in CASES]),
           "contrasts": np.array([c[4] for c in CASES])}
    for i, (h, w, flip, angle, fac) in enumerate(CASES):
        if i % 2:
            img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        else:       # smooth ramps: neighbouring pixels differ, so a one-pixel slip of the rotation shows
            img = ((np.add.outer(np.arange(h) * 3, np.arange(w) * 5)[..., None] + np.array([0, 40, 90])) % 256).astype(np.uint8)
        p = Image.fromarray(img)
        if flip:
            p = p.transpose(Image.FLIP_LEFT_RIGHT)
        p = p.rotate(angle, resample=Image.NEAREST, expand=False, fillcolor=0)
        p = ImageEnhance.Contrast(p).enhance(fac)
        out[f"in{i}"], out[f"out{i}"] = img, np.asarray(p)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "pil_augment.npz"), **out)
    print("wrote", len(CASES), "cases")


if __name__ == "__main__":
    main()
