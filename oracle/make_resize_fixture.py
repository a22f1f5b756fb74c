"""Write tests/golden/pil_resize.npz: input pixels and the output of Pillow's own BILINEAR resize (what the reference's
``transforms.Resize((224, 224))`` runs, data.py:170) for a few image sizes, plus ToTensor + Normalize computed with torch
the way torchvision does (data.py:171-172).  Inputs: two crops of the reference's qualitative test photos
(/root/reference/food_test/7.jpg, 5.jpg - reduced first so the fixture stays small), stored with their full outputs,
and seeded random pixels (``random_case`` regenerates the input; the fixture keeps the CRC-32 of Pillow's output and every
fourth pixel of it).  Build-container only."""
import os
import zlib

import numpy as np
import torch
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# up-scaling on both axes; identity vertical pass; 3.1x down-scaling (9 taps); one-pixel-wide; 18x down-scaling (39 taps)
RANDOM_CASES = [(97, 61), (224, 300), (700, 233), (50, 1), (4096, 300)]


def random_case(h, w):
    return np.random.default_rng(h * 10007 + w).integers(0, 256, (h, w, 3), dtype=np.uint8)


def main():
    photo = Image.open("/root/reference/food_test/7.jpg").convert("RGB")
    photo2 = Image.open("/root/reference/food_test/5.jpg").convert("RGB")
    cases = {
        "photo_169x274": np.asarray(photo.reduce(4).crop((0, 100, 225, 269)).resize((274, 169), Image.NEAREST)),   # the dataset's usual size
        "photo_240x240": np.asarray(photo2.reduce(6)),
    }
    out = {}
    mean = torch.tensor([0.485, 0.456, 0.406]).view(3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).view(3, 1, 1)
    for name, a in cases.items():
        a = np.ascontiguousarray(a, dtype=np.uint8)
        r = np.asarray(Image.fromarray(a).resize((224, 224), Image.BILINEAR))
        out[f"{name}_in"] = a
        out[f"{name}_out"] = r
        print(name, a.shape, "->", r.shape)
    for h, w in RANDOM_CASES:
        r = np.asarray(Image.fromarray(random_case(h, w)).resize((224, 224), Image.BILINEAR))
        out[f"rand_{h}x{w}_crc"] = np.array(zlib.crc32(r.tobytes()), dtype=np.uint64)
        out[f"rand_{h}x{w}_sub"] = np.ascontiguousarray(r[::4, ::4])
    x = torch.from_numpy(out["photo_169x274_out"].copy()).permute(2, 0, 1).float().div(255)     # ToTensor
    out["photo_169x274_norm"] = ((x - mean) / std).numpy()                                # Normalize
    path = os.path.join(ROOT, "tests", "golden", "pil_resize.npz")
    np.savez_compressed(path, **out)
    print(os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
