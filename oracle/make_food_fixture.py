"""Write tests/golden/food_test_u8.npz: the reference's eight qualitative test photos
(/root/reference/food_test/1-8.jpg) after the deterministic part of its preprocessing
(RGB, 224x224 bilinear resize; data.py:170), as uint8 NHWC.  Build-container only."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from image_captioning_amd.preprocess import load_image_u8  # noqa: E402

imgs = np.stack([load_image_u8(f"/root/reference/food_test/{i}.jpg") for i in range(1, 9)])
out = os.path.join(ROOT, "tests", "golden", "food_test_u8.npz")
np.savez_compressed(out, images=imgs)
print(imgs.shape, imgs.dtype, os.path.getsize(out))
