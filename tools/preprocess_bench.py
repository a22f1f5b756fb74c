"""Throughput of the device resize + normalise (icap_resize_normalize_u8_f32) on a batch of decoded images of the food
dataset's usual size (169 x 274), beside Pillow + torch doing the same on one host core.
usage: python tools/preprocess_bench.py [--batch 1024] [--h 169] [--w 274] [--iters 20]"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image_captioning_amd import preprocess  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--h", type=int, default=169)
    ap.add_argument("--w", type=int, default=274)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--layout", default="nchw")
    ap.add_argument("--cpu-images", type=int, default=64)
    args = ap.parse_args()
    B, H, W = args.batch, args.h, args.w
    a = np.random.default_rng(0).integers(0, 256, (B, H, W, 3), dtype=np.uint8)
    x = torch.from_numpy(a).cuda()
    for _ in range(3):
        y = preprocess.resize_normalize(x, layout=args.layout)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.iters):
        y = preprocess.resize_normalize(x, layout=args.layout)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / args.iters
    out_b = y.numel() * 4
    algo = B * H * W * 3 + out_b                              # pixels in, tensor out; the 8-bit intermediate is extra traffic
    tmp = 2 * B * H * 224 * 4
    res = {"batch": B, "in": [H, W], "layout": args.layout, "ms": ms, "images_per_s": B / ms * 1e3,
           "algorithmic_GBps": algo / ms / 1e6, "with_intermediate_GBps": (algo + tmp) / ms / 1e6}
    from PIL import Image
    n = min(args.cpu_images, B)
    torch.set_num_threads(1)
    t0 = time.perf_counter()
    for i in range(n):
        r = np.asarray(Image.fromarray(a[i]).resize((224, 224), Image.BILINEAR))
        preprocess.to_model_input(r)
    dt = time.perf_counter() - t0
    res["cpu_pillow_images_per_s_1core"] = n / dt
    print(json.dumps(res))


if __name__ == "__main__":
    main()
