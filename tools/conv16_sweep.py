"""Tile-configuration sweep of icap_conv2d_bf16_f32 on the VGG-16 / ResNet-18 layer shapes (B = 256).
usage: python tools/conv16_sweep.py [--batch 256]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image_captioning_amd import _native as N, ops  # noqa: E402

SHAPES = [(224, 64, 64, 3, 1), (112, 64, 128, 3, 1), (112, 128, 128, 3, 1), (56, 128, 256, 3, 1), (56, 256, 256, 3, 1), (28, 256, 512, 3, 1),
          (28, 512, 512, 3, 1), (14, 512, 512, 3, 1), (56, 64, 64, 3, 1), (28, 128, 128, 3, 1), (14, 256, 256, 3, 1), (7, 512, 512, 3, 1)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--cfgs", type=int, nargs="+", default=[0, 1, 2, 3, 4])
    ap.add_argument("--io", type=int, default=0, help="io_bf16 flags: 1 = bf16 input, 2 = bf16 output (3 = both)")
    args = ap.parse_args()
    lib = N.load()
    B = args.batch
    for H, Cin, Cout, K, s in SHAPES:
        x = torch.randn(B, H, H, Cin, device="cuda")
        if args.io & 1:
            x = x.to(torch.bfloat16)
        w = ops.to_bf16(torch.randn(Cout, K, K, Cin, device="cuda") * 0.05)
        bias = torch.randn(Cout, device="cuda")
        y = torch.empty(B, H, H, Cout, device="cuda", dtype=torch.bfloat16 if args.io & 2 else torch.float32)
        line = f"{H:4d} {Cin:4d}->{Cout:4d}:"
        for cfg in [-1] + args.cfgs:
            def run():
                N.check(lib.icap_conv2d_bf16_f32(x.data_ptr(), w.data_ptr(), None, bias.data_ptr(), None, y.data_ptr(), B, H, H, Cin, Cout, K, K, s,
                                                 1, 1, H, H, ops.ACT_RELU, args.io, cfg, ops._stream()), "conv2d_bf16")
            run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                run()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 5
            line += f"  cfg{cfg:2d} {ms:6.3f} ms {2.0 * B * H * H * Cout * K * K * Cin / ms / 1e9:5.0f} TF"
        print(line, flush=True)


if __name__ == "__main__":
    main()
