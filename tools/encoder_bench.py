"""Forward throughput of the three encoders (eval mode) at a given batch: images/s and the share of the fp32 MFMA peak
their algorithmic flops reach, with the per-launch table of the slowest shapes.
usage: python tools/encoder_bench.py [--batch 256] [--which resnet18 vgg16 effnet] [--top 12] [--dtype bf16]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image_captioning_amd import ops, seq2seqbasic as s2s, transformer_word_level as twl  # noqa: E402
from image_captioning_amd.synthetic import seeded_state_dict, synthetic_images  # noqa: E402

GFLOP_PER_IMAGE = {"resnet18": 3.63, "vgg16": 30.7, "effnet": 0.78}      # 2 * MAC, forward, 224 x 224


def build(which):
    if which == "effnet":
        m = twl.get_cnn_model()
    else:
        m = s2s.EncoderCNN(256, cnn_type=which)
    m.load_state_dict(seeded_state_dict(m, seed=3))
    return m.eval().cuda()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--which", nargs="+", default=["resnet18", "vgg16", "effnet"])
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--top", type=int, default=12)
    ap.add_argument("--dtype", choices=("f32", "bf16"), default="f32")
    args = ap.parse_args()
    ops.set_compute_dtype(args.dtype)
    x = synthetic_images(min(args.batch, 32), seed=1).repeat((args.batch + 31) // 32, 1, 1, 1)[:args.batch].contiguous().cuda()
    for which in args.which:
        m = build(which)
        with torch.no_grad():
            ops.set_autotune(True)
            m(x)
            torch.cuda.synchronize()
            ops.set_autotune(False)
            m(x)
            rec = []
            ops.set_trace(lambda n, s: rec.append((n, s, torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))) or rec[-1][2:])
            m(x)
            torch.cuda.synchronize()
            ops.set_trace(None)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                m(x)
            e1.record()
            torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.iters
        tf = GFLOP_PER_IMAGE[which] * args.batch / ms
        print(json.dumps({"encoder": which, "dtype": args.dtype, "batch": args.batch, "ms": ms, "images_per_s": args.batch / ms * 1e3, "TFLOP_per_s": tf,
                          "frac_f32_mfma_peak": tf / 157.3}))
        rows = sorted(((a.elapsed_time(b), n, s) for n, s, a, b in rec), reverse=True)
        tot = sum(r[0] for r in rows)
        print(f"  traced {len(rows)} launches, {tot:.3f} ms")
        for t, n, s in rows[:args.top]:
            extra = ""
            if n in ("conv2d", "conv2d_bf16"):
                B, H, W, Cin, Cout, K, st, Ho, Wo = s[:9]
                extra = f"  {2.0 * B * Ho * Wo * Cout * K * K * Cin / t / 1e9:6.1f} TFLOP/s"
            elif n == "linear":
                extra = f"  {2.0 * s[0] * s[1] * s[2] / t / 1e9:6.1f} TFLOP/s"
            print(f"  {t:8.4f} ms  {n}{list(s)}{extra}")
        del m
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
