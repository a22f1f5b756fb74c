"""The gated projection layers of EfficientNet-B0 at 14 x 14 / 7 x 7 for B images: the tall-activation split-product kernel
(linear_f32x3 cfg 3) against the tiled split-product kernel's configurations and the tuned fp32 kernel."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image_captioning_amd import ops
d = torch.device("cuda:0")


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
ONLY = sys.argv[2] if len(sys.argv) > 2 else ""          # substring of the layer names to run
shapes = [("28 240->40", 784, 240, 40, True), ("14 240->80", 196, 240, 80, False), ("14 480->80", 196, 480, 80, True),
          ("14 480->112", 196, 480, 112, False), ("14 672->112", 196, 672, 112, True), ("7 672->192", 49, 672, 192, False),
          ("7 1152->192", 49, 1152, 192, True), ("7 1152->320", 49, 1152, 320, False)]
for name, hw, K, N, use_res in shapes:
    if ONLY not in name:
        continue
    M = B * hw
    x = torch.randn(M, K, device=d)
    w = torch.randn(N, K, device=d) * K ** -0.5
    b, s = torch.randn(N, device=d), torch.rand(N, device=d) + 0.5
    gate = torch.rand(B, K, device=d)
    res = torch.randn(M, N, device=d) if use_res else None
    w3 = ops.split3(w)
    tiled = [timeit(lambda: ops.linear_f32x3(x, w3, b, s, residual=res, gate=gate, hw=hw, cfg=c)) for c in (0, 1, 2)]
    tt = timeit(lambda: ops.linear_f32x3(x, w3, b, s, residual=res, gate=gate, hw=hw, cfg=3))
    y0 = ops.linear_f32x3(x, w3, b, s, residual=res, gate=gate, hw=hw, cfg=0)
    y1 = ops.linear_f32x3(x, w3, b, s, residual=res, gate=gate, hw=hw, cfg=3)
    byts = 4.0 * M * (K + N * (2 if use_res else 1))
    flops = 2.0 * M * K * N * 6
    print(f"{name:12s} [{M:7d},{K:5d},{N:4d}] tiled " + " ".join(f"{t * 1e3:7.1f}" for t in tiled) + f" us | tall {tt * 1e3:7.1f} us = "
          f"{byts / tt / 1e6:7.1f} GB/s, {flops / tt / 1e9:6.1f} TFLOP/s (bf16 products) | max diff {(y0 - y1).abs().max().item():.2e}", flush=True)
