"""The thin projection layers of EfficientNet-B0 at B images: the streaming kernel (linear_thin) against the tuned implicit-GEMM
kernel and the split-product kernel."""
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
shapes = [("project 112", 12544, 32, 16, False), ("project 56 a", 3136, 96, 24, False), ("project 56 b", 3136, 144, 24, True)]
ops.set_autotune(True)
for name, hw, K, N, use_res in shapes:
    M = B * hw
    x = torch.randn(M, K, device=d)
    w = torch.randn(N, K, device=d) * K ** -0.5
    b, s = torch.randn(N, device=d), torch.rand(N, device=d) + 0.5
    gate = torch.rand(B, K, device=d)
    res = torch.randn(M, N, device=d) if use_res else None
    xx, rr = x.view(B, hw, 1, K), (res.view(B, hw, 1, N) if use_res else None)
    ops.conv2d(xx, w.reshape(N, 1, 1, K), s, b, residual=rr, gate=gate)        # tunes
    t32 = timeit(lambda: ops.conv2d(xx, w.reshape(N, 1, 1, K), s, b, residual=rr, gate=gate))
    tx3 = None
    if K % 8 == 0:
        w3 = ops.split3(w)
        tx3 = min(timeit(lambda: ops.linear_f32x3(x, w3, b, s, residual=res, gate=gate, hw=hw, cfg=c)) for c in (0, 1, 2))
    tt = timeit(lambda: ops.linear_thin(x, w, b, s, residual=res, gate=gate, hw=hw))
    y0 = ops.conv2d(xx, w.reshape(N, 1, 1, K), s, b, residual=rr, gate=gate).view(M, N)
    y1 = ops.linear_thin(x, w, b, s, residual=res, gate=gate, hw=hw)
    byts = 4.0 * M * (K + N * (2 if use_res else 1))
    print(f"{name:13s} [{M:8d},{K:4d},{N:3d}] implicit GEMM (tuned) {t32 * 1e3:7.1f} us | split products {tx3 * 1e3 if tx3 else float('nan'):7.1f} us | "
          f"streaming {tt * 1e3:7.1f} us = {byts / tt / 1e6:7.1f} GB/s | max diff {(y0 - y1).abs().max().item():.2e}", flush=True)
