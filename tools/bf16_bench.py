"""Time the bf16-compute linear kernel against the fp32 one on the captioning model's 1x1-convolution / Linear shapes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image_captioning_amd import ops
d = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
shapes = [(B * 196, 80, 480), (B * 196, 480, 80), (B * 196, 112, 672), (B * 196, 672, 112), (B * 49, 192, 1152), (B * 49, 1152, 192),
          (B * 49, 1152, 320), (B * 49, 320, 1280), (B, 512, 2048), (B, 2048, 512), (B, 512, 6756), (B * 784, 144, 40), (B * 3136, 96, 24)]
ops.set_autotune(True)
for M, K, N in shapes:
    x, w, b = torch.randn(M, K, device=d), torch.randn(N, K, device=d) * K ** -0.5, torch.randn(N, device=d)
    w16 = ops.to_bf16(w)
    res = []
    for name, fn in (("f32", lambda: ops.linear(x, w, b)), ("bf16 auto", lambda: ops.linear_bf16(x, w16, b)),
                     ("bf16 128x128", lambda: ops.linear_bf16(x, w16, b, cfg=0)), ("bf16 128x64", lambda: ops.linear_bf16(x, w16, b, cfg=1)),
                     ("bf16 64x64", lambda: ops.linear_bf16(x, w16, b, cfg=2))):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        res.append(f"{name} {us:7.1f} us ({2.0 * M * K * N / us / 1e6:6.1f} TF)")
    print(f"[{M:7d} x {K:4d} x {N:4d}] " + " | ".join(res), flush=True)
