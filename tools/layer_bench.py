"""Run one conv2d / dwconv2d launch shape repeatedly (for rocprofv3 counter passes and quick A/B timing).

    python tools/layer_bench.py conv  B H W Cin Cout K stride act [reps]
    python tools/layer_bench.py dw    B H W C K stride [reps]
"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image_captioning_amd import ops

d = torch.device("cuda:0")
kind = sys.argv[1]
a = [int(v) for v in sys.argv[2:]]
torch.manual_seed(0)
if kind == "conv":
    B, H, W, Cin, Cout, K, s, act = a[:8]
    reps = a[8] if len(a) > 8 else 20
    x = torch.randn(B, H, W, Cin, device=d)
    w = torch.randn(Cout, K, K, Cin, device=d) * (Cin * K * K) ** -0.5
    sc, bi = torch.rand(Cout, device=d) + 0.5, torch.randn(Cout, device=d) * 0.1
    pad = (K // 2,) * 4 if s == 1 else (0, K // 2, 0, K // 2) if K == 3 else (K // 2,) * 4
    fn = lambda: ops.conv2d(x, w, sc, bi, stride=s, pad=pad, act=act)
    out = fn()
    byts = 4 * (x.numel() + w.numel() + out.numel()); flops = 2 * out.numel() * K * K * Cin
else:
    B, H, W, Cc, K, s = a[:6]
    reps = a[6] if len(a) > 6 else 20
    x = torch.randn(B, H, W, Cc, device=d)
    w = torch.randn(K * K, Cc, device=d) / K
    sc, bi = torch.rand(Cc, device=d) + 0.5, torch.randn(Cc, device=d) * 0.1
    pad = (K // 2,) * 4 if s == 1 else ((K - 2) // 2, (K - 2) - (K - 2) // 2) * 2
    fn = lambda: ops.dwconv2d(x, w, sc, bi, ks=K, stride=s, pad=pad, act=2, want_partial=True)
    out = fn()[0]
    byts = 4 * (x.numel() + out.numel()); flops = 2 * out.numel() * K * K
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    fn()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
print(f"{kind} {a}: {ms:.4f} ms  {byts / ms / 1e6:.0f} GB/s  {flops / ms / 1e9:.2f} TFLOP/s")
