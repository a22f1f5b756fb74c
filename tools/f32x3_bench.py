"""fp32 MFMA kernel vs the split-product (bf16x3, fp32-accurate) kernel on the GEMM shapes of the captioning step."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image_captioning_amd import ops
d = torch.device("cuda:0")


def timeit(fn, reps=30):
    for _ in range(5):
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
shapes = [("vocab", B, 512, 6756), ("fc1", B, 512, 2048), ("fc2", B, 2048, 512), ("head 7x7", B * 49, 320, 1280), ("expand 7x7", B * 49, 192, 1152),
          ("project 7x7", B * 49, 1152, 192), ("expand 14x14", B * 196, 112, 672), ("project 14x14", B * 196, 672, 112),
          ("expand 14x14 b", B * 196, 80, 480), ("project 14x14 b", B * 196, 480, 80), ("project 28x28", B * 784, 240, 40), ("expand 28", B * 784, 40, 240)]
ops.set_autotune(True)
for name, M, K, N in shapes:
    x = torch.randn(M, K, device=d)
    w = torch.randn(N, K, device=d) * K ** -0.5
    b = torch.randn(N, device=d)
    w3 = ops.split3(w)
    ops.linear(x, w, b)
    t32 = timeit(lambda: ops.linear(x, w, b))
    best = None
    for cfg in (0, 1, 2):
        t = timeit(lambda: ops.linear_f32x3(x, w3, b, cfg=cfg))
        best = (t, cfg) if best is None or t < best[0] else best
    tauto = timeit(lambda: ops.linear_f32x3(x, w3, b))
    fl = 2.0 * M * K * N
    err = (ops.linear_f32x3(x, w3, b) - ops.linear(x, w, b)).abs().max().item()
    print(f"{name:16s} [{M:7d},{K:5d},{N:5d}]  fp32 MFMA {t32 * 1e3:8.1f} us {fl / t32 / 1e9:6.1f} TF | split best cfg {best[1]} {best[0] * 1e3:8.1f} us "
          f"{fl / best[0] / 1e9:6.1f} TF (auto {tauto * 1e3:8.1f} us) | x{t32 / best[0]:.2f} | max diff {err:.2e}")
