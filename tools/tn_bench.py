"""Time the weight-gradient GEMM dW = dY^T X (train_ops.gemm_tn) on the shapes of the training step.
python tools/tn_bench.py [K M N ...]   (default: the EfficientNet / decoder shapes at batch 256); run it under
rocprofv3 --kernel-trace --stats to split a launch into its partial-tile kernel and its split reduction."""
import os
import sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from image_captioning_amd import train_ops as T

SHAPES = [(50176, 672, 112), (50176, 112, 672), (12544, 1152, 192), (12544, 192, 1152), (50176, 480, 80), (5120, 2048, 512),
          (5120, 6756, 512), (12544, 1280, 320), (802816, 24, 144), (200704, 40, 144), (200704, 40, 240), (200704, 240, 40),
          (3211264, 96, 16), (3211264, 16, 32), (802816, 144, 24)]
a = sys.argv[1:]
if a:
    SHAPES = [tuple(int(v) for v in a[i:i + 3]) for i in range(0, len(a), 3)]
d = torch.device("cuda:0")
for K, M, N in SHAPES:
    x, y = torch.randn(K, M, device=d), torch.randn(K, N, device=d)
    for _ in range(2):
        T.gemm_tn(x, y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        T.gemm_tn(x, y)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"gemm_tn[{K}, {M}, {N}]  {ms * 1e3:8.1f} us  {2.0 * K * M * N / ms / 1e9:7.1f} TFLOP/s  {4.0 * K * (M + N) / ms / 1e6:7.0f} GB/s", flush=True)
    del x, y
