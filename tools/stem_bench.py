"""The fused EfficientNet stem + block-0 depthwise layer at B images of 224 x 224: the matrix-core kernel on the padded NHWC image,
the same on the planar image, and the vector-ALU kernel."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image_captioning_amd import ops
d = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
which = sys.argv[2] if len(sys.argv) > 2 else "all"


def timeit(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


C = 32
x = torch.randn(B, 3, 224, 224, device=d)
ws4 = torch.nn.functional.pad(torch.randn(C, 3, 3, 3, device=d) * 27 ** -0.5, (0, 1)).contiguous()
ss, sb = torch.rand(C, device=d) + 0.5, torch.randn(C, device=d) * 0.3
wd = (torch.randn(9, C, device=d) / 3).contiguous()
dsc, dbi = torch.rand(C, device=d) + 0.5, torch.randn(C, device=d) * 0.1
wk = ops.stem_weight_k(ws4)
kw = dict(stem_pad=(0, 1, 0, 1), pad=(1, 1, 1, 1), want_partial=True)
byts = 4.0 * B * (3 * 224 * 224 + 112 * 112 * C)
if which in ("all", "mfma"):
    x4 = ops.nchw_to_nhwc_pad(x, 4)
    t = timeit(lambda: ops.stem_dwconv2d(x4, ws4, ss, sb, wd, dsc, dbi, **kw))
    print(f"matrix cores, padded NHWC image  {t * 1e3:8.1f} us")
    t = timeit(lambda: ops.stem_dwconv2d(x, ws4, ss, sb, wd, dsc, dbi, nchw=True, **kw))
    print(f"matrix cores, planar image       {t * 1e3:8.1f} us = {byts / t / 1e6:7.1f} GB/s")
if which in ("all", "valu"):
    t = timeit(lambda: ops.stem_dwconv2d(x, ws4, ss, sb, wd, dsc, dbi, nchw=True, w_stem_k=wk, **kw))
    print(f"vector ALUs, planar image        {t * 1e3:8.1f} us = {byts / t / 1e6:7.1f} GB/s")
