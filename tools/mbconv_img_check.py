"""Whole-image fused expand + depthwise layers (14x14 / 7x7) of EfficientNet-B0: the split-product kernel with in-wave
depthwise interleave against the fp32 whole-image kernel - max relative error and time per launch.
python tools/mbconv_img_check.py [B]"""
import os
import sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from image_captioning_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
LAYERS = [(80, 480, 14, 3), (80, 480, 14, 5), (112, 672, 14, 5), (192, 1152, 7, 5), (192, 1152, 7, 3)]
only = os.environ.get("ICAP_ONLY_CIN")
d = torch.device("cuda:0")
for Cin, Cexp, H, ks in LAYERS:
    if only and int(only) != Cin:
        continue
    torch.manual_seed(Cin + ks)
    x = torch.randn(B, H, H, Cin, device=d)
    we = torch.randn(Cexp, Cin, device=d) * Cin ** -0.5
    es, eb = torch.rand(Cexp, device=d) + 0.5, torch.randn(Cexp, device=d) * 0.3
    wd = torch.randn(ks * ks, Cexp, device=d) / ks
    ds, db = torch.rand(Cexp, device=d) + 0.5, torch.randn(Cexp, device=d) * 0.1
    p = ks // 2
    res = {}
    for name, mode, img in (("f32", "f32", False), ("x3img", "f32x3", True)):
        ops.set_compute_dtype(mode)
        ops.IMG_X3 = img
        cache = {}
        run = lambda: ops.expand_dwconv2d(x, we, es, eb, wd, ds, db, ks=ks, stride=1, pad=(p, p, p, p), want_partial=True, cache=cache)
        y, part = run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            run()
        e1.record()
        torch.cuda.synchronize()
        res[name] = (e0.elapsed_time(e1) / 10, y, part)
    ops.set_compute_dtype("f32")
    ref, pref = res["f32"][1], res["f32"][2]
    y, part = res["x3img"][1], res["x3img"][2]
    err = float((y - ref).abs().max() / ref.abs().max())
    perr = float((part - pref).abs().max() / pref.abs().max())
    # partial batch (B not a multiple of the images per workgroup) and a NaN-prefilled output
    print(f"Cin={Cin} k{ks} H={H}: fp32 kernel {res['f32'][0]:.3f} ms, split+interleave {res['x3img'][0]:.3f} ms, "
          f"max rel err y {err:.1e} pooled {perr:.1e}", flush=True)
