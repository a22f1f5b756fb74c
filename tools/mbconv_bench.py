"""Time the five fused expand+depthwise layers of EfficientNet-B0 at 224x224 in each product mode.
python tools/mbconv_bench.py [B]"""
import os
import sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from image_captioning_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
LAYERS = [(16, 96, 112, 3, 2, (0, 1, 0, 1)), (24, 144, 56, 3, 1, (1, 1, 1, 1)), (24, 144, 56, 5, 2, (1, 2, 1, 2)),
          (40, 240, 28, 5, 1, (2, 2, 2, 2)), (40, 240, 28, 3, 2, (0, 1, 0, 1)),
          (80, 480, 14, 3, 1, (1, 1, 1, 1)), (80, 480, 14, 5, 1, (2, 2, 2, 2)), (112, 672, 14, 5, 1, (2, 2, 2, 2)),
          (192, 1152, 7, 5, 1, (2, 2, 2, 2)), (192, 1152, 7, 3, 1, (1, 1, 1, 1))]
if os.environ.get("ICAP_MB_EARLY"):
    LAYERS = LAYERS[:5]
if os.environ.get("ICAP_MB_LATE"):
    LAYERS = LAYERS[5:]
d = torch.device("cuda:0")
for Cin, Cexp, H, ks, s, pad in LAYERS:
    x = torch.randn(B, H, H, Cin, device=d)
    we = torch.randn(Cexp, Cin, device=d) * Cin ** -0.5
    es, eb = torch.rand(Cexp, device=d) + 0.5, torch.randn(Cexp, device=d) * 0.3
    wd = torch.randn(ks * ks, Cexp, device=d) / ks
    ds, db = torch.rand(Cexp, device=d) + 0.5, torch.randn(Cexp, device=d) * 0.1
    out = {}
    for mode in ("f32", "f32x3", "bf16")[:1 if H <= 14 else 3]:
        ops.set_compute_dtype(mode)
        run = lambda: ops.expand_dwconv2d(x, we, es, eb, wd, ds, db, ks=ks, stride=s, pad=pad, want_partial=True)
        y, _ = run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            run()
        e1.record()
        torch.cuda.synchronize()
        out[mode] = (e0.elapsed_time(e1) / 10, y.float())
    ops.set_compute_dtype("f32")
    ref = out["f32"][1]
    err = {m: float((out[m][1] - ref).abs().max() / ref.abs().max()) if m in out else 0.0 for m in ("f32x3", "bf16")}
    gb = (x.numel() + out["f32"][1].numel()) * 4 / 1e9
    print(f"Cin={Cin} k{ks} s{s} H={H}: " + "  ".join(f"{m} {out[m][0]:.3f} ms ({gb / out[m][0] * 1e3:.0f} GB/s)" for m in out)
          + f"  rel.err x3 {err['f32x3']:.1e} bf16 {err['bf16']:.1e}", flush=True)
