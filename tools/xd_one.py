"""Run the fused expand + depthwise kernel on one EfficientNet-B0 block shape a few times; meant to sit under rocprofv3 --pmc.
usage: xd_one.py <block 1..10> [batch]   (6..10: the whole-image kernel of the 14x14 / 7x7 blocks)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image_captioning_amd import ops
from image_captioning_amd.backbones import same_pad
SHAPES = {1: (112, 16, 96, 3, 2), 2: (56, 24, 144, 3, 1), 3: (56, 24, 144, 5, 2), 4: (28, 40, 240, 5, 1), 5: (28, 40, 240, 3, 2),
          6: (14, 80, 480, 3, 1), 7: (14, 80, 480, 5, 1), 8: (14, 112, 672, 5, 1), 9: (7, 192, 1152, 5, 1), 10: (7, 192, 1152, 3, 1)}
H, Cin, Cexp, ks, s = SHAPES[int(sys.argv[1])]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
d = torch.device("cuda:0")
x = torch.randn(B, H, H, Cin, device=d)
we, wd = torch.randn(Cexp, Cin, device=d) * Cin ** -0.5, torch.randn(ks * ks, Cexp, device=d) / ks
es, eb, ds, db = torch.rand(Cexp, device=d) + 0.5, torch.randn(Cexp, device=d) * 0.1, torch.rand(Cexp, device=d) + 0.5, torch.randn(Cexp, device=d) * 0.1
lo, hi = same_pad(ks, s)
for _ in range(3):
    ops.expand_dwconv2d(x, we, es, eb, wd, ds, db, ks=ks, stride=s, pad=(lo, hi, lo, hi), want_partial=True)
torch.cuda.synchronize()
t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0.record()
for _ in range(10):
    ops.expand_dwconv2d(x, we, es, eb, wd, ds, db, ks=ks, stride=s, pad=(lo, hi, lo, hi), want_partial=True)
t1.record(); torch.cuda.synchronize()
print(f"block {sys.argv[1]}: {t0.elapsed_time(t1) / 10 * 1e3:.1f} us")
