"""Time the split-product GEMMs of a decode step at B = 1024 (events over many launches): the vocabulary projection with
arg-max partials, and the two feed-forward layers on each tile configuration.  ICAP_X3_STAGGER is read by the library."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image_captioning_amd import ops
d = torch.device("cuda:0")
torch.manual_seed(0)


def timed(fn, n=200):
    for _ in range(20): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


M = 1024
x, w, b = torch.randn(M, 512, device=d), torch.randn(6756, 512, device=d) * 512 ** -0.5, torch.randn(6756, device=d)
x3, w3 = ops.split3(x), ops.split3(w)
out = [f"stagger {os.environ.get('ICAP_X3_STAGGER', '0'):>3}: vocab {timed(lambda: ops.linear_f32x3_argmax(x3, w3, b)):6.1f} us"]
for K, N in ((512, 2048), (2048, 512)):
    xx, ww, bb = torch.randn(M, K, device=d), torch.randn(N, K, device=d) * K ** -0.5, torch.randn(N, device=d)
    ww3 = ops.split3(ww)
    for cfg in (0, 1, 2):
        out.append(f"[{K}->{N}] cfg{cfg} {timed(lambda: ops.linear_f32x3(xx, ww3, bb, cfg=cfg)):5.1f}")
print("  ".join(out), flush=True)
