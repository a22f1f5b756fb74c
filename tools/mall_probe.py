"""Does a producer -> consumer pair run faster when the intermediate fits the 256 MB memory-side cache?
Times y = f(x); z = g(y) style copies at several working-set sizes (plain torch ops, bandwidth only)."""
import torch
d = torch.device("cuda:0")
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3
for mb in (8, 16, 32, 64, 96, 128, 192, 256, 512, 1024, 2048):
    n = mb * (1 << 20) // 4
    x = torch.empty(n, device=d).normal_(); y = torch.empty_like(x); z = torch.empty_like(x)
    def chain():
        torch.mul(x, 2.0, out=y)   # read x, write y
        torch.mul(y, 0.5, out=z)   # read y, write z
    s = t(chain)
    print("tensor %5d MB: chain of two r+w passes %.0f GB/s (%.1f us)" % (mb, 4 * n * 4 / s / 1e9, s * 1e6))
