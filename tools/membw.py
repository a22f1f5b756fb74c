"""Calibrate achievable HBM read / write / copy bandwidth on the box with plain torch ops."""
import torch, time
d = torch.device("cuda:0")
n = 1 << 30  # 4 GiB of fp32
x = torch.empty(n, device=d); y = torch.empty(n, device=d)
def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3
b = n * 4
print("fill  (write only) %.0f GB/s" % (b / t(lambda: x.fill_(1.0)) / 1e9))
print("copy  (r+w)        %.0f GB/s" % (2 * b / t(lambda: y.copy_(x)) / 1e9))
print("sum   (read only)  %.0f GB/s" % (b / t(lambda: x.sum()) / 1e9))
print("add   (2r+1w)      %.0f GB/s" % (3 * b / t(lambda: torch.add(x, y, out=y)) / 1e9))
