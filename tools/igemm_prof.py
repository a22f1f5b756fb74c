"""Per-K-step phase durations (shader clock ticks, wave 0 of the first 64 workgroups) of the implicit-GEMM kernel.
usage: ICAP_FORCE_CFG=<cfg without prefetch depth> igemm_prof.py M K N"""
import ctypes, os, sys
import numpy as np, torch
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, "igemm_prof.so"))
M, K, N = [int(v) for v in sys.argv[1:4]]
d = torch.device("cuda:0")
x, w, b, y = torch.randn(M, K, device=d), torch.randn(N, K, device=d), torch.randn(N, device=d), torch.empty(M, N, device=d)
P = lambda t: ctypes.c_void_p(t.data_ptr())
for _ in range(3):
    assert lib.ig_linear(P(x), P(w), P(b), P(y), M, K, N) == 0
out = np.zeros(64 * 32 * 4, dtype=np.uint64)
assert lib.ig_read(out.ctypes.data_as(ctypes.c_void_p)) == 0
t = out.reshape(64, 32, 4).astype(np.int64)
nk = min(32, (K + 15) // 16)
t = t[:, :nk]
t = t[t[:, 2, 0] > 0]                      # workgroups that ran
if (t[:, 3, 0] == 0).all():                # fragment-pipelined configurations stamp even steps only
    t = t[:, 0:nk - 4:2]
    nk = t.shape[1]
    print("(even steps only: 'step to step' spans two steps)")
print("per K step, mean over workgroups and steps 2..nk-2 (ticks):")
print("  issue loads + MFMAs :", (t[:, 2:nk - 1, 1] - t[:, 2:nk - 1, 0]).mean())
print("  wait loads + LDS store:", (t[:, 2:nk - 1, 2] - t[:, 2:nk - 1, 1]).mean())
print("  barrier              :", (t[:, 2:nk - 1, 3] - t[:, 2:nk - 1, 2]).mean())
print("  step to step         :", (t[:, 3:nk - 1, 0] - t[:, 2:nk - 2, 0]).mean())
print("  whole loop           :", (t[:, nk - 1, 3] - t[:, 0, 0]).mean())
