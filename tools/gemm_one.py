"""Run ops.linear(M, K, N) a few times (tile configuration from ICAP_FORCE_CFG); meant to sit under rocprofv3 --pmc."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image_captioning_amd import ops
M, K, N = [int(v) for v in sys.argv[1:4]]
d = torch.device("cuda:0")
x, w, b = torch.randn(M, K, device=d), torch.randn(N, K, device=d) * K ** -0.5, torch.randn(N, device=d)
for _ in range(5): ops.linear(x, w, b)
torch.cuda.synchronize()
