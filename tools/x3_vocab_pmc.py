"""The decoder's vocabulary projection alone (pre-split activations [3, 1024, 512] x split weights [3, 6756, 512], arg-max
partials in the epilogue), a few launches; meant to sit under rocprofv3 --pmc or --kernel-trace."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image_captioning_amd import ops
M, K, N = 1024, 512, 6756
d = torch.device("cuda:0")
torch.manual_seed(0)
x, w, b = torch.randn(M, K, device=d), torch.randn(N, K, device=d) * K ** -0.5, torch.randn(N, device=d)
x3, w3 = ops.split3(x), ops.split3(w)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
    ops.linear_f32x3_argmax(x3, w3, b)
torch.cuda.synchronize()
