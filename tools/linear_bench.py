"""Time ops.linear for one (M, K, N) with every tile configuration forced in turn (via the tuner) or the default."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image_captioning_amd import ops
M, K, N = [int(v) for v in sys.argv[1:4]]
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 50
tune = len(sys.argv) > 5 and sys.argv[5] == "tune"
d = torch.device("cuda:0")
x, w, b = torch.randn(M, K, device=d), torch.randn(N, K, device=d) * K ** -0.5, torch.randn(N, device=d)
ops.set_autotune(tune)
for _ in range(3): ops.linear(x, w, b)
ops.set_autotune(False)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps): ops.linear(x, w, b)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
from image_captioning_amd import _native
print(f"linear {M}x{K}x{N}: {ms*1e3:.1f} us  {2*M*K*N/ms/1e9:.1f} TFLOP/s cfg={_native.load().icap_conv2d_tuned_config(M,1,1,K,N,1,1,1,1,1)}")
