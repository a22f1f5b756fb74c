"""The dense 3x3 / 1x1 layers of VGG-16 and ResNet-18 at B images with bf16 tensors in and out: every tile configuration of
icap_conv2d_bf16_f32 (0-4: the four-wave tile kernel, 5: the 256-row eight-wave kernel with direct-to-LDS loads), TFLOP/s."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image_captioning_amd import ops
d = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ops.set_compute_dtype("bf16")


def timeit(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


shapes = [("vgg 224 64->64", 224, 64, 64, 3, 1), ("vgg 112 64->128", 112, 64, 128, 3, 1), ("vgg 112 128->128", 112, 128, 128, 3, 1),
          ("vgg 56 128->256", 56, 128, 256, 3, 1), ("vgg 56 256->256", 56, 256, 256, 3, 1), ("vgg 28 256->512", 28, 256, 512, 3, 1),
          ("vgg 28 512->512", 28, 512, 512, 3, 1), ("vgg 14 512->512", 14, 512, 512, 3, 1),
          ("res 56 64->64", 56, 64, 64, 3, 1), ("res 56 64->128 s2", 56, 64, 128, 3, 2), ("res 56 64->128 1x1 s2", 56, 64, 128, 1, 2),
          ("res 28 128->128", 28, 128, 128, 3, 1), ("res 28 128->256 s2", 28, 128, 256, 3, 2), ("res 14 256->256", 14, 256, 256, 3, 1),
          ("res 14 256->512 s2", 14, 256, 512, 3, 2), ("res 7 512->512", 7, 512, 512, 3, 1)]
for name, H, Cin, Cout, K, s in shapes:
    x = torch.randn(B, H, H, Cin, device=d).to(torch.bfloat16)
    w = torch.randn(Cout, K, K, Cin, device=d) * (Cin * K * K) ** -0.5
    sc, bi = torch.rand(Cout, device=d) + 0.5, torch.randn(Cout, device=d) * 0.1
    pad = (K // 2,) * 4
    Ho = ops.conv_out_size(H, K, s, pad[0], pad[1])
    flops = 2.0 * B * Ho * Ho * Cout * K * K * Cin
    cache, out = {}, []
    for cfg in (-1, 0, 1, 2, 3, 4, 5):
        try:
            t = timeit(lambda: ops.dense_conv(x, w, cache, "k", sc, bi, stride=s, pad=pad, act=ops.ACT_RELU, out_bf16=True, cfg=cfg))
            out.append(f"{flops / t / 1e9:6.0f}")
        except ops.IcapError:
            out.append("     -")
    print(f"{name:24s} cfg -1 | 0 1 2 3 4 | 5: {out[0]} | " + " ".join(out[1:6]) + f" | {out[6]}  TFLOP/s", flush=True)
