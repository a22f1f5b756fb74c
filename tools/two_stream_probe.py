"""Does splitting the caption batch over two HIP streams pay?  One stream, B = 1024 per step, against two streams with 512 each
(two model instances with the same weights, so no cached workspace is shared): the latency-bound decode launches of one half
can run beside the throughput-bound encoder launches of the other."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image_captioning_amd import ops
from image_captioning_amd import transformer_word_level as twl
from image_captioning_amd.synthetic import seeded_state_dict, synthetic_images
dev = torch.device("cuda:0")
ops.set_compute_dtype("f32x3")
VOCAB, LEN = 6756, 20
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
NS = int(sys.argv[2]) if len(sys.argv) > 2 else 2


def build():
    m = twl.build_model(VOCAB)
    m.load_state_dict(seeded_state_dict(m, seed=52))
    return m.eval().to(dev)


pool = synthetic_images(32, seed=1234)
images = pool.repeat(B // 32, 1, 1, 1).contiguous().to(dev)
models = [build() for _ in range(NS)]
parts = list(images.chunk(NS))
streams = [torch.cuda.Stream() for _ in range(NS)]


def run(model, x):
    return model.greedy_ids_from_features(model.cnn_model(x), LEN)


with torch.no_grad():
    ops.set_autotune(True)
    for _ in range(2):
        run(models[0], images)
        for m, x in zip(models, parts):
            run(m, x)
    torch.cuda.synchronize()
    ops.set_autotune(False)
    ref = run(models[0], images)
    torch.cuda.synchronize()
    K = 20
    t0 = time.perf_counter()
    for _ in range(K):
        run(models[0], images)
    torch.cuda.synchronize()
    one = (time.perf_counter() - t0) / K
    for s in streams:
        s.wait_stream(torch.cuda.current_stream())
    outs = [None] * NS
    t0 = time.perf_counter()
    for _ in range(K):
        for i, (m, x, s) in enumerate(zip(models, parts, streams)):
            with torch.cuda.stream(s):
                outs[i] = run(m, x)
    torch.cuda.synchronize()
    two = (time.perf_counter() - t0) / K
    same = torch.equal(torch.cat(outs), ref)
print(f"B={B}: one stream {one * 1e3:.2f} ms/step ({B / one:.0f} captions/s) | {NS} streams of {B // NS}: {two * 1e3:.2f} ms/step ({B / two:.0f} captions/s) | ids equal: {same}")
