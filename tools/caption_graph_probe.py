"""Does replaying the whole captioning step (encoder + 20 greedy steps, ~120 launches) from one HIP graph beat the eager launches?
python tools/caption_graph_probe.py [B]"""
import os
import sys
import time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from image_captioning_amd import ops
from image_captioning_amd import transformer_word_level as twl
from image_captioning_amd.synthetic import seeded_state_dict, synthetic_images

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = torch.device("cuda:0")
ops.set_compute_dtype("f32x3")
pool = synthetic_images(32, seed=1234)
images = pool.repeat((B + 31) // 32, 1, 1, 1)[:B].contiguous().to(dev)
model = twl.build_model(6756)
model.load_state_dict(seeded_state_dict(model, seed=52))
model = model.eval().to(dev)


def step():
    return model.greedy_ids_from_features(model.cnn_model(images), 20)


def timeit(fn, n=10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for _ in range(4):
    ref = step()
torch.cuda.synchronize()
print(f"eager   {timeit(step):.3f} ms / step")
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    step()
torch.cuda.current_stream().wait_stream(s)
with torch.cuda.graph(g):
    out = step()
g.replay()
torch.cuda.synchronize()
print("ids equal:", bool(torch.equal(out, ref)))
print(f"graph   {timeit(g.replay):.3f} ms / step")
print(f"eager   {timeit(step):.3f} ms / step")
