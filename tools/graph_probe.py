"""Does capturing a greedy driver call in a HIP graph (torch.cuda.CUDAGraph) remove the host launch bound?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import helpers as H
d = torch.device("cuda:0")
kind = sys.argv[1] if len(sys.argv) > 1 else "lstm"
if kind == "lstm":
    m = H.rnn_decoder("lstm", [256, 512, 6756, 2]).to(d)
    f = H.features(1, 256, 1).to(d)
    STEPS = 20
    m.sample_ids_ = m.sample_ids
else:
    m = H.tfm_model("char", [512, 512, 2, 4, 81]).to(d)
    f = H.features(1, 512, 2).to(d)
    STEPS = 58
    m.sample_ids = m.greedy_ids_from_features
ref = m.sample_ids(f, STEPS).clone()
torch.cuda.synchronize()
static_f = f.clone()
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2):
        m.sample_ids(static_f, STEPS)
torch.cuda.current_stream().wait_stream(s)
with torch.cuda.graph(g):
    out = m.sample_ids(static_f, STEPS)
g.replay(); torch.cuda.synchronize()
print("graph result equal:", torch.equal(out, ref))
def timeit(fn, reps=50):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
print("eager  %.1f us/step" % (timeit(lambda: m.sample_ids(f, STEPS)) * 1e3 / STEPS))
print("graph  %.1f us/step" % (timeit(lambda: g.replay()) * 1e3 / STEPS))
