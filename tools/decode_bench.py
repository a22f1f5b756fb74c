"""Per-step time and achieved weight-streaming bandwidth of the small-batch (lock-step <= 8) greedy decoders.

    python tools/decode_bench.py [NB]

Algorithmic bytes per step = parameters touched x 4 (fp32), the figures of the hot-path scope table:
LSTM 7.80 M, GRU 6.75 M, word transformer 5.56 M, Bahdanau 6.77 M, char transformer 0.57 M parameters.
"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import helpers as H

NB = int(sys.argv[1]) if len(sys.argv) > 1 else 1
d = torch.device("cuda:0")
V = 6756


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def report(name, ms, steps, params):
    us = ms * 1e3 / steps
    print(f"{name:12s} NB={NB}: {ms:8.3f} ms / {steps} steps = {us:7.2f} us/step  {params * 4 / us / 1e3:7.1f} GB/s "
          f"({params * 4 / us / 1e3 / 8000 * 100:4.1f}% of 8 TB/s)  {NB * 1e3 / ms:9.0f} captions/s")


for kind, params in (("lstm", 7.80e6), ("gru", 6.75e6)):
    m = H.rnn_decoder(kind, [256, 512, V, 2]).to(d)
    f = H.features(NB, 256, 1).to(d)
    report(kind, timeit(lambda: m.sample_ids(f, 20)), 20, params)
m = H.tfm_model("word", [512, 2048, 8, 8, V]).to(d)
f = H.features(NB, 512, 2).to(d)
report("word-tfm", timeit(lambda: m.greedy_ids_from_features(f, 20)), 20, 5.56e6)
m = H.tfm_model("char", [512, 512, 2, 4, 81]).to(d)
report("char-tfm", timeit(lambda: m.greedy_ids_from_features(f, 58)), 58, 0.57e6)
m = H.bah_model([512, 512, 512, 256, V, 512]).to(d)
report("bahdanau", timeit(lambda: m.greedy_ids_from_features(f, 20)), 20, 6.77e6)
