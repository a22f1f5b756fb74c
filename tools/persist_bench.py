"""us/step of the persistent whole-caption decoder against the step-per-launch driver (one caption and NB captions)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import helpers as H
from image_captioning_amd import ops
d = torch.device("cuda:0")


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


for which, dims, steps, params in (("char", [512, 512, 2, 4, 81], 58, 0.57e6), ("word", [512, 2048, 8, 8, 6756], 20, 5.56e6)):
    m = H.tfm_model(which, dims).to(d)
    f = H.features(8, 512, 2).to(d)
    wd = m._greedy_weights()
    ca = m.decoder.cross_attention_single(m.encoder.forward_independent(f))
    st = torch.full((8,), m.start_token, device=d, dtype=torch.int64)
    mx = ops.greedy_tfm_persist_max_nb(dims[0], dims[1], dims[4])
    for NB in sorted({1, min(4, mx), mx}):
        c, s = ca[:NB].contiguous(), st[:NB].contiguous()
        t_p = timeit(lambda: ops.greedy_tfm_persist(wd, c, s, steps))
        t_s = timeit(lambda: ops.greedy_tfm(wd, c, s, steps))
        ok = int(ops.greedy_tfm_persist(wd, c, s, steps)[0].min()) >= 0
        print(f"{which:5s} NB={NB}: persistent {t_p * 1e3 / steps:7.2f} us/step ({params * 4 / (t_p * 1e3 / steps) / 1e3:7.1f} GB/s algorithmic)   "
              f"step-per-launch {t_s * 1e3 / steps:7.2f} us/step   launch ok: {ok}")

# the char model's default path: next-token table + walk
m = H.tfm_model("char", [512, 512, 2, 4, 81]).to(d)
for NB in (1, 8, 64, 256):
    f = torch.cat([H.features(8, 512, 2)] * ((NB + 7) // 8))[:NB].to(d)
    t_t = timeit(lambda: m.greedy_ids_from_features(f, 58))
    print(f"char  NB={NB}: table walk {t_t * 1e3:8.1f} us per call = {t_t * 1e3 / 58:6.2f} us per step-equivalent, {NB * 1e3 / t_t:10.0f} captions/s")
