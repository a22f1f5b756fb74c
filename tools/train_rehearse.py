"""Two (or more) ranks of the data-parallel training step on ONE GPU with the gloo backend: a dry run of the multi-rank
control flow (hooks, buckets, side stream) where only one device is available.  Not a measurement.
Observed on the one-GPU box: completes with REH_BATCH=8 (parameters identical across ranks after 4 steps); with
REH_BATCH=64 both ranks stall inside gloo's all-reduce of GPU tensors from the second step on, with or without hooks
and with the device drained first - a property of this stand-in backend, which is why the run carries its own timeout.
A later run stalled the same way at REH_BATCH=8 (first step and its exchange complete on both ranks, the second step's
exchange never returns), so the batch size is not what decides it.
usage: python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 tools/train_rehearse.py
env: REH_OVERLAP=0|1 (hook-launched buckets), REH_SIDE_STREAM=0|1, REH_STEPS, REH_BATCH"""
import faulthandler
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image_captioning_amd import training  # noqa: E402
from image_captioning_amd import transformer_word_level_bahdnau as bah  # noqa: E402
from image_captioning_amd._bah_train import DropoutState  # noqa: E402
from image_captioning_amd.synthetic import seeded_state_dict, synthetic_captions, synthetic_images  # noqa: E402


def main():
    faulthandler.dump_traceback_later(float(os.environ.get("REH_DUMP", "25")), repeat=False, file=sys.stderr)
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    B = int(os.environ.get("REH_BATCH", "8"))
    steps = int(os.environ.get("REH_STEPS", "4"))
    model = bah.build_model(6756, dropout=0.3)
    model.load_state_dict(seeded_state_dict(model, seed=52))
    model = model.cuda()
    DropoutState.manual_seed(1000 + rank)
    tr = training.Trainer(model, lr=1e-4, weight_decay=1e-5, overlap=os.environ.get("REH_OVERLAP", "0") == "1")
    if os.environ.get("REH_SIDE_STREAM", "1") == "0":
        tr.reducer.comm_stream = None
    images = synthetic_images(B, seed=5 + rank).cuda()
    caps = synthetic_captions(B, 20, 6756, 2, 3, 0, seed=7 + rank).cuda()
    t0 = time.perf_counter()
    for i in range(steps):
        loss = tr.step(images, caps, 0)
        print(f"[r{rank} +{time.perf_counter() - t0:5.1f}s] step {i} enqueued, buckets {len(tr.reducer.buckets)}", file=sys.stderr, flush=True)
    torch.cuda.synchronize()
    # every rank must hold the same parameters after the same number of steps
    flat = tr.opt.flat.flat_p if hasattr(tr.opt.flat, "flat_p") else torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    s = flat.double().sum().reshape(1).cpu()
    both = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(both, s)
    print(f"[r{rank}] loss {float(loss):.5f} param checksum {float(s):.9f} equal across ranks: {all(float(b) == float(both[0]) for b in both)}",
          file=sys.stderr, flush=True)
    dist.destroy_process_group()
    faulthandler.cancel_dump_traceback_later()


if __name__ == "__main__":
    main()
