"""Two (or more) ranks of the data-parallel training step on ONE GPU with the gloo backend: a rehearsal of the multi-rank
control flow (hooks launching buckets during backward, bucket order, the exchange, the optimizer's 1/world factor) where
only one device is available.  Not a measurement.  RCCL refuses two ranks on one device, so gloo stands in; its device
tensors are staged through pinned host memory by GradientReducer (gloo's own device path stalls when the launch stream
is busy: tools/gloo_hip_allreduce_repro.py).

Checks, per rank:
  REH_SAME=1 (default): every rank trains on the SAME batch with the SAME dropout seed, so the averaged gradient equals
      the single-process gradient bit for bit (g + g = 2g, halved exactly for 2 ranks): the parameters after REH_STEPS
      data-parallel steps must equal those of a single-process trainer run first in the same process.
  REH_SAME=0: ranks train on different batches; the parameters must be identical across ranks afterwards.

usage: python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 tools/train_rehearse.py
env: REH_OVERLAP=1|0 (with REH_GRAPH=1: staged three-graph step | exchange after the whole backward), REH_GRAPH=0|1, REH_STEPS, REH_BATCH, REH_VOCAB, REH_BUCKET_MB"""
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


def run(rank, same, steps, B, V, graph, overlap, bucket_mb):
    model = bah.build_model(V, dropout=0.3)
    model.load_state_dict(seeded_state_dict(model, seed=52))
    model = model.cuda()
    DropoutState.manual_seed(1000 if same else 1000 + rank)
    tr = training.Trainer(model, lr=1e-3, weight_decay=1e-5, overlap=overlap, graph=graph, bucket_mb=bucket_mb)
    off = 0 if same else rank
    batches = [(synthetic_images(B, seed=5 + off + 10 * i).cuda(), synthetic_captions(B, 8, V, 2, 3, 0, seed=7 + off + 10 * i).cuda()) for i in range(2)]
    orders = []
    for i in range(steps):
        if os.environ.get("REH_TIMELINE") == "1" and i == steps - 1:
            tr.timeline = []
        loss = tr.step(*batches[i % 2], 0)
        if tr.reducer is not None:
            orders.append((list(tr.reducer.launch_order), tr.reducer.hooked))
    torch.cuda.synchronize()
    return tr.opt.flat.flat_p.detach().clone(), float(loss), tr, orders


def main():
    faulthandler.dump_traceback_later(float(os.environ.get("REH_DUMP", "100")), repeat=False, file=sys.stderr)
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    same = os.environ.get("REH_SAME", "1") == "1"
    steps, B, V = int(os.environ.get("REH_STEPS", "4")), int(os.environ.get("REH_BATCH", "4")), int(os.environ.get("REH_VOCAB", "300"))
    graph = os.environ.get("REH_GRAPH", "0") == "1"
    want = os.environ.get("REH_OVERLAP", "1") == "1"
    overlap = (None if want else False) if graph else want        # graph mode: None = the staged three-graph step
    bucket_mb = float(os.environ.get("REH_BUCKET_MB", "1.0"))
    t0 = time.perf_counter()
    ref = run(rank, same, steps, B, V, graph, overlap, bucket_mb)[0] if same else None      # no process group yet: single process
    dist.init_process_group("gloo")
    flat, loss, tr, orders = run(rank, same, steps, B, V, graph, overlap, bucket_mb)
    red = tr.reducer
    assert red is not None and red.world == world and red.host_staged and len(red.buckets) >= 3
    ok = True
    if overlap and not graph:
        # hooks launched the exchange during backward: buckets go out in the order backward fills them (last parameters first)
        # hooks launched (most of) the exchange during backward, every bucket exactly once, in the same order every step
        hooked = (all(sorted(o) == list(range(len(red.buckets))) for o, _ in orders) and all(o == orders[0][0] for o, _ in orders)
                  and all(2 * h >= len(red.buckets) for _, h in orders))
        ok = ok and hooked
        print(f"[r{rank}] launch order (first 8): {orders[0][0][:8]}, launched by hooks during backward: {orders[0][1]} of {len(red.buckets)}, "
              f"every bucket once and same order each step: {hooked}", file=sys.stderr, flush=True)
    if graph:
        staged = tr._graph_cnn is not None
        ok = ok and staged == want and (not staged or (len(tr._early_buckets) >= 1 and orders[-1][0][:len(tr._early_buckets)] == tr._early_buckets))
        print(f"[r{rank}] graph step staged: {staged} (wanted {want}); buckets launched between the backward graphs: "
              f"{len(tr._early_buckets)} of {len(red.buckets)}", file=sys.stderr, flush=True)
    if graph and tr.timeline:
        torch.cuda.synchronize()
        e_first, e_sent, e_cnn, running = tr.timeline[-1]
        print(f"[r{rank}] staged step timeline (ms after the first backward graph): early buckets' device-to-host copies done at "
              f"{e_first.elapsed_time(e_sent):.3f}, CNN backward graph done at {e_first.elapsed_time(e_cnn):.3f}; CNN backward still running when the "
              f"early buckets' all-reduce returned on the host: {running}", file=sys.stderr, flush=True)
    if same:
        err = float((flat - ref).abs().max())
        scale = float(ref.abs().max())
        ok = ok and err <= 1e-6 * scale
        print(f"[r{rank}] data-parallel vs single-process parameters after {steps} steps: max diff {err:.3e} (max |p| {scale:.3e})", file=sys.stderr, flush=True)
    s = flat.double().sum().reshape(1).cpu()
    both = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(both, s)
    equal = all(float(b) == float(both[0]) for b in both)
    ok = ok and equal and loss == loss
    print(f"[r{rank} +{time.perf_counter() - t0:5.1f}s] loss {loss:.5f} buckets {len(red.buckets)} overlap {red.overlap} graph {graph} "
          f"parameters equal across ranks: {equal}", file=sys.stderr, flush=True)
    dist.barrier()
    dist.destroy_process_group()
    faulthandler.cancel_dump_traceback_later()
    print(f"[r{rank}] REHEARSAL {'OK' if ok else 'FAILED'}", file=sys.stderr, flush=True)
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
