"""CPU, world_size 2 over gloo: the bucketed gradient exchange of the data-parallel training step.

Each rank back-propagates its own local batch through a small torch module whose parameters live in
FlatParameters; after GradientReducer.finish() the flat gradient times grad_scale must equal the gradient of
the mean of the two ranks' losses computed in one process."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from image_captioning_amd.training import FlatParameters, GradientReducer


def make_model():
    torch.manual_seed(0)
    return torch.nn.Sequential(torch.nn.Linear(24, 40), torch.nn.Tanh(), torch.nn.Linear(40, 40), torch.nn.Tanh(),
                               torch.nn.Linear(40, 8))


def batch(rank):
    g = torch.Generator().manual_seed(100 + rank)
    return torch.randn(6, 24, generator=g), torch.randn(6, 8, generator=g)


def _worker(rank, world, port, overlap, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    model = make_model()
    flat = FlatParameters(model.parameters())
    red = GradientReducer(flat, bucket_mb=0.0005, overlap=overlap)   # ~130 floats per bucket: several buckets
    assert len(red.buckets) >= 3
    outs = []
    for step in range(2):
        flat.zero_grad()
        red.prepare()
        x, y = batch(rank + 2 * step)
        ((model(x) - y) ** 2).mean().backward()
        red.finish()
        outs.append((flat.flat_g * flat.grad_scale).clone())
    if rank == 0:
        q.put([o.numpy() for o in outs])
    dist.barrier()
    dist.destroy_process_group()


def _run(overlap):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, overlap, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-process reference: mean over the two ranks' losses
    for step in range(2):
        model = make_model()
        loss = 0
        for r in range(2):
            x, y = batch(r + 2 * step)
            loss = loss + ((model(x) - y) ** 2).mean()
        (loss / 2).backward()
        flat = FlatParameters(make_model().parameters())
        want = torch.zeros(flat.n)
        for p, o in zip(model.parameters(), flat.offsets):
            want[o:o + p.numel()] = p.grad.reshape(-1)
        torch.testing.assert_close(torch.from_numpy(got[step]), want, rtol=1e-5, atol=1e-6)


def test_bucketed_allreduce_overlapped():
    _run(True)


def test_bucketed_allreduce_after_backward():
    _run(False)


def test_flat_parameters_views_and_touched_runs():
    model = make_model()
    before = [p.detach().clone() for p in model.parameters()]
    flat = FlatParameters(model.parameters())
    for p, b in zip(model.parameters(), before):
        assert torch.equal(p.detach(), b)
    flat.zero_grad()
    x, y = batch(0)
    # only the last layer gets a gradient
    h = model[:4](x).detach()
    ((model[4](h) - y) ** 2).mean().backward()
    runs = flat.touched_runs()
    assert runs == [(flat.offsets[4], flat.n)]
    assert flat.flat_g[:flat.offsets[4]].abs().sum() == 0
    assert flat.flat_g[flat.offsets[4]:].abs().sum() > 0
