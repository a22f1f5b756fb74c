"""CPU, world_size 2 over gloo: image-sharded captioning gathers to the same list as one process."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from image_captioning_amd import parallel


def fake_caption(image_id: int) -> list:
    """Stands in for the per-image greedy decode: any deterministic function of the image alone."""
    g = torch.Generator().manual_seed(image_id)
    n = int(torch.randint(1, 9, (1,), generator=g))
    return torch.randint(4, 6756, (n,), generator=g).tolist()


def _worker(rank, world, port, n_images, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = parallel.shard_range(n_images, rank, world)
    local = [fake_caption(i) for i in range(lo, hi)]
    dist.barrier()
    out = parallel.gather_captions(local)
    if rank == 0:
        q.put(out)
    else:
        assert out is None
    dist.destroy_process_group()


def test_two_rank_sharding_matches_single_process():
    n_images = 13
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_images, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got == [fake_caption(i) for i in range(n_images)]
    assert parallel.gather_captions([[1, 2]]) == [[1, 2]]      # no process group: identity
