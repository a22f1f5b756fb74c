"""Package-free reproducer for the stall seen when the data-parallel trainer was rehearsed with gloo on ONE GPU: two ranks
share device 0, each iteration launches four async all-reduces of 16 MB slices of one flat HIP buffer and waits for them.
Nothing from image_captioning_amd is imported.  REPRO_BUSY=1 keeps the launch stream busy (a long chain of kernels plus a
memset of the buffer queued right before the exchange, as backward and zero_grad are in the trainer).

    python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 tools/gloo_hip_allreduce_repro.py
"""
import faulthandler, os, sys, time
import torch, torch.distributed as dist

faulthandler.dump_traceback_later(float(os.environ.get("REPRO_DUMP", "40")), repeat=False, file=sys.stderr)
rank = int(os.environ["RANK"])
torch.cuda.set_device(0)
dist.init_process_group("gloo")
n = 4 * (16 << 20) // 4
flat = torch.full((n,), float(rank + 1), device="cuda")
busy = os.environ.get("REPRO_BUSY", "0") == "1"
a = torch.randn(2048, 2048, device="cuda")
t0 = time.perf_counter()
for it in range(int(os.environ.get("REPRO_ITERS", "4"))):
    if busy:
        for _ in range(200):
            a = torch.tanh(a @ a * 1e-3)
        flat.fill_(float(rank + 1))
    hs = [dist.all_reduce(flat[i * (n // 4):(i + 1) * (n // 4)], async_op=True) for i in range(4)]
    for h in hs:
        h.wait()
    torch.cuda.synchronize()
    print(f"[r{rank} +{time.perf_counter() - t0:5.1f}s] iteration {it} done, flat[0]={float(flat[0])} flat[-1]={float(flat[-1])}", file=sys.stderr, flush=True)
    flat.fill_(float(rank + 1))
dist.destroy_process_group()
faulthandler.cancel_dump_traceback_later()
print(f"[r{rank}] finished", file=sys.stderr, flush=True)
