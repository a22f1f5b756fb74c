"""One epoch of the word-level Bahdanau captioner on a data set of the reference's size, then greedy captioning of
its test split: the shape of `transformer_word_level_bahdnau.py:310-332` (13 455 usable images, 80 / 10 / 10 split,
loader batch 12, AdamW 1e-4 / 1e-5) with synthetic pixels and titles - the CSV of titles is absent from the reference
and no data set travels to the GPU box.

Per training step and rank: 12 decoded 169 x 274 uint8 images (the data set's typical size) already on the device ->
resize + normalise on the GPU -> Trainer.step (train-mode forward, compute_loss, backward, gradient exchange when a
process group is up, AdamW).  Then the 1 346 test images, sharded by rank, are resized and captioned greedily
(20 tokens, batches of --caption-batch).  JPEG decoding (host, PIL) is not part of it.

    python tools/epoch_bench.py [--batch 12] [--dtype f32|bf16] [--graph] [--images 13455]
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 tools/epoch_bench.py

Prints one JSON object (rank 0).
"""
import argparse
import json
import math
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

VOCAB, TOKENS = 6756, 20


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=13455)
    ap.add_argument("--batch", type=int, default=12, help="loader batch per GPU (the reference's: 12)")
    ap.add_argument("--caption-batch", type=int, default=256)
    ap.add_argument("--dtype", choices=("f32", "bf16"), default="f32")
    ap.add_argument("--graph", action="store_true", help="record the training step into HIP graphs (Trainer(graph=True))")
    ap.add_argument("--max-steps", type=int, default=0, help="stop the epoch early (0 = whole epoch)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)

    from image_captioning_amd import ops, parallel, preprocess
    from image_captioning_amd import transformer_word_level_bahdnau as bah
    from image_captioning_amd._bah_train import DropoutState
    from image_captioning_amd.synthetic import seeded_state_dict, synthetic_captions
    from image_captioning_amd.training import Trainer

    n_train = int(args.images * 0.8)
    n_test = args.images - n_train - int(args.images * 0.1)
    B = args.batch
    steps = math.ceil(n_train / (B * world))
    if args.max_steps:
        steps = min(steps, args.max_steps)

    g = torch.Generator().manual_seed(4321 + rank)
    pool = torch.randint(0, 256, (4 * B, 169, 274, 3), generator=g, dtype=torch.uint8).to(dev)   # decoded pixels, on the device
    caps = synthetic_captions(4 * B, TOKENS, VOCAB, 2, 3, 0, seed=99 + rank).to(dev)
    model = bah.build_model(VOCAB, dropout=0.5)                         # :329-332 (dropout 0.5)
    model.load_state_dict(seeded_state_dict(model, seed=52))
    model = model.to(dev)
    DropoutState.manual_seed(1000 + rank)
    trainer = Trainer(model, lr=1e-4, weight_decay=1e-5, dtype=args.dtype, graph=args.graph)

    def train_step(i):
        o = (i % 4) * B
        images = preprocess.resize_normalize(pool[o:o + B], 224)
        return trainer.step(images, caps[o:o + B], 0)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    ops.set_autotune(True)
    train_step(0)
    torch.cuda.synchronize()
    ops.set_autotune(False)
    for i in range(3):
        train_step(i)
    barrier()
    t0 = time.perf_counter()
    for i in range(steps):
        loss = train_step(i)
    barrier()
    t_train = time.perf_counter() - t0

    # test-split captioning, image-sharded, no collective
    lo, hi = parallel.shard_range(n_test, rank, world)
    CB = args.caption_batch
    big = pool.repeat((CB + pool.shape[0] - 1) // pool.shape[0], 1, 1, 1)[:CB].contiguous()
    model.eval()

    def caption(n):
        images = preprocess.resize_normalize(big[:n], 224)
        feats = model.cnn_model(images)
        return model.greedy_ids_from_features(feats, TOKENS)

    with torch.no_grad():
        ops.set_autotune(True)
        caption(CB)
        if (hi - lo) % CB:
            caption((hi - lo) % CB)
        torch.cuda.synchronize()
        ops.set_autotune(False)
        barrier()
        t1 = time.perf_counter()
        done = lo
        while done < hi:
            n = min(CB, hi - done)
            ids = caption(n)
            done += n
        barrier()
        t_cap = time.perf_counter() - t1
    if world > 1:
        t = torch.tensor([t_train, t_cap], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        t_train, t_cap = (float(v) for v in t.tolist())
    if rank == 0:
        print(json.dumps({
            "workload": "word-level Bahdanau captioner: one training epoch + greedy captioning of the test split, synthetic pixels and titles "
                        "of the reference data set's shape (BASELINE configs[4])",
            "n_gpus": world, "dtype": args.dtype, "hip_graph": bool(args.graph), "images": args.images, "train_images": n_train, "test_images": n_test,
            "loader_batch_per_gpu": B, "steps": steps, "epoch_train_s": t_train, "train_imgs_per_s": steps * B * world / t_train,
            "ms_per_step": t_train / steps * 1e3, "loss_last_step": float(loss), "caption_s": t_cap,
            "captions_per_s": n_test / t_cap, "caption_batch": CB, "caption_tokens": TOKENS, "ids_shape": list(ids.shape),
            "includes": "GPU resize + normalise of decoded 169x274 uint8 images; excludes JPEG decoding on the host"}), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
