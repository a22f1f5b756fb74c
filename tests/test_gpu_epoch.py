"""GPU: BASELINE configs[4] as one flow, at toy size - recipe CSV + JPEG folder -> FoodImageCaptionDataset -> the reference's
80 / 10 / 10 split -> iterate_batches(rank, world) -> collate_fn with the reference's random augmentations -> Trainer(graph=True)
(two eager steps, the recording, replays) -> greedy captioning of the test split sharded by image -> the same captions from a
fresh model loaded from the trained state_dict (transformer_word_level.py:260-271, transformer_word_level_bahdnau.py:218-233,298)."""
import csv
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

WORDS = ["roast", "chicken", "with", "lemon", "butter", "soup", "salad", "crispy", "potatoes", "miso", "garlic", "bread", "spicy",
         "noodles", "tomato", "basil", "cake", "chocolate", "and", "feta", "avocado", "rub", "chipotle", "cheese"]


def make_dataset(root, n=64, seed=5):
    from PIL import Image
    rng = np.random.RandomState(seed)
    img_dir = os.path.join(root, "Food Images")
    os.makedirs(img_dir)
    rows = []
    for i in range(n):
        h, w = (169, 274) if i % 5 else (180, 240)                 # the data set's usual size, and a second one
        base = rng.randint(0, 256, (h // 8 + 1, w // 8 + 1, 3)).astype(np.uint8)
        pix = np.kron(base, np.ones((8, 8, 1), dtype=np.uint8))[:h, :w]   # blocky: survives JPEG with structure left
        name = f"dish-{i:03d}"
        Image.fromarray(pix).save(os.path.join(img_dir, name + ".jpg"), quality=92)
        title = " ".join(rng.choice(WORDS, size=rng.randint(2, 12)))          # some longer than max_seq_length - 2: <END> is cut
        rows.append({"Title": title, "Image_Name": name})
    rows.append({"Title": "no picture for this one", "Image_Name": "missing"})     # dropped: the file does not exist
    path = os.path.join(root, "recipes.csv")
    with open(path, "w", newline="", encoding="utf-8") as f:
        wr = csv.DictWriter(f, fieldnames=["Title", "Image_Name"])
        wr.writeheader()
        wr.writerows(rows)
    return path, img_dir


def test_mini_epoch_dataset_to_sharded_captions(tmp_path):
    from image_captioning_amd import data, parallel
    from image_captioning_amd import transformer_word_level_bahdnau as bah
    from image_captioning_amd._bah_train import DropoutState
    from image_captioning_amd.synthetic import seeded_state_dict
    from image_captioning_amd.training import Trainer
    dev = torch.device("cuda:0")
    csv_path, img_dir = make_dataset(str(tmp_path))
    ds = data.FoodImageCaptionDataset(csv_path, img_dir, max_seq_length=10)
    assert len(ds) == 64
    tr, va, te = data.train_val_test_indices(len(ds))
    assert len(tr) == 51 and len(va) + len(te) == 13 and not (set(tr) & set(te)) and not (set(tr) & set(va))
    train, test = data.SubsetFoodImageCaptionDataset(ds, tr), data.SubsetFoodImageCaptionDataset(ds, te)
    V = len(ds.vocab.word2idx)
    assert V == 4 + len(WORDS)

    # two ranks' views of one epoch: same number of steps, disjoint samples, full batches only
    data.AUGMENT = True
    try:
        torch.manual_seed(3)
        per_rank = [list(data.iterate_batches(train, batch_size=12, shuffle=True, seed=1, rank=r, world=2)) for r in (0, 1)]
        assert [len(b) for b in per_rank] == [2, 2]
        for imgs, caps in per_rank[0] + per_rank[1]:
            assert imgs.shape == (12, 3, 224, 224) and imgs.is_cuda and imgs.dtype == torch.float32
            assert caps.shape == (12, 10) and (caps[:, 0] == 2).all()
        # augmentation is random per draw: the same batch collated again differs
        again = next(iter(data.iterate_batches(train, batch_size=12, shuffle=True, seed=1, rank=0, world=2)))
        assert not torch.equal(again[0], per_rank[0][0][0]) and torch.equal(again[1], per_rank[0][0][1])

        model = bah.build_model(V, dropout=0.5)
        model.load_state_dict(seeded_state_dict(model, seed=52))
        model = model.to(dev)
        before = {k: v.detach().clone() for k, v in model.state_dict().items()}
        DropoutState.manual_seed(11)
        trainer = Trainer(model, lr=1e-3, weight_decay=1e-5, graph=True)
        losses = []
        for epoch in range(2):                                       # 2 x 4 steps: eager, eager, recording + replay, replays
            for imgs, caps in data.iterate_batches(train, batch_size=12, shuffle=True, seed=epoch, drop_last=True):
                losses.append(float(trainer.step(imgs, caps.to(dev), 0)))
        assert len(losses) == 8 and all(np.isfinite(losses)) and trainer._graph is not None
        assert losses[-1] < losses[0]                                 # AdamW at 1e-3 on 48 samples: the loss moves down
    finally:
        data.AUGMENT = False
    moved = [k for k, v in model.state_dict().items() if v.is_floating_point() and not torch.equal(v, before[k])]
    assert any(k.startswith("cnn_model.") for k in moved) and any(k.startswith("decoder.") for k in moved)

    # test split, sharded by image over two "ranks" (no collective on this path), deterministic transform
    model.eval()
    imgs, caps = data.collate_fn([test[i] for i in range(len(test))])
    n = imgs.shape[0]
    sharded = []
    with torch.no_grad():
        for r in (0, 1):
            lo, hi = parallel.shard_range(n, r, 2)
            sharded += model.caption_ids(imgs[lo:hi], max_len=10)
        fresh = bah.build_model(V, dropout=0.5)
        fresh.load_state_dict(model.state_dict(), strict=True)        # what transformer_word_level_bahdnau.py:298 saves
        fresh = fresh.eval().to(dev)
        want = []
        for r in (0, 1):
            lo, hi = parallel.shard_range(n, r, 2)
            want += fresh.caption_ids(imgs[lo:hi], max_len=10)
        whole = fresh.caption_ids(imgs, max_len=10)
    assert len(sharded) == n and sharded == want          # the trained model's eval path runs on the UPDATED weights
    assert sum(a == b for a, b in zip(whole, want)) >= n - 1          # one batch or two: the images are independent (a near-tie may flip)
    words = [ds.vocab.decode(ids) for ids in sharded]
    assert all(isinstance(w, str) for row in words for w in row)
