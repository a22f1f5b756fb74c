"""CPU: the host-side data logic (vocabularies, splits, BLEU, checkpoints) and the resize oracle against Pillow and the
committed fixture."""
import os
import zlib

import numpy as np
import pytest
import torch

from image_captioning_amd import checkpoint, data as d, datachar as dc, preprocess
from oracle import pil_resize_ref as R
from oracle.make_resize_fixture import RANDOM_CASES, random_case
from tests import helpers as H


def test_vocabularies_match_the_reference_pairs():
    g = H.load("vocab")
    v = d.Vocabulary()
    sents = [["roast", "chicken", "with", "lemon"], ["chicken", "soup"], list("abcdefghij")]
    for s in sents:
        v.add_sentence(s)
    enc = [v.encode(s, 10) for s in sents] + [v.encode(["unknown", "chicken"], 10)]
    assert np.array_equal(np.array(enc), g["word_ids"])
    assert v.decode(enc[1]) == ["<START>", "chicken", "soup", "<END>"] + ["<PAD>"] * 6 and v.decode([999]) == ["<UNK>"]
    assert v.word_count["chicken"] == 2 and len(v.word2idx) == 4 + 15
    cv = dc.Vocabulary(dc.chars)
    titles = ("Miso-Butter Roast Chicken", "Crispy Salt & Pepper Potatoes!")
    encc = [cv.encode(dc.custom_standardization(t), dc.TEXT_MAX_LEN) for t in titles]
    assert np.array_equal(np.array(encc), g["char_ids"])
    assert "".join(cv.decode(encc[0][1:26])) == titles[0].lower() and dc.NUM_CHAR == 81 and dc.char2idx["<PAD>"] == 2
    long = cv.encode(list("x" * 100), 58)
    assert len(long) == 58 and long[0] == 0 and long[-1] == 1                      # <EOS> survives truncation (datachar.py:56)
    assert v.encode(list("abcdefghijk"), 10)[-1] != 3                              # ... but not in the word vocabulary (data.py:146)


def test_custom_standardization():
    assert d.custom_standardization("Miso-Butter Roast Chicken (2 ways)!") == ["misobutter", "roast", "chicken", "ways"]
    assert d.custom_standardization("I cannot; you wanna go, gimme 3") == ["i", "can", "not", "you", "wan", "na", "go", "gim", "me"]
    assert d.custom_standardization("  ") == []
    assert d.custom_standardization("all i wanna") == ["all", "i", "wan", "na"]      # NLTK pads with spaces: the end of the text counts
    assert d.custom_standardization("wannabe") == ["wannabe"]
    # the reference's pattern closes its character class at the first "]" and then splits at "|": one branch can never
    # match (a "^" anchor after a character), the other is the literal text }~] - so nothing else is ever removed, and
    # symbols outside the table become <UNK> at encode time
    assert dc.custom_standardization("Ab-c é€") == list("ab-c é€")
    assert dc.custom_standardization("a€^_`{|}~]b") == list("a€^_`{|b")
    assert dc.Vocabulary(dc.chars).encode(list("a€"), 5) == [0, dc.char2idx["a"], 3, 1, 2]


def test_split_matches_sklearn():
    sk = pytest.importorskip("sklearn.model_selection")
    for n in (10, 101, 13455):
        tr, te = sk.train_test_split(range(n), test_size=0.2, random_state=42)
        va, te2 = sk.train_test_split(te, test_size=0.5, random_state=42)
        a, b, c = d.train_val_test_indices(n)
        assert list(a) == tr and list(b) == va and list(c) == te2
        x = sk.train_test_split(list(range(n)), test_size=0.2, random_state=42)
        y = sk.train_test_split(x[1], test_size=0.02, random_state=42)
        assert d.train_val_split(list(range(n))) == (x[0], y[0], y[1])
    with pytest.raises(ValueError):
        d.split_indices(1, 0.5)


def test_sentence_bleu_hand_cases():
    ref = ["the", "cat", "is", "on", "the", "mat"]
    assert d.sentence_bleu([ref], ref) == pytest.approx(1.0)
    assert d.sentence_bleu([ref], ["the"] * 7, weights=(1, 0, 0, 0)) == pytest.approx(2 / 7)          # clipped counts
    assert d.sentence_bleu([ref], ["dog"], weights=(1, 0, 0, 0)) == 0.0                                  # no unigram overlap
    assert d.sentence_bleu([ref], [], weights=(1, 0, 0, 0)) == 0.0
    # p1 = 3/4, p2 = 1/3, brevity penalty exp(1 - 5/4)
    got = d.sentence_bleu([list("abcde")], list("abxd"), weights=(0.5, 0.5))
    assert got == pytest.approx(np.exp(-0.25) * np.sqrt(0.75 / 3))
    # a precision without matches becomes the smallest positive double: BLEU-4 collapses to ~0 but is not exactly 0
    tiny = d.sentence_bleu([list("abcde")], list("abxd"))
    assert 0.0 < tiny < 1e-70
    assert d.brevity_penalty(5, 6) == 1.0 and d.closest_ref_length([list("ab"), list("abcd")], 3) == 2


def test_checkpoint_round_trip(tmp_path):
    m = H.rnn_decoder("lstm", (16, 32, 29, 2))
    p = str(tmp_path / "RESNET18_LSTM_model.pth")
    checkpoint.save_checkpoint(m, p)
    raw = torch.load(p, weights_only=True)
    assert list(raw) == list(m.state_dict())                                           # bare state_dict, reference key order
    m2 = H.rnn_decoder("lstm", (16, 32, 29, 2))
    with torch.no_grad():
        for q in m2.parameters():
            q.zero_()
    checkpoint.load_checkpoint(m2, p)
    assert all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), m2.state_dict().values()))
    torch.save({"model": raw}, p)
    with pytest.raises(ValueError):
        checkpoint.load_checkpoint(m2, p)
    v = d.Vocabulary()
    v.add_sentence(["roast", "chicken", "roast"])
    checkpoint.save_vocab(v, str(tmp_path / "v.json"))
    v2 = checkpoint.load_vocab(str(tmp_path / "v.json"))
    assert v2.word2idx == v.word2idx and v2.idx2word == v.idx2word and v2.word_count["roast"] == 2
    cv = dc.Vocabulary(dc.chars)
    checkpoint.save_vocab(cv, str(tmp_path / "c.json"))
    assert checkpoint.load_vocab(str(tmp_path / "c.json")).char2idx == cv.char2idx


def test_dataset_reads_csv_and_images(tmp_path):
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(0)
    for name, (h, w) in {"a": (20, 30), "b": (25, 25)}.items():
        Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)).save(tmp_path / f"{name}.jpg")
    with open(tmp_path / "t.csv", "w", encoding="utf-8") as f:
        f.write(",Title,Ingredients,Image_Name\n0,Roast Chicken,x,a\n1,,x,b\n2,Chicken Soup (2),x,b\n3,Missing Image,x,zzz\n")
    ds = d.FoodImageCaptionDataset(str(tmp_path / "t.csv"), str(tmp_path))
    assert len(ds) == 2 and ds.vocab.word2idx["soup"] == 6
    img, cap = ds[1]
    assert img.shape == (25, 25, 3) and img.dtype == np.uint8 and cap.tolist() == [2, 5, 6, 3, 0, 0, 0, 0, 0, 0]
    dsc = dc.FoodImageCaptionDataset(str(tmp_path / "t.csv"), str(tmp_path))
    assert dsc[0][1].shape == (58,) and dsc[0][1][:3].tolist() == [0, dc.char2idx["r"], dc.char2idx["o"]]
    sub = d.SubsetFoodImageCaptionDataset(ds, [1])
    assert len(sub) == 1 and sub[0][1].tolist() == cap.tolist()


def test_dataset_title_filter_is_pandas_dropna(tmp_path):
    """data.py:26,31: read_csv + dropna(subset=['Title']) - pandas' default NA strings go, a whitespace-only title stays."""
    pd = pytest.importorskip("pandas")
    Image = pytest.importorskip("PIL.Image")
    Image.fromarray(np.zeros((8, 8, 3), np.uint8)).save(tmp_path / "a.jpg")
    titles = ["Roast", "NA", "null", "None", "  ", "n/a", "nan", "Soup", "", "NULL", "N/A", "na", "Null"]
    with open(tmp_path / "t.csv", "w", encoding="utf-8") as f:
        f.write(",Title,Ingredients,Image_Name\n" + "".join(f"{i},{t},x,a\n" for i, t in enumerate(titles)))
    want = pd.read_csv(tmp_path / "t.csv").dropna(subset=["Title"])["Title"].tolist()
    ds = d.FoodImageCaptionDataset(str(tmp_path / "t.csv"), str(tmp_path))
    assert [r["Title"] for r in ds.data] == want
    assert [r["Title"] for r in dc.FoodImageCaptionDataset(str(tmp_path / "t.csv"), str(tmp_path)).data] == want


@pytest.mark.parametrize("n,bs,world", [(7, 1, 4), (25, 4, 2), (12, 4, 3), (5, 2, 1), (3, 4, 2)])
def test_iterate_batches_gives_every_rank_the_same_number_of_steps(n, bs, world):
    """Each training step ends in an all-reduce all ranks join: unequal step counts would leave ranks waiting forever."""
    data = list(range(n))
    per_rank = [list(d.iterate_batches(data, batch_size=bs, collate=list, rank=r, world=world)) for r in range(world)]
    assert len({len(b) for b in per_rank}) == 1
    seen = [i for b in per_rank for batch in b for i in batch]
    assert len(seen) == len(set(seen))
    if world > 1:
        assert all(len(batch) == bs for b in per_rank for batch in b)          # short batches dropped by default
    else:
        assert sorted(seen) == data                                            # DataLoader default: nothing dropped
    with pytest.raises(ValueError):
        next(d.iterate_batches(data, rank=world, world=world))


# ------------------------------------------------------------------ the resize oracle

def test_resize_oracle_matches_the_fixture():
    g = H.load("pil_resize")
    for name in ("photo_169x274", "photo_240x240"):
        assert np.array_equal(R.resize_bilinear_u8(g[f"{name}_in"]), g[f"{name}_out"])
    for h, w in RANDOM_CASES:
        out = R.resize_bilinear_u8(random_case(h, w))
        assert zlib.crc32(out.tobytes()) == int(g[f"rand_{h}x{w}_crc"]) and np.array_equal(out[::4, ::4], g[f"rand_{h}x{w}_sub"])
    norm = R.to_tensor_normalize(g["photo_169x274_out"])
    assert np.array_equal(norm, g["photo_169x274_norm"])                              # fp32, bit for bit


def test_resize_oracle_matches_pillow_directly():
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(5)
    for (h, w), size in [((169, 274), 224), ((300, 180), 256), ((225, 223), 224), ((31, 977), 224), ((224, 224), 224)]:
        a = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        ref = np.asarray(Image.fromarray(a).resize((size, size), Image.BILINEAR))
        assert np.array_equal(R.resize_bilinear_u8(a, size), ref)


def test_host_tap_tables_agree_with_the_oracle():
    for n_in, n_out in [(274, 224), (169, 224), (224, 224), (4096, 224), (1, 224), (300, 256)]:
        taps, bounds = preprocess.bilinear_taps(n_in, n_out)
        ref = R.axis_taps(n_in, n_out)
        assert taps.shape[0] == n_out and taps.dtype == np.int32
        for i, (lo, k) in enumerate(ref):
            assert bounds[i, 0] == lo and bounds[i, 1] == len(k) and np.array_equal(taps[i, :len(k)], k) and not taps[i, len(k):].any()
        assert abs(int(taps.sum(1).min()) - (1 << 22)) <= taps.shape[1] and abs(int(taps.sum(1).max()) - (1 << 22)) <= taps.shape[1]
    with pytest.raises(Exception):
        preprocess.bilinear_taps(0, 224)
    if not torch.cuda.is_available():
        with pytest.raises(preprocess.N.IcapError):                                  # no CPU fallback
            preprocess.resize_normalize(np.zeros((4, 4, 3), np.uint8))


# ------------------------------------------------------------------ the augmentation oracle (flip, rotate, contrast)
def _augment_cases():
    g = H.load("pil_augment")
    for i, (h, w, flip) in enumerate(g["cases"]):
        yield g[f"in{i}"], bool(flip), float(g["angles"][i]), float(g["contrasts"][i]), g[f"out{i}"]


def test_augment_oracle_matches_the_pillow_fixture():
    from oracle import pil_augment_ref as A
    n = 0
    for img, flip, angle, fac, want in _augment_cases():
        assert np.array_equal(A.augment(img, flip, angle, fac), want), (img.shape, flip, angle, fac)
        n += 1
    assert n == 10


def test_augment_oracle_matches_pillow_directly():
    Image = pytest.importorskip("PIL.Image")
    from PIL import ImageEnhance
    from oracle import pil_augment_ref as A
    rng = np.random.default_rng(11)
    for trial in range(40):
        h, w = int(rng.integers(3, 120)), int(rng.integers(3, 120))
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        flip, angle, fac = bool(rng.integers(0, 2)), float(rng.uniform(-20, 20)), float(rng.uniform(0.7, 1.3))
        p = Image.fromarray(img)
        if flip:
            p = p.transpose(Image.FLIP_LEFT_RIGHT)
        p = ImageEnhance.Contrast(p.rotate(angle, resample=Image.NEAREST, expand=False, fillcolor=0)).enhance(fac)
        assert np.array_equal(A.augment(img, flip, angle, fac), np.asarray(p)), (h, w, flip, angle, fac)


def test_host_rotation_coefficients_agree_with_the_oracle():
    """preprocess.rotation_coeffs (what the kernel is given) against the oracle's restatement of Pillow, including the
    angles Pillow handles by exact transposes, which the host expresses as exact integer maps."""
    from oracle import pil_augment_ref as A
    from image_captioning_amd.preprocess import rotation_coeffs
    rng = np.random.default_rng(3)
    for w, h, angle in [(53, 37, 13.7), (37, 53, -19.99), (24, 24, 90.0), (24, 24, 270.0), (45, 31, 180.0), (31, 45, 0.0),
                        (48, 64, 90.0), (7, 5, 360.0 - 20.0), (9, 1, 3.0)]:
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        a0, a1, a2, a3, a4, a5 = rotation_coeffs(w, h, angle)
        ys, xs = np.mgrid[0:h, 0:w].astype(np.int64)
        sx, sy = (a2 + xs * a0 + ys * a1) >> 16, (a5 + xs * a3 + ys * a4) >> 16
        ok = (sx >= 0) & (sx < w) & (sy >= 0) & (sy < h)
        got = np.zeros_like(img)
        got[ok] = img[sy[ok], sx[ok]]
        assert np.array_equal(got, A.rotate_nearest(img, angle)), (w, h, angle)


def test_augment_draws_follow_the_documented_order():
    from image_captioning_amd.preprocess import draw_augment_params
    torch.manual_seed(5)
    flips, angles, factors = draw_augment_params(3)
    torch.manual_seed(5)
    for i in range(3):
        assert flips[i] == bool(torch.rand(1) < 0.5)
        assert angles[i] == float(torch.empty(1).uniform_(-20.0, 20.0))
        torch.randperm(4)
        assert factors[i] == float(torch.empty(1).uniform_(0.7, 1.3))
    assert all(-20 <= a <= 20 for a in angles) and all(0.7 <= f <= 1.3 for f in factors)
