"""Host-side data logic of the word-level models, mirroring the reference's ``data.py`` without its third-party
imports (pandas, sklearn, nltk, spacy, torchvision): ``Vocabulary``, ``custom_standardization``, ``collate_fn``, the
dataset classes, the train / validation / test split, and the metrics ``compute_loss`` / ``compute_accuracy`` /
``compute_bleu``.  Same names, arguments and results as data.py:24-173,244-276.

What differs, on purpose:

* ``image_transforms`` runs on the GPU.  By default it is the deterministic tail only (Resize + ToTensor + Normalize,
  data.py:170-172, ``preprocess.resize_normalize``), so evaluation inputs are reproducible; the reference's random
  flip / rotation / contrast jitter (data.py:167-169, ``preprocess.augment``) are applied with ``augment=True`` or, for
  everything that goes through ``collate_fn``, by setting the module switch ``AUGMENT = True`` (training loaders).
* the dataset returns decoded uint8 pixels and the collate function batches same-sized images through one resize
  launch; there is no per-item ``print`` (data.py:54).
* ``custom_standardization`` ends in NLTK's ``word_tokenize`` in the reference.  After lower-casing and removing
  everything but word characters and white space, what is left of that tokenizer is white-space splitting plus the
  Treebank contraction rules that need no apostrophe (cannot, gimme, gonna, gotta, lemme, wanna); those are restated
  here.  NLTK is not installed in this image, so the restatement is checked on hand-written cases only.
"""
from __future__ import annotations

import csv
import math
import os
import re
from collections import Counter, defaultdict

import numpy as np
import torch

from . import ops
from .training import compute_loss  # noqa: F401  (data.py:244-252: same name, HIP kernel behind it)

_SPLITS = [(re.compile(r"\b(can)(not)\b"), r"\1 \2"), (re.compile(r"\b(gim)(me)\b"), r"\1 \2"),
           (re.compile(r"\b(gon)(na)\b"), r"\1 \2"), (re.compile(r"\b(got)(ta)\b"), r"\1 \2"),
           (re.compile(r"\b(lem)(me)\b"), r"\1 \2"), (re.compile(r"\b(wan)(na)(?=\s|$)"), r"\1 \2")]   # NLTK pads the text with spaces first: a final 'wanna' splits too


# What pandas.read_csv turns into NaN by default (data.py:26), and dropna(subset=['Title']) (data.py:31) then removes;
# a whitespace-only title is not among them and stays.
_PANDAS_NA = frozenset(["", "#N/A", "#N/A N/A", "#NA", "-1.#IND", "-1.#QNAN", "-NaN", "-nan", "1.#IND", "1.#QNAN", "<NA>",
                        "N/A", "NA", "NULL", "NaN", "None", "n/a", "nan", "null"])


def custom_standardization(input_string: str) -> list[str]:
    """Lower-case, drop punctuation and digits, tokenize (data.py:98-115)."""
    s = input_string.lower()
    s = re.sub(r"[^\w\s]", "", s)
    s = re.sub(r"\d+", "", s)
    for pat, rep in _SPLITS:
        s = pat.sub(rep, s)
    return s.split()


class Vocabulary:
    """data.py:117-151."""

    def __init__(self):
        self.word2idx = {}
        self.idx2word = {}
        self.word_count = defaultdict(int)
        self.pad_token = "<PAD>"
        self.unk_token = "<UNK>"
        self.start_token = "<START>"
        self.end_token = "<END>"
        self.build_vocab()

    def build_vocab(self):
        self.word2idx = {self.pad_token: 0, self.unk_token: 1, self.start_token: 2, self.end_token: 3}
        self.idx2word = {idx: word for word, idx in self.word2idx.items()}

    def add_sentence(self, sentence):
        for word in sentence:
            self.word_count[word] += 1
            if word not in self.word2idx:
                idx = len(self.word2idx)
                self.word2idx[word] = idx
                self.idx2word[idx] = word

    def encode(self, sentence, max_length):
        """<START> + words + <END>, cut to max_length (the <END> may be cut off, as in the reference), padded with <PAD>."""
        tokens = [self.start_token] + list(sentence) + [self.end_token]
        unk = self.word2idx[self.unk_token]
        ids = [self.word2idx.get(w, unk) for w in tokens]
        return ids[:max_length] + [self.word2idx[self.pad_token]] * (max_length - len(ids))

    def decode(self, token_ids):
        return [self.idx2word.get(int(i), self.unk_token) for i in token_ids]

    # -- plain-data form for checkpoints (the reference pickles the object itself, seq2seqbasic.py:232-236)
    def state(self) -> dict:
        return {"kind": "word", "words": [self.idx2word[i] for i in range(len(self.idx2word))], "word_count": dict(self.word_count)}

    @classmethod
    def from_state(cls, st: dict) -> "Vocabulary":
        v = cls()
        if st["words"][:4] != [v.pad_token, v.unk_token, v.start_token, v.end_token]:
            raise ValueError("Vocabulary.from_state: the first four words must be the special tokens")
        v.add_sentence(st["words"][4:])
        v.word_count = defaultdict(int, st.get("word_count", {}))
        return v


def split_indices(n: int, test_size: float, random_state: int = 42):
    """(train, test) index arrays of sklearn's ``train_test_split(range(n), test_size=..., random_state=...)`` as the
    reference calls it (data.py:90-91, transformer_word_level.py:260-261): one Mersenne-Twister permutation of range(n),
    the first ceil(test_size * n) entries are the test set, the rest the training set, both in permutation order."""
    n_test = int(math.ceil(test_size * n))
    n_train = n - n_test
    if n_train <= 0 or n_test <= 0:
        raise ValueError(f"split_indices: n={n}, test_size={test_size} leaves an empty part")
    perm = np.random.RandomState(random_state).permutation(n)
    return perm[n_test:n_test + n_train], perm[:n_test]


def train_val_split(data, validation_size=0.2, test_size=0.02):
    """data.py:76-93 for any indexable sequence: (train, validation, test) lists."""
    items = list(data)
    tr, va = split_indices(len(items), validation_size)
    val_items = [items[i] for i in va]
    va2, te = split_indices(len(val_items), test_size)
    return [items[i] for i in tr], [val_items[i] for i in va2], [val_items[i] for i in te]


def train_val_test_indices(n: int):
    """The 80 / 10 / 10 split of transformer_word_level.py:260-261."""
    tr, va = split_indices(n, 0.2)
    va2, te = split_indices(len(va), 0.5)
    return tr, va[va2], va[te]


def image_transforms(u8_images, size: int = 224, augment=False) -> torch.Tensor:
    """data.py:166-173 on the GPU: uint8 [B, H, W, 3] -> fp32 [B, 3, size, size].

    augment=False (default): only the deterministic tail (Resize, ToTensor, Normalize) - reproducible inputs.
    augment=True: the reference's behaviour - RandomHorizontalFlip(0.5), RandomRotation(20), ColorJitter(contrast=0.3)
    drawn per image from torch's global generator (``preprocess.draw_augment_params``), then the tail.
    augment=(flips, angles, factors): those parameters, one per image."""
    from .preprocess import augment as run_augment, draw_augment_params, resize_normalize
    if augment is not False and augment is not None:
        n = u8_images.shape[0] if getattr(u8_images, "ndim", 4) == 4 else 1
        flips, angles, factors = draw_augment_params(n) if augment is True else augment
        u8_images = run_augment(u8_images, flips, angles, factors)
    return resize_normalize(u8_images, size=size)


class FoodImageCaptionDataset:
    """data.py:24-65: rows of the recipe CSV with a non-empty ``Title`` whose ``<image_dir>/<Image_Name>.jpg`` exists.
    ``__getitem__`` -> (uint8 [H, W, 3] decoded pixels, or ``transform(pixels)``; int64 caption ids)."""

    def __init__(self, csv_path, image_dir, transform=None, max_seq_length=10):
        self.image_dir = image_dir
        self.transform = transform
        self.max_seq_length = max_seq_length
        with open(csv_path, newline="", encoding="utf-8") as f:
            rows = [r for r in csv.DictReader(f) if r.get("Title") is not None and r["Title"] not in _PANDAS_NA]
        self.data = []
        for r in rows:
            path = os.path.join(image_dir, f"{r['Image_Name']}.jpg")
            if os.path.exists(path):
                self.data.append({"Title": r["Title"], "Image_Name": r["Image_Name"], "Image_Path": path})
        self.vocab = self._make_vocab()
        self._build_vocab()

    def _make_vocab(self):
        return Vocabulary()

    def _tokens(self, caption):
        return custom_standardization(caption)

    def _build_vocab(self):
        for row in self.data:
            self.vocab.add_sentence(self._tokens(row["Title"]))

    def __len__(self):
        return len(self.data)

    def __getitem__(self, idx):
        from .preprocess import decode_image_u8
        row = self.data[idx]
        image = decode_image_u8(row["Image_Path"])
        if self.transform:
            image = self.transform(image)
        ids = self.vocab.encode(self._tokens(row["Title"]), self.max_seq_length)
        return image, torch.tensor(ids, dtype=torch.long)


class SubsetFoodImageCaptionDataset:
    """data.py:154-163."""

    def __init__(self, full_dataset, indices):
        self.full_dataset = full_dataset
        self.indices = indices

    def __len__(self):
        return len(self.indices)

    def __getitem__(self, idx):
        return self.full_dataset[self.indices[idx]]


AUGMENT = False      # True: collate_fn draws the reference's random augmentations for every decoded image it batches


def _stack_images(images):
    """Tensors (already transformed) are stacked; decoded uint8 arrays are grouped by size, each group (augmented when
    AUGMENT is set, parameters drawn in batch order, and) resized and normalised by one launch, and put back in batch
    order."""
    if all(isinstance(im, torch.Tensor) and im.dtype != torch.uint8 for im in images):
        return torch.stack(list(images))
    arrs = [np.asarray(im.cpu()) if isinstance(im, torch.Tensor) else np.asarray(im) for im in images]
    groups = defaultdict(list)
    for i, a in enumerate(arrs):
        groups[a.shape].append(i)
    out = None
    drawn = None
    if AUGMENT:
        from .preprocess import draw_augment_params
        drawn = draw_augment_params(len(arrs))
    for idxs in groups.values():
        aug = tuple([v[i] for i in idxs] for v in drawn) if drawn is not None else False
        t = image_transforms(np.stack([arrs[i] for i in idxs]), augment=aug)
        if out is None:
            out = torch.empty((len(arrs),) + tuple(t.shape[1:]), device=t.device, dtype=t.dtype)
        out[torch.tensor(idxs, device=t.device)] = t
    return out


def collate_fn(batch):
    """data.py:67-72: images stacked, captions padded with 0 to the longest of the batch."""
    images, captions = zip(*batch)
    images = _stack_images(images)
    captions = [torch.as_tensor(c, dtype=torch.long) for c in captions]
    captions = torch.nn.utils.rnn.pad_sequence(captions, batch_first=True, padding_value=0)
    return images, captions


def iterate_batches(dataset, batch_size=12, shuffle=False, seed=0, collate=collate_fn, rank=0, world=1, drop_last=None):
    """Minimal stand-in for the reference's ``DataLoader(..., batch_size=12, collate_fn=...)`` calls
    (transformer_word_level.py:269-271); with world > 1 rank r takes batches r, r + world, ... .

    Every rank must take the same number of steps (each step ends in a gradient all-reduce that all ranks join), so
    with world > 1 the batch list is cut to a multiple of `world`.  drop_last drops a short final batch (what a
    graph-replayed step with its fixed shapes needs); its default is True for world > 1 - ranks with different local
    batch sizes would weight their samples differently in the averaged gradient - and False otherwise, the DataLoader
    default the reference uses."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError(f"iterate_batches: rank {rank} outside world {world}")
    if drop_last is None:
        drop_last = world > 1
    order = np.arange(len(dataset))
    if shuffle:
        order = np.random.RandomState(seed).permutation(len(dataset))
    starts = list(range(0, len(order), batch_size))
    if drop_last and starts and len(order) - starts[-1] < batch_size:
        starts.pop()
    starts = starts[:len(starts) // world * world][rank::world]
    for s in starts:
        yield collate([dataset[int(i)] for i in order[s:s + batch_size]])


def compute_accuracy(predictions, targets, padding_idx):
    """data.py:255-261: predictions [B, T, V] (the last position is dropped), targets [B, T-1]."""
    B, T, V = predictions.shape
    pred_ids = ops.argmax_rows(predictions.reshape(B * T, V).contiguous()).view(B, T)
    mask = (targets != padding_idx).float()
    correct = (pred_ids[:, :-1] == targets) * mask
    return (correct.sum() / mask.sum()).item()


# ---------------------------------------------------------------------------------------------------------------------
# BLEU.  The reference scores with nltk.translate.bleu_score.sentence_bleu (data.py:264-276, no smoothing function);
# NLTK is not installed here, so this is a restatement of its published algorithm (Papineni et al. 2002 with NLTK's
# conventions), checked on hand-computed cases: clipped n-gram precisions p_n as exact fractions, closest-reference-length
# brevity penalty, 0 when there is no unigram overlap, and - NLTK's "method 0" - a precision with zero matches replaced
# by the smallest positive double, so its weighted logarithm is a large finite negative number (or 0 for a zero weight).


def _ngrams(seq, n):
    return zip(*[seq[i:] for i in range(n)])


def modified_precision(references, hypothesis, n):
    counts = Counter(_ngrams(hypothesis, n)) if len(hypothesis) >= n else Counter()
    max_counts = {}
    for ref in references:
        rc = Counter(_ngrams(ref, n)) if len(ref) >= n else Counter()
        for g in counts:
            max_counts[g] = max(max_counts.get(g, 0), rc[g])
    num = sum(min(c, max_counts[g]) for g, c in counts.items())
    den = max(1, sum(counts.values()))
    return _Frac(num, den)


class _Frac:
    """Un-normalised fraction (NLTK keeps numerator and denominator as counted: the zero test looks at the numerator)."""

    def __init__(self, num, den):
        self.numerator, self.denominator = num, den

    def __float__(self):
        return self.numerator / self.denominator


def closest_ref_length(references, hyp_len):
    return min((len(r) for r in references), key=lambda rl: (abs(rl - hyp_len), rl))


def brevity_penalty(closest_ref_len, hyp_len):
    if hyp_len > closest_ref_len:
        return 1.0
    if hyp_len == 0:
        return 0.0
    return math.exp(1 - closest_ref_len / hyp_len)


def sentence_bleu(references, hypothesis, weights=(0.25, 0.25, 0.25, 0.25)) -> float:
    import sys
    hyp_len = len(hypothesis)
    p = [modified_precision(references, hypothesis, n) for n in range(1, len(weights) + 1)]
    if p[0].numerator == 0:
        return 0.0
    bp = brevity_penalty(closest_ref_length(references, hyp_len), hyp_len)
    vals = [float(pi) if pi.numerator != 0 else sys.float_info.min for pi in p]
    s = math.fsum(w * math.log(v) for w, v in zip(weights, vals) if v > 0)
    return bp * math.exp(s)


def compute_bleu(predictions, targets, idx2word, word2idx):
    """data.py:264-276: BLEU-1..4 of the FIRST caption of the batch (arg-max tokens against the unpadded target)."""
    T, V = predictions.shape[1], predictions.shape[2]
    pred_ids = ops.argmax_rows(predictions[0].reshape(T, V).contiguous()).tolist()
    pred_tokens = [idx2word[i] for i in pred_ids]
    pad = word2idx["<PAD>"]
    target_tokens = [idx2word[i] for i in targets[0].tolist() if i != pad]
    return {"bleu_1": sentence_bleu([target_tokens], pred_tokens, weights=(1, 0, 0, 0)),
            "bleu_2": sentence_bleu([target_tokens], pred_tokens, weights=(0.5, 0.5, 0, 0)),
            "bleu_3": sentence_bleu([target_tokens], pred_tokens, weights=(0.33, 0.33, 0.33, 0)),
            "bleu_4": sentence_bleu([target_tokens], pred_tokens, weights=(0.25, 0.25, 0.25, 0.25))}
