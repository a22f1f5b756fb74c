"""Host-side data logic of the character-level model, mirroring the reference's ``datachar.py:13-143`` without pandas /
torchvision: the fixed 81-symbol table, ``custom_standardization``, ``Vocabulary``, dataset classes, ``collate_fn``."""
from __future__ import annotations

import re
from collections import defaultdict

import torch

from . import data as _word
from .data import SubsetFoodImageCaptionDataset, image_transforms, _stack_images  # noqa: F401
from .training import compute_loss  # noqa: F401  (datachar.py:234-242)

# datachar.py:14-19
chars = ['<SOS>', '<EOS>', '<PAD>', '<UNK>', ' ', '!', '"', '#', '&', "'", '(', ')', ',', '-', '.'] + list("0123456789") + \
    [':', ';', '=', '?'] + [chr(c) for c in range(ord('A'), ord('Z') + 1)] + [chr(c) for c in range(ord('a'), ord('z') + 1)]
NUM_CHAR = len(chars)
idx2char = {k: v for k, v in enumerate(chars)}
char2idx = {v: k for k, v in enumerate(chars)}
TEXT_MAX_LEN = 58

_KEEP = re.compile(r"[^\w\s!\"#$%&'()*+,-./:;<=>?@[\\]^_`{|}~]")


def custom_standardization(input_string: str) -> list[str]:
    """Lower-case, apply the reference's clean-up pattern, split into characters (datachar.py:22-29).  The pattern is kept
    character for character: its class ends at the first "]" and the rest splits at "|" into a branch that cannot match
    and the literal text }~], so in practice nothing is removed; symbols outside the table map to <UNK> when encoded.
    Upper-case letters never occur in a caption although the table holds them."""
    return list(_KEEP.sub("", input_string.lower()))


class Vocabulary:
    """datachar.py:31-71."""

    def __init__(self, chars):
        self.char2idx = {c: i for i, c in enumerate(chars)}
        self.idx2char = {i: c for i, c in enumerate(chars)}
        self.char_count = defaultdict(int)
        self.pad_token = "<PAD>"
        self.unk_token = "<UNK>"
        self.sos_token = "<SOS>"
        self.eos_token = "<EOS>"

    def add_sentence(self, sentence):
        for c in sentence:
            if c in self.char2idx:
                self.char_count[c] += 1

    def encode(self, sentence, max_length):
        """<SOS> + at most max_length - 2 characters + <EOS>, padded with <PAD> to max_length."""
        tokens = [self.sos_token] + list(sentence[:max_length - 2]) + [self.eos_token]
        unk = self.char2idx[self.unk_token]
        ids = [self.char2idx.get(c, unk) for c in tokens]
        if len(ids) < max_length:
            ids += [self.char2idx[self.pad_token]] * (max_length - len(ids))
        else:
            ids = ids[:max_length]
        return ids

    def decode(self, token_ids):
        return [self.idx2char.get(int(i), self.unk_token) for i in token_ids]

    def state(self) -> dict:
        return {"kind": "char", "chars": [self.idx2char[i] for i in range(len(self.idx2char))], "char_count": dict(self.char_count)}

    @classmethod
    def from_state(cls, st: dict) -> "Vocabulary":
        v = cls(st["chars"])
        v.char_count = defaultdict(int, st.get("char_count", {}))
        return v


class FoodImageCaptionDataset(_word.FoodImageCaptionDataset):
    """datachar.py:73-114."""

    def __init__(self, csv_path, image_dir, transform=None, max_seq_length=TEXT_MAX_LEN):
        super().__init__(csv_path, image_dir, transform=transform, max_seq_length=max_seq_length)

    def _make_vocab(self):
        return Vocabulary(chars)

    def _tokens(self, caption):
        return custom_standardization(caption)


def collate_fn(batch):
    """datachar.py:117-121: captions are already padded to TEXT_MAX_LEN."""
    images, captions = zip(*batch)
    return _stack_images(images), torch.stack([torch.as_tensor(c, dtype=torch.long) for c in captions])
