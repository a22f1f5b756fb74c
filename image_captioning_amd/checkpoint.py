"""Checkpoint compatibility with the reference's files.

The reference writes ``torch.save(model.state_dict(), "<name>.pth")`` (transformer_word_level.py:246,
transformer_word_level_bahdnau.py:298, transformercharmiguel.py:248, seq2seqbasic.py:264) and reads it back with
``model.load_state_dict(torch.load(path, map_location=device))`` (TransformerTest.py:241, seq2seqtest.py:163).  The
modules here keep the reference's parameter names, so such a file loads with ``strict=True``; it is read with
``weights_only=True``, which executes nothing from the file.

The seq2seq script also pickles its ``Vocabulary`` object (seq2seqbasic.py:232-236).  A pickle of a class instance can
only be read by executing code named in the file, so it is not read here: vocabularies are stored as JSON
(``save_vocab`` / ``load_vocab``) holding the words in index order, from which the object is rebuilt.
"""
from __future__ import annotations

import json

import torch


def save_checkpoint(model: torch.nn.Module, path: str) -> None:
    """Same file the reference writes: the bare ``state_dict`` (CPU tensors)."""
    torch.save({k: v.detach().cpu() for k, v in model.state_dict().items()}, path)


def load_checkpoint(model: torch.nn.Module, path: str, strict: bool = True):
    """Load a reference-format ``.pth`` into ``model`` (on whatever device the model lives)."""
    sd = torch.load(path, map_location="cpu", weights_only=True)
    if not isinstance(sd, dict) or not all(isinstance(v, torch.Tensor) for v in sd.values()):
        raise ValueError(f"{path}: not a state_dict (a flat name -> tensor mapping)")
    return model.load_state_dict(sd, strict=strict)


def save_vocab(vocab, path: str) -> None:
    with open(path, "w", encoding="utf-8") as f:
        json.dump(vocab.state(), f, ensure_ascii=False)


def load_vocab(path: str):
    from . import data, datachar
    with open(path, encoding="utf-8") as f:
        st = json.load(f)
    return datachar.Vocabulary.from_state(st) if st.get("kind") == "char" else data.Vocabulary.from_state(st)
