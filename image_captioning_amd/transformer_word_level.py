"""Word-level transformer captioner (EfficientNet-B0 + encoder block + decoder block).

Drop-in for the model classes of the reference's transformer_word_level.py:11-112:
same names, constructor arguments, ``forward`` signatures and ``state_dict`` keys.
As in the reference, ``TransformerDecoderBlock`` and ``ImageCaptioningModel`` size
their vocabulary from the module-level ``vocab_size`` (transformer_word_level.py:69,89),
which the caller sets before constructing them; a ``vocab_size=`` keyword is also
accepted.  The arithmetic is in ``_tfm_common`` / ``libicap_hip.so``.
"""
from __future__ import annotations

from . import _tfm_common as _c
from .backbones import EfficientNet, get_cnn_model  # noqa: F401  (same public names as the reference module)

vocab_size = None  # set by the caller: len(dataset.vocab.word2idx) (transformer_word_level.py:273)


def _vocab(explicit):
    v = explicit if explicit is not None else vocab_size
    if v is None:
        raise NameError("vocab_size is not defined: set transformer_word_level.vocab_size before building the model")
    return int(v)


class TransformerEncoderBlock(_c.TransformerEncoderBlockBase):
    """transformer_word_level.py:28-49."""


class TransformerDecoderBlock(_c.TransformerDecoderBlockBase):
    """transformer_word_level.py:54-79."""

    def __init__(self, embed_dim, ff_dim, num_heads, dropout=0.3, vocab_size=None):
        super().__init__(embed_dim, ff_dim, num_heads, _vocab(vocab_size), dropout)


class ImageCaptioningModel(_c.ImageCaptioningModelBase):
    """transformer_word_level.py:82-112.  <START> = 2, <END> = 3 (data.py:130)."""

    start_token, end_token = 2, 3

    def __init__(self, cnn_model, encoder, decoder, num_captions_per_image=5, embed_dim=512, vocab_size=None):
        super().__init__(cnn_model, encoder, decoder, _vocab(vocab_size), num_captions_per_image, embed_dim)


def build_model(vocab: int, ff_dim: int = 2048, weights_path=None) -> ImageCaptioningModel:
    """The configuration trained by the reference (transformer_word_level.py:120-135): E=512, FF=2048, 8/8 heads."""
    enc = TransformerEncoderBlock(embed_dim=512, dense_dim=ff_dim, num_heads=8)
    dec = TransformerDecoderBlock(embed_dim=512, ff_dim=ff_dim, num_heads=8, vocab_size=vocab)
    return ImageCaptioningModel(get_cnn_model(weights_path), enc, dec, vocab_size=vocab)
