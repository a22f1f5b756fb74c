"""Char-level transformer captioner.

Drop-in for the model classes of the reference's transformercharmiguel.py:13-113:
same names, constructor arguments, ``forward`` signatures and ``state_dict`` keys.
The vocabulary is the fixed 81-symbol alphabet of datachar.py:14-15
(``<SOS>`` = 0, ``<EOS>`` = 1, ``<PAD>`` = 2), captions are up to 58 symbols
(datachar.py:19).  The arithmetic is in ``_tfm_common`` / ``libicap_hip.so``.
"""
from __future__ import annotations

from . import _tfm_common as _c
from .backbones import EfficientNet, get_cnn_model  # noqa: F401

NUM_CHAR = 81        # len(datachar.chars)
TEXT_MAX_LEN = 58    # datachar.TEXT_MAX_LEN


class TransformerEncoderBlock(_c.TransformerEncoderBlockBase):
    """transformercharmiguel.py:30-51."""


class TransformerDecoderBlock(_c.TransformerDecoderBlockBase):
    """transformercharmiguel.py:55-80 (vocabulary = d.NUM_CHAR)."""

    def __init__(self, embed_dim, ff_dim, num_heads, dropout=0.3):
        super().__init__(embed_dim, ff_dim, num_heads, NUM_CHAR, dropout)


class ImageCaptioningModel(_c.ImageCaptioningModelBase):
    """transformercharmiguel.py:83-113."""

    start_token, end_token = 0, 1

    def __init__(self, cnn_model, encoder, decoder, num_captions_per_image=5, embed_dim=512):
        super().__init__(cnn_model, encoder, decoder, NUM_CHAR, num_captions_per_image, embed_dim)


def build_model(weights_path=None) -> ImageCaptioningModel:
    """The configuration trained by the reference (transformercharmiguel.py:121-137): E=512, FF=512, heads 2 / 4."""
    enc = TransformerEncoderBlock(embed_dim=512, dense_dim=512, num_heads=2)
    dec = TransformerDecoderBlock(embed_dim=512, ff_dim=512, num_heads=4)
    return ImageCaptioningModel(get_cnn_model(weights_path), enc, dec)
