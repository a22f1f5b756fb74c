"""MI355X-native captioning hot path: CNN encoder forward and per-step decoder forward of the four
captioning models of Merumalady/Image_Captioning, behind the reference's own nn.Module classes.

Sub-modules mirror the reference's files:

    seq2seqbasic                   EncoderCNN, DecoderRNN, ImageCaptioningModel      (ResNet-18 / VGG-16 + LSTM / GRU)
    transformer_word_level         get_cnn_model, TransformerEncoderBlock, TransformerDecoderBlock, ImageCaptioningModel
    transformer_word_level_bahdnau TransformerEncoderBlock, Attention, BahdanauDecoder, ImageCaptioningModel
    transformercharmiguel          char-level variant of transformer_word_level

Every forward runs in ``libicap_hip.so`` (hand-written gfx950 kernels, C ABI in ``include/icap_hip.h``);
there is no CPU path in this package.
"""
__all__ = ["ops", "backbones", "seq2seqbasic", "transformer_word_level", "transformer_word_level_bahdnau",
           "transformercharmiguel", "synthetic", "preprocess"]
