"""CPU oracle for the CNN encoders.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

PARITY UNPINNED BY THE REFERENCE: the reference takes these networks from
``torchvision`` (resnet18, vgg16; seq2seqbasic.py:21-35) and ``efficientnet_pytorch``
(EfficientNet-B0; transformer_word_level.py:7,13), both un-vendored, version
unpinned, and absent from this image, and it holds no test or fixture for them.
This file restates their published architectures in plain ``torch.nn.functional``
(NCHW fp32, eval mode), keyed by those packages' ``state_dict`` names:

* torchvision ResNet-18: conv7x7/2 + BN + ReLU + maxpool3/2/1, four stages of two
  BasicBlocks (64/128/256/512; 1x1/2 + BN shortcut when the shape changes), global
  average pool; BatchNorm eps 1e-5.
* torchvision VGG-16 ``features``: 13 conv3x3(pad 1)+ReLU, five maxpool2/2, then
  ``AdaptiveAvgPool2d((7, 7))`` as the reference appends it.
* efficientnet_pytorch EfficientNet-B0: stem conv3x3/2 -> 16 MBConv blocks
  (expand 1x1 -> depthwise kxk -> squeeze-excite with ratio 0.25 of the block
  input -> project 1x1, identity skip at stride 1 with equal widths) -> head 1x1
  to 1280 -> global average pool; swish activations, BatchNorm eps 1e-3; every
  convolution uses TensorFlow 'SAME' padding fixed for a 224-px input (extra pixel
  on the bottom/right).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def _bn(sd, p, x, eps):
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"],
                        training=False, eps=eps)


# --------------------------------------------------------------------------- ResNet-18 trunk (keys "0", "1", "4".."7")


def resnet18_trunk(sd, x, p="", training=False, bufs=None):
    """torchvision resnet18 children[:-1].  training=True: BatchNorm uses the batch statistics and updates the running
    statistics in ``bufs`` (momentum 0.1) - what model.train() does to the reference's frozen backbone
    (seq2seqbasic.py:41 only switches gradients off, not the BatchNorm mode)."""
    def bn(name, t):
        if not training:
            return _bn(sd, name, t, 1e-5)
        return F.batch_norm(t, bufs[name + ".running_mean"], bufs[name + ".running_var"], sd[name + ".weight"], sd[name + ".bias"],
                            training=True, momentum=0.1, eps=1e-5)
    x = F.relu(bn(p + "1", F.conv2d(x, sd[p + "0.weight"], stride=2, padding=3)))
    x = F.max_pool2d(x, 3, 2, 1)
    for li in (4, 5, 6, 7):
        for bi in (0, 1):
            q = f"{p}{li}.{bi}."
            stride = 2 if (li > 4 and bi == 0) else 1
            idn = x
            if q + "downsample.0.weight" in sd:
                idn = bn(q + "downsample.1", F.conv2d(x, sd[q + "downsample.0.weight"], stride=stride))
            h = F.relu(bn(q + "bn1", F.conv2d(x, sd[q + "conv1.weight"], stride=stride, padding=1)))
            h = bn(q + "bn2", F.conv2d(h, sd[q + "conv2.weight"], padding=1))
            x = F.relu(h + idn)
    return F.adaptive_avg_pool2d(x, 1)


# --------------------------------------------------------------------------- VGG-16 trunk (keys = torchvision feature indices)

_VGG_CONV_IDX = [0, 2, 5, 7, 10, 12, 14, 17, 19, 21, 24, 26, 28]
_VGG_POOL_AFTER = {2, 7, 14, 21, 28}


def vgg16_trunk(sd, x, p=""):
    for i in _VGG_CONV_IDX:
        x = F.relu(F.conv2d(x, sd[f"{p}{i}.weight"], sd[f"{p}{i}.bias"], padding=1))
        if i in _VGG_POOL_AFTER:
            x = F.max_pool2d(x, 2, 2)
    return F.adaptive_avg_pool2d(x, (7, 7))


# --------------------------------------------------------------------------- EfficientNet-B0

_B0 = [(1, 3, 1, 1, 32, 16), (2, 3, 2, 6, 16, 24), (2, 5, 2, 6, 24, 40), (3, 3, 2, 6, 40, 80),
       (3, 5, 1, 6, 80, 112), (4, 5, 2, 6, 112, 192), (1, 3, 1, 6, 192, 320)]


def _swish(x):
    return x * torch.sigmoid(x)


def _same_pad_static(x, k, s, configured_size):
    """Conv2dStaticSamePadding: the pad is computed once from the CONFIGURED image size."""
    out = -(-configured_size // s)
    total = max((out - 1) * s + k - configured_size, 0)
    lo = total // 2
    return F.pad(x, (lo, total - lo, lo, total - lo))


def efficientnet_b0_features(sd, x, p=""):
    """[B,3,H,W] -> pooled [B,1280] (``_fc`` replaced by Identity as in transformer_word_level.py:16)."""
    size = 224
    x = _swish(_bn(sd, p + "_bn0", F.conv2d(_same_pad_static(x, 3, 2, size), sd[p + "_conv_stem.weight"], stride=2), 1e-3))
    size = -(-size // 2)
    bi = 0
    for (r, k, s, e, cin, cout) in _B0:
        for j in range(r):
            q = f"{p}_blocks.{bi}."
            stride = s if j == 0 else 1
            ci = cin if j == 0 else cout
            inp = x
            h = x
            if e != 1:
                h = _swish(_bn(sd, q + "_bn0", F.conv2d(h, sd[q + "_expand_conv.weight"]), 1e-3))
            h = F.conv2d(_same_pad_static(h, k, stride, size), sd[q + "_depthwise_conv.weight"], stride=stride, groups=h.shape[1])
            h = _swish(_bn(sd, q + "_bn1", h, 1e-3))
            size = -(-size // stride)
            sq = F.adaptive_avg_pool2d(h, 1)
            sq = F.conv2d(_swish(F.conv2d(sq, sd[q + "_se_reduce.weight"], sd[q + "_se_reduce.bias"])),
                          sd[q + "_se_expand.weight"], sd[q + "_se_expand.bias"])
            h = torch.sigmoid(sq) * h
            h = _bn(sd, q + "_bn2", F.conv2d(h, sd[q + "_project_conv.weight"]), 1e-3)
            if stride == 1 and ci == cout:
                h = h + inp
            x = h
            bi += 1
    x = _swish(_bn(sd, p + "_bn1", F.conv2d(x, sd[p + "_conv_head.weight"]), 1e-3))
    return F.adaptive_avg_pool2d(x, 1).flatten(1)


def _bn_train(sd, name, x, bufs, eps, momentum=0.01):
    """Train-mode BatchNorm2d: batch statistics, running statistics in ``bufs`` updated in place."""
    return F.batch_norm(x, bufs[name + ".running_mean"], bufs[name + ".running_var"], sd[name + ".weight"], sd[name + ".bias"],
                        training=True, momentum=momentum, eps=eps)


def efficientnet_b0_features_train(sd, x, bufs, masks=None, p="", drop_connect_rate=0.2):
    """Train-mode forward of efficientnet_pytorch's EfficientNet-B0 up to (and including) the dropout in front of
    ``_fc``: batch-statistic BatchNorm (eps 1e-3, momentum 0.01), drop-connect on the skip blocks with rate
    0.2 * block_index / 16 (one Bernoulli factor per sample), dropout 0.2 on the pooled feature.
    ``masks``: {"dc<i>": [B] factors, "cnn_drop": [B, 1280] factors}; a missing key means no dropping.
    ``bufs``: running statistics (cloned by the caller), updated in place."""
    m = masks or {}
    size = 224
    x = _swish(_bn_train(sd, p + "_bn0", F.conv2d(_same_pad_static(x, 3, 2, size), sd[p + "_conv_stem.weight"], stride=2), bufs, 1e-3))
    size = -(-size // 2)
    bi = 0
    for (r, k, s, e, cin, cout) in _B0:
        for j in range(r):
            q = f"{p}_blocks.{bi}."
            stride = s if j == 0 else 1
            ci = cin if j == 0 else cout
            inp = x
            h = x
            if e != 1:
                h = _swish(_bn_train(sd, q + "_bn0", F.conv2d(h, sd[q + "_expand_conv.weight"]), bufs, 1e-3))
            h = F.conv2d(_same_pad_static(h, k, stride, size), sd[q + "_depthwise_conv.weight"], stride=stride, groups=h.shape[1])
            h = _swish(_bn_train(sd, q + "_bn1", h, bufs, 1e-3))
            size = -(-size // stride)
            sq = F.adaptive_avg_pool2d(h, 1)
            sq = F.conv2d(_swish(F.conv2d(sq, sd[q + "_se_reduce.weight"], sd[q + "_se_reduce.bias"])),
                          sd[q + "_se_expand.weight"], sd[q + "_se_expand.bias"])
            h = torch.sigmoid(sq) * h
            h = _bn_train(sd, q + "_bn2", F.conv2d(h, sd[q + "_project_conv.weight"]), bufs, 1e-3)
            if stride == 1 and ci == cout:
                dc = m.get(f"dc{bi}")
                if dc is not None and drop_connect_rate * bi / 16 > 0:
                    h = h * dc.view(-1, 1, 1, 1)
                h = h + inp
            x = h
            bi += 1
    x = _swish(_bn_train(sd, p + "_bn1", F.conv2d(x, sd[p + "_conv_head.weight"]), bufs, 1e-3))
    x = F.adaptive_avg_pool2d(x, 1).flatten(1)
    if m.get("cnn_drop") is not None:
        x = x * m["cnn_drop"]
    return x


def cnn_model_train(sd, x, bufs, masks=None, p="cnn_model."):
    """get_cnn_model in train mode: EfficientNet-B0 (train) then Linear(1280, 512)."""
    bufs0 = {k[len(p + "0."):]: v for k, v in bufs.items() if k.startswith(p + "0.")}
    sd0 = {k[len(p + "0."):]: v for k, v in sd.items() if k.startswith(p + "0.")}
    f = efficientnet_b0_features_train(sd0, x, bufs0, masks)
    return f @ sd[p + "1.weight"].t() + sd[p + "1.bias"]


def cnn_model(sd, x, p="cnn_model."):
    """get_cnn_model: EfficientNet-B0 features then Linear(1280, 512) (transformer_word_level.py:19-22)."""
    f = efficientnet_b0_features(sd, x, p + "0.")
    return f @ sd[p + "1.weight"].t() + sd[p + "1.bias"]


def seq2seq_encoder(sd, images, cnn_type, p="encoder."):
    """EncoderCNN.forward (seq2seqbasic.py:40-45), eval mode."""
    if cnn_type == "resnet18":
        f = resnet18_trunk(sd, images, p + "cnn.")
    else:
        f = vgg16_trunk(sd, images, p + "cnn.")
    f = f.reshape(images.shape[0], -1)
    f = f @ sd[p + "fc.weight"].t() + sd[p + "fc.bias"]
    return F.batch_norm(f, sd[p + "bn.running_mean"], sd[p + "bn.running_var"], sd[p + "bn.weight"], sd[p + "bn.bias"],
                        training=False, eps=1e-5)
