"""CNN encoders of the captioning models on the MI355X kernels.

The reference takes its backbones from two packages that are not part of it:
``torchvision.models.resnet18 / vgg16`` (seq2seqbasic.py:21-35) and
``efficientnet_pytorch.EfficientNet`` (transformer_word_level.py:7,13).  The
classes here are *parameter containers* with those packages' attribute names, so a
``state_dict`` saved by the reference loads unchanged, plus a ``forward`` that
runs the whole network (eval mode) through ``libicap_hip.so``: NHWC fp32
activations, BatchNorm folded into the convolution epilogue, convolutions as
implicit GEMM on the matrix cores, depthwise / squeeze-excite / pooling as
bandwidth-bound kernels.

Pretrained weights cannot be downloaded here; ``pretrained=True`` /
``from_pretrained`` keep their spelling but initialise randomly unless a local
``weights_path`` is given.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from . import ops

# --------------------------------------------------------------------------- helpers


def require_eval(module: nn.Module, what: str) -> None:
    if module.training:
        raise RuntimeError(
            f"{what}: this method is the inference path and needs model.eval(); training goes through forward() / "
            "training.Trainer (train-mode BatchNorm, dropout and the hand-written backward)")


def _state_key(module: nn.Module):
    """Cheap fingerprint of every parameter/buffer: changes on load_state_dict, .to(), in-place updates through
    torch, and - through the generation counter - on every write the library makes by raw pointer (optimizer steps,
    BatchNorm running statistics of a train-mode forward, replayed training graphs)."""
    key = [ops.weights_generation()]
    for t in list(module.parameters()) + list(module.buffers()):
        key.append((t.data_ptr(), t._version))
    return tuple(key)


class PreparedCache:
    """Caches kernel-layout copies of a module's weights, rebuilt when the weights change."""

    def __init__(self):
        self._key = None
        self._val = None

    def get(self, module: nn.Module, builder):
        key = _state_key(module)
        if key != self._key:
            with torch.no_grad():
                self._val = builder()
            self._key = key
        return self._val


def fold_bn(bn: nn.modules.batchnorm._BatchNorm, conv_bias: torch.Tensor | None = None):
    """Eval-mode BatchNorm as y = x*scale + bias (optionally absorbing the preceding layer's bias)."""
    scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
    bias = bn.bias - bn.running_mean * scale
    if conv_bias is not None:
        bias = bias + conv_bias * scale
    return scale.float().contiguous(), bias.float().contiguous()


def conv_weight_khwc(w: torch.Tensor, cin_pad: int | None = None) -> torch.Tensor:
    """[Cout,Cin,KH,KW] -> [Cout,KH,KW,Cin(_pad)] contiguous."""
    w = w.permute(0, 2, 3, 1)
    if cin_pad is not None and cin_pad != w.shape[-1]:
        w = torch.nn.functional.pad(w, (0, cin_pad - w.shape[-1]))
    return w.float().contiguous()


# --------------------------------------------------------------------------- ResNet-18


class BasicBlock(nn.Module):
    """torchvision BasicBlock attribute layout (conv1, bn1, conv2, bn2, downsample)."""

    def __init__(self, inplanes, planes, stride=1):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = None
        if stride != 1 or inplanes != planes:
            self.downsample = nn.Sequential(nn.Conv2d(inplanes, planes, 1, stride, bias=False), nn.BatchNorm2d(planes))
        self.stride = stride


class ResNet18(nn.Module):
    """torchvision resnet18 attribute layout; ``children()`` order matches, so
    ``nn.Sequential(*list(backbone.children())[:-1])`` yields the reference's key names."""

    def __init__(self, pretrained: bool = False, weights_path: str | None = None):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.layer1 = nn.Sequential(BasicBlock(64, 64), BasicBlock(64, 64))
        self.layer2 = nn.Sequential(BasicBlock(64, 128, 2), BasicBlock(128, 128))
        self.layer3 = nn.Sequential(BasicBlock(128, 256, 2), BasicBlock(256, 256))
        self.layer4 = nn.Sequential(BasicBlock(256, 512, 2), BasicBlock(512, 512))
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512, 1000)
        if weights_path:
            self.load_state_dict(torch.load(weights_path, map_location="cpu", weights_only=True))


def resnet18(pretrained: bool = False, weights_path: str | None = None) -> ResNet18:
    return ResNet18(pretrained, weights_path)


class ResNetTrunk(nn.Sequential):
    """``nn.Sequential(*children[:-1])`` of :class:`ResNet18` with a HIP forward -> [B, 512, 1, 1]."""

    def __init__(self, backbone: ResNet18):
        super().__init__(*list(backbone.children())[:-1])
        self._prep = PreparedCache()

    def _build(self):
        conv1, bn1 = self[0], self[1]
        p = {"stem_w": conv_weight_khwc(conv1.weight, 4), "stem_sb": fold_bn(bn1), "blocks": []}
        for li in (4, 5, 6, 7):
            for blk in self[li]:
                d = {"w1": conv_weight_khwc(blk.conv1.weight), "sb1": fold_bn(blk.bn1),
                     "w2": conv_weight_khwc(blk.conv2.weight), "sb2": fold_bn(blk.bn2),
                     "stride": blk.stride, "down": None}
                if blk.downsample is not None:
                    d["down"] = (conv_weight_khwc(blk.downsample[0].weight), fold_bn(blk.downsample[1]))
                p["blocks"].append(d)
        return p

    def forward(self, images: torch.Tensor) -> torch.Tensor:
        require_eval(self, "ResNet-18 trunk")
        p = self._prep.get(self, self._build)
        x = ops.nchw_to_nhwc_pad(images.float().contiguous(), 4)
        # in bf16 compute mode the activations stay in bf16 from the stem to the global average pool
        k16 = ops.bf16_storage_ok
        x = ops.dense_conv(x, p["stem_w"], p, "stem16", *p["stem_sb"], stride=2, pad=(3, 3, 3, 3), act=ops.ACT_RELU,
                           out_bf16=k16(4, p["stem_w"].shape[0]))
        x = ops.maxpool2d(x, 3, 2, 1)
        for d in p["blocks"]:
            s = d["stride"]
            idn = x
            cin, cout = x.shape[3], d["w1"].shape[0]
            o16 = x.dtype == torch.bfloat16 and k16(cin, cout)
            if d["down"] is not None:
                idn = ops.dense_conv(x, d["down"][0], d, "down16", *d["down"][1], stride=s, out_bf16=o16)
            h = ops.dense_conv(x, d["w1"], d, "w1_16", *d["sb1"], stride=s, pad=(1, 1, 1, 1), act=ops.ACT_RELU, out_bf16=o16)
            x = ops.dense_conv(h, d["w2"], d, "w2_16", *d["sb2"], residual=idn, pad=(1, 1, 1, 1), act=ops.ACT_RELU, out_bf16=o16)
        x = ops.adaptive_avgpool2d(x, 1, 1)  # [B,1,1,512]
        return x.view(x.shape[0], 512, 1, 1)


# --------------------------------------------------------------------------- VGG-16

_VGG16_CFG = [64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512, "M"]


class VGG16(nn.Module):
    """torchvision vgg16 attribute layout (features / avgpool / classifier)."""

    def __init__(self, pretrained: bool = False, weights_path: str | None = None):
        super().__init__()
        layers, cin = [], 3
        for v in _VGG16_CFG:
            if v == "M":
                layers.append(nn.MaxPool2d(2, 2))
            else:
                layers += [nn.Conv2d(cin, v, 3, padding=1), nn.ReLU(inplace=True)]
                cin = v
        self.features = nn.Sequential(*layers)
        self.avgpool = nn.AdaptiveAvgPool2d((7, 7))
        self.classifier = nn.Sequential(nn.Linear(25088, 4096), nn.ReLU(True), nn.Dropout(), nn.Linear(4096, 4096),
                                        nn.ReLU(True), nn.Dropout(), nn.Linear(4096, 1000))
        if weights_path:
            self.load_state_dict(torch.load(weights_path, map_location="cpu", weights_only=True))


def vgg16(pretrained: bool = False, weights_path: str | None = None) -> VGG16:
    return VGG16(pretrained, weights_path)


class VGGTrunk(nn.Sequential):
    """``nn.Sequential(*features, AdaptiveAvgPool2d((7,7)))`` with a HIP forward.

    Returns NHWC-flattened features ``[B, 7*7*512]`` tagged ``nhwc_flat = True``; the
    Linear that follows must use :func:`vgg_fc_weight_nhwc` so the result equals the
    reference's NCHW ``view(B, -1)`` followed by its ``fc``."""

    def __init__(self, backbone: VGG16):
        super().__init__(*list(backbone.features), nn.AdaptiveAvgPool2d((7, 7)))
        self._prep = PreparedCache()

    def _build(self):
        plan = []
        first = True
        for m in self:
            if isinstance(m, nn.Conv2d):
                plan.append(("conv", conv_weight_khwc(m.weight, 4 if first else None), m.bias.float().contiguous(), {}))
                first = False
            elif isinstance(m, nn.MaxPool2d):
                plan.append(("pool",))
        return plan

    def forward(self, images: torch.Tensor) -> torch.Tensor:
        # no BatchNorm, no dropout in vgg16.features: train() and eval() compute the same thing
        plan = self._prep.get(self, self._build)
        x = ops.nchw_to_nhwc_pad(images.float().contiguous(), 4)
        for st in plan:
            if st[0] == "conv":
                # in bf16 compute mode the activations stay in bf16 up to the 7x7 pooling
                x = ops.dense_conv(x, st[1], st[3], "w16", None, st[2], pad=(1, 1, 1, 1), act=ops.ACT_RELU,
                                   out_bf16=ops.bf16_storage_ok(x.shape[3], st[1].shape[0]))
            else:
                x = ops.maxpool2d(x, 2, 2, 0)
        if x.shape[1] != 7 or x.shape[2] != 7 or x.dtype != torch.float32:
            x = ops.adaptive_avgpool2d(x, 7, 7)      # (also the bf16 -> fp32 step when the map is 7x7 already)
        return x.reshape(x.shape[0], -1)


def vgg_fc_weight_nhwc(w: torch.Tensor) -> torch.Tensor:
    """Reorder fc columns from the reference's (c, h, w) flatten to the (h, w, c) flatten used here."""
    out_f = w.shape[0]
    return w.view(out_f, 512, 7, 7).permute(0, 2, 3, 1).reshape(out_f, 25088).float().contiguous()


# --------------------------------------------------------------------------- EfficientNet-B0

# (repeats, kernel, stride, expand, in, out) of efficientnet-b0; squeeze-excite ratio 0.25 of the block input
_B0_STAGES = [(1, 3, 1, 1, 32, 16), (2, 3, 2, 6, 16, 24), (2, 5, 2, 6, 24, 40), (3, 3, 2, 6, 40, 80),
              (3, 5, 1, 6, 80, 112), (4, 5, 2, 6, 112, 192), (1, 3, 1, 6, 192, 320)]
_BN_EPS, _BN_MOM = 1e-3, 0.01


def same_pad(k: int, stride: int) -> tuple[int, int]:
    """TF 'SAME' padding as efficientnet_pytorch fixes it for its 224-px configuration: every strided
    convolution sees an even input, so the total is k - stride (split low = total // 2), and k - 1 at stride 1."""
    total = max(k - stride, 0) if stride > 1 else k - 1
    return total // 2, total - total // 2


class MBConvBlock(nn.Module):
    """efficientnet_pytorch MBConvBlock attribute layout."""

    def __init__(self, k, stride, expand, cin, cout):
        super().__init__()
        self.k, self.stride, self.expand, self.cin, self.cout = k, stride, expand, cin, cout
        mid = cin * expand
        if expand != 1:
            self._expand_conv = nn.Conv2d(cin, mid, 1, bias=False)
            self._bn0 = nn.BatchNorm2d(mid, eps=_BN_EPS, momentum=_BN_MOM)
        self._depthwise_conv = nn.Conv2d(mid, mid, k, stride, groups=mid, bias=False)
        self._bn1 = nn.BatchNorm2d(mid, eps=_BN_EPS, momentum=_BN_MOM)
        sq = max(1, int(cin * 0.25))
        self._se_reduce = nn.Conv2d(mid, sq, 1)
        self._se_expand = nn.Conv2d(sq, mid, 1)
        self._project_conv = nn.Conv2d(mid, cout, 1, bias=False)
        self._bn2 = nn.BatchNorm2d(cout, eps=_BN_EPS, momentum=_BN_MOM)


class EfficientNetB0(nn.Module):
    """efficientnet_pytorch ``EfficientNet`` ('efficientnet-b0') attribute layout with a HIP forward.

    ``_fc`` exists so the reference's ``base_model._fc = nn.Identity()`` (transformer_word_level.py:16)
    works as written; the forward returns the pooled 1280-d feature when ``_fc`` is an Identity."""

    def __init__(self):
        super().__init__()
        self._conv_stem = nn.Conv2d(3, 32, 3, 2, bias=False)
        self._bn0 = nn.BatchNorm2d(32, eps=_BN_EPS, momentum=_BN_MOM)
        blocks = []
        for (r, k, s, e, i, o) in _B0_STAGES:
            for j in range(r):
                blocks.append(MBConvBlock(k, s if j == 0 else 1, e, i if j == 0 else o, o))
        self._blocks = nn.ModuleList(blocks)
        self._conv_head = nn.Conv2d(320, 1280, 1, bias=False)
        self._bn1 = nn.BatchNorm2d(1280, eps=_BN_EPS, momentum=_BN_MOM)
        self._avg_pooling = nn.AdaptiveAvgPool2d(1)
        self._dropout = nn.Dropout(0.2)
        self._fc = nn.Linear(1280, 1000)
        self._prep = PreparedCache()

    @classmethod
    def from_pretrained(cls, model_name: str = "efficientnet-b0", weights_path: str | None = None, **_):
        if model_name != "efficientnet-b0":
            raise ValueError("only efficientnet-b0 is used by the captioning models")
        m = cls()
        if weights_path:
            m.load_state_dict(torch.load(weights_path, map_location="cpu", weights_only=True), strict=False)
        return m

    @classmethod
    def from_name(cls, model_name: str = "efficientnet-b0", **_):
        return cls.from_pretrained(model_name)

    def _build(self):
        p = {"stem_w": conv_weight_khwc(self._conv_stem.weight, 4), "stem_sb": fold_bn(self._bn0), "blocks": []}
        p["stem_wk"] = ops.stem_weight_k(p["stem_w"])           # k-major copy for the vector-ALU stem kernel
        for b in self._blocks:
            mid = b.cin * b.expand
            d = {"k": b.k, "stride": b.stride, "skip": b.stride == 1 and b.cin == b.cout, "expand": None}
            if b.expand != 1:
                d["expand"] = (conv_weight_khwc(b._expand_conv.weight), fold_bn(b._bn0))
            d["dw_w"] = b._depthwise_conv.weight.view(mid, b.k * b.k).t().float().contiguous()  # [k*k, C]
            d["dw_sb"] = fold_bn(b._bn1)
            sq = b._se_reduce.weight.shape[0]
            d["se"] = (b._se_reduce.weight.view(sq, mid).float().contiguous(), b._se_reduce.bias.float().contiguous(),
                       b._se_expand.weight.view(mid, sq).t().float().contiguous(), b._se_expand.bias.float().contiguous())   # [sq, mid]
            d["proj"] = (conv_weight_khwc(b._project_conv.weight), fold_bn(b._bn2))
            p["blocks"].append(d)
        p["head_w"] = conv_weight_khwc(self._conv_head.weight)
        p["head_sb"] = fold_bn(self._bn1)
        return p

    def extract_pooled(self, images: torch.Tensor) -> torch.Tensor:
        """[B,3,H,W] NCHW -> [B,1280] (stem, 16 MBConv blocks, head, global average pool)."""
        require_eval(self, "EfficientNet-B0")
        p = self._prep.get(self, self._build)
        lo, hi = same_pad(3, 2)
        blocks = p["blocks"]
        d0 = blocks[0]
        fused_stem = d0["expand"] is None and d0["k"] == 3 and d0["stride"] == 1 and not d0["skip"]
        # the fused stem kernel reads the planar image as it comes (fp32 modes); every other path takes it interleaved and padded
        planar = fused_stem and ops.COMPUTE_DTYPE != "bf16" and ops.STEM_NCHW
        x = images.float().contiguous() if planar else ops.nchw_to_nhwc_pad(images.float().contiguous(), 4)
        if fused_stem:
            # the stem's 112x112x32 output feeds only block 0's depthwise convolution: both run in one kernel
            # (in bf16 mode the fused kernels hand their result to the project convolution as a bf16 tensor)
            h, part = ops.stem_dwconv2d(x, p["stem_w"], *p["stem_sb"], d0["dw_w"], *d0["dw_sb"], stem_pad=(lo, hi, lo, hi),
                                        pad=(1, 1, 1, 1), want_partial=True, nchw=planar, w_stem_k=p.get("stem_wk"),
                                        out_bf16=ops.bf16_activations_ok(d0["dw_w"].shape[1], d0["proj"][0].shape[0]))
            gate = ops.se_gate(part, h.shape[1] * h.shape[2], *d0["se"])
            x = ops.pointwise_conv(h, d0["proj"][0], d0, "proj16", *d0["proj"][1], gate=gate)
            blocks = blocks[1:]
        else:
            x = ops.conv2d(x, p["stem_w"], *p["stem_sb"], stride=2, pad=(lo, hi, lo, hi), act=ops.ACT_SWISH)
        for d in blocks:
            inp = x
            h = x
            lo, hi = same_pad(d["k"], d["stride"])
            pad = (lo, hi, lo, hi)
            if d["expand"] is not None and ops.expand_dwconv_supported(h.shape[1], h.shape[2], h.shape[3], d["dw_w"].shape[1],
                                                                       d["k"], d["stride"], pad):
                # early blocks: the 6x expanded activations stay on chip between the 1x1 and the depthwise convolution
                h, part = ops.expand_dwconv2d(h, d["expand"][0], *d["expand"][1], d["dw_w"], *d["dw_sb"], ks=d["k"],
                                              stride=d["stride"], pad=pad, want_partial=True,
                                              out_bf16=ops.bf16_activations_ok(d["dw_w"].shape[1], d["proj"][0].shape[0]), cache=d)
            else:
                if d["expand"] is not None:
                    h = ops.pointwise_conv(h, d["expand"][0], d, "expand16", *d["expand"][1], act=ops.ACT_SWISH)
                h, part = ops.dwconv2d(h, d["dw_w"], *d["dw_sb"], ks=d["k"], stride=d["stride"], pad=pad,
                                       act=ops.ACT_SWISH, want_partial=True)
            gate = ops.se_gate(part, h.shape[1] * h.shape[2], *d["se"])
            x = ops.pointwise_conv(h, d["proj"][0], d, "proj16", *d["proj"][1], residual=inp if d["skip"] else None, gate=gate)
        x = ops.pointwise_conv(x, p["head_w"], p, "head16", *p["head_sb"], act=ops.ACT_SWISH)
        x = ops.adaptive_avgpool2d(x, 1, 1)
        return x.view(x.shape[0], 1280)

    def forward(self, images: torch.Tensor) -> torch.Tensor:
        x = self.extract_pooled(images)
        if isinstance(self._fc, nn.Identity):
            return x
        return ops.linear(x, self._fc.weight.float().contiguous(), self._fc.bias.float().contiguous())


# Name used by the reference's import line (transformer_word_level.py:7).
EfficientNet = EfficientNetB0


class CnnModel(nn.Sequential):
    """``nn.Sequential(base_model, nn.Linear(1280, 512))`` of get_cnn_model (transformer_word_level.py:19-22)."""

    dropout_masks = None   # tests: {"dc<i>": [B], "cnn_drop": [B, 1280]} factor tensors instead of generated masks

    def forward(self, images: torch.Tensor) -> torch.Tensor:
        if self.training:
            from ._effnet_train import cnn_train
            ops.bump_weights_generation()          # BatchNorm running statistics are written by raw pointer
            return cnn_train(self, images, self.dropout_masks)
        feat = self[0](images)
        lin = self[1]
        return ops.linear(feat, lin.weight.detach().float().contiguous(), lin.bias.detach().float().contiguous())


def get_cnn_model(weights_path: str | None = None) -> CnnModel:
    """EfficientNet-B0 without its classifier + Linear(1280, 512)  (transformer_word_level.py:11-24)."""
    base_model = EfficientNetB0.from_pretrained("efficientnet-b0", weights_path=weights_path)
    base_model._fc = nn.Identity()
    return CnnModel(base_model, nn.Linear(1280, 512))
