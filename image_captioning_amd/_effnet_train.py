"""Training-mode forward and hand-written backward of get_cnn_model: EfficientNet-B0 + Linear(1280, 512)
(transformer_word_level_bahdnau.py:12-25 under ``model.train()``; the reference optimises every CNN parameter).

efficientnet_pytorch train-mode semantics: BatchNorm with batch statistics (eps 1e-3, momentum 0.01), swish,
squeeze-excite, drop-connect on the identity-skip blocks with rate 0.2 * block_index / 16 (one Bernoulli factor
per sample), dropout 0.2 on the pooled feature.

The whole CNN is ONE ``torch.autograd.Function``.  The forward records a tape of closures, one per layer, each
holding what its gradient needs; the backward runs the tape in reverse.  Activations are NHWC fp32 viewed as
[B*H*W, C] matrices, so a 1x1 convolution is a GEMM (forward: the MFMA implicit-GEMM kernel; input gradient:
``gemm_nn``; weight gradient: ``gemm_tn`` with the contraction split across workgroups).
"""
from __future__ import annotations

import torch

from . import ops
from . import train_ops as T
from ._bah_train import DropoutState
from .backbones import conv_weight_khwc, same_pad
from .ops import ACT_NONE, ACT_SIGMOID, ACT_SWISH

DROP_CONNECT_RATE = 0.2   # efficientnet-b0 global parameter
HEAD_DROPOUT = 0.2


class _Tape:
    def __init__(self):
        self.steps = []
        self.grads = {}
        self.counters = []         # num_batches_tracked of every BatchNorm passed: bumped together when the forward ends

    def add(self, fn):
        self.steps.append(fn)

    def grad(self, param, value):
        self.grads[id(param)] = value.view(param.shape) if value.shape != param.shape else value


def _w2d(conv):
    w = conv.weight.detach()
    return w.reshape(w.shape[0], w.shape[1]).float().contiguous()


def _bn_act(tape, z, bn, act, parts=None):
    """parts: the column statistics the kernel that produced z left behind (ops.StatParts), when it has them."""
    mom = bn.momentum if bn.momentum is not None else 0.1
    y, saved = T.batchnorm_train(z, bn.weight.detach(), bn.bias.detach(), bn.running_mean, bn.running_var, bn.eps, mom, act=act,
                                 parts=parts)
    tape.counters.append(bn.num_batches_tracked)

    def bwd(dy):
        dx, dg, db = T.batchnorm_act_bwd(dy, z, saved, act=act)
        tape.grad(bn.weight, dg)
        tape.grad(bn.bias, db)
        return dx
    return y, bwd


def _conv1x1(tape, x, conv, need_dx=True):
    """-> (z, column statistics of z for the BatchNorm that follows, backward closure)"""
    w = _w2d(conv)
    z, parts = ops.linear_stats(x, w)

    def bwd(dz, add=None):
        """add: the gradient an identity skip brings to the same input, added in the input-gradient GEMM's epilogue"""
        tape.grad(conv.weight, T.gemm_tn(dz, x))
        return T.gemm_nn(dz, w, add=add) if need_dx else None
    return z, parts, bwd


def _mbconv(tape, blk, x, B, H, W, idx, masks):
    """One MBConv block on x [B*H*W, cin]; returns (y, Ho, Wo) and pushes its backward closure."""
    cin, cout, k, stride = blk.cin, blk.cout, blk.k, blk.stride
    mid = cin * blk.expand
    inp = x
    bw = []                                  # layer closures of this block, forward order
    a0 = x
    if blk.expand != 1:
        z0, st0, f = _conv1x1(tape, x, blk._expand_conv); bw.append(f)
        a0, f = _bn_act(tape, z0, blk._bn0, ACT_SWISH, st0); bw.append(f)
    lo, hi = same_pad(k, stride)
    dw_w = blk._depthwise_conv.weight.detach().view(mid, k * k).t().float().contiguous()   # [k*k, C]
    z1, st1 = ops.dwconv2d_stats(a0.view(B, H, W, mid), dw_w, ks=k, stride=stride, pad=(lo, hi, lo, hi))
    Ho, Wo = z1.shape[1], z1.shape[2]
    HW = Ho * Wo
    z1 = z1.view(B * HW, mid)
    # BatchNorm + swish of the depthwise result: a first pass over z1 takes the squeeze-excite pooling of the activation
    # without writing it, a second writes the GATED activation; the ungated one never exists in memory (the backward pass
    # recomputes it from z1 where it needs it)
    bn1 = blk._bn1
    _, bn1_saved, pooled = T.batchnorm_train(z1, bn1.weight.detach(), bn1.bias.detach(), bn1.running_mean, bn1.running_var, bn1.eps,
                                             bn1.momentum if bn1.momentum is not None else 0.1, act=ACT_SWISH, parts=st1, pool=(B, HW),
                                             pool_only=True)
    tape.counters.append(bn1.num_batches_tracked)
    s1, b1 = bn1_saved[2], bn1_saved[3]
    # squeeze-excite
    sq = blk._se_reduce.weight.shape[0]
    wr, br = _w2d(blk._se_reduce), blk._se_reduce.bias.detach().float().contiguous()
    we, be = _w2d(blk._se_expand), blk._se_expand.bias.detach().float().contiguous()
    r_pre, e, gate = T.se_train_fwd(pooled, wr, br, we, be)          # reduce FC + swish + expand FC + sigmoid in one launch
    a2 = T.act_gate(z1, s1, b1, ACT_SWISH, gate, B, HW)
    z2, st2, proj_bwd = _conv1x1(tape, a2, blk._project_conv)
    y, bn2_bwd = _bn_act(tape, z2, blk._bn2, ACT_NONE, st2)
    skip = stride == 1 and cin == cout
    dc_mask = None
    if skip:
        rate = DROP_CONNECT_RATE * idx / 16.0
        given = (masks or {}).get(f"dc{idx}") if masks is not None else None
        if rate > 0 and given is not None:
            dc_mask = given.reshape(-1).float().contiguous()
            y = T.mul_mask(y, dc_mask, group=HW * cout, add=inp)
        elif rate > 0 and masks is None:
            y, dc_mask = T.dropout(y, rate, DropoutState.seed, DropoutState.take(B), group=HW * cout)
            T.axpby(inp, y)
        else:
            y = T.scale_bias_act(y, res=inp)

    def bwd(dy):
        dyb = T.mul_mask(dy, dc_mask, group=HW * cout) if dc_mask is not None else dy
        dz2 = bn2_bwd(dyb)
        da2 = proj_bwd(dz2)
        dgate = T.image_dot_act(da2, z1, s1, b1, ACT_SWISH, B, HW, mid)
        dpooled, dwe, dbe, dwr, dbr = T.se_train_bwd(dgate, e, r_pre, pooled, wr, we)
        tape.grad(blk._se_expand.weight, dwe)
        tape.grad(blk._se_expand.bias, dbe)
        tape.grad(blk._se_reduce.weight, dwr)
        tape.grad(blk._se_reduce.bias, dbr)
        # the gate's and the pooled mean's gradients are applied to da2 inside the BatchNorm-gradient passes
        dz1, dg1, db1 = T.batchnorm_act_bwd_se(da2, z1, bn1_saved, gate, dpooled, 1.0 / HW, HW, act=ACT_SWISH)
        tape.grad(bn1.weight, dg1)
        tape.grad(bn1.bias, db1)
        dww = T.dwconv_bwd_weight(dz1, a0, B, H, W, mid, k, stride, lo, lo, Ho, Wo)
        tape.grad(blk._depthwise_conv.weight, dww.t().contiguous().view(mid, 1, k, k))
        d = T.dwconv_bwd_data(dz1, dw_w, B, H, W, mid, k, stride, lo, lo, Ho, Wo)
        if bw:
            d = bw[1](d)                                     # BatchNorm + swish of the expand convolution
            d = bw[0](d, dy if skip else None)               # its input gradient (+ the skip's gradient in the epilogue)
        elif skip:
            T.axpby(dy, d)
        return d
    tape.add(bwd)
    return y, Ho, Wo


class CnnTrain(torch.autograd.Function):
    """images [B, 3, H, W] -> [B, 512] in train mode; parameters are passed so that autograd routes their gradients."""

    @staticmethod
    def forward(ctx, images, model, masks, *params):
        eff, lin = model[0], model[1]
        tape = _Tape()
        B, _, H, W = images.shape
        x4 = ops.nchw_to_nhwc_pad(images.float().contiguous(), 4)                       # [B, H, W, 4]
        lo, hi = same_pad(3, 2)
        w_stem = conv_weight_khwc(eff._conv_stem.weight.detach(), 4)
        z, st_stem = ops.conv2d_stats(x4, w_stem, stride=2, pad=(lo, hi, lo, hi))
        Ho, Wo = z.shape[1], z.shape[2]
        z = z.view(B * Ho * Wo, 32)
        a, bn0_bwd = _bn_act(tape, z, eff._bn0, ACT_SWISH, st_stem)

        def stem_bwd(da):
            dz = bn0_bwd(da)
            # patches gathered from the image inside the GEMM: [32, (kh, kw, c)] -> the parameter's [32, c, kh, kw]
            dw = T.conv_wgrad_gather(dz, x4, B, H, W, 4, 3, 3, 3, 2, lo, lo, Ho, Wo).view(32, 3, 3, 3).permute(0, 3, 1, 2).contiguous()
            tape.grad(eff._conv_stem.weight, dw)
            return None
        tape.add(stem_bwd)
        x, H2, W2 = a, Ho, Wo
        for idx, blk in enumerate(eff._blocks):
            x, H2, W2 = _mbconv(tape, blk, x, B, H2, W2, idx, masks)
        HW = H2 * W2
        zh, st_head, head_bwd = _conv1x1(tape, x, eff._conv_head)
        ah, bn1_bwd = _bn_act(tape, zh, eff._bn1, ACT_SWISH, st_head)
        pooled = T.image_dot(ah, None, B, HW, 1280, scale=1.0 / HW)
        given = (masks or {}).get("cnn_drop") if masks is not None else None
        if given is not None:
            m_drop = given.reshape(-1).float().contiguous()
            pd = T.mul_mask(pooled, m_drop)
        elif masks is None and HEAD_DROPOUT > 0:
            pd, m_drop = T.dropout(pooled, HEAD_DROPOUT, DropoutState.seed, DropoutState.take(pooled.numel()))
        else:
            pd, m_drop = pooled, None
        lw, lb = lin.weight.detach().float().contiguous(), lin.bias.detach().float().contiguous()
        out = ops.linear(pd, lw, lb)

        def head(dout):
            tape.grad(lin.weight, T.gemm_tn(dout, pd))
            tape.grad(lin.bias, T.col_sum(dout))
            dpd = T.gemm_nn(dout, lw)
            dpool = T.mul_mask(dpd, m_drop) if m_drop is not None else dpd
            da = T.chan_affine(None, None, dpool, 1.0 / HW, B, HW, 1280)
            return head_bwd(bn1_bwd(da))
        tape.add(head)
        # the closures hold the tape (for tape.grad); the tape must not hold them back, or a step's activations would
        # sit in a reference cycle until Python's cycle collector happens to run
        ctx.steps, tape.steps = tape.steps, None
        torch._foreach_add_(tape.counters, 1)
        tape.counters = []
        ctx.tape = tape
        ctx.params = params
        return out

    @staticmethod
    def backward(ctx, dout):
        tape = ctx.tape
        d = dout.float().contiguous()
        for f in reversed(ctx.steps):
            d = f(d)
        grads = tuple(tape.grads.get(id(p)) for p in ctx.params)
        tape.grads = {}
        return (None, None, None) + T.accumulate_into_grads(ctx.params, grads)


def cnn_train(model, images, masks=None):
    """masks: None -> generated drop-connect / dropout; a dict -> exactly the given factors (missing key: no dropping)."""
    params = [p for p in model.parameters() if p.requires_grad]
    return CnnTrain.apply(images, model, masks, *params)
