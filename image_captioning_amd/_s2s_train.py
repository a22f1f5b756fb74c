"""Training-mode forward and hand-written backward of the seq2seq captioner (seq2seqbasic.py:14-103 under
``model.train()``, the loop of lines 216-223).

* ``EncoderCNN``: the backbone runs under ``torch.no_grad()`` in the reference (line 41), so it gets no gradient -
  but ``model.train()`` still puts its BatchNorm layers in training mode: ResNet-18 normalises with the batch's
  statistics and updates its running statistics.  ``resnet_trunk_train`` reproduces that forward (VGG-16 has no
  BatchNorm: its eval forward is its train forward).  ``fc`` + ``BatchNorm1d(momentum=0.01)`` are trained.
* ``DecoderRNN.forward``: embedding of ``captions[:, :-1]`` behind the image feature, ``feature_fc``, stacked
  LSTM / GRU, ``fc``; the backward is back-propagation through time, layer by layer, with the input-side GEMMs of a
  layer batched over all steps.

Each module is one ``torch.autograd.Function``; all arithmetic is in ``libicap_hip.so``.
"""
from __future__ import annotations

import torch

from . import ops
from . import train_ops as T
from .backbones import conv_weight_khwc
from .ops import ACT_NONE, ACT_RELU


# ---------------------------------------------------------------------------- frozen ResNet-18 with train-mode BatchNorm
def _bn_fwd(x4, bn, act, res=None):
    B, H, W, C = x4.shape
    mom = bn.momentum if bn.momentum is not None else 0.1
    y, _ = T.batchnorm_train(x4.view(B * H * W, C), bn.weight.detach(), bn.bias.detach(), bn.running_mean, bn.running_var, bn.eps,
                             mom, act=ACT_NONE if res is not None else act)
    bn.num_batches_tracked += 1
    if res is not None:                                    # relu(bn(x) + identity)
        y = T.scale_bias_act(T.scale_bias_act(y, res=res.view(B * H * W, C)), act=act)
    return y.view(B, H, W, C)


def resnet_trunk_train(trunk, images):
    """[B, 3, H, W] -> [B, 512, 1, 1]: ResNet-18 children[:-1] with batch-statistic BatchNorm, no gradient."""
    with torch.no_grad():
        x = ops.nchw_to_nhwc_pad(images.float().contiguous(), 4)
        x = ops.conv2d(x, conv_weight_khwc(trunk[0].weight.detach(), 4), stride=2, pad=(3, 3, 3, 3))
        x = _bn_fwd(x, trunk[1], ACT_RELU)
        x = ops.maxpool2d(x, 3, 2, 1)
        for li in (4, 5, 6, 7):
            for blk in trunk[li]:
                s = blk.stride
                idn = x
                if blk.downsample is not None:
                    idn = ops.conv2d(x, conv_weight_khwc(blk.downsample[0].weight.detach()), stride=s)
                    idn = _bn_fwd(idn, blk.downsample[1], ACT_NONE)
                h = ops.conv2d(x, conv_weight_khwc(blk.conv1.weight.detach()), stride=s, pad=(1, 1, 1, 1))
                h = _bn_fwd(h, blk.bn1, ACT_RELU)
                h = ops.conv2d(h, conv_weight_khwc(blk.conv2.weight.detach()), pad=(1, 1, 1, 1))
                x = _bn_fwd(h, blk.bn2, ACT_RELU, res=idn)
        x = ops.adaptive_avgpool2d(x, 1, 1)
        return x.view(x.shape[0], 512, 1, 1)


class EncoderHeadTrain(torch.autograd.Function):
    """feats [B, F] (no gradient) -> BatchNorm1d(fc(feats)) with batch statistics (seq2seqbasic.py:43-44)."""

    @staticmethod
    def forward(ctx, feats, w, b, bn_g, bn_b, bn, vgg):
        wk = w.detach().float()
        if vgg:       # the VGG trunk hands over (h, w, c)-flattened features: reorder the columns of fc.weight to match
            wk = wk.view(wk.shape[0], 512, 7, 7).permute(0, 2, 3, 1).reshape(wk.shape[0], -1)
        z = ops.linear(feats, wk.contiguous(), b.detach().float().contiguous())
        mom = bn.momentum if bn.momentum is not None else 0.1
        y, saved = T.batchnorm_train(z, bn_g.detach(), bn_b.detach(), bn.running_mean, bn.running_var, bn.eps, mom)
        bn.num_batches_tracked += 1
        ctx.saved = (feats, z, saved)
        ctx.w_shape = w.shape
        ctx.vgg = vgg
        return y

    @staticmethod
    def backward(ctx, dy):
        feats, z, saved = ctx.saved
        dz, dg, db_bn = T.batchnorm_act_bwd(dy.float().contiguous(), z, saved)
        dw = T.gemm_tn(dz, feats)                         # in the kernel's column order
        if ctx.vgg:
            dw = dw.view(dw.shape[0], 7, 7, 512).permute(0, 3, 1, 2).reshape(dw.shape[0], -1).contiguous()
        return None, dw.view(ctx.w_shape), T.col_sum(dz), dg, db_bn, None, None


# ---------------------------------------------------------------------------- decoder
class DecoderRNNTrain(torch.autograd.Function):
    """features [B, E], captions [B, T] -> logits [B, T, V] (seq2seqbasic.py:59-66), LSTM or GRU, L layers."""

    @staticmethod
    def forward(ctx, features, emb_w, ff_w, ff_b, fc_w, fc_b, captions, mod, *rnn_params):
        B, Tn = captions.shape
        H, L = mod.hidden_size, mod.num_layers
        lstm = mod.rnn_type == "lstm"
        G = 4 if lstm else 3
        dev = features.device
        E = emb_w.shape[1]
        idx = captions[:, :-1].t().contiguous().view(-1)                            # time-major, (T-1)*B tokens
        xin = torch.empty(Tn * B, E, device=dev)                                    # step 0: image feature
        xin[:B].copy_(features.float())
        if Tn > 1:
            emb, _ = ops.embedding(emb_w.detach().float().contiguous(), idx)
            xin[B:].copy_(emb)
        x = ops.linear(xin, ff_w.detach().float().contiguous(), ff_b.detach().float().contiguous())   # [T*B, H]
        layers = []
        cur = x
        for l in range(L):
            w_ih, w_hh, b_ih, b_hh = [p.detach().float().contiguous() for p in rnn_params[4 * l:4 * l + 4]]
            gi_all = ops.linear(cur, w_ih, b_ih).view(Tn, B, G * H)
            h = torch.zeros(B, H, device=dev)
            c = torch.zeros(B, H, device=dev)
            Hprev = torch.empty(Tn, B, H, device=dev)
            Hs = torch.empty(Tn, B, H, device=dev)
            Cs = torch.empty(Tn + 1, B, H, device=dev) if lstm else None
            Gs = torch.empty(Tn, B, G * H, device=dev)                              # LSTM: full gates; GRU: recurrent half gh
            if lstm:
                Cs[0].zero_()
            for t in range(Tn):
                Hprev[t].copy_(h)
                if lstm:
                    g = ops.linear(h, w_hh, b_hh, residual=gi_all[t])
                    h, c = ops.lstm_pointwise(g, c)
                    Cs[t + 1].copy_(c)
                else:
                    g = ops.linear(h, w_hh, b_hh)
                    h = ops.gru_pointwise(gi_all[t], g, h)
                Gs[t].copy_(g)
                Hs[t].copy_(h)
            layers.append((cur, w_ih, w_hh, gi_all, Hprev, Hs, Cs, Gs))
            cur = Hs.view(Tn * B, H)
        hs_bm = Hs.permute(1, 0, 2).contiguous().view(B * Tn, H)                    # batch-major rows for [B, T, V] logits
        fcw = fc_w.detach().float().contiguous()
        logits = ops.linear(hs_bm, fcw, fc_b.detach().float().contiguous())
        ctx.saved = (xin, idx, layers, hs_bm, fcw, ff_w.detach().float().contiguous())
        ctx.dims = (B, Tn, H, L, G, lstm, emb_w.shape[0], E)
        return logits.view(B, Tn, -1)

    @staticmethod
    def backward(ctx, dout):
        xin, idx, layers, hs_bm, fcw, ffw = ctx.saved
        B, Tn, H, L, G, lstm, V, E = ctx.dims
        dev = xin.device
        dl = dout.float().contiguous().view(B * Tn, -1)
        dfc_w = T.gemm_tn(dl, hs_bm)
        dfc_b = T.col_sum(dl)
        dhs = T.gemm_nn(dl, fcw).view(B, Tn, H).permute(1, 0, 2).contiguous()       # time-major [T, B, H]
        rnn_grads = [None] * (4 * L)
        for l in reversed(range(L)):
            cur, w_ih, w_hh, gi_all, Hprev, Hs, Cs, Gs = layers[l]
            dGi = torch.empty(Tn, B, G * H, device=dev)
            dGh = dGi if lstm else torch.empty(Tn, B, G * H, device=dev)
            dh_next = torch.zeros(B, H, device=dev)
            dc_next = torch.zeros(B, H, device=dev) if lstm else None
            for t in reversed(range(Tn)):
                dh = T.axpby(dhs[t], dh_next)
                if lstm:
                    dg, dc_next = T.lstm_bwd(Gs[t], Cs[t], Cs[t + 1], dh, dc_next)
                    dGi[t].copy_(dg)
                    dh_next = T.gemm_nn(dg, w_hh)
                else:
                    dgi, dgh, dhd = T.gru_bwd(gi_all[t], Gs[t], Hprev[t], dh)
                    dGi[t].copy_(dgi); dGh[t].copy_(dgh)
                    dh_next = T.gemm_nn(dgh, w_hh, out=dhd, accumulate=True)
            dGi2, dGh2 = dGi.view(Tn * B, G * H), dGh.view(Tn * B, G * H)
            rnn_grads[4 * l + 0] = T.gemm_tn(dGi2, cur)
            rnn_grads[4 * l + 1] = T.gemm_tn(dGh2, Hprev.view(Tn * B, H))
            rnn_grads[4 * l + 2] = T.col_sum(dGi2)
            rnn_grads[4 * l + 3] = T.col_sum(dGh2) if not lstm else rnn_grads[4 * l + 2].clone()
            dhs = T.gemm_nn(dGi2, w_ih).view(Tn, B, -1)                             # gradient of this layer's input sequence
        dx = dhs.view(Tn * B, H)
        dff_w = T.gemm_tn(dx, xin)
        dff_b = T.col_sum(dx)
        dxin = T.gemm_nn(dx, ffw)                                                   # [T*B, E]
        dfeat = dxin[:B].contiguous()
        demb = T.embedding_bwd(idx, dxin[B:], V) if Tn > 1 else torch.zeros(V, E, device=dev)
        return (dfeat, demb, dff_w, dff_b, dfc_w, dfc_b, None, None) + tuple(rnn_grads)


def decoder_train(mod, features, captions):
    rp = []
    for l in range(mod.num_layers):
        rp += [getattr(mod.rnn, f"weight_ih_l{l}"), getattr(mod.rnn, f"weight_hh_l{l}"), getattr(mod.rnn, f"bias_ih_l{l}"),
               getattr(mod.rnn, f"bias_hh_l{l}")]
    return DecoderRNNTrain.apply(features, mod.embed.weight, mod.feature_fc.weight, mod.feature_fc.bias, mod.fc.weight, mod.fc.bias,
                                 captions, mod, *rp)
