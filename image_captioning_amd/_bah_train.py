"""Training-mode forward and hand-written backward of the Bahdanau captioner's encoder block and decoder
(transformer_word_level_bahdnau.py:53-146 under ``model.train()``).

Each module is ONE ``torch.autograd.Function``: the forward keeps what the backward needs, the backward is
back-propagation through time written out against the kernels of ``train_ops`` (no autograd inside).  torch
records the two nodes and hands the parameter gradients to ``.grad``; all arithmetic is in ``libicap_hip.so``.

Dropout masks come from the library's counter-based generator, keyed by ``(seed, offset)`` drawn from
``DropoutState`` - or are supplied by the caller (``masks=``), which is how the parity tests feed the CPU oracle
and this path identical masks.
"""
from __future__ import annotations

import torch

from . import ops
from . import train_ops as T
from .ops import ACT_NONE, ACT_RELU


class DropoutState:
    """Seed and running offset of the dropout generator (one per process; rank-dependent seed under data parallel)."""

    seed = 0x5EED
    offset = 0

    @classmethod
    def manual_seed(cls, seed: int) -> None:
        cls.seed, cls.offset = int(seed), 0

    @classmethod
    def take(cls, n: int) -> int:
        o = cls.offset
        cls.offset += int(n)
        return o


def _drop(x, p, mask):
    """Dropout of a 2-D tensor: returns (y, mask).  p == 0 -> identity with mask None."""
    if mask is not None:
        return T.mul_mask(x, mask.reshape(-1)), mask.reshape(-1)
    if p <= 0.0:
        return x, None
    return T.dropout(x, p, DropoutState.seed, DropoutState.take(x.numel()))


def _undrop(dy, mask):
    return dy if mask is None else T.mul_mask(dy, mask)


def _lin(x, w, b=None, act=ACT_NONE, residual=None):
    return ops.linear(x, w.detach(), None if b is None else b.detach(), act=act, residual=residual)


# ---------------------------------------------------------------------------- encoder block
class EncoderBlockTrain(torch.autograd.Function):
    """rows [B, E] -> LayerNorm2(2 * fc(LayerNorm1(rows + dropout(attention(rows))))) as the reference computes it
    (lines 70-90): one key per image, so attention = out_proj(W_v x + b_v); W_q / W_k get zero gradient."""

    @staticmethod
    def forward(ctx, rows, in_w, in_b, out_w, out_b, ln1_g, ln1_b, ln2_g, ln2_b, w0, b0, bn_g, bn_b, w4, b4, mod, masks):
        E = rows.shape[1]
        p = mod.dropout.p
        m = masks or {}
        rows = rows.float().contiguous()
        wv, bv = in_w.detach()[2 * E:].contiguous(), in_b.detach()[2 * E:].contiguous()
        v = ops.linear(rows, wv, bv)
        att = _lin(v, out_w, out_b)
        att_d, m_att = _drop(att, p, m.get("enc_att"))
        x1, ln1_saved = T.layernorm_train(att_d, ln1_g.detach(), ln1_b.detach(), res=rows, eps=mod.layer_norm1.eps)
        h = _lin(x1, w0, b0, act=ACT_RELU)
        h_d, m_h = _drop(h, mod.fc[2].p, m.get("enc_fc1"))
        bn = mod.fc[3]
        mom = bn.momentum if bn.momentum is not None else 0.1
        hb, bn_saved = T.batchnorm_train(h_d, bn_g.detach(), bn_b.detach(), bn.running_mean, bn.running_var, bn.eps, mom)
        bn.num_batches_tracked += 1
        f = _lin(hb, w4, b4)
        f_d, m_f = _drop(f, mod.fc[5].p, m.get("enc_fc2"))
        out, ln2_saved = T.layernorm_train(f_d, ln2_g.detach(), ln2_b.detach(), eps=mod.layer_norm2.eps, pre_scale=2.0)
        ctx.saved = (rows, wv, v, m_att, ln1_saved, x1, h, m_h, h_d, bn_saved, hb, m_f, ln2_saved)
        ctx.params = (in_w, out_w, ln1_g, ln2_g, w0, w4)
        ctx.all_params = (in_w, in_b, out_w, out_b, ln1_g, ln1_b, ln2_g, ln2_b, w0, b0, bn_g, bn_b, w4, b4)
        return out

    @staticmethod
    def backward(ctx, dout):
        rows, wv, v, m_att, ln1_saved, x1, h, m_h, h_d, bn_saved, hb, m_f, ln2_saved = ctx.saved
        in_w, out_w, ln1_g, ln2_g, w0, w4 = ctx.params
        E = rows.shape[1]
        dout = dout.float().contiguous()
        df_d, dln2_g, dln2_b = T.layernorm_bwd(dout, ln2_saved, ln2_g.detach(), post_scale=2.0)
        df = _undrop(df_d, m_f)
        dw4 = T.gemm_tn(df, hb)
        db4 = T.col_sum(df)
        dhb = T.gemm_nn(df, w4.detach())
        dh_d, dbn_g, dbn_b = T.batchnorm_act_bwd(dhb, h_d, bn_saved)
        dh = _undrop(dh_d, m_h)
        dz = T.act_bwd(dh, h, ACT_RELU)               # relu'(z) = [h > 0]; h = relu(z) has the same sign pattern
        dw0 = T.gemm_tn(dz, x1)
        db0 = T.col_sum(dz)
        dx1 = T.gemm_nn(dz, w0.detach())
        dsum, dln1_g, dln1_b = T.layernorm_bwd(dx1, ln1_saved, ln1_g.detach())
        datt = _undrop(dsum, m_att)
        dout_w = T.gemm_tn(datt, v)
        dout_b = T.col_sum(datt)
        dv = T.gemm_nn(datt, out_w.detach())
        din_w = torch.zeros_like(in_w)
        din_b = torch.zeros(3 * E, device=rows.device)
        T.gemm_tn(dv, rows, out=din_w[2 * E:])
        T.col_sum(dv, out=din_b[2 * E:])
        drows = T.gemm_nn(dv, wv)
        T.axpby(dsum, drows)                           # residual branch of LayerNorm1
        # parameter gradients go into their buffers by one multi-tensor add (hooks included) where buffers exist
        pg = T.accumulate_into_grads(ctx.all_params, (din_w, din_b, dout_w, dout_b, dln1_g, dln1_b, dln2_g, dln2_b, dw0, db0, dbn_g, dbn_b,
                                                      dw4, db4))
        return (drows,) + pg + (None, None)


def encoder_block_train(mod, rows, masks=None):
    a = mod.attention
    return EncoderBlockTrain.apply(rows, a.in_proj_weight, a.in_proj_bias, a.out_proj.weight, a.out_proj.bias,
                                   mod.layer_norm1.weight, mod.layer_norm1.bias, mod.layer_norm2.weight, mod.layer_norm2.bias,
                                   mod.fc[0].weight, mod.fc[0].bias, mod.fc[3].weight, mod.fc[3].bias, mod.fc[4].weight,
                                   mod.fc[4].bias, mod, masks)


# ---------------------------------------------------------------------------- decoder
class BahdanauDecoderTrain(torch.autograd.Function):
    """captions [B, T], feats [B, E] -> logits [B, T, V], reference lines 127-141 in train mode: per step
    embedding, additive attention over the batch's images, LSTMCell, dropout, Linear, BatchNorm1d(V) with the
    statistics of that step's B rows (running statistics updated once per step, in step order)."""

    @staticmethod
    def forward(ctx, feats, W_w, W_b, U_w, U_b, A_w, A_b, emb_w, w_ih, w_hh, b_ih, b_hh, fc_w, fc_b, bn_g, bn_b, captions, mod, masks):
        B, Tn = captions.shape
        E, H = mod.embed_dim, mod.decoder_dim
        ENC = feats.shape[1]
        V = fc_w.shape[0]
        dev = feats.device
        feats = feats.float().contiguous()
        idx = captions.t().contiguous().view(-1)                                    # time-major token ids
        emb_all, _ = ops.embedding(emb_w.detach(), idx)                             # [T*B, E]
        w_ih_d = w_ih.detach()
        w_ih_e, w_ih_c = w_ih_d[:, :E].contiguous(), w_ih_d[:, E:].contiguous()
        gi_emb = ops.linear(emb_all, w_ih_e, b_ih.detach()).view(Tn, B, 4 * H)
        u = _lin(feats, U_w, U_b)
        a_w = A_w.detach().view(-1).contiguous()
        a_b = 0.0   # the scoring bias shifts every score of a softmax row alike: no effect on alpha, and no host read-back
        # every per-step result is written by its kernel straight into the slice the backward pass reads: h_t into
        # Hall[t + 1] (so h_{t-1} of all steps is Hall[:T] and h_t of all steps Hall[1:]), c_t into Cs[t + 1]
        Hall = torch.empty(Tn + 1, B, H, device=dev)
        Hall[0].zero_()
        Hprev, Hs = Hall[:Tn], Hall[1:]
        Cs = torch.empty(Tn + 1, B, H, device=dev)     # c_{t-1} at [t], c_t at [t+1]
        Cs[0].zero_()
        Ws = torch.empty(Tn, B, u.shape[1], device=dev)
        Al = torch.empty(Tn, B, B, device=dev)
        Ctx = torch.empty(Tn, B, ENC, device=dev)
        G = torch.empty(Tn, B, 4 * H, device=dev)
        al_direct = (B * B) % 4 == 0                   # a [B, B] slice starts 16-byte aligned only then
        W_wd, W_bd, w_hh_d, b_hh_d = W_w.detach(), W_b.detach(), w_hh.detach(), b_hh.detach()
        for t in range(Tn):
            h = Hall[t]
            ops.linear(h, W_wd, W_bd, out=Ws[t])
            alpha, _ = ops.bahdanau_attention(u, Ws[t], a_w, a_b, feats, alpha_out=Al[t] if al_direct else None, ctx_out=Ctx[t])
            if not al_direct:
                Al[t].copy_(alpha)
            g = ops.linear(Ctx[t], w_ih_c, residual=gi_emb[t])
            ops.linear(h, w_hh_d, b_hh_d, residual=g, out=G[t])
            ops.lstm_pointwise(G[t], Cs[t], h_out=Hall[t + 1], c_out=Cs[t + 1])
        hs2 = Hs.view(Tn * B, H)
        m = (masks or {}).get("dec_out")
        if m is not None:                                                           # given as [B, T, H]
            m = m.permute(1, 0, 2).contiguous()
        hd, m_out = _drop(hs2, mod.dropout.p, m)
        pre = _lin(hd, fc_w, fc_b)                                                  # [T*B, V]
        out = torch.empty(B, Tn, V, device=dev)
        bn = mod.fc[1]
        mom = bn.momentum if bn.momentum is not None else 0.1
        if V % 4 == 0:
            # every step normalised with the statistics of its own B rows, all steps per launch
            bn_saved = T.batchnorm_steps_fwd(pre, bn_g.detach(), bn_b.detach(), bn.running_mean, bn.running_var, bn.eps, mom, B, Tn, out)
        else:
            bn_saved = []
            for t in range(Tn):
                x_t = pre[t * B:(t + 1) * B]
                # one pass over the step's rows (moments about the first row), the running statistics updated in step order
                _, saved = T.batchnorm_train(x_t, bn_g, bn_b, bn.running_mean, bn.running_var, bn.eps, mom, out=out[:, t, :])
                bn_saved.append(saved)
        bn.num_batches_tracked += Tn
        ctx.saved = (feats, idx, emb_all, w_ih_e, w_ih_c, u, a_w, Hprev, Cs, Ws, Al, Ctx, G, hd, m_out, pre, bn_saved)
        ctx.params = (W_w, U_w, w_ih, w_hh, fc_w, emb_w)
        ctx.all_params = (W_w, W_b, U_w, U_b, A_w, A_b, emb_w, w_ih, w_hh, b_ih, b_hh, fc_w, fc_b, bn_g, bn_b)
        ctx.dims = (B, Tn, E, H, ENC, V)
        return out

    @staticmethod
    def backward(ctx, dout):
        feats, idx, emb_all, w_ih_e, w_ih_c, u, a_w, Hprev, Cs, Ws, Al, Ctx, G, hd, m_out, pre, bn_saved = ctx.saved
        W_w, U_w, w_ih, w_hh, fc_w, emb_w = ctx.params
        B, Tn, E, H, ENC, V = ctx.dims
        dev = feats.device
        A = u.shape[1]
        dout = dout.float().contiguous()
        # BatchNorm1d(V) of every step
        lib = T.N.load()
        if not isinstance(bn_saved, list):
            dpre, DG, DB = T.batchnorm_steps_bwd(dout, pre, bn_saved, B, Tn)
        else:
            dpre = torch.empty(Tn * B, V, device=dev)
            DG = torch.empty(Tn, V, device=dev); DB = torch.empty(Tn, V, device=dev)     # per-step dgamma / dbeta, summed after the loop
        for t in range(Tn if isinstance(bn_saved, list) else 0):
            mean, rstd, s, b = bn_saved[t]
            dy_t = dout[:, t, :]
            x_t = pre[t * B:(t + 1) * B]
            dbeta, dgamma = T.col_reduce(T.COL_BN_BWD, dy_t, x=x_t, v1=mean, v2=rstd, v3=s, v4=b, out1=DB[t], out2=DG[t])
            dx_t = dpre[t * B:(t + 1) * B]
            T.N.check(lib.icap_bn_act_bwd_f32(dy_t.data_ptr(), dy_t.stride(0), x_t.data_ptr(), x_t.stride(0), s.data_ptr(), b.data_ptr(),
                                              mean.data_ptr(), rstd.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), ACT_NONE, B, V,
                                              dx_t.data_ptr(), dx_t.stride(0), ops._stream()), "bn_act_bwd")
        dbn_g, dbn_b = T.col_sum(DG), T.col_sum(DB)
        dfc_w = T.gemm_tn(dpre, hd)
        dfc_b = T.col_sum(dpre)
        dhd = T.gemm_nn(dpre, fc_w.detach())
        dHs = _undrop(dhd, m_out).view(Tn, B, H)
        # back-propagation through time
        dG = torch.empty(Tn, B, 4 * H, device=dev)
        dWs = torch.empty(Tn, B, A, device=dev)
        dApart = torch.empty(Tn, B, A, device=dev)
        dAb = torch.empty(Tn, B, device=dev)
        du = torch.zeros(B, A, device=dev)
        dfeat = torch.zeros(B, ENC, device=dev)
        dh_next = torch.zeros(B, H, device=dev)
        dc_next = torch.zeros(B, H, device=dev)
        w_ih_d, w_hh_d, W_w_d = w_ih.detach(), w_hh.detach(), W_w.detach()
        direct = B % 4 == 0
        for t in reversed(range(Tn)):
            dh = T.axpby(dHs[t], dh_next)                                          # dh_next += dHs[t]
            dg, dc_next = T.lstm_bwd(G[t], Cs[t], Cs[t + 1], dh, dc_next, dg_out=dG[t])
            dctx = T.gemm_nn(dg, w_ih_d[:, E:])
            dh_prev = T.gemm_nn(dg, w_hh_d)
            if direct:      # the per-step slices start 16-byte aligned: the kernels write them in place
                dw, _, _ = T.attention_bwd(u, Ws[t], a_w, Al[t], feats, dctx, du, dfeat, accumulate=True,
                                           outs=(dWs[t], dApart[t], dAb[t]))
            else:
                dw, da_part, dab_part = T.attention_bwd(u, Ws[t], a_w, Al[t], feats, dctx, du, dfeat, accumulate=True)
                dWs[t].copy_(dw); dApart[t].copy_(da_part); dAb[t].copy_(dab_part)
            T.gemm_nn(dw, W_w_d, out=dh_prev, accumulate=True)
            dh_next = dh_prev
        dG2 = dG.view(Tn * B, 4 * H)
        Hp2 = Hprev.view(Tn * B, H)
        dEmb = T.gemm_nn(dG2, w_ih_e)              # no step needs it: one GEMM over all T*B rows after the loop
        dw_ih = torch.empty_like(w_ih)
        T.gemm_tn(dG2, emb_all, out=dw_ih[:, :E])
        T.gemm_tn(dG2, Ctx.view(Tn * B, ENC), out=dw_ih[:, E:])
        dw_hh = T.gemm_tn(dG2, Hp2)
        db_g = T.col_sum(dG2)                                                      # bias_ih and bias_hh share it
        dW_w = T.gemm_tn(dWs.view(Tn * B, A), Hp2)
        dW_b = T.col_sum(dWs.view(Tn * B, A))
        dA_w = T.col_sum(dApart.view(Tn * B, A)).view(1, A)
        dA_b = T.col_sum(dAb.view(Tn * B, 1)).view(1)
        dU_w = T.gemm_tn(du, feats)
        dU_b = T.col_sum(du)
        T.gemm_nn(du, U_w.detach(), out=dfeat, accumulate=True)
        demb_w = T.embedding_bwd(idx, dEmb, emb_w.shape[0])
        pg = T.accumulate_into_grads(ctx.all_params, (dW_w, dW_b, dU_w, dU_b, dA_w, dA_b, demb_w, dw_ih, dw_hh, db_g, db_g, dfc_w, dfc_b,
                                                      dbn_g, dbn_b))
        if pg[9] is not None and pg[10] is not None:
            pg = pg[:10] + (db_g.clone(),) + pg[11:]          # autograd must not see one tensor as two gradients
        return (dfeat,) + pg + (None, None, None)


def decoder_train(mod, captions, feats, masks=None):
    a = mod.attention
    return BahdanauDecoderTrain.apply(feats, a.W.weight, a.W.bias, a.U.weight, a.U.bias, a.A.weight, a.A.bias, mod.embedding.weight,
                                      mod.lstm.weight_ih, mod.lstm.weight_hh, mod.lstm.bias_ih, mod.lstm.bias_hh, mod.fc[0].weight,
                                      mod.fc[0].bias, mod.fc[1].weight, mod.fc[1].bias, captions, mod, masks)
