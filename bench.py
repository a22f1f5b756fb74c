#!/usr/bin/env python3
"""Headline benchmark: greedy captioning throughput of the word-level transformer captioner
(EfficientNet-B0 encoder + transformer encoder block + 20 greedy decoder steps) on synthetic
224x224 images, one process per GPU.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
    python bench.py --mode train [--batch 256]        # the second half of the metric: training imgs/s

--mode train times the teacher-forced training step of the Bahdanau captioner (EfficientNet-B0 in train mode,
encoder block, 20-step LSTMCell decoder, masked cross entropy, backward, AdamW; data-parallel with a bucketed
RCCL all-reduce when launched on N ranks) and prints the same kind of JSON line with unit imgs/s.

A "step" is one pass of the hot path over one batch of --batch images already resident in HBM:
NCHW fp32 images -> encoder features -> 20 greedy tokens per image (every image decoded as its own
batch of one, the reference's generate_caption semantics).  Inference shards by image: each rank
captions its own images and no data-path collective exists, so scaling is weak.  Rank 0 prints ONE
JSON line (metric, value, roofline of the dominant kernel, CPU baseline).  After the timed region the
line collects two secondary measurements on the same ranks: "bf16_mode" (the same captioning step with
bf16 products) and "train" (the --mode train step, data-parallel when N > 1).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

VOCAB = 6756          # word vocabulary of the full dataset (notebook cell 5)
CAPTION_LEN = 20      # greedy tokens per caption (BASELINE: 20-token titles)
HBM_PEAK_GBS = 8000.0
F32_MFMA_PEAK_TFLOPS = 157.3
BF16_MFMA_PEAK_TFLOPS = 2500.0      # dense bf16 (no sparsity)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--mode", choices=("caption", "train"), default="caption")
    ap.add_argument("--dtype", choices=("f32", "f32x3", "bf16"), default="f32x3",
                    help="compute dtype of the GEMMs: f32 = the fp32 matrix-core instruction; f32x3 (captioning default) = fp32-accurate "
                         "products on the bf16 matrix cores (three-way bf16 operand splits, six exact products per multiply, fp32 "
                         "accumulation) where that is faster, the fp32 instruction elsewhere; bf16 = bf16 products with fp32 accumulation and "
                         "fp32 tensors in memory (training mode: forward GEMMs and input gradients; fp32 weight gradients)")
    ap.add_argument("--graph", action="store_true",
                    help="training mode: record the step into HIP graphs (Trainer(graph=True); pays at small batches)")
    ap.add_argument("--no-bf16-mode", action="store_true", help="skip the secondary bf16-mode measurement")
    ap.add_argument("--no-plain-f32", action="store_true", help="skip the secondary plain-fp32 measurement of the captioning step")
    ap.add_argument("--no-train-mode", action="store_true", help="skip the secondary training-step measurement")
    ap.add_argument("--train-batch", type=int, default=256, help="images per step per GPU of the secondary training measurement")
    ap.add_argument("--train-deadline", type=float, default=300.0,
                    help="seconds the secondary training measurement may take before the line is printed without it")
    ap.add_argument("--batch", type=int, default=None, help="images per step per GPU (default 1024 captioning, 256 training)")
    ap.add_argument("--breakdown", action="store_true", help="print a per-launch time table of one step and exit")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true", help="skip the check of the timed step's token ids (untuned rerun + CPU oracle)")
    ap.add_argument("--no-decode-b1", action="store_true", help="skip the single-caption decode-step measurement (configs[3] and the other decoders)")
    ap.add_argument("--rehearse", action="store_true",
                    help="multi-rank dry run on fewer GPUs than ranks: gloo instead of RCCL (which refuses two ranks on one device), ranks "
                         "share devices, gradient buckets staged through host memory - the control flow is the multi-GPU one, the numbers are not results")
    ap.add_argument("--cpu-sample", type=int, default=384, help="images in the CPU-baseline sample")
    return ap.parse_args()


# ---------------------------------------------------------------- algorithmic work of one launch
def conv_work(sig):
    """(flops, bytes) of a conv2d launch from its trace signature: every input, weight and output element once."""
    B, H, W, Cin, Cout, KH, stride, Ho, Wo, has_res, has_gate = sig
    m = B * Ho * Wo
    flops = 2.0 * m * Cout * KH * KH * Cin
    byts = 4.0 * (B * H * W * Cin + Cout * KH * KH * Cin + m * Cout * (2 if has_res else 1))
    return flops, byts


def dw_work(sig):
    B, H, W, Cc, ks, stride, Ho, Wo = sig
    return 2.0 * B * Ho * Wo * Cc * ks * ks, 4.0 * (B * H * W * Cc + B * Ho * Wo * Cc)


def linear_work(sig):
    M, K, Nn, has_res = sig
    return 2.0 * M * K * Nn, 4.0 * (M * K + K * Nn + M * Nn * (2 if has_res else 1))


def linear_bf16_work(sig):
    M, K, Nn, has_res = sig[:4]
    a_bytes = 2.0 if (len(sig) > 5 and sig[5]) else 4.0          # bf16 or fp32 activation tensor
    return 2.0 * M * K * Nn, a_bytes * M * K + 2.0 * K * Nn + 4.0 * M * Nn * (2 if has_res else 1)


def gemm_work(sig):
    a, b, c = sig
    return 2.0 * a * b * c, 4.0 * (a * b + b * c + a * c)


def expand_dw_work(sig):
    """(B, H, W, Cin, Cexp, ks, stride, Ho, Wo): the expand GEMM + the depthwise taps; input read once, output written once."""
    B, H, W, Cin, Cexp, ks, stride, Ho, Wo = sig[:9]
    ob = 2.0 if (len(sig) > 9 and sig[9]) else 4.0                # bf16 or fp32 result
    return (2.0 * B * H * W * Cin * Cexp + 2.0 * B * Ho * Wo * Cexp * ks * ks,
            4.0 * (B * H * W * Cin + Cexp * (Cin + ks * ks)) + ob * B * Ho * Wo * Cexp)


def stem_dw_work(sig):
    """(B, H, W, C, Hs, Ws, Ho, Wo, bf16 result, planar input): 3x3x3 stem + 3x3 depthwise; the image (three planes, or four
    interleaved channels) read once, the output written once."""
    B, H, W, C, Hs, Ws, Ho, Wo = sig[:8]
    ob = 2.0 if (len(sig) > 8 and sig[8]) else 4.0
    cin = 3 if (len(sig) > 9 and sig[9]) else 4
    return 2.0 * B * Hs * Ws * 27 * C + 2.0 * B * Ho * Wo * C * 9, 4.0 * B * H * W * cin + ob * B * Ho * Wo * C


def linear_x3_work(sig):
    """(M, K, N, residual, gate): fp32 activations, three bf16 weight planes."""
    M, K, Nn, has_res = sig[:4]
    return 2.0 * M * K * Nn, 4.0 * M * K + 6.0 * K * Nn + 4.0 * M * Nn * (2 if has_res else 1)


def linear_thin_work(sig):
    """(M, K, N, residual, gate): fp32 activations and weights."""
    M, K, Nn, has_res = sig[:4]
    return 2.0 * M * K * Nn, 4.0 * M * K + 4.0 * K * Nn + 4.0 * M * Nn * (2 if has_res else 1)


WORK = {"linear_f32x3": linear_x3_work, "linear_thin": linear_thin_work, "linear_bf16": lambda s: linear_bf16_work(s), "conv2d_bf16": lambda s: conv_work(tuple(s[:10]) + (False,)), "stem_dwconv2d": stem_dw_work, "conv2d": conv_work, "dwconv2d": dw_work, "linear": linear_work, "expand_dwconv2d": expand_dw_work,
        # training-step launches: (flops, algorithmic bytes)
        "gemm_nn": gemm_work, "gemm_tn": gemm_work,
        "col_reduce": lambda s: (2.0 * s[1] * s[2], 4.0 * s[1] * s[2] * (2 if s[0] >= 2 else 1)),
        "scale_bias_act": lambda s: (2.0 * s[0] * s[1], 4.0 * s[0] * s[1] * (3 if s[2] else 2)),
        "bn_act_bwd": lambda s: (8.0 * s[0] * s[1], 4.0 * s[0] * s[1] * 3),
        # both passes of the gated form: the sums read dy and x, the apply reads them again and writes dx
        "bn_act_bwd_se": lambda s: (18.0 * s[0] * s[1], 4.0 * s[0] * s[1] * 5),
        # BatchNorm1d of every decode step: statistics pass + apply; gradient sums + apply
        "bn_steps_fwd": lambda s: (6.0 * s[0] * s[1] * s[2], 4.0 * s[0] * s[1] * s[2] * 3),
        "bn_steps_bwd": lambda s: (12.0 * s[0] * s[1] * s[2], 4.0 * s[0] * s[1] * s[2] * 5),
        # (B, H, W, C, Cv, K, S, Ho, Wo, Cout): dz and the image read once
        "conv_wgrad_gather": lambda s: (2.0 * s[0] * s[7] * s[8] * s[9] * s[5] * s[5] * s[4], 4.0 * s[0] * (s[7] * s[8] * s[9] + s[1] * s[2] * s[3])),
        "act_pool": lambda s: (3.0 * s[0] * s[1] * s[2], 4.0 * s[0] * s[1] * s[2] * (2 if (len(s) < 4 or s[3]) else 1)),
        "act_gate": lambda s: (4.0 * s[0] * s[1] * s[2], 4.0 * s[0] * s[1] * s[2] * 2),
        "chan_affine": lambda s: (2.0 * s[0] * s[1] * s[2], 4.0 * s[0] * s[1] * s[2] * (2 if s[3] else 1)),
        "se_train_fwd": lambda s: (4.0 * s[0] * s[1] * s[2], 4.0 * (3 * s[0] * s[1] + 2 * s[1] * s[2])),
        "se_train_bwd": lambda s: (8.0 * s[0] * s[1] * s[2], 4.0 * (5 * s[0] * s[1] + 4 * s[1] * s[2])),
        "image_dot": lambda s: (2.0 * s[0] * s[1] * s[2], 4.0 * s[0] * s[1] * s[2] * (2 if s[3] else 1)),
        "dwconv_bwd_data": lambda s: (2.0 * s[0] * s[6] * s[7] * s[3] * s[4] * s[4], 4.0 * s[3] * s[0] * (s[1] * s[2] + s[6] * s[7])),
        "dwconv_bwd_weight": lambda s: (2.0 * s[0] * s[6] * s[7] * s[3] * s[4] * s[4], 4.0 * s[3] * s[0] * (s[1] * s[2] + s[6] * s[7]))}


F32X3_MFMA_PEAK_TFLOPS = BF16_MFMA_PEAK_TFLOPS / 6.0     # six bf16 products per fp32-accurate multiply-add


def mfma_peak_for(name):
    """Dense matrix-core peak for the arithmetic a launch uses, in TFLOP/s of ALGORITHMIC (2 M N K) work."""
    if name in ("linear_bf16", "conv2d_bf16"):
        return BF16_MFMA_PEAK_TFLOPS
    if name == "linear_f32x3":
        return F32X3_MFMA_PEAK_TFLOPS
    return F32_MFMA_PEAK_TFLOPS


# ---------------------------------------------------------------- roofline of a whole traced step
def launch_roof_ms(name, sig):
    """(bound, milliseconds the launch would take AT its binding roofline): the larger of algorithmic flops / dense MFMA peak of
    the dtype it multiplies in and algorithmic bytes / HBM peak."""
    flops, byts = WORK[name](sig)
    peak = mfma_peak_for(name)
    t_mfma, t_hbm = flops / (peak * 1e12) * 1e3, byts / (HBM_PEAK_GBS * 1e9) * 1e3
    return ("mfma", t_mfma) if t_mfma >= t_hbm else ("hbm", t_hbm)


def step_roofline(rec, step_ms):
    """rec: (phase, name, sig, ms) of every traced launch of ONE step; step_ms: the measured time of a step.  Returns the
    time-weighted roofline fraction of the whole step (time outside the traced launches - row kernels, gaps - counts as zero
    work), the same per phase, and the kernel shape that loses the most time against its roofline."""
    tot_ms = sum(r[3] for r in rec)
    tot_roof = sum(launch_roof_ms(r[1], r[2])[1] for r in rec)
    phases, shapes = {}, {}
    for ph, n, sg, ms in rec:
        bound, roof = launch_roof_ms(n, sg)
        fl, by = WORK[n](sg)
        d = phases.setdefault(ph, {"ms": 0.0, "roof_ms": 0.0, "flops": 0.0, "bytes": 0.0, "launches": 0})
        d["ms"] += ms; d["roof_ms"] += roof; d["flops"] += fl; d["bytes"] += by; d["launches"] += 1
        k = shapes.setdefault((n, sg), {"ms": 0.0, "roof_ms": 0.0, "n": 0, "bound": bound})
        k["ms"] += ms; k["roof_ms"] += roof; k["n"] += 1
    (wn, ws), wk = max(shapes.items(), key=lambda kv: kv[1]["ms"] - kv[1]["roof_ms"])
    return {
        "step": {"frac": tot_roof / step_ms, "traced_frac": tot_roof / tot_ms, "step_ms": step_ms, "traced_ms": tot_ms,
                 "untraced_ms": max(step_ms - tot_ms, 0.0), "launches": len(rec),
                 "definition": "sum over traced launches of (time at the binding roofline: max(flops/MFMA peak, bytes/HBM peak)) / step time"},
        "phases": {ph: {"ms": d["ms"], "frac": d["roof_ms"] / d["ms"], "tflops": d["flops"] / d["ms"] / 1e9, "gbs": d["bytes"] / d["ms"] / 1e6,
                        "launches": d["launches"]} for ph, d in phases.items()},
        "worst": {"kernel": f"{wn}{list(ws)}", "bound": wk["bound"], "launches_per_step": wk["n"], "ms_per_step": wk["ms"],
                  "frac": wk["roof_ms"] / wk["ms"], "lost_ms_per_step": wk["ms"] - wk["roof_ms"]},
    }


def traced_step_roofline(torch, ops, step, step_ms, phase="train"):
    """One more step with an event pair around every traced launch (events on the launch stream) -> the `roofline` object of
    that step: the launch shape with the largest accumulated time priced against its binding roofline, plus
    step / phases / worst for the whole step (step_ms: the step time measured without events)."""
    rec = []
    PHASE[0] = phase
    ops.set_trace(lambda n, sg: rec.append((PHASE[0], n, sg, torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))) or rec[-1][3:])
    step()
    torch.cuda.synchronize()
    ops.set_trace(None)
    rows = [(ph, n, sg, e0.elapsed_time(e1)) for ph, n, sg, e0, e1 in rec]
    agg = {}
    for _, n, sg, ms in rows:
        a = agg.setdefault((n, sg), [0.0, 0])
        a[0] += ms
        a[1] += 1
    (n, sg), (tot, cnt) = max(agg.items(), key=lambda kv: kv[1][0])
    dom_ms = tot / cnt
    flops, byts = WORK[n](sg)
    peak = mfma_peak_for(n)
    if byts / (HBM_PEAK_GBS * 1e9) >= flops / (peak * 1e12):
        roof = {"bound": "hbm", "achieved": byts / (dom_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s"}
    else:
        roof = {"bound": "mfma", "achieved": flops / (dom_ms * 1e-3) / 1e12, "peak": peak, "unit": "TFLOP/s"}
    roof["frac"] = roof["achieved"] / roof["peak"]
    roof["kernel"] = f"{n}{list(sg)}"
    roof["launches_per_step"] = cnt
    roof["avg_launch_ms"] = dom_ms
    roof["traffic"] = pmc_traffic(roof["kernel"])
    roof.update(step_roofline(rows, step_ms))
    return roof


# ---------------------------------------------------------------- verification of what was timed
GAP_EXEMPT = 1e-4      # ids may differ only from a step on whose top-2 logit gap is below this (BASELINE north_star)


def verify_caption(torch, ops, model, images, ids_timed, n_oracle=16):
    """The timed step ran tile configurations chosen by the tuner.  Check its token ids (a) for all images against a
    rerun on the untuned (heuristic) configurations with logits kept, and (b) for the first n_oracle images against the
    CPU oracle's prefix-re-run greedy decode on the oracle's own encoder features; ids must agree up to the first step
    whose top-2 logit gap is below GAP_EXEMPT."""
    from oracle import backbones_ref as BR, torch_ref as R
    B = images.shape[0]
    mode = ops.compute_dtype()
    ops.set_compute_dtype("f32")              # the rerun is plain fp32 on the heuristic configurations, whatever mode was timed
    try:
        with ops.untuned():
            feats = model.cnn_model(images)
            ids_h, logits_h = model.greedy_ids_from_features(feats, CAPTION_LEN, want_logits=True)
    finally:
        ops.set_compute_dtype(mode)
    top2 = torch.topk(logits_h, 2, dim=2).values                              # [T, B, 2]
    gaps = (top2[..., 0] - top2[..., 1]).t().contiguous()                     # [B, T]
    diff = ids_timed != ids_h
    first_small = torch.where((gaps < GAP_EXEMPT).any(1), (gaps < GAP_EXEMPT).float().argmax(1), torch.full((B,), CAPTION_LEN, device=gaps.device))
    first_diff = torch.where(diff.any(1), diff.float().argmax(1), torch.full((B,), CAPTION_LEN, device=gaps.device))
    bad_rerun = int(((first_diff < first_small) & diff.any(1)).sum().item())     # differing at the small-gap step itself is allowed
    res = {"rerun_untuned_images": B, "rerun_rows_differing": int(diff.any(1).sum().item()), "rerun_mismatch_outside_gap": bad_rerun,
           "min_top2_gap": float(gaps.min().item())}
    sd = {k: (v.detach().cpu().float() if v.is_floating_point() else v.detach().cpu()) for k, v in model.state_dict().items()}
    n = min(n_oracle, B)
    bad_oracle, exempt, feat_err = 0, 0, 0.0
    with torch.no_grad():
        f_cpu = BR.cnn_model(sd, images[:n].detach().cpu())
        feat_err = float(((feats[:n].cpu() - f_cpu).abs().max() / f_cpu.abs().max()).item())
        got = ids_timed[:n].cpu().tolist()
        for r in range(n):
            want, lg = R.tfm_greedy(sd, f_cpu[r], 8, 8, CAPTION_LEN, 2, 3, False)
            t2 = torch.topk(lg, 2, dim=1).values
            g = (t2[:, 0] - t2[:, 1]).tolist()
            for t in range(CAPTION_LEN):
                if g[t] < GAP_EXEMPT:
                    exempt += 1
                    break
                if got[r][t] != want[t]:
                    bad_oracle += 1
                    break
    res.update({"oracle_images": n, "oracle_mismatch_outside_gap": bad_oracle, "oracle_rows_cut_by_gap_rule": exempt,
                "encoder_features_max_rel_err_vs_oracle": feat_err, "gap_exempt": GAP_EXEMPT})
    res["ok"] = bad_rerun == 0 and bad_oracle == 0 and feat_err < 1e-3
    return res


def verify_train(torch, ops, dev, rank, B, tuned, seed_state):
    """The training secondary's first steps ran tuned tile configurations (and, in f32x3 mode, split-product GEMMs) in the
    forward pass, the backward pass, the weight-gradient GEMMs and AdamW.  The same three steps (same weights, inputs,
    dropout seed) in plain fp32 on the untuned configurations must give the same loss at step 1 (forward), the same norm
    of the flat gradient after step 1 (backward and weight gradients) and the same loss at step 3 (two optimizer updates
    later).  tuned = {"loss1", "grad_norm1", "loss3"} of the timed trainer."""
    from image_captioning_amd._bah_train import DropoutState
    with ops.untuned():
        _, step = build_train_step(torch, dev, rank, B)
        DropoutState.manual_seed(seed_state)
        loss1 = float(step())
        gn1 = float(step.trainer.opt.flat.flat_g.double().norm().item())
        step()
        loss3 = float(step())
    del step
    torch.cuda.empty_cache()

    def rel(a, b):
        return abs(a - b) / max(abs(b), 1e-12)
    r1, rg, r3 = rel(tuned["loss1"], loss1), rel(tuned["grad_norm1"], gn1), rel(tuned["loss3"], loss3)
    return {"loss_tuned": tuned["loss1"], "loss_untuned": loss1, "rel_diff": r1,
            "grad_norm_step1_tuned": tuned["grad_norm1"], "grad_norm_step1_untuned": gn1, "grad_norm_rel_diff": rg,
            "loss_step3_tuned": tuned["loss3"], "loss_step3_untuned": loss3, "loss_step3_rel_diff": r3,
            "tolerances": {"loss_step1": 1e-4, "grad_norm_step1": TRAIN_VERIFY_GRAD_TOL, "loss_step3": TRAIN_VERIFY_LOSS3_TOL},
            "ok": r1 < 1e-4 and rg < TRAIN_VERIFY_GRAD_TOL and r3 < TRAIN_VERIFY_LOSS3_TOL}


TRAIN_VERIFY_GRAD_TOL = 2e-3      # norm of all 16.6 M gradients, split-product (fp32-accurate) against plain fp32 products
TRAIN_VERIFY_LOSS3_TOL = 5e-4     # loss after two AdamW updates


# ---------------------------------------------------------------- one caption at a time (weight-streaming decode steps)
DECODE_PARAMS = {"char_transformer": 0.57e6, "lstm": 7.80e6, "gru": 6.75e6, "word_transformer": 5.56e6, "bahdanau": 6.77e6}


def decode_b1(torch, dev):
    """BASELINE configs[3] (char-level long-sequence decode) and the other three decoders with ONE caption in flight:
    microseconds per greedy step and the fraction of the HBM roofline that algorithmic bytes (parameters touched per step
    x 4) / step time reaches.  Events on torch's current stream, which is the stream the drivers launch on."""
    from image_captioning_amd import seq2seqbasic as s2s, transformer_word_level as twl, transformer_word_level_bahdnau as bah
    from image_captioning_amd import transformercharmiguel as chr_
    from image_captioning_amd.synthetic import seeded_state_dict
    import torch.nn as nn

    class Pass(nn.Module):
        def forward(self, x):
            return x

    def seeded(m, seed):
        m.load_state_dict(seeded_state_dict(m, seed=seed))
        return m.eval().to(dev)

    def timeit(fn, reps=20):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    g = torch.Generator().manual_seed(3)
    f512, f256 = torch.randn(1, 512, generator=g).to(dev), torch.randn(1, 256, generator=g).to(dev)
    out = {}
    cm = seeded(chr_.ImageCaptioningModel(Pass(), chr_.TransformerEncoderBlock(512, 512, 2), chr_.TransformerDecoderBlock(512, 512, 4), embed_dim=512), 21)
    jobs = [("char_transformer", lambda: cm.greedy_ids_from_features(f512, 58), 58)]
    lstm_models = []
    for kind in ("lstm", "gru"):
        m = seeded(s2s.DecoderRNN(256, 512, VOCAB, num_layers=2, rnn_type=kind), 11)
        lstm_models.append((kind, m))
        jobs.append((kind, (lambda m=m: m.sample_ids(f256, CAPTION_LEN)), CAPTION_LEN))
    twl.vocab_size = VOCAB
    wm = seeded(twl.ImageCaptioningModel(Pass(), twl.TransformerEncoderBlock(512, 2048, 8), twl.TransformerDecoderBlock(512, 2048, 8), embed_dim=512), 21)
    jobs.append(("word_transformer", lambda: wm.greedy_ids_from_features(f512, CAPTION_LEN), CAPTION_LEN))
    bm = seeded(bah.ImageCaptioningModel(Pass(), bah.TransformerEncoderBlock(512, 512, 8, dropout=0.5), bah.BahdanauDecoder(512, 512, 512, 256, VOCAB, dropout=0.5)), 31)
    jobs.append(("bahdanau", lambda: bm.greedy_ids_from_features(f512, CAPTION_LEN), CAPTION_LEN))
    modes = {"char_transformer": "next-token table: one batched decoder pass over 81 rows, then 58 look-ups (the decoder has no self-attention or positional encoding, so logits depend on the last token and the image only)",
             "lstm": "2 launches per step (layer 0's cell in the prologue of layer 1's input rows; layer 1's cell in the prologue of the vocabulary projection, which also carries both layers' recurrent rows for the next step)",
             "gru": "2 launches per step (as lstm)", "word_transformer": "3 launches per step",
             "bahdanau": "1 launch per step (cell in the prologue; vocabulary projection + recurrent rows for the next step)"}
    for name, fn, steps in jobs:
        ms = timeit(fn)
        us = ms * 1e3 / steps
        gbs = DECODE_PARAMS[name] * 4 / us / 1e3
        out[name] = {"us_per_step": us, "steps": steps, "captions_per_s": 1e3 / ms, "bytes_per_step": DECODE_PARAMS[name] * 4,
                     "achieved_gbs": gbs, "hbm_frac": gbs / HBM_PEAK_GBS, "mode": modes[name]}
        if name == "char_transformer":
            # the table path reads the weights once per CAPTION (one batched pass over the 81 tokens), not once per step: its
            # figure of merit is captions/s; the bandwidth fraction of the char model is that of the step-by-step forms below
            for k in ("bytes_per_step", "achieved_gbs", "hbm_frac"):
                del out[name][k]
            out[name]["us_per_step_equivalent"] = out[name].pop("us_per_step")
    # the same captions replayed from a whole-caption HIP graph (ops.set_decode_graph): launches inside a graph follow each
    # other faster than stream-ordered ones; launches per step as in `mode`
    from image_captioning_amd import ops as _ops
    launches = {"lstm": 2, "gru": 2, "word_transformer": 3, "bahdanau": 1}
    # the recurrent decoders with one launch per cell (three per step for the two-layer LSTM / GRU, two for the LSTMCell decoder)
    _ops.set_decode_fused(False)
    try:
        for name, fn, steps in jobs[1:]:
            if name == "word_transformer":
                continue
            ms = timeit(fn)
            us = ms * 1e3 / steps
            gbs = DECODE_PARAMS[name] * 4 / us / 1e3
            out[name]["launch_per_cell"] = {"us_per_step": us, "achieved_gbs": gbs, "hbm_frac": gbs / HBM_PEAK_GBS,
                                            "launches_per_step": launches[name] + 1}
    finally:
        _ops.set_decode_fused(True)
    # what the fused forms actually stream per step: W_ih of the token-fed cell has become a table row (4 H floats per step)
    streamed = {"lstm": DECODE_PARAMS["lstm"] - 4 * 512 * 512, "gru": DECODE_PARAMS["gru"] - 3 * 512 * 512,
                "bahdanau": 512 * VOCAB + 4 * 512 * 512 + 2 * VOCAB}
    for name in launches:
        out[name]["us_per_dependent_launch"] = out[name]["us_per_step"] / launches[name]
        if name in streamed:
            out[name]["streamed_bytes_per_step"] = streamed[name] * 4
            out[name]["hbm_frac_streamed"] = streamed[name] * 4 / out[name]["us_per_step"] / 1e3 / HBM_PEAK_GBS
    _ops.set_decode_graph(True)
    try:
        for name, fn, steps in jobs[1:]:
            ms = timeit(fn)
            us = ms * 1e3 / steps
            gbs = DECODE_PARAMS[name] * 4 / us / 1e3
            out[name]["graph"] = {"us_per_step": us, "achieved_gbs": gbs, "hbm_frac": gbs / HBM_PEAK_GBS,
                                  "us_per_dependent_launch": us / launches[name],
                                  "note": "whole caption (every step's launches) replayed from one HIP graph; includes the copy into "
                                          "the graph's input buffer and the copy of the ids out of it"}
    finally:
        _ops.set_decode_graph(False)
    # bf16 weight streaming (icap_greedy_*_w16_f32): the weight matrices are read as bf16 copies, everything else stays fp32
    _ops.set_decode_weights("bf16")
    try:
        for name, fn, steps in jobs[1:]:
            ms = timeit(fn)
            us = ms * 1e3 / steps
            gbs = DECODE_PARAMS[name] * 2 / us / 1e3
            out[name]["bf16_weights"] = {"us_per_step": us, "bytes_per_step": DECODE_PARAMS[name] * 2, "achieved_gbs": gbs,
                                         "hbm_frac": gbs / HBM_PEAK_GBS,
                                         # the bf16-weight drivers are launch-per-cell forms: compared with the fp32 form of the same structure
                                         "speedup_vs_f32_weights_same_driver": out[name].get("launch_per_cell", out[name])["us_per_step"] / us,
                                         "tolerance": "logits within 2e-2 of max |logit|, ids equal outside twice the logit error "
                                                      "(tests/test_gpu_decode_w16.py)"}
    finally:
        _ops.set_decode_weights(None)
    # the char model step by step, for comparison with its table path: the step-per-launch driver and the persistent launch
    for mode, label in (("steps", "step_by_step"), ("persistent", "persistent")):
        ms = timeit(lambda: cm.greedy_ids_from_features(f512, 58, mode=mode))
        us = ms * 1e3 / 58
        gbs = DECODE_PARAMS["char_transformer"] * 4 / us / 1e3
        out["char_transformer"][label] = {"us_per_step": us, "bytes_per_step": DECODE_PARAMS["char_transformer"] * 4, "achieved_gbs": gbs,
                                         "hbm_frac": gbs / HBM_PEAK_GBS}
    # the decoder calls alone, without the per-image work in front of them (encoder block over the image's row, cross-attention /
    # feature projection: ~10 launches per caption that the model-level figures above include)
    st1 = torch.full((1,), 2, device=dev, dtype=torch.int64)
    for kind in ("lstm", "gru"):
        m = [j for j in jobs if j[0] == kind][0][1].__defaults__[0]
        wd_r = m._weights()
        x0 = _ops.linear(f256, wd_r["ff_w"], bias=wd_r["ff_b"])
        out[kind]["decoder_call_only_us_per_step"] = timeit(lambda: _ops.greedy_rnn(wd_r, x0, CAPTION_LEN)) * 1e3 / CAPTION_LEN
    wd_t = wm._greedy_weights()
    ca1 = wm.decoder.cross_attention_single(wm.encoder.forward_independent(f512))
    out["word_transformer"]["decoder_call_only_us_per_step"] = timeit(lambda: _ops.greedy_tfm(wd_t, ca1, st1, CAPTION_LEN)) * 1e3 / CAPTION_LEN
    try:
        ok_ids = bool(torch.equal(_ops.greedy_tfm_persist(wd_t, ca1, st1, CAPTION_LEN)[0], _ops.greedy_tfm(wd_t, ca1, st1, CAPTION_LEN)[0]))
        out["word_transformer"]["persistent"] = {
            "decoder_call_only_us_per_step": timeit(lambda: _ops.greedy_tfm_persist(wd_t, ca1, st1, CAPTION_LEN, check=False)) * 1e3 / CAPTION_LEN,
            "ids_equal_default_path": ok_ids}
    except Exception as e:      # noqa: BLE001
        out["word_transformer"]["persistent"] = {"error": f"{type(e).__name__}: {e}"}
    # the LSTMCell decoder's caption as one persistent launch (weights LDS-resident, one exchange between workgroups per step)
    try:
        same = bool(torch.equal(bm.greedy_ids_from_features(f512, CAPTION_LEN, mode="persistent"), bm.greedy_ids_from_features(f512, CAPTION_LEN)))
        ms = timeit(lambda: bm.greedy_ids_from_features(f512, CAPTION_LEN, mode="persistent", check=False))
        us = ms * 1e3 / CAPTION_LEN
        # the decoder call alone (encoder-block rows of the image computed beforehand): at 133 us of GPU time per caption the
        # model-level call above is bound by the host's ~10 launches in front of the decoder, not by the decoder
        pw = bm.decoder._weights()
        ctx1 = bm.encoder.rows_forward(f512)
        st1 = torch.full((1,), bm.start_token, device=dev, dtype=torch.int64)
        us_dec = timeit(lambda: _ops.greedy_bah_persist(pw, ctx1, st1, CAPTION_LEN, check=False)) * 1e3 / CAPTION_LEN
        us_dec_fused = timeit(lambda: _ops.greedy_bah(pw, ctx1, st1, CAPTION_LEN)) * 1e3 / CAPTION_LEN
        gbs = DECODE_PARAMS["bahdanau"] * 4 / us / 1e3
        out["bahdanau"]["persistent"] = {"us_per_step": us, "achieved_gbs": gbs, "hbm_frac": gbs / HBM_PEAK_GBS,
                                         "decoder_call_only": {"us_per_step": us_dec, "hbm_frac": DECODE_PARAMS["bahdanau"] * 4 / us_dec / 1e3 / HBM_PEAK_GBS,
                                                               "default_path_us_per_step": us_dec_fused},
                                         "ids_equal_default_path": same,
                                         "note": "one launch per caption incl. its set-up (18 MB of weights into LDS, exchange buffers cleared); timed "
                                                 "without the read-back of the ids that detects a launch which could not run (checked once before); "
                                                 "opt-in (mode=\"persistent\"): needs the device to itself"}
    except Exception as e:      # noqa: BLE001 - a secondary figure must not take the line down
        out["bahdanau"]["persistent"] = {"error": f"{type(e).__name__}: {e}"}
    # the two-layer LSTM's / GRU's caption as one persistent launch (weights LDS-resident, two exchanges between workgroups per step)
    for kind, lm in lstm_models:
        try:
            same = bool(torch.equal(lm.sample_ids(f256, CAPTION_LEN, mode="persistent"), lm.sample_ids(f256, CAPTION_LEN)))
            ms = timeit(lambda: lm.sample_ids(f256, CAPTION_LEN, mode="persistent", check=False))
            us = ms * 1e3 / CAPTION_LEN
            lw = lm._weights()
            x1 = _ops.linear(f256, lw["ff_w"], bias=lw["ff_b"])
            us_dec = timeit(lambda: _ops.greedy_rnn2_persist(lw, x1, CAPTION_LEN, check=False)) * 1e3 / CAPTION_LEN
            gbs = DECODE_PARAMS[kind] * 4 / us / 1e3
            out[kind]["persistent"] = {"us_per_step": us, "achieved_gbs": gbs, "hbm_frac": gbs / HBM_PEAK_GBS,
                                       "decoder_call_only": {"us_per_step": us_dec, "hbm_frac": DECODE_PARAMS[kind] * 4 / us_dec / 1e3 / HBM_PEAK_GBS},
                                       "ids_equal_default_path": same,
                                       "note": "one launch per caption incl. its set-up (20-26 MB of weights into LDS, exchange buffers cleared), two "
                                               "exchanges between the workgroups per step; timed without the read-back of the ids that detects a "
                                               "launch which could not run (checked once before); opt-in (mode=\"persistent\"): needs the device to itself"}
        except Exception as e:      # noqa: BLE001 - a secondary figure must not take the line down
            out[kind]["persistent"] = {"error": f"{type(e).__name__}: {e}"}
    out["note"] = ("one caption in flight, fp32 weights, whole caption enqueued without host round trips; bytes = parameters touched per step x 4; "
                   "the char model's default (table) path is reported as captions/s and us per step-equivalent (caption time / 58): its weights are read once per caption; its bandwidth figures are those of the step-by-step forms (steps: one launch chain per step; persistent: one launch per caption)")
    return out



CAPTION_DTYPES = {"f32": "f32", "f32x3": "f32x3"}
DTYPE_NOTES = {"f32": "every product on the fp32 matrix-core instruction, fp32 tensors",
               "f32x3": "fp32-accurate, not the plain-fp32 instruction path of the `f32` lines (`plain_f32`): the MFMA-bound GEMMs (1x1 convolutions "
                        "with a long contraction, the fused MBConv expand GEMMs, the decoder's fc1 / vocabulary projection) as six exact bf16 "
                        "products per multiply from three-way operand splits on the bf16 matrix cores, fp32 accumulate; every other kernel plain "
                        "fp32; fp32 tensors; the plain-fp32 figure of the same step is in `plain_f32`",
               "bf16": "bf16 products / f32 accumulate / f32 storage except the fused expand+depthwise results (bf16)"}

TRAIN_WORKLOAD = ("transformer_word_level_bahdnau: teacher-forced training step (EfficientNet-B0 train mode + encoder block + "
                  "Bahdanau LSTMCell decoder, compute_loss, backward, AdamW)")
TRAIN_DTYPES = {"f32": "f32",
                "f32x3": "f32 (fp32-accurate: forward GEMMs and input gradients of the large layers as six exact bf16 products per multiply "
                         "from three-way operand splits, fp32 accumulate; weight gradients and every other kernel plain fp32)",
                "bf16": "bf16 products in the forward GEMMs and input gradients / f32 accumulation, master weights, activations, "
                        "weight gradients and optimizer"}
TRAIN_BF16_TOLERANCE = ("loss within 2e-2, gradients within 0.1 relative L2 (cosine > 0.99) of the f32 step "
                        "(tests/test_gpu_train_bf16.py)")


def build_train_step(torch, dev, rank, B, dtype="f32", graph=False):
    """The Bahdanau captioner in train mode on `dev` with its Trainer (data-parallel when a process group is up) and one
    synthetic batch -> (model, step)."""
    from image_captioning_amd import transformer_word_level_bahdnau as bah
    from image_captioning_amd._bah_train import DropoutState
    from image_captioning_amd.synthetic import seeded_state_dict, synthetic_captions, synthetic_images
    from image_captioning_amd.training import Trainer
    pool = synthetic_images(min(B, 32), seed=1234 + rank)
    images = pool.repeat((B + pool.shape[0] - 1) // pool.shape[0], 1, 1, 1)[:B].contiguous().to(dev)
    model = bah.build_model(VOCAB, dropout=0.3)                       # transformer_word_level_bahdnau.py:171,183-188
    model.load_state_dict(seeded_state_dict(model, seed=52))
    model = model.to(dev)
    DropoutState.manual_seed(1000 + rank)                             # dropout differs per rank, weights do not
    trainer = Trainer(model, lr=1e-4, weight_decay=1e-5, dtype=dtype, graph=graph)  # :191
    caps = synthetic_captions(B, CAPTION_LEN, VOCAB, 2, 3, 0, seed=77 + rank).to(dev)

    def step():
        return trainer.step(images, caps, 0)

    step.trainer = trainer
    return model, step


def train_parallelism(model, world):
    if world == 1:
        return "single GPU, no collective"
    return f"data-parallel x{world}, bucketed all-reduce of {sum(p.numel() for p in model.parameters()) * 4 / 1e6:.1f} MB of gradients"


def train_secondary(torch, dist, ops, dev, world, rank, steps, B=256, dtype="f32", partial=None, cpu_base=False):
    note(rank, f"train secondary: building the model, batch {B}, {dtype}")
    """The second half of the metric, measured after the contract's timed region on the same ranks: K training steps
    between barriers, max over ranks -> the "train" object of the JSON line."""
    from image_captioning_amd._bah_train import DropoutState
    model, step = build_train_step(torch, dev, rank, B, dtype)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    ops.set_autotune(os.environ.get("ICAP_BENCH_NO_TUNE") != "1")
    DropoutState.manual_seed(1000 + rank)
    first_loss = float(step())
    torch.cuda.synchronize()
    ops.set_autotune(False)
    note(rank, "train secondary: first step (tuning) done")
    tuned = {"loss1": first_loss, "grad_norm1": float(step.trainer.opt.flat.flat_g.double().norm().item())}
    step()
    tuned["loss3"] = float(step())
    check = verify_train(torch, ops, dev, rank, B, tuned, 1000 + rank)      # every rank: the step contains the gradient exchange
    note(rank, f"train secondary: verification {check}")
    DropoutState.manual_seed(2000 + rank)
    for _ in range(2):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    barrier()
    elapsed = time.perf_counter() - t0
    note(rank, f"train secondary: {steps} steps in {elapsed:.2f} s")
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    red = step.trainer.reducer
    res = {"value": world * B * steps / elapsed, "unit": "imgs/s", "ms_per_step": elapsed / steps * 1e3, "steps": steps, "warmup": 5,
           "images_per_step_per_gpu": B, "dtype": dtype, "dtype_note": TRAIN_DTYPES[dtype], "scaling": "weak", "loss_last_step": float(loss),
           "workload": TRAIN_WORKLOAD, "parallelism": train_parallelism(model, world),
           "gradient_exchange": "none (single GPU)" if red is None else
                                ("bucketed all-reduce launched by hooks during backward" if red.overlap else "bucketed all-reduce after backward")}
    res["verified"], res["verification"] = check["ok"], check                    # rank 0's own check is the one reported
    # roofline of this step: every rank runs the traced step (it contains the gradient exchange), rank 0 reports it
    res["roofline"] = traced_step_roofline(torch, ops, step, res["ms_per_step"])

    if partial is not None:
        partial.update(res)            # what a later stall must not lose

    def timed_as(kind, label):
        step.trainer.dtype = kind
        for _ in range(2):
            step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            last = step()
        barrier()
        dt = time.perf_counter() - t0
        note(rank, f"train secondary, {label}: {steps} steps in {dt:.2f} s")
        if world > 1:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return {"value": world * B * steps / dt, "unit": "imgs/s", "ms_per_step": dt / steps * 1e3, "dtype": TRAIN_DTYPES[kind],
                "loss_last_step": float(last)}

    if red is not None and not red.overlap and not red.host_staged:
        # RCCL: the default keeps the exchange after backward (launched from the calling thread); the hook-launched, overlapped
        # form is measured next, after the figure above is safe - it has never run on multi-GPU hardware before this line
        red.overlap = True
        try:
            ov = timed_as(dtype, "overlapped gradient exchange (hooks)")
            ov["gradient_exchange"] = "bucketed all-reduce launched by hooks during backward"
            ov["buckets_launched_by_hooks_last_step"] = red.hooked
            res["overlapped"] = ov
            if ov["value"] < res["value"]:
                red.overlap = False        # the remaining variants run in the faster of the two forms
        except Exception as e:      # noqa: BLE001 - reported in the line
            res["overlapped"] = {"error": f"{type(e).__name__}: {e}"[:300]}
            red.overlap = False
        if partial is not None:
            partial.update(res)
    if dtype == "f32x3":        # the same trainer on the fp32 matrix-core instruction only, for comparison
        res["plain_f32"] = timed_as("f32", "plain fp32 products")
    # the same trainer with bf16 products (BASELINE configs[2] names bf16), same K steps between the same barriers
    res["bf16"] = dict(timed_as("bf16", "bf16 products"), tolerance=TRAIN_BF16_TOLERANCE)
    if partial is not None:
        partial.update(res)
    if cpu_base and rank == 0:
        res["cpu_baseline"] = cpu_train_baseline(torch, model, 12)
        note(rank, "train secondary: CPU baseline measured")
    # the step at the reference's loader batch (12), replayed from HIP graphs (Trainer(graph=True)): one graph in a single
    # process; under data parallelism three graphs with the finished buckets' all-reduce launched between the backward replays
    # (the staged step) - measured last: its first run over RCCL is this one
    if True:
        try:
            del model, step
            torch.cuda.empty_cache()
            gb = 12
            _, gstep = build_train_step(torch, dev, rank, gb, "f32", graph=True)
            ops.set_autotune(True)
            gstep()
            torch.cuda.synchronize()
            ops.set_autotune(False)
            for _ in range(4):              # second eager step, the recording with its first replay, two more replays
                gstep()
            barrier()
            t0 = time.perf_counter()
            for _ in range(5 * steps):
                loss = gstep()
            barrier()
            elapsed = time.perf_counter() - t0
            if world > 1:
                t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                elapsed = float(t.item())
            gtr = gstep.trainer
            res["graph_b12"] = {"value": world * gb * 5 * steps / elapsed, "unit": "imgs/s", "ms_per_step": elapsed / (5 * steps) * 1e3, "steps": 5 * steps,
                                "images_per_step_per_gpu": gb, "dtype": "f32", "loss_last_step": float(loss),
                                "note": "the reference's loader batch; zero_grad + forward + loss + backward + AdamW replayed from one HIP graph"
                                        if world == 1 else
                                        f"the reference's loader batch per GPU; three graphs (forward + backward to the CNN's output | CNN backward | AdamW), "
                                        f"{len(gtr._early_buckets)} of {len(gtr.reducer.buckets)} gradient buckets all-reduced under the CNN's backward"
                                        if gtr._graph_cnn is not None else "two graphs with the gradient exchange between them"}
            note(rank, f"train secondary, graph-replayed step at batch {gb}: {5 * steps} steps in {elapsed:.2f} s")
        except Exception as e:      # noqa: BLE001 - reported in the line
            res["graph_b12"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    return res


def with_deadline(seconds, on_timeout, fn):
    """fn() unless it takes longer than `seconds`: then on_timeout() runs on a timer thread and the process ends there
    (a rank stuck in a collective cannot be interrupted any other way)."""
    done = threading.Event()

    def fire():
        if not done.is_set():
            code = 3             # a stall is a failure: the line says so and so does the exit status ...
            try:
                code = on_timeout() or 3
            finally:
                os._exit(code)   # ... unless on_timeout says what stalled was an optional extra

    timer = threading.Timer(seconds, fire)
    timer.daemon = True
    timer.start()
    try:
        return fn()
    finally:
        done.set()
        timer.cancel()


T_START = time.perf_counter()
PHASE = ["step"]      # which part of the step the traced launches belong to (set by the step function)


def note(rank, msg):
    """Progress on stderr (stdout carries only the JSON line); every rank reports, so a stuck rank can be told apart."""
    print(f"[bench r{rank} +{time.perf_counter() - T_START:6.1f}s] {msg}", file=sys.stderr, flush=True)


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    if args.rehearse:
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ.get("ICAP_BENCH_DUMP_AFTER", "1e9")), repeat=True, file=sys.stderr)   # where a stuck rank is
        local %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        if args.rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    from image_captioning_amd import ops
    from image_captioning_amd import transformer_word_level as twl
    if args.mode != "train":
        ops.set_compute_dtype(args.dtype)
    from image_captioning_amd.synthetic import seeded_state_dict, synthetic_captions, synthetic_images

    train = args.mode == "train"
    B = args.batch or (256 if train else 1024)
    # synthetic images: a small seeded pool tiled to the batch (content does not change the work)
    pool = synthetic_images(min(B, 32), seed=1234 + rank)
    images = pool.repeat((B + pool.shape[0] - 1) // pool.shape[0], 1, 1, 1)[:B].contiguous().to(dev)
    if train:
        model, step = build_train_step(torch, dev, rank, B, args.dtype, args.graph)
    else:
        model = twl.build_model(VOCAB)
        model.load_state_dict(seeded_state_dict(model, seed=52))
        model = model.eval().to(dev)

        def step():
            PHASE[0] = "encoder"
            feats = model.cnn_model(images)
            PHASE[0] = "decode"
            return model.greedy_ids_from_features(feats, CAPTION_LEN)

    if args.breakdown:
        breakdown(torch, ops, step, B)
        return

    # ---- dominant-kernel timing with events on the launch stream (torch's current stream) ----
    dom_events = []
    dom = {"name": None, "sig": None}

    step_rec = []

    def find_dominant():
        rec = []
        ops.set_trace(lambda n, s: rec.append((PHASE[0], n, s, torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))) or rec[-1][3:])
        step()
        torch.cuda.synchronize()
        ops.set_trace(None)
        agg = {}
        for ph, n, s, e0, e1 in rec:
            ms = e0.elapsed_time(e1)
            step_rec.append((ph, n, s, ms))
            agg[(n, s)] = agg.get((n, s), 0.0) + ms
        (n, s), _ = max(agg.items(), key=lambda kv: kv[1])
        dom["name"], dom["sig"] = n, s

    def dom_trace(n, s):
        if n == dom["name"] and s == dom["sig"]:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            dom_events.append(ev)
            return ev
        return None

    # --graph: tuning and the dominant-kernel events come from eager steps (a replayed graph makes no traced launches; it
    # runs the same kernels), then the step is recorded and the timed region replays it
    graph_mode = train and args.graph
    if graph_mode:
        step.trainer.graph = False
    note(rank, f"process group and model ready ({'train' if train else 'caption'} mode, batch {B}, world {world})")
    ops.set_autotune(True)      # the first warm-up pass times the tile configurations of every GEMM shape once
    for _ in range(max(args.warmup, 1)):
        step()
    torch.cuda.synchronize()
    ops.set_autotune(False)
    note(rank, "warm-up and tuning done")
    find_dominant()
    ops.set_trace(dom_trace)
    if graph_mode:
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        ops.set_trace(None)
        step.trainer.graph = True
        for _ in range(4):          # two eager steps, the recording (with its first replay), one more replay
            step()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ids = step()
    barrier()
    elapsed = time.perf_counter() - t0
    ops.set_trace(None)
    note(rank, f"timed region: {args.steps} steps in {elapsed:.2f} s")
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert train or ids.shape == (B, CAPTION_LEN)
    verified = None
    if not train and rank == 0 and args.dtype in ("f32", "f32x3") and not args.no_verify:
        verified = verify_caption(torch, ops, model, images, ids)
        note(rank, f"verification of the timed step's ids: {verified}")

    # Secondary, after the contract's timed region: the same captioning step in the opt-in bf16-product mode (fp32
    # accumulation and storage), same K steps and barriers, reported beside the fp32 headline as "bf16_mode".
    alt = None
    if not train and args.dtype in ("f32", "f32x3") and not args.no_bf16_mode:
        ops.set_compute_dtype("bf16")
        for _ in range(2):
            step()
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
        barrier()
        alt_elapsed = time.perf_counter() - t1
        ops.set_compute_dtype(args.dtype)
        note(rank, f"bf16-mode secondary: {args.steps} steps in {alt_elapsed:.2f} s")
        if world > 1:
            t = torch.tensor([alt_elapsed], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            alt_elapsed = float(t.item())
        alt = {"dtype": "bf16 products / f32 accumulate / f32 storage except the fused expand+depthwise results (bf16)", "value": world * B * args.steps / alt_elapsed,
               "unit": "captions/s", "ms_per_step": alt_elapsed / args.steps * 1e3,
               "tolerance": "features 1e-2, logits 2e-2 of their largest magnitude vs the f32 path (tests/test_gpu_bf16.py)"}

    # the same step on the plain fp32 matrix-core instruction (the `f32` path), same K steps and barriers
    plain = None
    if not train and args.dtype == "f32x3" and not args.no_plain_f32:
        ops.set_compute_dtype("f32")
        ops.set_autotune(True)
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        ops.set_autotune(False)
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
        barrier()
        pl_elapsed = time.perf_counter() - t1
        ops.set_compute_dtype(args.dtype)
        note(rank, f"plain-fp32 secondary: {args.steps} steps in {pl_elapsed:.2f} s")
        if world > 1:
            t = torch.tensor([pl_elapsed], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            pl_elapsed = float(t.item())
        plain = {"dtype": "f32", "dtype_note": DTYPE_NOTES["f32"], "value": world * B * args.steps / pl_elapsed, "unit": "captions/s",
                 "ms_per_step": pl_elapsed / args.steps * 1e3}

    ops.set_compute_dtype("f32")      # everything after the timed region (single-caption decode, training) runs in its own modes
    out = None
    if rank == 0:
        captions = world * B * args.steps
        dom_ms = sum(e0.elapsed_time(e1) for e0, e1 in dom_events) / max(len(dom_events), 1)
        flops, byts = WORK[dom["name"]](dom["sig"])
        hbm_t = byts / (HBM_PEAK_GBS * 1e9)
        mfma_peak = mfma_peak_for(dom["name"])
        mfma_t = flops / (mfma_peak * 1e12)
        if hbm_t >= mfma_t:
            roof = {"bound": "hbm", "achieved": byts / (dom_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s"}
        else:
            roof = {"bound": "mfma", "achieved": flops / (dom_ms * 1e-3) / 1e12, "peak": mfma_peak, "unit": "TFLOP/s"}
        roof["frac"] = roof["achieved"] / roof["peak"]
        roof["kernel"] = f"{dom['name']}{list(dom['sig'])}"
        if dom["name"] == "linear_f32x3":
            roof["frac_of_f32_mfma_peak"] = roof["achieved"] / F32_MFMA_PEAK_TFLOPS     # the figure comparable with an fp32-MFMA kernel's frac
            roof["peak_note"] = ("fp32-accurate GEMM on the bf16 matrix cores: six bf16 products per multiply-add, so the peak for algorithmic "
                                 "2MNK flops is the dense bf16 peak 2500 / 6; the fp32 MFMA instruction's own peak is 157.3")
        roof["traffic"] = pmc_traffic(roof["kernel"])
        roof["avg_launch_ms"] = dom_ms
        roof.update(step_roofline(step_rec, elapsed / args.steps * 1e3))
        if train:
            workload = TRAIN_WORKLOAD
            par = train_parallelism(model, world) + (", step replayed from HIP graphs" if args.graph else "")
        else:
            workload = ("transformer_word_level: EfficientNet-B0 encoder + word-level transformer decoder, "
                        "batched greedy captioning (BASELINE configs[1])")
            par = f"image-sharded x{world}, no collective"
        out = {
            "metric": "captions/sec greedy + train imgs/sec, 224px food imgs, 1/2/4/8 MI355X",
            "value": captions / elapsed, "unit": "imgs/s" if train else "captions/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, **({"rehearsal": "gloo, shared devices: not a result"} if args.rehearse else {}), "dtype": TRAIN_DTYPES[args.dtype] if train else (CAPTION_DTYPES[args.dtype] if args.dtype != "bf16" else "bf16"),
            "dtype_note": TRAIN_DTYPES[args.dtype] if train else DTYPE_NOTES[args.dtype], "data": "synthetic",
            "config": {"workload": workload, "images_per_step_per_gpu": B, "image": "3x224x224 fp32", "caption_tokens": CAPTION_LEN,
                       "vocab": VOCAB, "weights": "seeded random init", "parallelism": par,
                       "images": f"{min(B, 32)} unique seeded synthetic images tiled to the batch of {B} (distinct addresses: same HBM traffic as {B} different images)"},
            "roofline": roof,
        }
        if verified is not None:
            out["verified"] = verified["ok"]
            out["verification"] = verified
        if plain is not None:
            out["plain_f32"] = plain
        if alt is not None:
            out["bf16_mode"] = alt
        if not train and world == 1 and not args.no_decode_b1:
            out["decode_b1"] = decode_b1(torch, dev)
            note(rank, "single-caption decode steps measured")
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_train_baseline(torch, model, min(B, 12)) if train else cpu_baseline(torch, model, args.cpu_sample)

    # Secondary, after everything the contract asks for: the training half of the metric on the same ranks, so that the
    # data-parallel path is measured wherever this line is.  It cannot take the line down with it: an exception becomes an
    # "error" entry, and a rank that does not come back within the deadline prints the line without it and exits.
    if not train and not args.no_train_mode:
        partial = {}

        def emit_without():
            # the default-form training figure (value, verified) is in `partial` once it is measured: a stall after that hit one of
            # the optional variants (hook-launched exchange, staged graphs - neither has run over RCCL before) and is reported as
            # `incomplete`, not as a failed line
            main_done = "value" in partial and partial.get("verified") is not False
            msg = f"not finished within {args.train_deadline:.0f} s" + (" (the fields present were measured before the stall)" if partial else "")
            if rank == 0:
                out["train"] = dict(partial, incomplete=msg) if main_done else dict(partial, error=msg)
                print(json.dumps(out), flush=True)
            return 0 if main_done and out.get("verified") is not False else 3

        def run():
            try:
                return train_secondary(torch, dist, ops, dev, world, rank, max(1, min(args.steps, 10)), args.train_batch,
                                       "f32x3" if args.dtype == "f32x3" else "f32", partial=partial,
                                       cpu_base=world == 1 and not args.no_cpu_baseline)
            except Exception as e:      # noqa: BLE001 - reported in the line (rank 0) and on stderr (every rank)
                import traceback
                note(rank, "train secondary failed: " + "".join(traceback.format_exception(type(e), e, e.__traceback__))[-1500:])
                return {"error": f"{type(e).__name__}: {e}"[:300]}

        res = with_deadline(args.train_deadline, emit_without, run)
        if rank == 0:
            out["train"] = res
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()
    if rank == 0 and (out.get("verified") is False or "error" in (out.get("train") or {}) or (out.get("train") or {}).get("verified") is False):
        sys.exit(3)              # the line is printed either way; a failed check or secondary must not look like success


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed counter passes (FETCH_SIZE / WRITE_SIZE cannot be read from
    inside this process: they come from separate `rocprofv3 --pmc` runs of this same command), or None."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            return json.load(f).get(kernel, {}).get("bytes_per_launch")
    except (OSError, ValueError):
        return None


def breakdown(torch, ops, step, B):
    """Per-launch table of one step (after tuning) and the step's roofline summary computed from it - the same
    numbers the JSON line carries as roofline.step / phases / worst."""
    ops.set_autotune(True)
    for _ in range(2):
        step()
    ops.set_autotune(False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    step_ms = (time.perf_counter() - t0) / 5 * 1e3
    rec = []
    ops.set_trace(lambda n, s: rec.append((PHASE[0], n, s, torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))) or rec[-1][3:])
    step()
    torch.cuda.synchronize()
    ops.set_trace(None)
    rows = [(ph, n, s, e0.elapsed_time(e1)) for ph, n, s, e0, e1 in rec]
    print(f"step {step_ms:.3f} ms (untraced mean of 5), traced launches {len(rows)} sum {sum(r[3] for r in rows):.3f} ms, batch {B}")
    lib = __import__("image_captioning_amd._native", fromlist=["load"]).load()
    for ph, n, s, ms in rows:
        fl, by = WORK[n](s)
        bound, roof = launch_roof_ms(n, s)
        cfg = ""
        if n == "conv2d":
            cfg = f" cfg={lib.icap_conv2d_tuned_config(s[0], s[1], s[2], s[3], s[4], s[5], s[5], s[6], s[7], s[8])}"
        elif n == "linear":
            cfg = f" cfg={lib.icap_conv2d_tuned_config(s[0], 1, 1, s[1], s[2], 1, 1, 1, 1, 1)}"
        print(f"{ms:9.4f} ms  {fl / ms / 1e9:8.2f} TFLOP/s {by / ms / 1e6:9.1f} GB/s  {bound:4s} {roof / ms:5.3f}  {ph:7s} {n}{list(s)}{cfg}")
    print("roofline summary: " + json.dumps(step_roofline(rows, step_ms)))


def host_cpu_model():
    """Model string and core counts of the host: /proc/cpuinfo's model name (with family / model / stepping numbers: some
    hosts publish a shortened name), physical cores = distinct (physical id, core id) pairs, logical CPUs, and how many of
    them this process may use."""
    name, fam, mod, step, phys, cur_pkg = "unknown", "?", "?", "?", set(), None
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                k, _, v = line.partition(":")
                k, v = k.strip(), v.strip()
                if k == "model name":
                    name = v
                elif k == "cpu family":
                    fam = v
                elif k == "model":
                    mod = v
                elif k == "stepping":
                    step = v
                elif k == "physical id":
                    cur_pkg = v
                elif k == "core id":
                    phys.add((cur_pkg, v))
    except OSError:
        pass
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count() or 0
    return {"model_name": name, "family_model_stepping": f"{fam}/{mod}/{step}", "physical_cores": len(phys) or None,
            "logical_cpus": os.cpu_count(), "cpus_usable_by_this_process": usable}


def cpu_baseline(torch, model, n):
    """The CPU oracle on the host cores over a bounded sample: n synthetic images -> EfficientNet-B0 features
    (once per image) -> 20 greedy steps per image with the reference's call pattern for the decoder (the model's
    decoder side re-run on the growing prefix, one image at a time).  Run with torch's default thread count (the cores
    this process may use) and, on a quarter of the sample, with 8 threads."""
    from image_captioning_amd.synthetic import synthetic_images
    from oracle import backbones_ref as BR, torch_ref as R
    sd = {k: (v.detach().cpu().float() if v.is_floating_point() else v.detach().cpu()) for k, v in model.state_dict().items()}
    threads = torch.get_num_threads()

    def run(count, seed):
        imgs = synthetic_images(count, seed=seed)
        with torch.no_grad():
            t0 = time.perf_counter()
            feats = torch.cat([BR.cnn_model(sd, imgs[i:i + 16]) for i in range(0, count, 16)])
            for f in feats:
                R.tfm_greedy(sd, f, 8, 8, CAPTION_LEN, 2, 3, False)
            return time.perf_counter() - t0
    dt = run(n, 99)
    res = {"value": n / dt, "unit": "captions/s", "cores": threads, "kind": "port", "host_cpu": host_cpu_model(),
           "sample": f"{n} synthetic 224x224 images: oracle EfficientNet-B0 (batches of 16) + per-image prefix-re-run greedy decode, "
                     f"{CAPTION_LEN} tokens, fp32, torch CPU with {threads} threads, {dt:.1f} s"}
    if threads != 8:
        n8 = max(16, n // 4 // 16 * 16)
        torch.set_num_threads(8)
        try:
            dt8 = run(n8, 98)
        finally:
            torch.set_num_threads(threads)
        res["threads_8"] = {"value": n8 / dt8, "unit": "captions/s", "cores": 8, "sample": f"{n8} images, same pipeline, torch CPU with 8 threads, {dt8:.1f} s"}
    return res


def cpu_train_baseline(torch, model, B):
    """The CPU oracle's training step (train-mode forward with torch autograd backward and torch.optim.AdamW) on the
    host cores: one warm-up and two timed steps at the reference's batch size."""
    from image_captioning_amd.synthetic import synthetic_captions, synthetic_images
    from oracle import backbones_ref as BR, torch_ref as R, train_ref as TR
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    params = {k: v.float().requires_grad_(True) for k, v in sd.items() if v.is_floating_point() and "running" not in k}
    bufs = {k: v.float() for k, v in sd.items() if "running" in k}
    full = dict(sd)
    full.update(params)
    opt = torch.optim.AdamW(list(params.values()), lr=1e-4, weight_decay=1e-5)
    imgs = synthetic_images(B, seed=99)
    caps = synthetic_captions(B, CAPTION_LEN, VOCAB, 2, 3, 0, seed=98)
    threads = torch.get_num_threads()
    enc_bn = {"running_mean": bufs["encoder.fc.3.running_mean"], "running_var": bufs["encoder.fc.3.running_var"]}
    dec_bn = {"running_mean": bufs["decoder.fc.1.running_mean"], "running_var": bufs["decoder.fc.1.running_var"]}

    def one():
        opt.zero_grad()
        feats = BR.cnn_model_train(full, imgs, bufs, None)
        pred = TR.bah_train_forward(full, feats, caps, enc_bn, dec_bn, None)
        R.compute_loss(pred[:, :-1], caps[:, 1:], 0).backward()
        opt.step()
    one()
    nsteps = 0
    t0 = time.perf_counter()
    while nsteps < 40 and (nsteps < 2 or time.perf_counter() - t0 < 10.0):      # a bounded sample: about 10 s of CPU work
        one()
        nsteps += 1
    dt = (time.perf_counter() - t0) / nsteps
    return {"value": B / dt, "unit": "imgs/s", "cores": threads, "kind": "port", "host_cpu": host_cpu_model(),
            "sample": f"{nsteps} training steps of {B} synthetic images x {CAPTION_LEN} tokens (dropout masks omitted), oracle forward + torch "
                      f"autograd + torch.optim.AdamW, fp32, torch CPU with {threads} threads, {dt:.1f} s per step"}


if __name__ == "__main__":
    main()
