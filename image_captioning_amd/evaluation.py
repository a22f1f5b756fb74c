"""Teacher-forced evaluation of a captioning model on a data split: the reference's ``evaluate_model``
(TransformerTest_bahdnau.py:216-256, TransformerTest.py:189-229) - mean loss and accuracy over batches, sentence BLEU-1..4
over all captions.  ROUGE-L of the reference uses the ``rouge_score`` package with Porter stemming, which is not
available here and is left out."""
from __future__ import annotations

import numpy as np
import torch

from . import ops
from .data import sentence_bleu


def evaluate_model(model, dataloader, device, idx2word, vocab, loss_fn, padding_idx):
    """-> (avg_loss, avg_accuracy, {"BLEU-1".."BLEU-4": mean}).  ``loss_fn(logits [rows, V], targets [rows])`` as in the
    reference's call (``nn.CrossEntropyLoss(ignore_index=...)``: ``training.CrossEntropyLoss`` here)."""
    model.eval()
    total_loss = 0.0
    total_accuracy = 0.0
    n_batches = 0
    bleu = {f"BLEU-{n}": [] for n in range(1, 5)}
    with torch.no_grad():
        for images, captions in dataloader:
            images, captions = images.to(device), captions.to(device)
            outputs = model(images, captions[:, :-1].contiguous())
            targets = captions[:, 1:].contiguous()
            B, T, V = outputs.shape
            flat = outputs.reshape(B * T, V).contiguous()
            total_loss += float(loss_fn(flat, targets.reshape(-1)))
            pred_ids = ops.argmax_rows(flat).view(B, T)
            mask = (targets != padding_idx).float()
            total_accuracy += float(((pred_ids == targets) * mask).sum() / mask.sum())
            n_batches += 1
            pred_l, tgt_l = pred_ids.tolist(), targets.tolist()
            for i in range(B):
                pred_caption = [idx2word[t] for t in pred_l[i]]
                target_caption = [idx2word[t] for t in tgt_l[i] if t != padding_idx]
                for n in range(1, 5):
                    bleu[f"BLEU-{n}"].append(sentence_bleu([target_caption], pred_caption, weights=[1 / n] * n))
    if n_batches == 0:
        raise ValueError("evaluate_model: the data loader produced no batch")
    return total_loss / n_batches, total_accuracy / n_batches, {k: float(np.mean(v)) for k, v in bleu.items()}
