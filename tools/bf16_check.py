"""bf16 compute mode vs fp32 mode of the captioning path: feature / logit differences and token agreement."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image_captioning_amd import ops, transformer_word_level as twl
from image_captioning_amd.synthetic import seeded_state_dict, synthetic_images
d = torch.device("cuda:0")
model = twl.build_model(6756)
model.load_state_dict(seeded_state_dict(model, seed=3))
model = model.to(d).eval()
imgs = synthetic_images(64).to(d)
out = {}
for mode in ("f32", "bf16"):
    ops.set_compute_dtype(mode)
    feats = model.cnn_model(imgs)
    ids, logits = model.greedy_ids_from_features(feats, 20, want_logits=True)
    out[mode] = (feats, ids, logits)
f0, i0, l0 = out["f32"]; f1, i1, l1 = out["bf16"]
print("feature max abs diff", float((f0 - f1).abs().max()), "max |f32|", float(f0.abs().max()), "rel", float((f0 - f1).abs().max() / f0.abs().max()))
print("logits step 0 max abs diff", float((l0[0] - l1[0]).abs().max()), "max |f32|", float(l0[0].abs().max()))
top2 = torch.topk(l0[0], 2, dim=-1).values
print("step-0 top-2 gap: min", float((top2[:, 0] - top2[:, 1]).min()), "median", float((top2[:, 0] - top2[:, 1]).median()))
print("token agreement", float((i0 == i1).float().mean()), "first-step agreement", float((i0[:, 0] == i1[:, 0]).float().mean()))
