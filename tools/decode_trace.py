"""One caption in flight through the three recurrent / transformer decoders, a few captions each: run under
`rocprofv3 --kernel-trace --stats` to read the per-launch durations of the decode kernels."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn as nn
import bench
from image_captioning_amd import seq2seqbasic as s2s, transformer_word_level_bahdnau as bah
from image_captioning_amd.synthetic import seeded_state_dict

dev = torch.device("cuda:0")


class Pass(nn.Module):
    def forward(self, x):
        return x


def seeded(m, seed):
    m.load_state_dict(seeded_state_dict(m, seed=seed))
    return m.eval().to(dev)


which = sys.argv[1] if len(sys.argv) > 1 else "bahdanau"
g = torch.Generator().manual_seed(3)
f512, f256 = torch.randn(1, 512, generator=g).to(dev), torch.randn(1, 256, generator=g).to(dev)
if which == "bahdanau":
    bm = seeded(bah.ImageCaptioningModel(Pass(), bah.TransformerEncoderBlock(512, 512, 8, dropout=0.5),
                                         bah.BahdanauDecoder(512, 512, 512, 256, bench.VOCAB, dropout=0.5)), 31)
    fn = lambda: bm.greedy_ids_from_features(f512, bench.CAPTION_LEN)
elif which == "bahdanau_persistent":
    bm = seeded(bah.ImageCaptioningModel(Pass(), bah.TransformerEncoderBlock(512, 512, 8, dropout=0.5),
                                         bah.BahdanauDecoder(512, 512, 512, 256, bench.VOCAB, dropout=0.5)), 31)
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else bench.CAPTION_LEN
    fn = lambda: bm.greedy_ids_from_features(f512, steps, mode="persistent", check=False)
else:
    m = seeded(s2s.DecoderRNN(256, 512, bench.VOCAB, num_layers=2, rnn_type=which), 11)
    fn = lambda: m.sample_ids(f256, bench.CAPTION_LEN)
for _ in range(10):
    fn()
torch.cuda.synchronize()
