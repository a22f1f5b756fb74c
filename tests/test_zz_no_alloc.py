"""GPU: icap_greedy_rnn_static_f32 - the greedy seq2seq decode whose workspace is a fixed array inside the library, for
callers that cannot allocate - against icap_greedy_rnn_f32 (caller workspace), through the C ABI: identical token ids
and logits on two calls in a row, the second smaller than the first (the shared array still holds the first call's
state), for the LSTM and the GRU decoder."""
import ctypes as C

import pytest
import torch

from tests import helpers as H

pytestmark = pytest.mark.gpu
D = "cuda:0"
V = 6756


@pytest.mark.parametrize("kind", ["lstm", "gru"])
def test_static_workspace_entry_matches_the_caller_workspace_one(kind):
    from image_captioning_amd import ops
    m = H.rnn_decoder(kind, [256, 512, V, 2]).to(D)
    wd = m._weights()
    for NB, steps, seed in ((8, 20, 1), (3, 7, 2), (8, 20, 1)):      # big, smaller, big again
        x0 = ops.linear(H.features(NB, 256, seed).to(D), wd["ff_w"], bias=wd["ff_b"])
        want_ids, want_logits = ops.greedy_rnn(wd, x0, steps, want_logits=True)
        ids = torch.full((NB, steps), -7, device=D, dtype=torch.int64)
        logits = torch.full((steps, NB, V), float("nan"), device=D)
        torch.cuda.synchronize()
        allocs = torch.cuda.memory_stats()["allocation.all.allocated"]       # count of allocation events so far
        got_ids, got_logits = ops.greedy_rnn(wd, x0, steps, ids=ids, logits=logits, static_ws=True)
        assert torch.cuda.memory_stats()["allocation.all.allocated"] == allocs   # the binding allocated nothing either
        assert got_ids.data_ptr() == ids.data_ptr()
        assert torch.equal(got_ids, want_ids), f"{kind} NB={NB} steps={steps}"
        assert torch.equal(got_logits, want_logits)
        ids2, none = ops.greedy_rnn(wd, x0, steps, static_ws=True)  # without logits
        assert none is None and torch.equal(ids2, want_ids)


def test_static_workspace_entry_rejects_sizes_beyond_its_buffer():
    from image_captioning_amd import _native as N
    lib = N.load()
    cap = lib.icap_greedy_rnn_static_ws_floats()
    assert cap == 1 << 18
    w = N.RnnWeights()
    w.layers, w.H, w.I, w.V, w.is_lstm = 4, 8192, 512, V, 1
    assert lib.icap_greedy_rnn_ws_floats(8, 4, 8192, 512, V) > cap
    rc = lib.icap_greedy_rnn_static_f32(C.byref(w), None, None, None, 8, 20, None)     # refused before any pointer is used
    assert rc == 3 and b"built-in buffer" in lib.icap_last_error()
