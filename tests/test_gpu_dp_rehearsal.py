"""GPU: the data-parallel training step with TWO ranks on the one GPU of the test box (gloo stands in for RCCL, which
refuses two ranks on one device; device buckets are staged through pinned host memory).  Run as a child job through
torch.distributed.run, exactly as the multi-GPU benchmark is launched.  tools/train_rehearse.py holds the checks:
same batch on both ranks => parameters after 4 steps equal the single-process trainer's bit for bit; different batches =>
parameters identical across ranks; hooks launch most buckets during backward (overlap is the default)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _rehearse(port, **env):
    e = dict(os.environ, **{k: str(v) for k, v in env.items()})
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "tools", "train_rehearse.py")],
                       env=e, capture_output=True, text=True, timeout=170)
    tail = "\n".join(l for l in r.stderr.splitlines() if l.startswith("[r"))
    assert r.returncode == 0 and tail.count("REHEARSAL OK") == 2, tail or r.stderr[-2000:]
    return tail


def test_two_ranks_equal_the_single_process_trainer():
    out = _rehearse(29561, REH_SAME=1, REH_OVERLAP=1)
    assert "max diff 0.000e+00" in out


def test_two_ranks_on_different_batches_stay_in_step():
    _rehearse(29562, REH_SAME=0, REH_OVERLAP=1)


def test_exchange_after_backward_and_graph_replayed_step():
    _rehearse(29563, REH_SAME=1, REH_OVERLAP=0)
    _rehearse(29564, REH_SAME=1, REH_GRAPH=1, REH_OVERLAP=0, REH_STEPS=5)         # two graphs, the whole exchange between them


def test_staged_graph_step_exchanges_the_decoder_buckets_under_the_cnn_backward():
    """graph=True under data parallelism: three graphs (forward + backward to the CNN's output | CNN backward | optimizer);
    the buckets the first completes are launched between the two backward replays.  Parameters after 5 steps (2 eager, the
    recording, 2 replays) equal the single-process one-graph trainer's bit for bit; different batches stay in step."""
    out = _rehearse(29565, REH_SAME=1, REH_GRAPH=1, REH_STEPS=5)
    assert "graph step staged: True" in out and "max diff 0.000e+00" in out
    _rehearse(29566, REH_SAME=0, REH_GRAPH=1, REH_STEPS=6)
