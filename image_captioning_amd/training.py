"""Training-step pieces: the reference's two masked cross-entropy losses, AdamW and the
data-parallel gradient exchange.

* ``compute_loss(predictions, targets, padding_idx)`` - data.py:244-252 / datachar.py:234-242.
* ``CrossEntropyLoss(ignore_index=pad)`` - the criterion of seq2seqbasic.py:204,219.
* ``AdamW`` - torch.optim.AdamW as configured at transformer_word_level_bahdnau.py:191,
  transformer_word_level.py:138, transformercharmiguel.py:140 and seq2seqbasic.py:205-210:
  every parameter of the model lives in ONE flat fp32 buffer (parameter ``.data`` and
  ``.grad`` are views into flat buffers), so the update is a single kernel launch and the
  gradient exchange works on contiguous slices.
* ``GradientReducer`` - data-parallel training across one process per GPU: bucketed
  all-reduce (RCCL through ``torch.distributed``; gloo on CPU tensors in tests) of the flat
  gradient buffer, launched per bucket as soon as backward has produced every gradient in it.

The loss and the optimizer arithmetic run in ``libicap_hip.so``; torch supplies memory,
the autograd graph and the process group.
"""
from __future__ import annotations

import torch

from . import _native as N
from . import ops
from .ops import _ck, _stream


# ---------------------------------------------------------------------------- losses
def _row_view(pred: torch.Tensor):
    """(base tensor pointer info) of a [..., V] logits tensor whose rows are either contiguous or the
    first rows of equally sized groups (a [B, T-1, V] slice of a contiguous [B, T, V] tensor)."""
    if pred.dim() < 2 or pred.stride(-1) != 1:
        raise N.IcapError("loss: logits need a contiguous last dimension")
    V = pred.size(-1)
    if pred.dim() == 2:
        if pred.stride(0) < V:
            raise N.IcapError("loss: overlapping logits rows")
        return pred.size(0), V, pred.stride(0), pred.size(0), pred.size(0)
    if pred.dim() == 3:
        B, T1, _ = pred.shape
        ld = pred.stride(1) if T1 > 1 else V
        if ld < V or (B > 1 and pred.stride(0) % ld):
            raise N.IcapError("loss: unsupported logits strides")
        gs = pred.stride(0) // ld if B > 1 else T1
        if gs < T1:
            raise N.IcapError("loss: unsupported logits strides")
        return B * T1, V, ld, T1, gs
    raise N.IcapError("loss: logits must be [R, V] or [B, T, V]")


class _MaskedCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, targets, ignore_index, mean_over_valid):
        if not pred.is_cuda or pred.dtype != torch.float32:
            raise N.IcapError("loss: logits must be an fp32 tensor on the HIP device (no CPU path)")
        R, V, ld, gr, gs = _row_view(pred)
        tg = targets.reshape(-1)
        if tg.numel() != R or tg.dtype != torch.int64 or not tg.is_cuda:
            raise N.IcapError("loss: targets must be int64 on the device, one per logits row")
        tg = tg.contiguous()
        dev = pred.device
        row_loss = torch.empty(R, device=dev, dtype=torch.float32)
        row_lse = torch.empty(R, device=dev, dtype=torch.float32)
        out = torch.empty(2, device=dev, dtype=torch.float32)
        N.check(N.load().icap_ce_loss_fwd_f32(pred.data_ptr(), ld, gr, gs, tg.data_ptr(), R, V, int(ignore_index),
                                              int(mean_over_valid), row_loss.data_ptr(), row_lse.data_ptr(),
                                              out.data_ptr(), _stream()), "ce_loss_fwd")
        ctx.save_for_backward(pred, tg, row_lse, out)
        ctx.geom = (R, V, ld, gr, gs, int(ignore_index))
        return out[0]

    @staticmethod
    def backward(ctx, grad_out):
        pred, tg, row_lse, out = ctx.saved_tensors
        R, V, ld, gr, gs, ignore = ctx.geom
        # the gradient has the shape of `pred`; laid out densely, rows addressed as groups of gr
        dl = torch.empty(pred.shape, device=pred.device, dtype=torch.float32)
        g = grad_out.reshape(1).contiguous().float()
        N.check(N.load().icap_ce_loss_bwd_f32(pred.data_ptr(), ld, gr, gs, tg.data_ptr(), R, V, ignore, row_lse.data_ptr(),
                                              out.data_ptr(), g.data_ptr(), dl.data_ptr(), V, _stream()), "ce_loss_bwd")
        return dl, None, None, None


def compute_loss(predictions: torch.Tensor, targets: torch.Tensor, padding_idx: int) -> torch.Tensor:
    """data.py:244-252: CE per position, pad positions zeroed, mean over ALL positions (pads in the denominator)."""
    return _MaskedCE.apply(predictions, targets, padding_idx, False)


class CrossEntropyLoss(torch.nn.Module):
    """nn.CrossEntropyLoss(ignore_index=...) as used at seq2seqbasic.py:204: mean over non-ignored rows."""

    def __init__(self, ignore_index: int = -100):
        super().__init__()
        self.ignore_index = ignore_index

    def forward(self, outputs: torch.Tensor, targets: torch.Tensor) -> torch.Tensor:
        return _MaskedCE.apply(outputs, targets, self.ignore_index, True)


# ---------------------------------------------------------------------------- optimizer
class FlatParameters:
    """Moves the trainable parameters of a model into ONE flat fp32 tensor.

    ``param.data`` becomes a view into ``flat_p`` (values kept) and ``param.grad`` a view into ``flat_g``, so
    backward accumulates straight into the flat gradient, clearing gradients is one memset, the optimizer
    update is one kernel launch and the data-parallel exchange works on contiguous slices.  Pure bookkeeping:
    works on any device.  ``touched[i]`` records whether backward produced a gradient for parameter i since the
    last ``zero_grad`` (torch's optimizers skip parameters whose ``.grad`` is None; the reference's models carry
    parameters that never get one, e.g. the unused self-attention of the word-level decoder).
    """

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        dev = self.params[0].device
        self.offsets, n = [], 0
        for p in self.params:
            if p.dtype != torch.float32 or p.device != dev:
                raise ValueError("fp32 parameters on one device required")
            self.offsets.append(n)
            n += (p.numel() + 3) // 4 * 4          # every parameter starts 16-byte aligned
        self.n = n
        self.flat_p = torch.zeros(n, device=dev)
        self.flat_g = torch.zeros(n, device=dev)
        self.touched = [False] * len(self.params)
        with torch.no_grad():
            for i, (p, o) in enumerate(zip(self.params, self.offsets)):
                self.flat_p[o:o + p.numel()].copy_(p.data.reshape(-1))
                p.data = self.flat_p[o:o + p.numel()].view(p.shape)
                p.grad = self.flat_g[o:o + p.numel()].view(p.shape)
                p.register_post_accumulate_grad_hook(self._mark(i))
        self.grad_scale = 1.0

    def _mark(self, i):
        def hook(_p):
            self.touched[i] = True
        return hook

    def zero_grad(self, set_to_none: bool = True) -> None:
        self.flat_g.zero_()
        self.touched = [False] * len(self.params)
        for p, o in zip(self.params, self.offsets):   # re-attach views a caller may have dropped
            if p.grad is None or p.grad.data_ptr() != self.flat_g.data_ptr() + 4 * o:
                p.grad = self.flat_g[o:o + p.numel()].view(p.shape)

    def touched_runs(self, keys=None):
        """Maximal [lo, hi) slices of the flat buffer covering consecutive parameters that received a gradient
        (and, with `keys`, share the same key); returns (lo, hi) or (lo, hi, key) tuples."""
        runs, lo, hi, key = [], None, None, None
        for i, (p, o) in enumerate(zip(self.params, self.offsets)):
            end = o + (p.numel() + 3) // 4 * 4
            k = keys[i] if keys is not None else None
            if self.touched[i] and lo is not None and k == key:
                hi = end
                continue
            if lo is not None:
                runs.append((lo, hi) if keys is None else (lo, hi, key))
                lo = None
            if self.touched[i]:
                lo, hi, key = o, end, k
        if lo is not None:
            runs.append((lo, hi) if keys is None else (lo, hi, key))
        return runs


class AdamW:
    """torch.optim.AdamW(params, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2) over ``FlatParameters``.

    ``step`` launches ``icap_adamw_f32`` once per run of parameters that received a gradient (normally one or
    two launches); parameters without a gradient are left untouched and their own step count (which sets the
    bias correction) does not advance, as in torch.
    """

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        self.flat = params if isinstance(params, FlatParameters) else FlatParameters(params)
        if self.flat.flat_p.device.type != "cuda":
            raise N.IcapError("AdamW: parameters must live on the HIP device (no CPU path)")
        self.lr, self.betas, self.eps, self.weight_decay = float(lr), (float(betas[0]), float(betas[1])), float(eps), float(weight_decay)
        self.m = torch.zeros_like(self.flat.flat_p)
        self.v = torch.zeros_like(self.flat.flat_p)
        self.steps = 0
        self.param_steps = [0] * len(self.flat.params)

    @property
    def params(self):
        return self.flat.params

    def zero_grad(self, set_to_none: bool = True) -> None:
        self.flat.zero_grad(set_to_none)

    def step(self) -> None:
        self.steps += 1
        ops.bump_weights_generation()       # the kernel writes the parameters by raw pointer: eval-mode weight caches must rebuild
        f = self.flat
        lib = N.load()
        self.param_steps = [s + 1 if t else s for s, t in zip(self.param_steps, f.touched)]
        for lo, hi, step in f.touched_runs(self.param_steps):
            N.check(lib.icap_adamw_f32(f.flat_p.data_ptr() + 4 * lo, f.flat_g.data_ptr() + 4 * lo, self.m.data_ptr() + 4 * lo,
                                       self.v.data_ptr() + 4 * lo, hi - lo, self.lr, self.betas[0], self.betas[1], self.eps,
                                       self.weight_decay, step, float(f.grad_scale), _stream()), "adamw")

    def record_step(self, step_dev: torch.Tensor) -> None:
        """The launches of ``step`` with the step count read from `step_dev` (int32 [1] on the device, holding the count
        of the step being taken) instead of passed by value - what a step recorded in a HIP graph needs.  Changes no
        host state; ``advance_recorded`` does that once per replay."""
        f = self.flat
        lib = N.load()
        nxt = [s + 1 if t else s for s, t in zip(self.param_steps, f.touched)]
        for lo, hi, step in f.touched_runs(nxt):
            N.check(lib.icap_adamw_dev_f32(f.flat_p.data_ptr() + 4 * lo, f.flat_g.data_ptr() + 4 * lo, self.m.data_ptr() + 4 * lo,
                                           self.v.data_ptr() + 4 * lo, hi - lo, self.lr, self.betas[0], self.betas[1], self.eps,
                                           self.weight_decay, step_dev.data_ptr(), step - (self.steps + 1), float(f.grad_scale),
                                           _stream()), "adamw_dev")

    def advance_recorded(self, touched) -> None:
        self.steps += 1
        ops.bump_weights_generation()
        self.param_steps = [s + 1 if t else s for s, t in zip(self.param_steps, touched)]

    def state_dict(self):
        return {"step": self.steps, "param_steps": list(self.param_steps), "exp_avg": self.m, "exp_avg_sq": self.v, "lr": self.lr,
                "betas": self.betas, "eps": self.eps, "weight_decay": self.weight_decay}

    def load_state_dict(self, state) -> None:
        """Resume: moments, the global step and the per-parameter steps (a parameter that never received a gradient has
        its own count, which sets its bias correction)."""
        if state["exp_avg"].numel() != self.m.numel() or len(state["param_steps"]) != len(self.param_steps):
            raise ValueError("AdamW.load_state_dict: state belongs to a different parameter set")
        self.m.copy_(state["exp_avg"])
        self.v.copy_(state["exp_avg_sq"])
        self.steps = int(state["step"])
        self.param_steps = [int(s) for s in state["param_steps"]]
        self.lr, self.betas, self.eps, self.weight_decay = float(state["lr"]), tuple(state["betas"]), float(state["eps"]), float(state["weight_decay"])


class Adam(AdamW):
    """torch.optim.Adam(params, lr) as selected by ``optimizer_type="adam"`` at seq2seqbasic.py:205-210: with its
    default weight_decay = 0 the update is AdamW's without the decay term."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=0.0)


class SGD:
    """torch.optim.SGD(params, lr) without momentum (``optimizer_type="sgd"``, seq2seqbasic.py:208): p -= lr * g over
    the flat buffer, parameters without a gradient untouched."""

    def __init__(self, params, lr=1e-3):
        self.flat = params if isinstance(params, FlatParameters) else FlatParameters(params)
        if self.flat.flat_p.device.type != "cuda":
            raise N.IcapError("SGD: parameters must live on the HIP device (no CPU path)")
        self.lr = float(lr)

    @property
    def params(self):
        return self.flat.params

    def zero_grad(self, set_to_none: bool = True) -> None:
        self.flat.zero_grad(set_to_none)

    def step(self) -> None:
        f = self.flat
        lib = N.load()
        ops.bump_weights_generation()
        for lo, hi in f.touched_runs():
            N.check(lib.icap_axpby_f32(f.flat_g.data_ptr() + 4 * lo, -self.lr * float(f.grad_scale), 1.0, f.flat_p.data_ptr() + 4 * lo,
                                       hi - lo, _stream()), "sgd")


# ---------------------------------------------------------------------------- data-parallel gradient exchange
class GradientReducer:
    """Sums the flat gradient buffer of ``FlatParameters`` (or of an ``AdamW``) across the ranks of a process
    group in buckets.

    One process per GPU; every rank computes the reference's loss on its own local batch, the gradients are
    summed and divided by the world size (the optimizer applies the 1/world factor through ``grad_scale``), i.e.
    the mean of the per-rank mean losses.  Buckets are contiguous slices of the flat gradient, filled in reverse
    parameter order by backward.

    ``overlap=True``: post-accumulate hooks launch the all-reduce of a bucket as soon as its last
    gradient has been accumulated, so the exchange of the late layers' gradients runs while backward is still working
    on the early layers (collectives are launched from autograd's worker thread, in the same order on every rank because
    every rank runs the same graph); ``finish`` launches whatever the hooks did not (buckets holding parameters without
    a gradient this step) and makes the launch stream wait for all of them.  ``break_at``: parameter indices at which a
    bucket must start whatever its size (the staged-graph trainer puts one where the CNN's parameters end, so that no bucket
    straddles the two backward graphs).  ``overlap=False`` reduces every bucket in
    ``finish``, after backward, from the thread that called ``step``.  ``overlap=None`` (the default) picks by backend:
    True for gloo, whose hook path is exercised by the two-rank tests, and False for RCCL ("nccl") unless the
    environment sets ICAP_DP_OVERLAP=1 - the hook path HAS NOT YET RUN OVER RCCL on real multi-GPU hardware (collectives
    launched from autograd's worker thread, ordering of the process group's stream against the launch stream through
    ``wait``), so it stays opt-in there; ``overlap`` may also be switched between steps (the hooks are always
    registered and look at the flag).

    Backends.  RCCL ("nccl"): ``all_reduce(async_op=True)`` runs on the process group's own stream, ordered after the
    stream current at the call, and ``wait`` orders the current stream after it - no host blocking, no extra stream
    here.  gloo with device tensors (rehearsals of the multi-rank control flow on fewer GPUs than ranks) is staged
    through pinned host memory by this class: gloo's own handling of device tensors stalls when the launch stream still
    has work queued at the call (tools/gloo_hip_allreduce_repro.py reproduces that without this package), so the bucket
    is copied to the host on a side stream at launch time, and ``finish`` reduces the host copy and copies it back.
    CPU tensors (the gloo unit tests) go to gloo directly.
    """

    def __init__(self, flat, group=None, bucket_mb: float = 16.0, overlap=None, break_at=()):
        import os
        import torch.distributed as dist
        self.dist = dist
        optimizer = flat.flat if isinstance(flat, (AdamW, SGD)) else flat
        self.opt = optimizer
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        optimizer.grad_scale = 1.0 / self.world
        per = max(4, int(bucket_mb * (1 << 20) / 4) // 4 * 4)
        # buckets in reverse parameter order: the last parameters get their gradients first
        self.buckets = []          # (lo, hi) float offsets
        self.bucket_of = {}
        hi = optimizer.n
        cur_lo = hi
        idxs = []
        for i in reversed(range(len(optimizer.params))):
            cur_lo = optimizer.offsets[i]
            idxs.append(i)
            if hi - cur_lo >= per or i == 0 or i in break_at:       # break_at: parameter indices a bucket must start at
                b = len(self.buckets)
                self.buckets.append((cur_lo, hi))
                for j in idxs:
                    self.bucket_of[j] = b
                idxs, hi = [], cur_lo
        self.pending = [0] * len(self.buckets)
        self.fired = [False] * len(optimizer.params)
        self.counts = [0] * len(self.buckets)
        for j, b in self.bucket_of.items():
            self.counts[b] += 1
        self.handles = []
        on_device = optimizer.flat_g.is_cuda
        backend = dist.get_backend(group) if dist.is_initialized() else ""
        if overlap is None:
            overlap = backend != "nccl" or os.environ.get("ICAP_DP_OVERLAP") == "1"
        self.overlap = bool(overlap) and self.world > 1
        self.host_staged = on_device and self.world > 1 and backend == "gloo"
        self.comm_stream = torch.cuda.Stream() if self.host_staged else None
        self.host = [torch.empty(hi - lo, dtype=torch.float32).pin_memory() for lo, hi in self.buckets] if self.host_staged else None
        self.staged = []
        self.launched = [False] * len(self.buckets)
        self.launch_order = []     # bucket indices in the order their exchange was launched this step (tests, diagnostics)
        self.hooked = 0            # how many of them were launched by hooks, i.e. while backward was still running
        if self.world > 1:
            for i, p in enumerate(optimizer.params):
                p.register_post_accumulate_grad_hook(self._make_hook(i))

    def _make_hook(self, i):
        def hook(_param):
            if not self.overlap:
                return
            b = self.bucket_of[i]
            # once per parameter and backward pass: a Function that adds its parameter gradients itself
            # (train_ops.accumulate_into_grads) runs these hooks when the sums are in place, and autograd runs them a second
            # time for the None it is handed - counted twice, a bucket shared with later parameters would leave early
            if self.fired[i]:
                return
            self.fired[i] = True
            self.pending[b] += 1
            if self.pending[b] == self.counts[b]:
                self.hooked += 1
                self._launch(b)
        return hook

    def _launch(self, b):
        lo, hi = self.buckets[b]
        t = self.opt.flat_g[lo:hi]
        self.launched[b] = True
        self.launch_order.append(b)
        if self.host_staged:
            self.comm_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.comm_stream):
                self.host[b].copy_(t, non_blocking=True)
            self.staged.append(b)
        else:
            self.handles.append(self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group, async_op=True))

    def launch_buckets(self, buckets):
        """Start the exchange of these buckets now, from the calling thread (the staged-graph step calls this between two
        graph replays with the buckets the first graph has completed).  With the host-staged stand-in the host all-reduce
        of what has been launched so far can be run right away by ``reduce_staged``; RCCL needs nothing more."""
        for b in buckets:
            if not self.launched[b]:
                self._launch(b)

    def reduce_staged(self):
        """Host-staged backend only: reduce the buckets whose device-to-host copies have been launched so far and start
        their copies back, all beside whatever the launch stream is running; ``finish`` orders the launch stream after
        them.  (The caller has queued later work on the launch stream already, so the host blocks here on the COPIES'
        event, not on that work.)"""
        if not self.host_staged or not self.staged:
            return
        ev = torch.cuda.Event()
        ev.record(self.comm_stream)
        ev.synchronize()
        with torch.cuda.stream(self.comm_stream):
            for b in self.staged:                                # same order on every rank
                self.dist.all_reduce(self.host[b], op=self.dist.ReduceOp.SUM, group=self.group)
                lo, hi = self.buckets[b]
                self.opt.flat_g[lo:hi].copy_(self.host[b], non_blocking=True)
        self.staged = []
        self.early_reduced = True

    def buckets_within(self, param_indices):
        """Buckets all of whose parameters are in ``param_indices`` (indices into the optimizer's parameter list), in
        launch order (the order backward fills them)."""
        inside = set(param_indices)
        whole = [True] * len(self.buckets)
        for j, b in self.bucket_of.items():
            if j not in inside:
                whole[b] = False
        return [b for b in range(len(self.buckets)) if whole[b]]

    def prepare(self):
        """Call before backward of every step."""
        self.early_reduced = False
        self.pending = [0] * len(self.buckets)
        self.fired = [False] * len(self.fired)
        self.launched = [False] * len(self.buckets)
        self.handles = []
        self.staged = []
        self.launch_order = []
        self.hooked = 0

    def finish(self):
        """Call after backward: reduces whatever the hooks did not (parameters without gradient this step,
        or overlap disabled) and orders the launch stream after all buckets."""
        if self.world == 1:
            return
        for b in range(len(self.buckets)):
            if not self.launched[b]:
                self._launch(b)
        for h in self.handles:
            h.wait()
        if self.host_staged:
            self.comm_stream.synchronize()                       # the device-to-host copies have landed
            for b in self.staged:                                # same order on every rank
                self.dist.all_reduce(self.host[b], op=self.dist.ReduceOp.SUM, group=self.group)
                lo, hi = self.buckets[b]
                self.opt.flat_g[lo:hi].copy_(self.host[b], non_blocking=True)
            self.staged = []
            if getattr(self, "early_reduced", False):            # copies back started by reduce_staged run on the side stream
                torch.cuda.current_stream().wait_stream(self.comm_stream)
        self.handles = []


# ---------------------------------------------------------------------------- one training step
class Trainer:
    """The loop body of transformer_word_level_bahdnau.py:218-233 (and of the word / char transformer and seq2seq
    loops, which differ only in the loss call): zero_grad, forward in train mode, loss, backward, optimizer step -
    with the gradient exchange of data-parallel training between backward and the step when a process group is up.

        trainer = Trainer(model, lr=1e-4, weight_decay=1e-5)          # AdamW as at line 191
        loss = trainer.step(images, captions, padding_idx)

    ``step`` enqueues everything on the current stream and returns the loss as a device scalar (no host sync).
    """

    def __init__(self, model, lr=1e-4, weight_decay=1e-5, betas=(0.9, 0.999), eps=1e-8, group=None, bucket_mb=16.0,
                 loss="compute_loss", overlap=None, dtype="f32", graph=False):
        import torch.distributed as dist
        if dtype not in ("f32", "f32x3", "bf16"):
            raise ValueError(f"Trainer: dtype must be 'f32', 'f32x3' or 'bf16', not {dtype!r}")
        self.model = model
        self.graph = bool(graph)
        self._graph, self._graph_opt, self._eager_steps = None, None, 0
        self.dtype = dtype                   # "bf16": bf16 products in forward and input gradients, fp32 master weights
                                             # / accumulation / weight gradients / optimizer (ops.train_step)
        self.opt = AdamW(model.parameters(), lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        # graph=True: hooks do not run in a replay, so the reducer's hook path stays off; "overlap" then means the STAGED
        # step - the backward recorded as two graphs split where the gradient reaches the CNN's output, the finished buckets'
        # exchange launched between the two replays (see _graph_step).  Default on; overlap=False keeps the exchange after
        # the whole backward.
        self._want_staged = self.graph and overlap is not False
        if self.graph:
            overlap = False
        break_at = ()
        if self._want_staged and hasattr(model, "cnn_model"):
            cnn = {id(p) for p in model.cnn_model.parameters()}
            idx = [i for i, p in enumerate(self.opt.params) if id(p) not in cnn]
            break_at = (idx[0],) if idx and idx[0] > 0 else ()
        self.reducer = GradientReducer(self.opt, group=group, bucket_mb=bucket_mb, overlap=overlap, break_at=break_at) if dist.is_available() and dist.is_initialized() else None
        self.loss_kind = loss
        self.staged = (self._want_staged and self.reducer is not None and self.reducer.world > 1
                       and hasattr(model, "cnn_model") and hasattr(model, "forward_features")
                       and any(p.requires_grad for p in model.cnn_model.parameters()))
        self._graph_cnn, self._early_buckets = None, []
        self.timeline = None                 # a list: the staged step appends its events to it (tools/train_rehearse.py)

    def _forward_backward(self, images, captions, padding_idx, exchange=True):
        exchange = exchange and self.reducer is not None
        self.model.train()
        self.opt.zero_grad()
        if exchange:
            self.reducer.prepare()
        with ops.train_step(self.dtype):      # per-step weight copies live until backward is done
            predictions = self.model(images, captions)
            loss = self._loss_of(predictions, captions, padding_idx)
            loss.backward()
        if exchange:
            self.reducer.finish()
        return loss.detach()

    def _loss_of(self, predictions, captions, padding_idx):
        if self.loss_kind == "compute_loss":
            return compute_loss(predictions[:, :-1], captions[:, 1:], padding_idx)
        return CrossEntropyLoss(ignore_index=padding_idx)(predictions.reshape(-1, predictions.size(-1)), captions.reshape(-1))

    def step(self, images, captions, padding_idx):
        if self.graph:
            return self._graph_step(images, captions, padding_idx)
        loss = self._forward_backward(images, captions, padding_idx)
        self.opt.step()
        return loss

    # ---- the step as one HIP graph (graph=True): for small per-GPU batches, where ~2000 launches per step make the
    # host the bottleneck.  Two eager steps first (allocator pools, the set of parameters that receive gradients), then
    # the third call records zero_grad + forward + loss + backward + AdamW into a graph on static input buffers; from
    # then on a step is two copies into those buffers and one graph launch.  What changes from step to step lives in
    # device memory and is advanced inside the graph: the dropout generator's offset base and AdamW's step count.
    # Under data parallelism no collective is ever recorded: the step is THREE graphs - forward + backward down to the CNN's
    # output | the CNN's backward | the optimizer - and the exchange is launched between the replays in the ordinary way:
    # the buckets the first graph completes (decoder, encoder block: three quarters of the gradient bytes of the Bahdanau
    # captioner) travel while the second graph runs, the rest after it.  overlap=False, or a model without a trainable
    # `cnn_model` / `forward_features`, keeps two graphs with the whole exchange between them.
    def _graph_step(self, images, captions, padding_idx):
        if self._graph is None:
            if self._eager_steps < 2:
                self._eager_steps += 1
                loss = self._forward_backward(images, captions, padding_idx)
                self.opt.step()
                return loss
            self._record(images, captions, padding_idx)
        if images.shape != self._images.shape or captions.shape != self._caps.shape or padding_idx != self._pad:
            raise ValueError("Trainer(graph=True): every step must have the shapes (and padding index) of the recorded one")
        from ._bah_train import DropoutState
        self._images.copy_(images)
        self._caps.copy_(captions)
        self._graph.replay()
        if self._graph_opt is not None:
            self.reducer.prepare()
            if self._graph_cnn is not None:
                # staged: the decoder / encoder-block buckets are complete - their exchange runs under the CNN's backward
                tl = self.timeline
                if tl is not None:
                    e_first = torch.cuda.Event(enable_timing=True)
                    e_first.record()
                self.reducer.launch_buckets(self._early_buckets)
                if tl is not None:
                    e_sent = torch.cuda.Event(enable_timing=True)
                    e_sent.record(self.reducer.comm_stream if self.reducer.comm_stream is not None else torch.cuda.current_stream())
                self._graph_cnn.replay()
                if tl is not None:
                    e_cnn = torch.cuda.Event(enable_timing=True)
                    e_cnn.record()
                self.reducer.reduce_staged()
                if tl is not None:       # (events, was the CNN's backward still running when the early buckets' host reduction returned)
                    tl.append((e_first, e_sent, e_cnn, not e_cnn.query()))
            self.reducer.finish()
            self._graph_opt.replay()
        DropoutState.offset += self._draws_per_step
        self.opt.advance_recorded(self._touched)
        return self._loss.clone()

    def _record(self, images, captions, padding_idx):
        from . import train_ops as T
        from ._bah_train import DropoutState
        dev = images.device
        self._images, self._caps, self._pad = images.clone(), captions.clone(), padding_idx
        self._offset_base = torch.zeros(1, device=dev, dtype=torch.int64)
        self._step_dev = torch.full((1,), self.opt.steps, device=dev, dtype=torch.int32)
        o0 = DropoutState.offset
        tune, ops.AUTOTUNE = ops.AUTOTUNE, False          # tuning synchronises; a recording must not
        T.DROPOUT_BASE = self._offset_base
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        graph_opt = torch.cuda.CUDAGraph() if self.reducer is not None else None

        def record_optimizer():
            self._step_dev.add_(1)
            self.opt.record_step(self._step_dev)
            self._draws_per_step = DropoutState.offset - o0
            self._offset_base.add_(self._draws_per_step)

        graph_cnn = torch.cuda.CUDAGraph() if self.staged else None
        try:
            if graph_cnn is None:
                with torch.cuda.graph(graph):
                    self._loss = self._forward_backward(self._images, self._caps, padding_idx, exchange=False)
                    if graph_opt is None:
                        record_optimizer()
            else:
                # Three graphs.  1: zero_grad, forward, loss, backward down to the CNN's output (the gradient stops at a
                # detached leaf standing for that output); 2: the CNN's backward from the leaf's gradient, recorded into the
                # first graph's memory pool - it reads the activations the first graph saved; 3: the optimizer.  The
                # per-step weight copies (ops.train_step) stay alive across both backward recordings.
                self.model.train()
                with ops.train_step(self.dtype):
                    with torch.cuda.graph(graph):
                        self.opt.zero_grad()
                        feats = self.model.cnn_model(self._images)
                        leaf = feats.detach().requires_grad_()
                        predictions = self.model.forward_features(leaf, self._caps)
                        loss = self._loss_of(predictions, self._caps, padding_idx)
                        loss.backward()
                        self._loss = loss.detach()
                    with torch.cuda.graph(graph_cnn, pool=graph.pool()):
                        feats.backward(leaf.grad)
                    del feats, leaf, predictions, loss
            if graph_opt is not None:
                with torch.cuda.graph(graph_opt):
                    record_optimizer()
        finally:
            T.DROPOUT_BASE = None
            ops.AUTOTUNE = tune
        DropoutState.offset = o0                          # recording ran nothing: the first replay makes these draws
        self._touched = list(self.opt.flat.touched)
        self._graph, self._graph_opt, self._graph_cnn = graph, graph_opt, graph_cnn
        if graph_cnn is not None:
            cnn = {id(p) for p in self.model.cnn_model.parameters()}
            self._early_buckets = self.reducer.buckets_within([i for i, p in enumerate(self.opt.params) if id(p) not in cnn])
