"""Tensor-level wrappers over the training-step entry points of the C ABI (backward GEMMs, column
reductions, BatchNorm / LayerNorm in training mode, dropout, cell / attention / embedding gradients).

Same contract as ``ops``: fp32 contiguous tensors on the HIP device, no CPU path, every arithmetic
operation runs in ``libicap_hip.so``.
"""
from __future__ import annotations

import torch

from . import _native as N
from . import ops
from .ops import ACT_NONE, _Span, _ck, _stream

COL_SUM, COL_SQDEV, COL_BN_BWD, COL_LN_BWD, COL_MOMENTS = 0, 1, 2, 3, 4


def _p(t):
    return t.data_ptr() if t is not None else None


def _ckr(t, name):
    """fp32 matrix on the device whose rows are contiguous (a row stride larger than the width is fine)."""
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise N.IcapError(f"{name}: expected a tensor on the HIP device; the MI355X path has no CPU fallback")
    if t.dtype != torch.float32 or t.dim() != 2 or (t.size(1) > 1 and t.stride(1) != 1) or (t.size(0) > 1 and t.stride(0) < t.size(1)):
        raise N.IcapError(f"{name}: expected an fp32 [rows, cols] matrix with contiguous rows")


def _ws(n: int, dev) -> torch.Tensor | None:
    return torch.empty(max(int(n), 1), device=dev, dtype=torch.float32) if n > 0 else None


def accumulate_into_grads(params, grads):
    """Adds grads[i] into params[i].grad for every parameter that already holds a gradient buffer of the same shape - ONE
    multi-tensor add instead of autograd's one small add per parameter (213 of them for the EfficientNet) - and then runs
    the parameters' post-accumulate hooks, as autograd would have.  Returns the gradients still to be handed to autograd:
    None where the sum was taken here, grads[i] elsewhere (no buffer yet, other shape or dtype).
    autograd runs the same hooks once more for the None it receives in place of such a gradient (observed on torch 2.10), so
    hooks that count must ignore a parameter's second call within one backward pass (training.GradientReducer does)."""
    if not hasattr(torch.Tensor, "_post_accumulate_grad_hooks"):
        return tuple(grads)         # no way to run the hooks ourselves on this torch: leave every gradient to autograd
    dst, src, idx = [], [], []
    for i, (p, g) in enumerate(zip(params, grads)):
        if g is None or p.grad is None or p.grad.shape != g.shape or p.grad.dtype != g.dtype or p.grad.device != g.device:
            continue
        dst.append(p.grad); src.append(g); idx.append(i)
    if not dst:
        return tuple(grads)
    torch._foreach_add_(dst, src)
    left = list(grads)
    for i in idx:
        left[i] = None
        hooks = getattr(params[i], "_post_accumulate_grad_hooks", None)
        if hooks:
            for h in list(hooks.values()):
                h(params[i])
    return tuple(left)


def transpose(a: torch.Tensor) -> torch.Tensor:
    """[R, C] (rows may be strided) -> dense [C, R]."""
    _ckr(a, "a")
    R, C = a.shape
    out = torch.empty(C, R, device=a.device, dtype=torch.float32)
    N.check(N.load().icap_transpose_f32(a.data_ptr(), R, C, a.stride(0), out.data_ptr(), _stream()), "transpose")
    return out


# rows from which gemm_nn goes through the tile kernel of ops.linear (one launch, split-K inside) rather than the generic
# backward kernel + its split-K reduction
GEMM_NN_TILE_MIN_ROWS = 8


def gemm_nn(a: torch.Tensor, b: torch.Tensor, out: torch.Tensor | None = None, accumulate: bool = False,
            add: torch.Tensor | None = None) -> torch.Tensor:
    """out[M, N] (+)= a[M, K] @ b[K, N]  (input gradient of a Linear: dX = dY @ W).  add [M, N]: a second gradient flowing
    into the same tensor (an identity skip), added in the GEMM's epilogue instead of by a pass of its own."""
    _ckr(a, "a"); _ckr(b, "b")
    M, K = a.shape
    K2, Nn = b.shape
    if K != K2:
        raise N.IcapError(f"gemm_nn: inner sizes differ ({K} vs {K2})")
    if add is not None:
        if out is not None or accumulate or tuple(add.shape) != (M, Nn) or not add.is_contiguous():
            raise N.IcapError("gemm_nn: `add` needs a fresh result and a contiguous [M, N] tensor")
        if K % 4 == 0 and Nn % 4 == 0 and a.stride(0) == K and M >= GEMM_NN_TILE_MIN_ROWS:
            bt = ops.step_cached("transposed", b, lambda: transpose(b))
            return ops.linear(a, bt, residual=add)
        return axpby(add, gemm_nn(a, b))
    if out is None:
        out = torch.empty(M, Nn, device=a.device, dtype=torch.float32)
        accumulate = False
    lib = N.load()
    if K % 4 == 0 and Nn % 4 == 0 and a.stride(0) == K and out.stride(0) == Nn and M >= GEMM_NN_TILE_MIN_ROWS:
        # the tuned implicit-GEMM kernel wants the contraction axis of both operands contiguous: a transposed copy of the
        # (small) second operand buys its 60-100 TFLOP/s instead of the k-major staging of the generic backward kernel
        # (inside a training step the copy is made once per weight, not once per time step)
        bt = ops.step_cached("transposed", b, lambda: transpose(b))
        return ops.linear(a, bt, residual=out if accumulate else None, out=out)
    ws = _ws(lib.icap_gemm_bwd_ws_floats(M, Nn, K, int(accumulate)), a.device)
    span = _Span("gemm_nn", (M, K, Nn))
    N.check(lib.icap_gemm_bwd_f32(a.data_ptr(), a.stride(0), 0, b.data_ptr(), b.stride(0), out.data_ptr(), out.stride(0), M, Nn, K,
                                  int(accumulate), _p(ws), _stream()), "gemm_nn")
    span.done()
    return out


def gemm_tn(a: torch.Tensor, b: torch.Tensor, out: torch.Tensor | None = None, accumulate: bool = False) -> torch.Tensor:
    """out[M, N] (+)= a[K, M].T @ b[K, N]  (weight gradient of a Linear: dW = dY.T @ X)."""
    _ckr(a, "a"); _ckr(b, "b")
    K, M = a.shape
    K2, Nn = b.shape
    if K != K2:
        raise N.IcapError(f"gemm_tn: contraction sizes differ ({K} vs {K2})")
    if out is None:
        out = torch.empty(M, Nn, device=a.device, dtype=torch.float32)
        accumulate = False
    lib = N.load()
    ws = _ws(lib.icap_gemm_bwd_ws_floats(M, Nn, K, int(accumulate)), a.device)
    span = _Span("gemm_tn", (K, M, Nn))
    N.check(lib.icap_gemm_bwd_f32(a.data_ptr(), a.stride(0), 1, b.data_ptr(), b.stride(0), out.data_ptr(), out.stride(0), M, Nn, K,
                                  int(accumulate), _p(ws), _stream()), "gemm_tn")
    span.done()
    return out


def col_reduce(mode: int, a, x=None, v1=None, v2=None, v3=None, v4=None, r1=None, r2=None, act=ACT_NONE, scale=1.0,
               out1=None, out2=None, accumulate=False):
    """Column reductions over a [M, C] matrix (see icap_col_reduce_f32); returns (out1, out2)."""
    _ckr(a, "a")
    M, C = a.shape
    dev = a.device
    two = mode >= COL_BN_BWD
    if out1 is None:
        out1 = torch.empty(C, device=dev, dtype=torch.float32)
        accumulate = False
    if two and out2 is None:
        out2 = torch.empty(C, device=dev, dtype=torch.float32)
    lib = N.load()
    ws = _ws(lib.icap_col_reduce_ws_floats(M, C), dev)
    span = _Span("col_reduce", (mode, M, C))
    N.check(lib.icap_col_reduce_f32(mode, a.data_ptr(), a.stride(0), _p(x), x.stride(0) if x is not None else 0, _p(v1), _p(v2), _p(v3),
                                    _p(v4), _p(r1), _p(r2), act, M, C, float(scale), int(accumulate), out1.data_ptr(), _p(out2),
                                    ws.data_ptr(), _stream()), "col_reduce")
    span.done()
    return out1, out2


def col_sum(a, out=None, accumulate=False, scale=1.0):
    return col_reduce(COL_SUM, a, out1=out, accumulate=accumulate, scale=scale)[0]


def batchnorm_train(x, gamma, beta, run_mean, run_var, eps, momentum, act=ACT_NONE, res=None, out=None, parts=None, pool=None,
                    pool_only=False):
    """BatchNorm over the rows of x [M, C] with batch statistics (+ activation, + residual); updates the running
    statistics in place.  Returns (y, saved) with saved = (mean, rstd, s, b) for the backward pass.
    parts (ops.StatParts): the column statistics the kernel that produced x left behind - the pass over x that would
    otherwise take them is skipped.
    pool = (B, HW): the rows are B images of HW positions and the per-image channel means of y are wanted too (taken by
    the pass that writes y); returns (y, saved, pooled [B, C]).  pool_only: y is not written at all (None is returned
    for it) - for a consumer that recomputes the activation from x with saved's scale and bias (act_gate)."""
    _ckr(x, "x")
    M, C = x.shape
    dev = x.device
    if parts is not None:
        mean = torch.empty(C, device=dev); rstd = torch.empty(C, device=dev); s = torch.empty(C, device=dev); b = torch.empty(C, device=dev)
        lib = N.load()
        ws = _ws(lib.icap_bn_stats_from_parts_ws_floats(parts.P, C), dev)
        N.check(lib.icap_bn_stats_from_parts_f32(parts.part.data_ptr(), _p(parts.cnt), parts.P, parts.rows_per_part, M, C, _p(gamma), _p(beta),
                                                 float(eps), float(momentum), mean.data_ptr(), rstd.data_ptr(), s.data_ptr(), b.data_ptr(),
                                                 _p(run_mean), _p(run_var), ws.data_ptr(), _stream()), "bn_stats_from_parts")
        if pool is not None:
            y, pooled = act_pool(x, s, b, act, pool[0], pool[1], write=not pool_only)
            return y, (mean, rstd, s, b), pooled
        y = scale_bias_act(x, s, b, act=act, res=res, out=out)
        return y, (mean, rstd, s, b)
    # one pass over x: moments about its first row (a sample of every column, so the subtraction stays benign); the second
    # pass of that reduction also derives mean / rstd / the fused scale and bias and updates the running statistics
    mean = torch.empty(C, device=dev); rstd = torch.empty(C, device=dev); s = torch.empty(C, device=dev); b = torch.empty(C, device=dev)
    lib = N.load()
    ws = _ws(lib.icap_col_reduce_ws_floats(M, C), dev)
    span = _Span("col_reduce", (COL_MOMENTS, M, C))
    N.check(lib.icap_bn_train_stats_f32(x.data_ptr(), x.stride(0), _p(gamma), _p(beta), float(eps), float(momentum), M, C,
                                        mean.data_ptr(), rstd.data_ptr(), s.data_ptr(), b.data_ptr(), _p(run_mean), _p(run_var),
                                        ws.data_ptr(), _stream()), "bn_train_stats")
    span.done()
    if pool is not None:
        y, pooled = act_pool(x, s, b, act, pool[0], pool[1], write=not pool_only)
        return y, (mean, rstd, s, b), pooled
    y = scale_bias_act(x, s, b, act=act, res=res, out=out)
    return y, (mean, rstd, s, b)


def scale_bias_act(x, s=None, b=None, act=ACT_NONE, res=None, out=None):
    _ckr(x, "x")
    M, C = x.shape
    if out is None:
        out = torch.empty(M, C, device=x.device, dtype=torch.float32)
    span = _Span("scale_bias_act", (M, C, res is not None))
    N.check(N.load().icap_scale_bias_act_f32(x.data_ptr(), x.stride(0), _p(s), _p(b), _p(res), res.stride(0) if res is not None else 0,
                                             act, M, C, out.data_ptr(), out.stride(0), _stream()), "scale_bias_act")
    span.done()
    return out


def batchnorm_steps_fwd(x, gamma, beta, run_mean, run_var, eps, momentum, B, Tn, out):
    """BatchNorm1d over the columns of x [Tn*B, V] (step-major) with the statistics of each step's own B rows, all steps in
    one launch each; out [B, Tn, V] receives the result, the running statistics one momentum update per step in step
    order.  Returns saved = (mean, rstd, s, b), each [Tn, V]."""
    _ckr(x, "x")
    V = x.shape[1]
    if tuple(x.shape) != (Tn * B, V) or tuple(out.shape) != (B, Tn, V) or not (x.is_contiguous() and out.is_contiguous()):
        raise N.IcapError("batchnorm_steps_fwd: x must be a dense [T*B, V] matrix and out a dense [B, T, V] tensor")
    dev = x.device
    mean, rstd, s, b, uvar = (torch.empty(Tn, V, device=dev, dtype=torch.float32) for _ in range(5))
    span = _Span("bn_steps_fwd", (B, Tn, V))
    N.check(N.load().icap_bn_steps_fwd_f32(x.data_ptr(), _p(gamma), _p(beta), float(eps), float(momentum), B, Tn, V, out.data_ptr(),
                                           mean.data_ptr(), rstd.data_ptr(), s.data_ptr(), b.data_ptr(), uvar.data_ptr(),
                                           _p(run_mean), _p(run_var), _stream()), "bn_steps_fwd")
    span.done()
    return mean, rstd, s, b


def batchnorm_steps_bwd(dy, x, saved, B, Tn):
    """Gradient of batchnorm_steps_fwd: dy [B, Tn, V] -> (dx [Tn*B, V], dgamma_t [Tn, V], dbeta_t [Tn, V])."""
    mean, rstd, s, _ = saved
    V = x.shape[1]
    if tuple(dy.shape) != (B, Tn, V) or not (dy.is_contiguous() and x.is_contiguous()):
        raise N.IcapError("batchnorm_steps_bwd: dy must be a dense [B, T, V] tensor")
    dev = x.device
    dx = torch.empty(Tn * B, V, device=dev, dtype=torch.float32)
    dg = torch.empty(Tn, V, device=dev, dtype=torch.float32); db = torch.empty(Tn, V, device=dev, dtype=torch.float32)
    span = _Span("bn_steps_bwd", (B, Tn, V))
    N.check(N.load().icap_bn_steps_bwd_f32(dy.data_ptr(), x.data_ptr(), mean.data_ptr(), rstd.data_ptr(), s.data_ptr(), B, Tn, V,
                                           dg.data_ptr(), db.data_ptr(), dx.data_ptr(), _stream()), "bn_steps_bwd")
    span.done()
    return dx, dg, db


def act_pool(x, s, b, act, B, HW, write=True):
    """y = act(x * s + b) over x [B*HW, C] and the per-image channel means of y, in one pass -> (y, pooled [B, C]).
    write=False: only the means (y is returned as None)."""
    _ckr(x, "x")
    M, C = x.shape
    if M != B * HW or not x.is_contiguous():
        raise N.IcapError("act_pool: x must be a dense [B*HW, C] matrix")
    lib = N.load()
    y = torch.empty(M, C, device=x.device, dtype=torch.float32) if write else None
    pooled = torch.empty(B, C, device=x.device, dtype=torch.float32)
    ws = _ws(lib.icap_act_pool_ws_floats(B, HW, C), x.device)
    span = _Span("act_pool", (B, HW, C, write))
    N.check(lib.icap_act_pool_f32(x.data_ptr(), s.data_ptr(), b.data_ptr(), act, B, HW, C, 1.0 / HW, _p(y), pooled.data_ptr(),
                                  ws.data_ptr(), _stream()), "act_pool")
    span.done()
    return y, pooled


def act_gate(x, s, b, act, g, B, HW):
    """act(x * s + b) * g[image] over x [B*HW, C] with g [B, C]: BatchNorm apply, activation and squeeze-excite gate in one pass."""
    _ckr(x, "x")
    M, C = x.shape
    if M != B * HW or not x.is_contiguous() or tuple(g.shape) != (B, C):
        raise N.IcapError("act_gate: x must be a dense [B*HW, C] matrix and g [B, C]")
    y = torch.empty(M, C, device=x.device, dtype=torch.float32)
    span = _Span("act_gate", (B, HW, C))
    N.check(N.load().icap_act_gate_f32(x.data_ptr(), s.data_ptr(), b.data_ptr(), act, g.data_ptr(), B, HW, C, y.data_ptr(), _stream()),
            "act_gate")
    span.done()
    return y


def image_dot_act(x, z, s, b, act, B, HW, C, scale=1.0):
    """[B, C] per-image sums of x * act(z * s + b): image_dot against an activation recomputed from its BatchNorm input."""
    lib = N.load()
    out = torch.empty(B, C, device=x.device, dtype=torch.float32)
    ws = _ws(lib.icap_image_dot_ws_floats(B, HW, C), x.device)
    span = _Span("image_dot", (B, HW, C, True))
    N.check(lib.icap_image_dot_act_f32(x.data_ptr(), z.data_ptr(), s.data_ptr(), b.data_ptr(), act, B, HW, C, float(scale), out.data_ptr(),
                                       ws.data_ptr(), _stream()), "image_dot_act")
    span.done()
    return out


def batchnorm_act_bwd_se(dy, x, saved, gate, dpool, alpha, HW, act=ACT_NONE):
    """batchnorm_act_bwd for a gradient that still has to pass a squeeze-excite gate: the gradient used is
    dy * gate[image] + dpool[image] * alpha, formed inside the two passes.  Returns (dx, dgamma, dbeta)."""
    mean, rstd, s, b = saved
    M, C = x.shape
    if not (dy.is_contiguous() and x.is_contiguous() and tuple(dy.shape) == (M, C)):
        raise N.IcapError("batchnorm_act_bwd_se: dense [M, C] operands required")
    lib = N.load()
    dev = x.device
    dx = torch.empty(M, C, device=dev, dtype=torch.float32)
    dgamma = torch.empty(C, device=dev, dtype=torch.float32); dbeta = torch.empty(C, device=dev, dtype=torch.float32)
    ws = _ws(lib.icap_col_reduce_ws_floats(M, C), dev)
    span = _Span("bn_act_bwd_se", (M, C))
    N.check(lib.icap_bn_act_bwd_se_f32(dy.data_ptr(), x.data_ptr(), s.data_ptr(), b.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                       gate.data_ptr(), dpool.data_ptr(), float(alpha), HW, act, M, C, dgamma.data_ptr(), dbeta.data_ptr(),
                                       dx.data_ptr(), ws.data_ptr(), _stream()), "bn_act_bwd_se")
    span.done()
    return dx, dgamma, dbeta


def batchnorm_act_bwd(dy, x, saved, act=ACT_NONE, affine=True):
    """Gradient of y = act(BatchNorm_train(x)): returns (dx, dgamma, dbeta)."""
    mean, rstd, s, b = saved
    M, C = x.shape
    dbeta, dgamma = col_reduce(COL_BN_BWD, dy, x=x, v1=mean, v2=rstd, v3=s, v4=b, act=act)
    dx = torch.empty(M, C, device=x.device, dtype=torch.float32)
    span = _Span("bn_act_bwd", (M, C))
    N.check(N.load().icap_bn_act_bwd_f32(dy.data_ptr(), dy.stride(0), x.data_ptr(), x.stride(0), s.data_ptr(), b.data_ptr(),
                                         mean.data_ptr(), rstd.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), act, M, C,
                                         dx.data_ptr(), dx.stride(0), _stream()), "bn_act_bwd")
    span.done()
    return dx, dgamma, dbeta


def act_bwd(dy, z, act, s=None, b=None):
    """dy * act'(z*s + b) * s  (activation after an eval-style affine, or a bare activation with s = b = None)."""
    M, C = z.shape
    dx = torch.empty(M, C, device=z.device, dtype=torch.float32)
    N.check(N.load().icap_bn_act_bwd_f32(dy.data_ptr(), dy.stride(0), z.data_ptr(), z.stride(0), _p(s), _p(b), None, None, None, None,
                                         act, M, C, dx.data_ptr(), dx.stride(0), _stream()), "act_bwd")
    return dx


def axpby(x, y, alpha=1.0, beta=1.0):
    """y = alpha*x + beta*y in place."""
    N.check(N.load().icap_axpby_f32(x.data_ptr(), float(alpha), float(beta), y.data_ptr(), x.numel(), _stream()), "axpby")
    return y


# While a training step is being recorded into a HIP graph (training.Trainer(graph=True)) this holds a one-element
# int64 device tensor: the dropout kernels add its value to their offset, and the graph advances it at the end of every
# replay, so a recorded step draws new masks each time it runs.
DROPOUT_BASE = None


def dropout(x, p, seed, offset, group=1):
    """Returns (y, mask); mask has x.numel() // group entries (0 or 1/(1-p))."""
    _ck(x, "x")
    n = x.numel()
    y = torch.empty_like(x)
    mask = torch.empty(n // group, device=x.device, dtype=torch.float32)
    if DROPOUT_BASE is not None:
        N.check(N.load().icap_dropout_dev_f32(x.data_ptr(), float(p), int(seed) & (2 ** 64 - 1), int(offset), DROPOUT_BASE.data_ptr(), n,
                                              group, y.data_ptr(), mask.data_ptr(), _stream()), "dropout_dev")
        return y, mask
    N.check(N.load().icap_dropout_f32(x.data_ptr(), float(p), int(seed) & (2 ** 64 - 1), int(offset), n, group, y.data_ptr(),
                                      mask.data_ptr(), _stream()), "dropout")
    return y, mask


def mul_mask(x, mask, group=1, add=None):
    y = torch.empty_like(x)
    N.check(N.load().icap_mul_mask_f32(x.data_ptr(), mask.data_ptr(), group, _p(add), x.numel(), y.data_ptr(), _stream()), "mul_mask")
    return y


def layernorm_train(x, gamma, beta, res=None, eps=1e-5, pre_scale=1.0):
    """Returns (y, saved) with saved = (xs, mean, rstd)."""
    _ck(x, "x")
    R, E = x.shape
    dev = x.device
    y = torch.empty(R, E, device=dev); xs = torch.empty(R, E, device=dev)
    mean = torch.empty(R, device=dev); rstd = torch.empty(R, device=dev)
    N.check(N.load().icap_ln_train_fwd_f32(x.data_ptr(), _p(res), gamma.data_ptr(), beta.data_ptr(), float(eps), float(pre_scale), R, E,
                                           y.data_ptr(), xs.data_ptr(), mean.data_ptr(), rstd.data_ptr(), _stream()), "ln_train_fwd")
    return y, (xs, mean, rstd)


def layernorm_bwd(dy, saved, gamma, post_scale=1.0):
    """Returns (dx, dgamma, dbeta); dx is the gradient with respect to x (and to the residual) times post_scale."""
    xs, mean, rstd = saved
    R, E = xs.shape
    dx = torch.empty(R, E, device=xs.device)
    N.check(N.load().icap_ln_bwd_f32(dy.data_ptr(), xs.data_ptr(), gamma.data_ptr(), mean.data_ptr(), rstd.data_ptr(), float(post_scale),
                                     R, E, dx.data_ptr(), _stream()), "ln_bwd")
    dbeta, dgamma = col_reduce(COL_LN_BWD, dy, x=xs, r1=mean, r2=rstd)
    return dx, dgamma, dbeta


def lstm_bwd(gates, c_in, c_out, dh, dc, dg_out=None):
    """Returns (dgates [B, 4H], dc_prev [B, H]); dg_out: optional contiguous [B, 4H] destination of the first."""
    B, H4 = gates.shape
    H = H4 // 4
    dg = dg_out if dg_out is not None else torch.empty(B, H4, device=gates.device)
    dcp = torch.empty(B, H, device=gates.device)
    N.check(N.load().icap_lstm_bwd_f32(gates.data_ptr(), c_in.data_ptr(), c_out.data_ptr(), dh.data_ptr(), _p(dc), B, H, dg.data_ptr(),
                                       dcp.data_ptr(), _stream()), "lstm_bwd")
    return dg, dcp


def gru_bwd(gi, gh, h_in, dh):
    """Returns (dgi [B, 3H], dgh [B, 3H], dh_direct [B, H])."""
    B, H3 = gi.shape
    H = H3 // 3
    dgi = torch.empty(B, H3, device=gi.device); dgh = torch.empty(B, H3, device=gi.device)
    dhd = torch.empty(B, H, device=gi.device)
    N.check(N.load().icap_gru_bwd_f32(gi.data_ptr(), gh.data_ptr(), h_in.data_ptr(), dh.data_ptr(), B, H, dgi.data_ptr(), dgh.data_ptr(),
                                      dhd.data_ptr(), _stream()), "gru_bwd")
    return dgi, dgh, dhd


def attention_bwd(u, w, a_w, alpha, feat, dctx, du, dfeat, accumulate, outs=None):
    """Returns (dw [B, A], da_part [B, A], dab_part [B]); du / dfeat are written or accumulated in place.
    outs: optional (dw, da_part, dab_part) contiguous destinations."""
    B, A = u.shape
    E = feat.shape[1]
    dev = u.device
    dscore = torch.empty(B, B, device=dev)
    if outs is not None:
        dw, da_part, dab_part = outs
    else:
        dw = torch.empty(B, A, device=dev); da_part = torch.empty(B, A, device=dev); dab_part = torch.empty(B, device=dev)
    N.check(N.load().icap_att_bwd_f32(u.data_ptr(), w.data_ptr(), a_w.data_ptr(), alpha.data_ptr(), feat.data_ptr(), dctx.data_ptr(), B, A, E,
                                      int(accumulate), dscore.data_ptr(), dw.data_ptr(), da_part.data_ptr(), dab_part.data_ptr(),
                                      du.data_ptr(), dfeat.data_ptr(), _stream()), "att_bwd")
    return dw, da_part, dab_part


def mha_bwd(q, k, v, dO, heads):
    """Gradient of softmax(q k^T / sqrt(d)) v per head: q, dO [R, E]; k, v [Bk, E] -> (dq, dk, dv)."""
    R, E = q.shape
    Bk = k.shape[0]
    dev = q.device
    dq = torch.empty(R, E, device=dev); dk = torch.empty(Bk, E, device=dev); dv = torch.empty(Bk, E, device=dev)
    ws = torch.empty(2 * R * heads * Bk, device=dev)
    N.check(N.load().icap_mha_bwd_f32(q.data_ptr(), k.data_ptr(), v.data_ptr(), dO.data_ptr(), R, Bk, E, heads, dq.data_ptr(), dk.data_ptr(),
                                      dv.data_ptr(), ws.data_ptr(), _stream()), "mha_bwd")
    return dq, dk, dv


def embedding_bwd(idx, dy, V):
    """dtable [V, E] = scatter-add of dy rows by idx (deterministic)."""
    P, E = dy.shape
    dt = torch.zeros(V, E, device=dy.device)
    N.check(N.load().icap_embedding_bwd_f32(idx.data_ptr(), dy.data_ptr(), dy.stride(0), P, E, V, dt.data_ptr(), _stream()), "embedding_bwd")
    return dt


# ---------------------------------------------------------------------------- convolutional encoder
def chan_affine(x, g, a, alpha, B, HW, C):
    """y[b,p,c] = x*g[b][c] + a[b][c]*alpha over an NHWC tensor viewed as [B*HW, C] (x, g, a optional)."""
    ref = x if x is not None else a
    y = torch.empty(B * HW, C, device=ref.device, dtype=torch.float32)
    span = _Span("chan_affine", (B, HW, C, x is not None))
    N.check(N.load().icap_chan_affine_f32(_p(x), _p(g), _p(a), float(alpha), B, HW, C, y.data_ptr(), _stream()), "chan_affine")
    span.done()
    return y


def image_dot(x, y, B, HW, C, scale=1.0):
    out = torch.empty(B, C, device=x.device, dtype=torch.float32)
    lib = N.load()
    ws = _ws(lib.icap_image_dot_ws_floats(B, HW, C), x.device)
    span = _Span("image_dot", (B, HW, C, y is not None))
    N.check(lib.icap_image_dot_f32(x.data_ptr(), _p(y), B, HW, C, float(scale), out.data_ptr(), ws.data_ptr(), _stream()), "image_dot")
    span.done()
    return out


def se_train_fwd(pooled, wr, br, we, be):
    """Squeeze-excite branch in training mode, one launch: pooled [B, C], wr [S, C], br [S], we [C, S], be [C]
    -> (r_pre [B, S], e [B, C], gate [B, C]) with r_pre = wr.pooled + br, e = we.swish(r_pre) + be, gate = sigmoid(e)."""
    B, C = pooled.shape
    S = wr.shape[0]
    for t, shape, name in ((wr, (S, C), "wr"), (br, (S,), "br"), (we, (C, S), "we"), (be, (C,), "be")):
        if tuple(t.shape) != shape or not t.is_contiguous() or t.dtype != torch.float32 or not t.is_cuda:
            raise N.IcapError(f"se_train_fwd: {name} must be a contiguous fp32 device tensor of shape {shape}")
    _ckr(pooled, "pooled")
    dev = pooled.device
    r_pre, e, gate = torch.empty(B, S, device=dev), torch.empty(B, C, device=dev), torch.empty(B, C, device=dev)
    span = _Span("se_train_fwd", (B, C, S))
    N.check(N.load().icap_se_train_fwd_f32(pooled.data_ptr(), wr.data_ptr(), br.data_ptr(), we.data_ptr(), be.data_ptr(),
                                           r_pre.data_ptr(), e.data_ptr(), gate.data_ptr(), B, C, S, _stream()), "se_train_fwd")
    span.done()
    return r_pre, e, gate


def se_train_bwd(dgate, e, r_pre, pooled, wr, we):
    """Gradient of se_train_fwd, two launches: -> (dpooled [B, C], dwe [C, S], dbe [C], dwr [S, C], dbr [S])."""
    B, C = dgate.shape
    S = wr.shape[0]
    for t, name in ((dgate, "dgate"), (e, "e"), (r_pre, "r_pre"), (pooled, "pooled"), (wr, "wr"), (we, "we")):
        if not t.is_contiguous() or t.dtype != torch.float32 or not t.is_cuda:
            raise N.IcapError(f"se_train_bwd: {name} must be a contiguous fp32 device tensor")
    dev = dgate.device
    de, drp, dpooled = torch.empty(B, C, device=dev), torch.empty(B, S, device=dev), torch.empty(B, C, device=dev)
    dwe, dbe, dwr, dbr = torch.empty(C, S, device=dev), torch.empty(C, device=dev), torch.empty(S, C, device=dev), torch.empty(S, device=dev)
    span = _Span("se_train_bwd", (B, C, S))
    N.check(N.load().icap_se_train_bwd_f32(dgate.data_ptr(), e.data_ptr(), r_pre.data_ptr(), pooled.data_ptr(), wr.data_ptr(),
                                           we.data_ptr(), de.data_ptr(), drp.data_ptr(), dpooled.data_ptr(), dwe.data_ptr(),
                                           dbe.data_ptr(), dwr.data_ptr(), dbr.data_ptr(), B, C, S, _stream()), "se_train_bwd")
    span.done()
    return dpooled, dwe, dbe, dwr, dbr


def dwconv_bwd_data(dy, w, B, H, W, C, K, S, pad_t, pad_l, Ho, Wo):
    if S == 1 and Ho == H and Wo == W and C % 4 == 0 and dy.is_contiguous():
        # stride 1: the input gradient is itself a depthwise convolution - of dy with the taps in reverse order and the
        # padding mirrored - so it runs on the forward strip kernel (no BatchNorm, no activation)
        dx, _ = ops.dwconv2d(dy.view(B, Ho, Wo, C), w.flip(0).contiguous(), ks=K, stride=1,
                             pad=(K - 1 - pad_t, pad_t, K - 1 - pad_l, pad_l))
        return dx.view(B * H * W, C)
    dx = torch.empty(B * H * W, C, device=dy.device, dtype=torch.float32)
    span = _Span("dwconv_bwd_data", (B, H, W, C, K, S, Ho, Wo))
    N.check(N.load().icap_dwconv_bwd_data_f32(dy.data_ptr(), w.data_ptr(), B, H, W, C, K, S, pad_t, pad_l, Ho, Wo, dx.data_ptr(), _stream()),
            "dwconv_bwd_data")
    span.done()
    return dx


def dwconv_bwd_weight(dy, x, B, H, W, C, K, S, pad_t, pad_l, Ho, Wo):
    lib = N.load()
    dw = torch.empty(K * K, C, device=dy.device, dtype=torch.float32)
    ws = _ws(lib.icap_dwconv_bwd_weight_ws_floats(B, Ho, Wo, C, K), dy.device)
    span = _Span("dwconv_bwd_weight", (B, H, W, C, K, S, Ho, Wo))
    N.check(lib.icap_dwconv_bwd_weight_f32(dy.data_ptr(), x.data_ptr(), B, H, W, C, K, S, pad_t, pad_l, Ho, Wo, dw.data_ptr(), ws.data_ptr(),
                                           _stream()), "dwconv_bwd_weight")
    span.done()
    return dw


def conv_wgrad_gather(dz, x, B, H, W, C, Cv, KH, KW, S, pad_t, pad_l, Ho, Wo):
    """Weight gradient of a few-channel convolution (the stem) without a patch matrix: dz [B*Ho*Wo, Cout], x [B, H, W, C]
    with Cv real channels -> dw [Cout, KH*KW*Cv] (tap-major, channel fastest).  Cout and KH*KW*Cv at most 32."""
    _ckr(dz, "dz")
    Cout = dz.shape[1]
    if dz.shape[0] != B * Ho * Wo or x.numel() != B * H * W * C or not x.is_contiguous():
        raise N.IcapError("conv_wgrad_gather: dz must be [B*Ho*Wo, Cout] and x a dense [B, H, W, C] tensor")
    lib = N.load()
    Nn = KH * KW * Cv
    dw = torch.empty(Cout, Nn, device=x.device, dtype=torch.float32)
    ws = _ws(max(lib.icap_conv_wgrad_gather_ws_floats(B, Ho, Wo, Cout, Nn), 1), x.device)
    span = _Span("conv_wgrad_gather", (B, H, W, C, Cv, KH, S, Ho, Wo, Cout))
    N.check(lib.icap_conv_wgrad_gather_f32(dz.data_ptr(), dz.stride(0), x.data_ptr(), B, H, W, C, Cv, KH, KW, S, pad_t, pad_l, Ho, Wo, Cout,
                                           dw.data_ptr(), ws.data_ptr(), _stream()), "conv_wgrad_gather")
    span.done()
    return dw


def im2col(x, B, H, W, C, KH, KW, S, pad_t, pad_l, Ho, Wo):
    col = torch.empty(B * Ho * Wo, KH * KW * C, device=x.device, dtype=torch.float32)
    N.check(N.load().icap_im2col_f32(x.data_ptr(), B, H, W, C, KH, KW, S, pad_t, pad_l, Ho, Wo, col.data_ptr(), _stream()), "im2col")
    return col
