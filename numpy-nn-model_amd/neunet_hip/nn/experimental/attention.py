"""HIPMultiHeadAttention -- the attention block of examples/gpt.ipynb cell 2 on the strided-batched MFMA GEMM
(batched Tensor.matmul neunet/autograd.py:192-230 + where-mask + Softmax(-1)).

The reference reshapes/transposes q,k,v to (B,H,T,dh) on the host and materialises every intermediate.  Here
the three projections stay in their [B,T,H*dh] buffers: nnhipGemmF32Ex addresses head h of batch b through a
two-level batch stride (T*D, dh) with row stride D, so QK^T, attn*V and all four backward GEMMs read and write
that layout directly -- no transpose or .contiguous() copies; 1/sqrt(d_model) and the pad/causal mask are
fused into the softmax kernel.  Same parameters, same order (wq, wk, wv, fc) and the same math as the notebook.
"""
import ctypes
import itertools
import math

import numpy as np

from ...autograd import Tensor
from ..modules import Module
from .embedding import HIPDropout, check_capture_seed, process_dropout_seed
from .linear import HIPLinear, _finish_param, hip_linear_module_backward, hip_linear_module_forward
from ..._lib import AttentionOptions, StridedView
from .utils import call_hip_function, get_current_stream_ptr


def _gemm_ex(A, B, C, M, N, K, lda, ldb, ldc, akm, bkm, b1, sA1, sB1, sC1, b2, sA2, sB2, sC2, alpha=1.0):
    call_hip_function("nnhipGemmF32Ex", A, B, C, None, M, N, K, lda, ldb, ldc, akm, bkm, b1, sA1, sB1, sC1,
                      b2, sA2, sB2, sC2, float(alpha), get_current_stream_ptr())


def attention_forward(q, k, v, key_valid, n_heads, scale, causal, drop_mask=None, dense_mask=None):
    """q,k,v: device arrays [B,T,D] (D = H*dh).  Returns (ctx [B,Tq,D], attn [B,H,Tq,Tk], attn_used) where attn_used is
    attn * drop_mask (the notebook's `self.dropout(softmax(scores))`, cell 2) or attn itself without dropout.
    dense_mask: int32 device array [B,Tq,Tk] (0 = masked), applied on top of (key_valid, causal)."""
    import torch
    B, Tq, D = q.shape
    Tk = k.shape[1]
    H, dh = n_heads, D // n_heads
    scores = torch.empty((B, H, Tq, Tk), dtype=torch.float32, device=q.device)
    # scores[b,h] = q[b,:,h,:] (Tq x dh, k-major, lda=D)  x  k[b,:,h,:]^T (B operand k-major, ldb=D)
    _gemm_ex(q, k, scores, Tq, Tk, dh, D, D, Tk, 1, 1, B, Tq * D, Tk * D, H * Tq * Tk, H, dh, dh, Tq * Tk)
    attn = scores  # softmax in place over the scores buffer
    call_hip_function("nnhipMaskedSoftmaxForwardEx", attn, scores, key_valid, dense_mask, B, H, Tq, Tk, 1.0 / scale,
                      int(causal), get_current_stream_ptr())
    used = attn
    if drop_mask is not None:
        used = torch.empty_like(attn)
        call_hip_function("nnhipMul", used, attn, drop_mask, attn.numel(), get_current_stream_ptr())
    ctx = torch.empty((B, Tq, D), dtype=torch.float32, device=q.device)
    # ctx[b,:,h,:] = used[b,h] (Tq x Tk, k-major) x v[b,:,h,:] (Tk x dh, outer-major, ldb=D) -> written in [B,T,D]
    _gemm_ex(used, v, ctx, Tq, dh, Tk, Tk, D, D, 1, 0, B, H * Tq * Tk, Tk * D, Tq * D, H, Tq * Tk, dh, dh)
    return ctx, attn, used


def _row_stride(*ts):
    """Common row stride (floats) of [B,T,D] arrays that are dense tensors or column blocks of one wider buffer."""
    ld = ts[0].stride(1)
    for t in ts:
        if t.stride(2) != 1 or t.stride(1) != ld or t.stride(0) != ld * t.shape[1]:
            raise ValueError("fused attention needs q, k, v with unit column stride and one common row stride")
    return ld


FUSED_HEAD_DIMS = (32, 64, 128)
_SEEDS = itertools.count(1)


class FusedAttentionOptions:
    """What the fused attention kernels can do beyond (key_valid, causal): a dense mask (packed bits) and attention
    dropout (injected mask or hash RNG) -- the Python side of struct nnhipAttentionOptions.  One object is shared by the
    forward and the backward of a step, so both see the same mask and the same dropout multipliers."""

    def __init__(self, mask_bits=None, mask_bitsT=None, row_any=None, dropout_mask=None, dropout_p=0.0, seed=0,
                 seed_dev=None):
        self.mask_bits, self.mask_bitsT, self.row_any = mask_bits, mask_bitsT, row_any
        self.dropout_mask, self.dropout_p, self.seed, self.seed_dev = dropout_mask, float(dropout_p), int(seed), seed_dev

    def cstruct(self):
        ptr = lambda t: None if t is None else t.data_ptr()  # noqa: E731
        return AttentionOptions(ptr(self.mask_bits), ptr(self.mask_bitsT), ptr(self.row_any), ptr(self.dropout_mask),
                                self.dropout_p if self.dropout_mask is None else 0.0, self.seed & 0xFFFFFFFF, ptr(self.seed_dev))


def pack_attention_mask(mask):
    """Dense mask (device int array [B,Tq,Tk] or [B,1,Tq,Tk], non-zero = visible: the notebook's
    get_pad_mask(x) & get_sub_mask(x), examples/gpt.ipynb cell 7) -> (mask_bits, mask_bitsT, row_any) for the fused
    kernels (nnhipAttentionPackMask)."""
    import torch
    if mask.dim() == 4:
        if mask.shape[1] != 1:
            raise ValueError("a per-head mask is not supported: expected [B,1,Tq,Tk] or [B,Tq,Tk]")
        mask = mask[:, 0]
    if mask.dim() != 3:
        raise ValueError("mask must be [B,Tq,Tk] or [B,1,Tq,Tk]")
    mask = mask.to(torch.int32).contiguous()
    B, Tq, Tk = mask.shape
    bits = torch.empty((B, Tq, (Tk + 63) // 64), dtype=torch.int64, device=mask.device)
    bitsT = torch.empty((B, Tk, (Tq + 63) // 64), dtype=torch.int64, device=mask.device)
    row_any = torch.empty((B, Tq), dtype=torch.uint8, device=mask.device)
    call_hip_function("nnhipAttentionPackMask", mask, bits, bitsT, row_any, B, Tq, Tk, get_current_stream_ptr())
    return bits, bitsT, row_any


def attention_dropout_mask(B, H, Tq, Tk, p, seed, device="cuda"):
    """The [B,H,Tq,Tk] multipliers (0 or 1/(1-p)) the fused kernels' hash RNG yields for (p, seed)."""
    import torch
    out = torch.empty((B, H, Tq, Tk), dtype=torch.float32, device=device)
    call_hip_function("nnhipAttentionDropoutMask", out, B, H, Tq, Tk, float(p), int(seed) & 0xFFFFFFFF, get_current_stream_ptr())
    return out


def fused_attention_forward(q, k, v, key_valid, n_heads, scale, causal, opts=None):
    """Flash-style forward (nnhipAttentionForward[Ex], head_dim 32 / 64 / 128): returns (ctx [B,Tq,D], lse [B,H,Tq,2]);
    the score matrix is never materialised.  q, k, v may be column blocks of one fused [B,T,3D] projection buffer.
    opts: FusedAttentionOptions (dense mask, attention dropout) or None."""
    import torch
    B, Tq, D = q.shape
    Tk = k.shape[1]
    ctx = torch.empty((B, Tq, D), dtype=torch.float32, device=q.device)
    lse = torch.empty((B, n_heads, Tq, 2), dtype=torch.float32, device=q.device)   # (row max, log2 row sum)
    if opts is None:
        call_hip_function("nnhipAttentionForward", StridedView(q), StridedView(k), StridedView(v), key_valid, ctx, lse, B,
                          n_heads, Tq, Tk, D // n_heads, _row_stride(q, k, v), 1.0 / scale, int(causal),
                          get_current_stream_ptr())
    else:
        cs = opts.cstruct()
        call_hip_function("nnhipAttentionForwardEx", StridedView(q), StridedView(k), StridedView(v), key_valid, ctx, lse, B,
                          n_heads, Tq, Tk, D // n_heads, _row_stride(q, k, v), 1.0 / scale, int(causal), ctypes.byref(cs),
                          get_current_stream_ptr())
    return ctx, lse


def fused_attention_backward(q, k, v, key_valid, ctx, lse, n_heads, scale, causal, dctx, out=None, opts=None):
    """Flash-style backward (nnhipAttentionBackward[Ex]): (dq, dk, dv) from the saved ctx and row statistics.
    out = (dq, dk, dv) destinations laid out like q, k, v (e.g. column blocks of one [B,T,3D] buffer).
    opts must be the object the forward used."""
    import torch
    B, Tq, D = q.shape
    Tk = k.shape[1]
    dq, dk, dv = out if out is not None else (torch.empty_like(q), torch.empty_like(k), torch.empty_like(v))
    ld = _row_stride(q, k, v)
    if _row_stride(dq, dk, dv) != ld:
        raise ValueError("dq/dk/dv must share the row stride of q/k/v")
    if opts is None:
        call_hip_function("nnhipAttentionBackward", StridedView(q), StridedView(k), StridedView(v), key_valid, ctx, dctx,
                          lse, StridedView(dq), StridedView(dk), StridedView(dv), B, n_heads, Tq, Tk, D // n_heads, ld,
                          1.0 / scale, int(causal), get_current_stream_ptr())
    else:
        cs = opts.cstruct()
        call_hip_function("nnhipAttentionBackwardEx", StridedView(q), StridedView(k), StridedView(v), key_valid, ctx, dctx,
                          lse, StridedView(dq), StridedView(dk), StridedView(dv), B, n_heads, Tq, Tk, D // n_heads, ld,
                          1.0 / scale, int(causal), ctypes.byref(cs), get_current_stream_ptr())
    return dq, dk, dv


def attention_backward(q, k, v, attn, key_valid, n_heads, scale, causal, dctx, need=(True, True, True),
                       drop_mask=None, attn_used=None, dense_mask=None):
    """Returns (dq, dk, dv) in the [B,T,D] layout of the projections.  With attention dropout: attn_used = attn*mask
    feeds dV, and the gradient of the attention map is multiplied by the same mask before the softmax backward."""
    import torch
    B, Tq, D = q.shape
    Tk = k.shape[1]
    H, dh = n_heads, D // n_heads
    dattn = torch.empty_like(attn)
    # dattn[b,h] = dctx[b,:,h,:] (Tq x dh, k-major lda=D) x v[b,:,h,:]^T (k-major ldb=D)
    _gemm_ex(dctx, v, dattn, Tq, Tk, dh, D, D, Tk, 1, 1, B, Tq * D, Tk * D, H * Tq * Tk, H, dh, dh, Tq * Tk)
    dv = None
    if need[2]:
        dv = torch.empty_like(v)
        # dv[b,:,h,:] = attn[b,h]^T (A outer-major, lda=Tk) x dctx[b,:,h,:] (outer-major, ldb=D)
        _gemm_ex(attn if attn_used is None else attn_used, dctx, dv, Tk, dh, Tq, Tk, D, D, 0, 0, B, H * Tq * Tk, Tq * D,
                 Tk * D, H, Tq * Tk, dh, dh)
    if drop_mask is not None:
        call_hip_function("nnhipMul", dattn, dattn, drop_mask, dattn.numel(), get_current_stream_ptr())
    # dscores (in place over dattn) = where(mask, 0, softmax_bwd(dattn, attn)) / scale
    call_hip_function("nnhipMaskedSoftmaxBackwardEx", dattn, dattn, attn, key_valid, dense_mask, B, H, Tq, Tk, 1.0 / scale,
                      int(causal), get_current_stream_ptr())
    ds = dattn
    dq = dk = None
    if need[0]:
        dq = torch.empty_like(q)
        _gemm_ex(ds, k, dq, Tq, dh, Tk, Tk, D, D, 1, 0, B, H * Tq * Tk, Tk * D, Tq * D, H, Tq * Tk, dh, dh)
    if need[1]:
        dk = torch.empty_like(k)
        _gemm_ex(ds, q, dk, Tk, dh, Tq, Tk, D, D, 0, 0, B, H * Tq * Tk, Tq * D, Tk * D, H, Tq * Tk, dh, dh)
    return dq, dk, dv


class _HIPAttentionTensor(Tensor):
    def __init__(self, data, args, op, device):
        super().__init__(data, args, op, device=device, _nocopy=True)

        def grad_fn(q: Tensor, k: Tensor, v: Tensor, attn, key_valid, n_heads, scale, causal, drop_mask, attn_used,
                    dense_mask, grad):
            grad = grad if grad.is_contiguous() else grad.contiguous()
            dq, dk, dv = attention_backward(q.data, k.data, v.data, attn, key_valid, n_heads, scale, causal, grad,
                                            (q.requires_grad, k.requires_grad, v.requires_grad), drop_mask, attn_used,
                                            dense_mask)
            if dq is not None:
                q.apply_grad(dq)
            if dk is not None:
                k.apply_grad(dk)
            if dv is not None:
                v.apply_grad(dv)

        self.grad_fn = grad_fn


class _HIPFusedAttentionTensor(Tensor):
    def __init__(self, data, args, op, device):
        super().__init__(data, args, op, device=device, _nocopy=True)
        ctx = data   # the closure owns the output ARRAY, not the Tensor (no tensor -> grad_fn -> tensor cycle)

        def grad_fn(q: Tensor, k: Tensor, v: Tensor, lse, key_valid, n_heads, scale, causal, opts, grad):
            grad = grad if grad.is_contiguous() else grad.contiguous()
            dq, dk, dv = fused_attention_backward(q.data, k.data, v.data, key_valid, ctx, lse, n_heads, scale,
                                                  causal, grad, opts=opts)
            if q.requires_grad:
                q.apply_grad(dq)
            if k.requires_grad:
                k.apply_grad(dk)
            if v.requires_grad:
                v.apply_grad(dv)

        self.grad_fn = grad_fn


class _HIPQKVProjTensor(Tensor):
    """qkv[B,T,3D] = X [Wq;Wk;Wv]^T + [bq,bk,bv]: the three projections of self-attention as ONE GEMM (three
    generations of tiles instead of three single-generation launches), one K = 3D dX GEMM and one split-K dW GEMM."""

    def __init__(self, data, args, op, device):
        super().__init__(data, args, op, device=device, _nocopy=True)

        def grad_fn(X: Tensor, weights, biases, Wqkv, rows, D, grad):
            import torch
            grad = grad if grad.is_contiguous() else grad.contiguous()
            gW = _fused_grad_dest(weights, (3 * D, D))
            gb = _fused_grad_dest(biases, (3, D))
            grad_X = torch.empty_like(X.data) if X.requires_grad else None
            held = X.foldable_grad() if grad_X is not None else None
            hook = getattr(weights[0], "_grad_hook", None)
            if hook is not None and grad_X is not None:    # DP overlap: parameter gradients first (see linear.py)
                hip_linear_module_backward(X.data, Wqkv, grad, None, gW, gb, rows, D, 3 * D)
            else:
                hip_linear_module_backward(X.data, Wqkv, grad, grad_X, gW, gb, rows, D, 3 * D, grad_X_addend=held)
            for i, (w, b) in enumerate(zip(weights, biases)):
                _finish_param(w, gW[i * D:(i + 1) * D])
                _finish_param(b, gb[i:i + 1])
            if hook is not None and grad_X is not None:
                hip_linear_module_backward(X.data, Wqkv, grad, grad_X, None, None, rows, D, 3 * D, grad_X_addend=held)
            if grad_X is not None:
                if held is not None:
                    X.grad = grad_X
                else:
                    X.apply_grad(grad_X)

        self.grad_fn = grad_fn


def _adjacent(ts):
    """Back-to-back views of ONE storage (separately allocated tensors can be neighbours by accident)."""
    base = ts[0].untyped_storage().data_ptr()
    return all(t.is_contiguous() and t.untyped_storage().data_ptr() == base for t in ts) and all(
        ts[i + 1].data_ptr() == ts[i].data_ptr() + ts[i].numel() * 4 for i in range(len(ts) - 1))


def _fused_grad_dest(params, shape):
    """One buffer for the gradients of `params` (in order): their DP bucket slots when those are free and back-to-back
    (GradBucket keeps a `_bucket_group` adjacent), else a fresh allocation the per-parameter views point into."""
    import torch
    slots = [getattr(p, "_grad_slot", None) for p in params]
    if all(s is not None for s in slots) and all(p.grad is None for p in params) and _adjacent(slots):
        return torch.as_strided(slots[0], shape, (shape[1], 1))
    return torch.empty(shape, dtype=torch.float32, device=params[0].data.device)


class _HIPFusedSelfAttentionTensor(Tensor):
    """ctx = attention(q, k, v) with q, k, v the three column blocks of one [B,T,3D] projection tensor."""

    def __init__(self, data, args, op, device):
        super().__init__(data, args, op, device=device, _nocopy=True)
        ctx = data   # the closure owns the output ARRAY, not the Tensor (no tensor -> grad_fn -> tensor cycle)

        def grad_fn(qkv: Tensor, lse, key_valid, n_heads, scale, causal, opts, grad):
            import torch
            grad = grad if grad.is_contiguous() else grad.contiguous()
            D = ctx.shape[-1]
            x = qkv.data
            dqkv = torch.empty_like(x)
            fused_attention_backward(x[..., 0:D], x[..., D:2 * D], x[..., 2 * D:], key_valid, ctx, lse, n_heads,
                                     scale, causal, grad, out=(dqkv[..., 0:D], dqkv[..., D:2 * D], dqkv[..., 2 * D:]),
                                     opts=opts)
            qkv.apply_grad(dqkv)

        self.grad_fn = grad_fn


class HIPMultiHeadAttention(Module):
    def __init__(self, d_model, n_heads, dropout=0.0, device="cuda"):
        super().__init__()
        if d_model % n_heads != 0:
            raise ValueError("d_model must be divisible by n_heads")
        self.d_model, self.n_heads = d_model, n_heads
        self.scale = math.sqrt(d_model)  # the notebook divides by sqrt(d_model), not sqrt(depth)
        self.dropout = HIPDropout(dropout)
        self.depth = d_model // n_heads
        self.wq = HIPLinear(d_model, d_model, device=device)
        self.wk = HIPLinear(d_model, d_model, device=device)
        self.wv = HIPLinear(d_model, d_model, device=device)
        self.fc = HIPLinear(d_model, d_model, device=device)
        self.fuse_qkv = True     # self-attention + need_weights=False + a fused head dim: one GEMM for the q|k|v projections
        self.dropout_seed_dev = None     # optional device int32 tensor added to the dropout seed inside the kernels (set it to
        #                                  a per-step counter when the step is replayed from a hipGraph)
        self._seed_base = (next(_SEEDS) * 0x9E3779B9) & 0x7FFFFFFF
        self._calls = 0
        if self.depth in FUSED_HEAD_DIMS and device == "cuda":
            self._pack_qkv()     # now, so a GradBucket built before the first forward already sees the slot groups

    def _pack_qkv(self):
        """Make wq/wk/wv weights (and biases) three views of one [3D,D] ([3,D]) buffer, so the fused projection reads
        them as one operand.  Values, Parameter objects, their order and state_dict are unchanged; re-packed if someone
        re-bound .data (e.g. Module.to)."""
        import torch
        D = self.d_model
        lins = (self.wq, self.wk, self.wv)
        ws, bs = [lin.weight for lin in lins], [lin.bias for lin in lins]
        if not _adjacent([w.data for w in ws]):
            buf = torch.empty((3 * D, D), dtype=torch.float32, device=ws[0].data.device)
            for i, w in enumerate(ws):
                buf[i * D:(i + 1) * D].copy_(w.data)
                w.data = buf[i * D:(i + 1) * D]
        if not _adjacent([b.data for b in bs]):
            buf = torch.empty((3, D), dtype=torch.float32, device=bs[0].data.device)
            for i, b in enumerate(bs):
                buf[i:i + 1].copy_(b.data.reshape(1, D))
                b.data = buf[i:i + 1]
        ws[0]._bucket_group, bs[0]._bucket_group = ws, bs      # GradBucket keeps their slots back-to-back
        return ws, bs, torch.as_strided(ws[0].data, (3 * D, D), (D, 1)), torch.as_strided(bs[0].data, (1, 3 * D), (3 * D, 1))

    def _forward_fused_qkv(self, x: Tensor, key_valid, causal, residual, opts):
        import torch
        D = self.d_model
        ws, bs, Wqkv, bqkv = self._pack_qkv()
        rows = int(np.prod(x.shape[:-1]))
        qkv = torch.empty(tuple(x.shape[:-1]) + (3 * D,), dtype=torch.float32, device=x.data.device)
        hip_linear_module_forward(x.data, Wqkv, bqkv, qkv, rows, D, 3 * D)
        qkv_t = _HIPQKVProjTensor(qkv, (x, ws, bs, Wqkv, rows, D), "qkv_proj", device="cuda")
        q3 = qkv.reshape(-1, x.shape[-2], 3 * D) if qkv.dim() != 3 else qkv
        ctx, lse = fused_attention_forward(q3[..., 0:D], q3[..., D:2 * D], q3[..., 2 * D:], key_valid, self.n_heads,
                                           self.scale, causal, opts)
        ctx_t = _HIPFusedSelfAttentionTensor(ctx, (qkv_t, lse, key_valid, self.n_heads, self.scale, causal, opts),
                                             "fused_self_attention", device="cuda")
        return self.fc(ctx_t, residual=residual)

    def _fused_options(self, mask, drop_mask, dropping):
        """FusedAttentionOptions for this call (None when neither a dense mask nor dropout is involved)."""
        if mask is None and not dropping:
            return None
        opts = FusedAttentionOptions()
        if mask is not None:
            m = mask.data if isinstance(mask, Tensor) else mask
            opts.mask_bits, opts.mask_bitsT, opts.row_any = pack_attention_mask(m)
        if drop_mask is not None:
            opts.dropout_mask = drop_mask if drop_mask.is_contiguous() else drop_mask.contiguous()
        elif dropping:
            check_capture_seed(self.dropout_seed_dev, "HIPMultiHeadAttention")
            self._calls += 1
            opts.dropout_p, opts.seed, opts.seed_dev = (self.dropout.p, (self._seed_base + self._calls + process_dropout_seed()) & 0xFFFFFFFF,
                                                        self.dropout_seed_dev)
        return opts

    def forward(self, q: Tensor, k: Tensor, v: Tensor, key_valid=None, causal=True, need_weights=True, residual=None,
                drop_mask=None, mask=None):
        """key_valid: int32 device array [B,Tk] (1 = real token, 0 = padding) or None.  The notebook's dense
        mask get_pad_mask(x) & get_sub_mask(x) (cell 7) is exactly (key_valid, causal=True); an arbitrary dense
        `mask` ([B,Tq,Tk] or [B,1,Tq,Tk], non-zero = visible -- what the notebook's forward takes) is accepted too
        and then replaces (key_valid, causal).

        drop_mask ([B,H,Tq,Tk], entries 0 or 1/(1-p)) injects the attention-dropout mask (parity tests); with
        dropout p > 0 in training mode the fused kernels draw it from a hash of (seed, b, h, q, k).

        need_weights=False (training steps that never look at the attention map) takes the fused flash-style
        kernels when head_dim is 32, 64 or 128 and returns (out, None): scores/attn/dattn are never written to HBM.
        residual (extension): out = residual + fc(ctx), folded into the output projection's epilogue."""
        dropping = drop_mask is not None or (self.dropout.p != 0 and self.dropout.training)
        fusable = not need_weights and self.depth in FUSED_HEAD_DIMS
        if fusable:
            opts = self._fused_options(mask, drop_mask, dropping)
            if (q is k and k is v and self.fuse_qkv and self.wq.bias is not None and q.dtype == "float32"
                    and q.data.is_contiguous()):
                return self._forward_fused_qkv(q, key_valid, causal, residual, opts), None
            qp, kp, vp = self.wq(q), self.wk(k), self.wv(v)
            ctx, lse = fused_attention_forward(qp.data, kp.data, vp.data, key_valid, self.n_heads, self.scale, causal, opts)
            ctx_t = _HIPFusedAttentionTensor(ctx, (qp, kp, vp, lse, key_valid, self.n_heads, self.scale, causal, opts),
                                             "fused_attention", device="cuda")
            return self.fc(ctx_t, residual=residual), None
        dense = None
        if mask is not None:          # the dense mask replaces (key_valid, causal), as in the fused path
            import torch
            dense = mask.data if isinstance(mask, Tensor) else mask
            dense = (dense[:, 0] if dense.dim() == 4 else dense).to(torch.int32).contiguous()
            key_valid, causal = None, False
        qp, kp, vp = self.wq(q), self.wk(k), self.wv(v)
        if dropping and drop_mask is None:
            # attention dropout (cell 2: self.dropout(softmax(scores))): the multipliers of the library's counter hash of
            # (seed [+ device step word], b, h, q, k) -- the very ones the fused kernels would draw for this call
            import torch
            shape = (qp.shape[0], self.n_heads, qp.shape[1], kp.shape[1])
            drop_mask = torch.empty(shape, dtype=torch.float32, device=qp.data.device)
            check_capture_seed(self.dropout_seed_dev, "HIPMultiHeadAttention")
            self._calls += 1
            call_hip_function("nnhipAttentionDropoutMaskEx", drop_mask, *shape, float(self.dropout.p),
                              (self._seed_base + self._calls + process_dropout_seed()) & 0xFFFFFFFF, self.dropout_seed_dev,
                              get_current_stream_ptr())
        ctx, attn, used = attention_forward(qp.data, kp.data, vp.data, key_valid, self.n_heads, self.scale, causal, drop_mask,
                                            dense)
        ctx_t = _HIPAttentionTensor(ctx, (qp, kp, vp, attn, key_valid, self.n_heads, self.scale, causal, drop_mask,
                                          used if drop_mask is not None else None, dense), "attention", device="cuda")
        return self.fc(ctx_t, residual=residual), used
