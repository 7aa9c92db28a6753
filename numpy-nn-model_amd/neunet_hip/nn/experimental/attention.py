"""HIPMultiHeadAttention -- the attention block of examples/gpt.ipynb cell 2 on the strided-batched MFMA GEMM
(batched Tensor.matmul neunet/autograd.py:192-230 + where-mask + Softmax(-1)).

The reference reshapes/transposes q,k,v to (B,H,T,dh) on the host and materialises every intermediate.  Here
the three projections stay in their [B,T,H*dh] buffers: nnhipGemmF32Ex addresses head h of batch b through a
two-level batch stride (T*D, dh) with row stride D, so QK^T, attn*V and all four backward GEMMs read and write
that layout directly -- no transpose or .contiguous() copies; 1/sqrt(d_model) and the pad/causal mask are
fused into the softmax kernel.  Same parameters, same order (wq, wk, wv, fc) and the same math as the notebook.
"""
import math

import numpy as np

from ...autograd import Tensor
from ..modules import Module
from .embedding import HIPDropout
from .linear import HIPLinear
from .utils import call_hip_function, get_current_stream_ptr


def _gemm_ex(A, B, C, M, N, K, lda, ldb, ldc, akm, bkm, b1, sA1, sB1, sC1, b2, sA2, sB2, sC2, alpha=1.0):
    call_hip_function("nnhipGemmF32Ex", A, B, C, None, M, N, K, lda, ldb, ldc, akm, bkm, b1, sA1, sB1, sC1,
                      b2, sA2, sB2, sC2, float(alpha), get_current_stream_ptr())


def attention_forward(q, k, v, key_valid, n_heads, scale, causal):
    """q,k,v: device arrays [B,T,D] (D = H*dh).  Returns (ctx [B,Tq,D], attn [B,H,Tq,Tk])."""
    import torch
    B, Tq, D = q.shape
    Tk = k.shape[1]
    H, dh = n_heads, D // n_heads
    scores = torch.empty((B, H, Tq, Tk), dtype=torch.float32, device=q.device)
    # scores[b,h] = q[b,:,h,:] (Tq x dh, k-major, lda=D)  x  k[b,:,h,:]^T (B operand k-major, ldb=D)
    _gemm_ex(q, k, scores, Tq, Tk, dh, D, D, Tk, 1, 1, B, Tq * D, Tk * D, H * Tq * Tk, H, dh, dh, Tq * Tk)
    attn = scores  # softmax in place over the scores buffer
    call_hip_function("nnhipMaskedSoftmaxForward", attn, scores, key_valid, B, H, Tq, Tk, 1.0 / scale,
                      int(causal), get_current_stream_ptr())
    ctx = torch.empty((B, Tq, D), dtype=torch.float32, device=q.device)
    # ctx[b,:,h,:] = attn[b,h] (Tq x Tk, k-major) x v[b,:,h,:] (Tk x dh, outer-major, ldb=D) -> written in [B,T,D]
    _gemm_ex(attn, v, ctx, Tq, dh, Tk, Tk, D, D, 1, 0, B, H * Tq * Tk, Tk * D, Tq * D, H, Tq * Tk, dh, dh)
    return ctx, attn


def fused_attention_forward(q, k, v, key_valid, n_heads, scale, causal):
    """Flash-style forward (nnhipAttentionForward, head_dim 64): returns (ctx [B,Tq,D], lse [B,H,Tq,2]); the score
    matrix is never materialised."""
    import torch
    B, Tq, D = q.shape
    Tk = k.shape[1]
    ctx = torch.empty((B, Tq, D), dtype=torch.float32, device=q.device)
    lse = torch.empty((B, n_heads, Tq, 2), dtype=torch.float32, device=q.device)   # (row max, log row sum)
    call_hip_function("nnhipAttentionForward", q, k, v, key_valid, ctx, lse, B, n_heads, Tq, Tk, D // n_heads,
                      1.0 / scale, int(causal), get_current_stream_ptr())
    return ctx, lse


def fused_attention_backward(q, k, v, key_valid, ctx, lse, n_heads, scale, causal, dctx):
    """Flash-style backward (nnhipAttentionBackward): (dq, dk, dv) from the saved ctx and row log-sum-exp."""
    import torch
    B, Tq, D = q.shape
    Tk = k.shape[1]
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    call_hip_function("nnhipAttentionBackward", q, k, v, key_valid, ctx, dctx, lse, dq, dk, dv, B, n_heads, Tq, Tk,
                      D // n_heads, 1.0 / scale, int(causal), get_current_stream_ptr())
    return dq, dk, dv


def attention_backward(q, k, v, attn, key_valid, n_heads, scale, causal, dctx, need=(True, True, True)):
    """Returns (dq, dk, dv) in the [B,T,D] layout of the projections."""
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
        _gemm_ex(attn, dctx, dv, Tk, dh, Tq, Tk, D, D, 0, 0, B, H * Tq * Tk, Tq * D, Tk * D, H, Tq * Tk, dh, dh)
    # dscores (in place over dattn) = where(mask, 0, softmax_bwd(dattn, attn)) / scale
    call_hip_function("nnhipMaskedSoftmaxBackward", dattn, dattn, attn, key_valid, B, H, Tq, Tk, 1.0 / scale,
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

        def grad_fn(q: Tensor, k: Tensor, v: Tensor, attn, key_valid, n_heads, scale, causal, grad):
            grad = grad if grad.is_contiguous() else grad.contiguous()
            dq, dk, dv = attention_backward(q.data, k.data, v.data, attn, key_valid, n_heads, scale, causal, grad,
                                            (q.requires_grad, k.requires_grad, v.requires_grad))
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

        def grad_fn(q: Tensor, k: Tensor, v: Tensor, lse, key_valid, n_heads, scale, causal, grad):
            grad = grad if grad.is_contiguous() else grad.contiguous()
            dq, dk, dv = fused_attention_backward(q.data, k.data, v.data, key_valid, self.data, lse, n_heads, scale,
                                                  causal, grad)
            if q.requires_grad:
                q.apply_grad(dq)
            if k.requires_grad:
                k.apply_grad(dk)
            if v.requires_grad:
                v.apply_grad(dv)

        self.grad_fn = grad_fn


FUSED_HEAD_DIM = 64


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

    def forward(self, q: Tensor, k: Tensor, v: Tensor, key_valid=None, causal=True, need_weights=True, residual=None):
        """key_valid: int32 device array [B,Tk] (1 = real token, 0 = padding) or None.  The notebook's dense
        mask get_pad_mask(x) & get_sub_mask(x) (cell 7) is exactly (key_valid, causal=True).

        need_weights=False (training steps that never look at the attention map) takes the fused flash-style
        kernels when head_dim == 64 and returns (out, None): scores/attn/dattn are never written to HBM.
        residual (extension): out = residual + fc(ctx), folded into the output projection's epilogue."""
        if self.dropout.p != 0 and self.dropout.training:
            raise NotImplementedError("attention dropout > 0 is not implemented on the HIP path yet")
        qp, kp, vp = self.wq(q), self.wk(k), self.wv(v)
        if not need_weights and self.depth == FUSED_HEAD_DIM:
            ctx, lse = fused_attention_forward(qp.data, kp.data, vp.data, key_valid, self.n_heads, self.scale, causal)
            ctx_t = _HIPFusedAttentionTensor(ctx, (qp, kp, vp, lse, key_valid, self.n_heads, self.scale, causal),
                                             "fused_attention", device="cuda")
            return self.fc(ctx_t, residual=residual), None
        ctx, attn = attention_forward(qp.data, kp.data, vp.data, key_valid, self.n_heads, self.scale, causal)
        ctx_t = _HIPAttentionTensor(ctx, (qp, kp, vp, attn, key_valid, self.n_heads, self.scale, causal),
                                    "attention", device="cuda")
        return self.fc(ctx_t, residual=residual), attn
