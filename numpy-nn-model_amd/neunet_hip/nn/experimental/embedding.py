"""HIPEmbedding / HIPDropout / HIPPositionalEncoding (the callers either side of the hot path in
the GPT step, examples/gpt.ipynb cells 5-6).

Embedding: forward = weight[ids] (neunet/nn/layers/embedding.py:61-75); the gradient reproduces the
reference's assignment semantics -- for repeated ids the LAST occurrence wins (neunet/autograd.py:905-912).
`scale` and a positional-encoding table can be fused into the gather (cell 6: emb * sqrt(d) + pe[:, :T])."""
import itertools
import math

import numpy as np

from ...autograd import Tensor
from ..modules import Module
from ..parameter import Parameter
from .linear import _finish_param, _grad_out
from .utils import call_hip_function, get_current_stream_ptr


def hip_embedding_forward(out, weight, ids, pe, seq_len, scale):
    n_ids, dim = ids.numel(), weight.shape[1]
    call_hip_function("nnhipEmbeddingForward", out, weight, ids, pe, n_ids, dim, seq_len, weight.shape[0],
                      float(scale), get_current_stream_ptr())
    return out


def hip_embedding_backward(grad_weight, grad_out, ids, scale):
    call_hip_function("nnhipEmbeddingBackward", grad_weight, grad_out, ids, ids.numel(), grad_weight.shape[1],
                      grad_weight.shape[0], float(scale), get_current_stream_ptr())
    return grad_weight


class _HIPEmbeddingTensor(Tensor):
    def __init__(self, data, args, op, device):
        super().__init__(data, args, op, device=device, _nocopy=True)

        def grad_fn(weight: Tensor, ids, scale, grad):
            grad = grad if grad.is_contiguous() else grad.contiguous()
            grad_weight = _grad_out(weight, weight.data)
            hip_embedding_backward(grad_weight, grad, ids, scale)
            _finish_param(weight, grad_weight)

        self.grad_fn = grad_fn


class HIPEmbedding(Module):
    def __init__(self, num_embeddings: int, embedding_dim: int, device="cuda"):
        super().__init__()
        self.num_embeddings, self.embedding_dim = num_embeddings, embedding_dim
        self.weight = Parameter(Tensor(np.random.randn(num_embeddings, embedding_dim), dtype=np.float32))
        self.to(device)

    def forward(self, X: Tensor, scale: float = 1.0, pe=None) -> Tensor:
        """X: int32 ids of any shape (..., T).  Optional fused `* scale + pe[:T]` (pe: device array (max_len, dim))."""
        import torch
        if X.device != "cuda":
            raise NotImplementedError("HIPEmbedding runs on the HIP device only (no CPU fallback)")
        ids = X.data
        if ids.dtype != torch.int32:
            ids = ids.to(torch.int32)
        ids = ids.contiguous()
        T = ids.shape[-1] if ids.ndim else 1
        if pe is not None and pe.shape[0] < T:
            raise ValueError("positional table shorter than the sequence")
        out = torch.empty(tuple(ids.shape) + (self.embedding_dim,), dtype=torch.float32, device=ids.device)
        hip_embedding_forward(out, self.weight.data, ids, pe, max(T, 1), scale)
        return _HIPEmbeddingTensor(out, (self.weight, ids, scale), "embedding", device="cuda")


class HIPPositionalEncoding(Module):
    """examples/gpt.ipynb cell 5: sinusoidal table built on the host in fp32, resident on the device."""

    def __init__(self, d_model, max_len=5000, device="cuda"):
        super().__init__()
        import torch
        pe = np.zeros((max_len, d_model), dtype=np.float32)
        position = np.arange(0, max_len, dtype=np.float32)[:, None]
        div_term = np.exp(np.arange(0, d_model, 2, dtype=np.float32) * np.float32(-math.log(10000.0) / d_model)).astype(np.float32)
        pe[:, 0::2] = np.sin(position * div_term)
        pe[:, 1::2] = np.cos(position * div_term)
        self.table = torch.from_numpy(pe).to("cuda" if device == "cuda" else "cpu")

    def to(self, device):
        self.table = self.table.to("cuda" if device == "cuda" else "cpu")
        return self


class _HIPDropoutTensor(Tensor):
    def __init__(self, data, args, op, device):
        super().__init__(data, args, op, device=device, _nocopy=True)

        def grad_fn(X: Tensor, mask, grad):
            if mask is None:
                X.apply_grad(grad)
                return
            g = X.xp.empty_like(X.data)
            grad = grad if grad.is_contiguous() else grad.contiguous()
            if isinstance(mask, tuple):          # (p, seed, seed_dev): the hash mask is regenerated, never stored
                p_, seed, seed_dev = mask
                call_hip_function("nnhipDropout", g, grad, g.numel(), float(p_), int(seed), seed_dev, get_current_stream_ptr())
            else:
                call_hip_function("nnhipMul", g, grad, mask, g.numel(), get_current_stream_ptr())
            X.apply_grad(g)

        self.grad_fn = grad_fn


_DROPOUT_SEEDS = itertools.count(1)
_seed_cache = {}


def process_dropout_seed() -> int:
    """32-bit word mixed into every hash-dropout seed of this process: torch.initial_seed() -- what `torch.manual_seed(s)` set,
    read at call time so a seed set after the model was built still counts -- and the data-parallel rank, so that a run is
    reproducible under the user's seed and the ranks of a DP job drop DIFFERENT positions (advisor, round 3: the bare module
    counter gave every run and every rank the same masks).  The reference draws its masks from the host NumPy generator
    (neunet/nn/layers/dropout.py:17-37); drawing from it here would shift the layers' weight initialisation, which comes
    from the same generator, so the device RNG keys off torch's seed instead."""
    import os
    import torch
    rank = int(os.environ.get("RANK", "0"))
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            rank = dist.get_rank()
    except Exception:  # noqa: BLE001
        pass
    key = (torch.initial_seed(), rank)
    v = _seed_cache.get(key)
    if v is None:
        x = (key[0] ^ (key[0] >> 32)) & 0xFFFFFFFF
        x = (x * 0x85EBCA6B + 0x9E3779B9 * (rank + 1)) & 0xFFFFFFFF
        x ^= x >> 15
        v = _seed_cache[key] = (x * 0xC2B2AE35) & 0xFFFFFFFF
    return v


def check_capture_seed(seed_dev, what):
    """A hash-dropout launch recorded into a hipGraph without a device step word replays the SAME mask on every step."""
    import torch
    if seed_dev is None and torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
        raise RuntimeError(f"{what}: dropout inside a captured step needs a device step seed, or every replay draws the same mask "
                           "-- build the step as GraphedTrainStep(..., step_seed=attach_step_seed(model))")


class HIPDropout(Module):
    """neunet/nn/layers/dropout.py:17-37.  p == 0 or eval: identity (no kernel, no mask tensor).  p > 0 in
    training: out = X * mask, mask = Bernoulli(1-p)/(1-p) from nnhipDropout's counter hash of (seed, element index) --
    regenerated by the backward pass, never stored (the reference draws it with the host NumPy RNG, so streams cannot
    match; `forward(X, mask=...)` injects a mask for parity tests).  Every call draws a new seed; `seed_dev` (an optional
    device uint32 tensor, e.g. a per-step counter) is added to it inside the kernel, so a step replayed from a hipGraph
    gets a fresh mask per replay."""

    def __init__(self, p: float = 0.5):
        super().__init__()
        self.p = p
        self.scale = 1 / (1 - p) if p < 1 else 0.0
        self.seed_dev = None
        self._base = (next(_DROPOUT_SEEDS) * 0x9E3779B9) & 0x7FFFFFFF
        self._calls = 0

    def forward(self, X: Tensor, mask=None) -> Tensor:
        if mask is None and (not self.training or self.p == 0):
            return X
        out = X.xp.empty_like(X.data)
        xd = X.data if X.data.is_contiguous() else X.data.contiguous()
        if mask is None:
            check_capture_seed(self.seed_dev, "HIPDropout")
            self._calls += 1
            seed = (self._base + self._calls * 0x632BE5AB + process_dropout_seed()) & 0xFFFFFFFF
            call_hip_function("nnhipDropout", out, xd, out.numel(), float(self.p), seed, self.seed_dev, get_current_stream_ptr())
            return _HIPDropoutTensor(out, (X, (self.p, seed, self.seed_dev)), "dropout", device=X.device)
        call_hip_function("nnhipMul", out, xd, mask, out.numel(), get_current_stream_ptr())
        return _HIPDropoutTensor(out, (X, mask), "dropout", device=X.device)

    def train(self, mode=True):
        self.training = mode

    def eval(self):
        self.training = False
