"""GPT-tiny on the HIP path -- user-level model code with the module graph of the reference's
examples/gpt.ipynb (cells 2-7, 11-12): Embedding*sqrt(d) + sinusoidal PE -> N x [RMSNorm -> masked
multi-head self-attention -> residual -> RMSNorm -> Linear -> Swish -> Linear -> residual] -> Linear ->
CrossEntropy(ignore_index=PAD) -> Adam.  Same attribute names / creation order, so Module.parameters()
lists parameters in the notebook's order (including the never-called cross_attn of every DecoderLayer,
whose 8 parameters never receive a gradient -- SURVEY 3.4).

Used by bench.py (--workload c4) and the tests; BASELINE C4 = d_model 512, 6 layers, 8 heads, d_ff 2048,
vocab 15000, batch 64 x seq 256.
"""
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "numpy-nn-model_amd"))
import neunet_hip  # noqa: E402,F401
import neunet_hip.nn as nn  # noqa: E402
from neunet_hip import Tensor  # noqa: E402


class PositionwiseFeedForward(nn.Module):
    def __init__(self, d_model, d_ff, dropout=0.0, fused=True):
        super().__init__()
        self.fused = fused
        if fused:   # fc_1 + Swish in one GEMM epilogue (fused Linear->Swish, a6); same parameters
            self.fc_1 = nn.LinearSwish(d_model, d_ff, swish_beta=1.0, save_preactivation=True)
        else:
            self.fc_1 = nn.Linear(d_model, d_ff)
        self.fc_2 = nn.Linear(d_ff, d_model)
        self.dropout = nn.Dropout(dropout)
        self.activation = nn.Swish()

    def forward(self, x, residual=None):
        x = self.fc_1(x)
        if not self.fused:
            x = self.activation(x)
        x = self.dropout(x)
        return self.fc_2(x) if residual is None else self.fc_2(x, residual=residual)


class DecoderLayer(nn.Module):
    def __init__(self, d_model, n_heads, d_ff, dropout=0.0, fused=True, fused_attention=True):
        super().__init__()
        self.need_weights = not fused_attention   # fused flash-style attention never materialises the map
        self.fused = fused
        self.self_attn = nn.MultiHeadAttention(d_model, n_heads, dropout)
        self.cross_attn = nn.MultiHeadAttention(d_model, n_heads, dropout)   # constructed, never called
        self.ffn = PositionwiseFeedForward(d_model, d_ff, dropout, fused)
        self.norm1 = nn.RMSNorm(d_model)
        self.norm2 = nn.RMSNorm(d_model)
        self.dropout = nn.Dropout(dropout)

    def forward(self, x, key_valid):
        nx1 = self.norm1(x)
        if self.fused and self.dropout.p == 0:
            # x + sublayer(norm(x)) with the add folded into the sublayer's last GEMM (dropout p = 0 is the identity)
            x, attn = self.self_attn(nx1, nx1, nx1, key_valid, causal=True, need_weights=self.need_weights, residual=x)
            x = self.ffn(self.norm2(x), residual=x)
            return x, attn
        _x, attn = self.self_attn(nx1, nx1, nx1, key_valid, causal=True, need_weights=self.need_weights)
        x = x + self.dropout(_x)
        nx2 = self.norm2(x)
        _x = self.ffn(nx2)
        x = x + self.dropout(_x)
        return x, attn


class Decoder(nn.Module):
    def __init__(self, tgt_vocab_size, d_model, n_heads, d_ff, n_layers, dropout=0.0, max_len=5000, fused=True,
                 fused_attention=True):
        super().__init__()
        self.token_embedding = nn.Embedding(tgt_vocab_size, d_model)
        self.position_embedding = nn.PositionalEncoding(d_model, max_len)
        self.layers = nn.ModuleList([DecoderLayer(d_model, n_heads, d_ff, dropout, fused, fused_attention) for _ in range(n_layers)])
        self.fc_out = nn.Linear(d_model, tgt_vocab_size)
        self.dropout = nn.Dropout(dropout)
        self.scale = math.sqrt(d_model)

    def forward(self, ids, key_valid):
        # emb * sqrt(d) + pe[:, :T] fused into the gather
        x = self.token_embedding(ids, scale=self.scale, pe=self.position_embedding.table)
        x = self.dropout(x)
        attn = None
        for layer in self.layers:
            x, attn = layer(x, key_valid)
        return self.fc_out(x), attn


class GPT(nn.Module):
    def __init__(self, decoder: Decoder, pad_idx: int):
        super().__init__()
        self.decoder = decoder
        self.pad_idx = pad_idx

    def forward(self, x):
        """x: host int array (batch, seq) or a device int32 Tensor.  The notebook's dense mask
        get_pad_mask(x) & get_sub_mask(x) is carried as (key_valid = x != pad, causal=True)."""
        import torch
        ids = x if isinstance(x, Tensor) else Tensor(np.asarray(x), dtype=np.int32, requires_grad=False, device="cuda")
        if ids.data.dtype == torch.int32 and ids.data.is_contiguous():
            from neunet_hip._lib import call_hip_function, get_current_stream_ptr
            key_valid = torch.empty_like(ids.data)
            call_hip_function("nnhipNotEqualInt32", key_valid, ids.data, ids.data.numel(), int(self.pad_idx), get_current_stream_ptr())
        else:
            key_valid = (ids.data != self.pad_idx).to(torch.int32)
        return self.decoder(ids, key_valid)


def build_gpt(vocab=15000, d_model=512, n_heads=8, d_ff=2048, n_layers=6, pad_idx=0, max_len=1024, fused=True,
              fused_attention=None, dropout=0.0):
    """fused: Linear->Swish epilogue fusion in the FFN.  fused_attention (default = fused): flash-style attention
    kernels (head_dim 64 only; other head sizes keep the GEMM + masked-softmax path); the model then returns
    attn=None, which the training loop of cell 12 never reads.  dropout > 0 (the notebook trains with 0.1, cell 11) draws
    its masks with the device RNG and keeps the un-fused residual / attention paths."""
    fused_attention = fused if fused_attention is None else fused_attention
    dec = Decoder(vocab, d_model, n_heads, d_ff, n_layers, dropout=dropout, max_len=max_len, fused=fused,
                  fused_attention=fused_attention)
    return GPT(dec, pad_idx)


def train_step(model, optimizer, loss_fn, batch_ids, bucket=None):
    """cell 12's loop body: forward on batch[:, :-1], CE against batch[:, 1:], backward, step, zero_grad."""
    output, _ = model.forward(batch_ids[:, :-1])
    output = output.reshape(output.shape[0] * output.shape[1], output.shape[2])
    targets = Tensor(np.ascontiguousarray(batch_ids[:, 1:]).reshape(-1), dtype=np.int32, requires_grad=False, device="cuda")
    loss = loss_fn(output, targets)
    loss.backward()
    if bucket is not None:
        bucket.all_reduce()
    optimizer.step()
    optimizer.zero_grad()
    return loss


def synthetic_batches(vocab, batch, seq, steps, seed=0):
    """A learnable toy language: x[t+1] = (5 x[t] + 3) mod (vocab - 1) + 1 from a random start token (0 = PAD is never
    produced; the last 10 % of some rows are padded, as real batches are)."""
    rng = np.random.default_rng(seed)
    for _ in range(steps):
        x = np.empty((batch, seq + 1), np.int32)
        x[:, 0] = rng.integers(1, vocab, batch)
        for t in range(seq):
            x[:, t + 1] = (5 * x[:, t] + 3) % (vocab - 1) + 1
        x[: max(1, batch // 8), -max(1, seq // 10):] = 0
        yield x


if __name__ == "__main__":
    import argparse
    from neunet_hip.optim import Adam
    ap = argparse.ArgumentParser(description="Train the notebook's GPT on a synthetic next-token task (HIP backend).")
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--vocab", type=int, default=512)
    ap.add_argument("--d-model", type=int, default=256)
    ap.add_argument("--heads", type=int, default=4)
    ap.add_argument("--layers", type=int, default=2)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--seq", type=int, default=64)
    ap.add_argument("--dropout", type=float, default=0.1)
    a = ap.parse_args()
    np.random.seed(0)
    model = build_gpt(a.vocab, a.d_model, a.heads, 4 * a.d_model, a.layers, pad_idx=0, max_len=a.seq + 1, dropout=a.dropout)
    opt = Adam(model.parameters(), lr=1e-3, betas=(0.9, 0.98), eps=1e-9)
    loss_fn = nn.CrossEntropyLoss(ignore_index=0)
    for i, batch in enumerate(synthetic_batches(a.vocab, a.batch, a.seq, a.steps)):
        loss = train_step(model, opt, loss_fn, batch)
        if i % 20 == 0 or i == a.steps - 1:
            print(f"step {i:4d}  loss {loss.item():.4f}", flush=True)
