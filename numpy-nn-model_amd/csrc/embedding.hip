// embedding.hip -- Embedding gather and its last-write-wins gradient.
// CPU semantics: Embedding.forward = weight[ids] (neunet/nn/layers/embedding.py:71-72) through
// Tensor.__getitem__, whose backward ASSIGNS into zeros -- `_grad[index] = grad`
// (neunet/autograd.py:905-912): for repeated token ids only the LAST occurrence (C order) survives.
// That is reproduced exactly and deterministically here (atomicMax of positions, then one copy per
// vocabulary row); it is the reference's behaviour, not the mathematical gradient.
// The decoder's `emb * sqrt(d_model) + pe[:, :T]` (examples/gpt.ipynb cells 5-6) is fused into the gather.
#include "common.h"

namespace nnhip {

// out[p, :] = W[ids[p], :] * scale + (pe ? pe[(p % T), :] : 0)        one wave per row, float4 lanes
template <bool VEC>
__global__ __launch_bounds__(256) void embedding_fwd_kernel(float* __restrict__ out,
                                                            const float* __restrict__ W,
                                                            const int32_t* __restrict__ ids,
                                                            const float* __restrict__ pe, int64_t n,
                                                            int64_t dim, int64_t T, int64_t vocab, float scale) {
    const int64_t p = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (p >= n) return;
    int64_t id = ids[p];
    if (id < 0) id += vocab;                      // python negative indexing
    const bool ok = id >= 0 && id < vocab;        // out-of-range ids read as zero rows
    const float* src = W + id * dim;
    const float* per = pe ? pe + (p % T) * dim : nullptr;
    float* dst = out + p * dim;
    if constexpr (VEC) {
        for (int64_t c = lane * 4; c < dim; c += 256) {
            float4 v = ok ? *reinterpret_cast<const float4*>(src + c) : make_float4(0.f, 0.f, 0.f, 0.f);
            v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale;
            if (per) {
                const float4 q = *reinterpret_cast<const float4*>(per + c);
                v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
            }
            *reinterpret_cast<float4*>(dst + c) = v;
        }
    } else {
        for (int64_t c = lane; c < dim; c += 64) {
            float v = (ok ? src[c] : 0.f) * scale;
            if (per) v += per[c];
            dst[c] = v;
        }
    }
}

// (a plain kernel rather than hipMemsetAsync: memset nodes misbehave under hipGraph replay on ROCm 7.2)
__global__ __launch_bounds__(256) void fill_i32_kernel(int32_t* __restrict__ p, int64_t n, int32_t v) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = v;
}

// out[i] = ids[i] != value  (the notebook's get_pad_mask, examples/gpt.ipynb cell 7: (x != pad_idx).astype(int))
__global__ __launch_bounds__(256) void not_equal_i32_kernel(int32_t* __restrict__ out, const int32_t* __restrict__ ids,
                                                            int64_t n, int32_t value) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = ids[i] != value ? 1 : 0;
}

__global__ __launch_bounds__(256) void embedding_last_kernel(const int32_t* __restrict__ ids, int64_t n,
                                                             int64_t vocab, int32_t* __restrict__ last) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    int64_t id = ids[p];
    if (id < 0) id += vocab;
    if (id >= 0 && id < vocab) atomicMax(&last[id], (int32_t)p);
}

// dW[v, :] = last[v] >= 0 ? scale * g[last[v], :] : 0        one wave per vocabulary row
template <bool VEC>
__global__ __launch_bounds__(256) void embedding_bwd_kernel(float* __restrict__ dW,
                                                            const float* __restrict__ g,
                                                            const int32_t* __restrict__ last, int64_t vocab,
                                                            int64_t dim, float scale) {
    const int64_t v = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (v >= vocab) return;
    const int32_t src = last[v];
    float* dst = dW + v * dim;
    if constexpr (VEC) {
        for (int64_t c = lane * 4; c < dim; c += 256) {
            float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
            if (src >= 0) {
                q = *reinterpret_cast<const float4*>(g + (int64_t)src * dim + c);
                q.x *= scale; q.y *= scale; q.z *= scale; q.w *= scale;
            }
            *reinterpret_cast<float4*>(dst + c) = q;
        }
    } else {
        for (int64_t c = lane; c < dim; c += 64) dst[c] = src >= 0 ? g[(int64_t)src * dim + c] * scale : 0.f;
    }
}

// ---- argmax (neunet.argmax -> int32, neunet/__init__.py:132-139 = np.argmax): index of the FIRST maximum along one axis of an
// [outer, n, inner] view; a NaN counts as the maximum (NumPy: the first NaN wins).  Index results are bit-exact by construction:
// a pure comparison network, (value, index) pairs ordered by (value desc, index asc).
struct ArgBest {
    float v;
    int32_t i;
};
__device__ __forceinline__ bool arg_better(float v, int32_t i, float bv, int32_t bi) {
    const bool vn = v != v, bn = bv != bv;
    if (vn != bn) return vn;                     // a NaN beats any number
    if (vn || v == bv) return i < bi;            // two NaNs, or a tie: the earlier index
    return v > bv;
}
__device__ __forceinline__ ArgBest arg_wave_reduce(ArgBest b) {
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {
        const float ov = __shfl_xor(b.v, m);
        const int32_t oi = __shfl_xor(b.i, m);
        if (arg_better(ov, oi, b.v, b.i)) { b.v = ov; b.i = oi; }
    }
    return b;
}

// last-axis rows (inner == 1): blockDim = 256; ROWS_PER_BLOCK = 4 (one wave per row, n <= 4096) or 1 (the block strides the row).
// chunks > 1 (one long row, e.g. axis=None): block (row, c) scans its slice and writes a partial pair; arg_finish_kernel closes.
template <int RPB>
__global__ __launch_bounds__(256) void argmax_rows_kernel(const float* __restrict__ x, int32_t* __restrict__ out, int64_t rows,
                                                          int64_t n, int chunks, float* __restrict__ pv, int32_t* __restrict__ pi) {
    __shared__ float sv[4];
    __shared__ int32_t si[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    ArgBest b{-INFINITY, 0x7fffffff};
    if constexpr (RPB == 4) {
        const int64_t r = (int64_t)blockIdx.x * 4 + w;
        if (r >= rows) return;
        const float* p = x + r * n;
        for (int64_t i = lane; i < n; i += 64) {
            const float v = p[i];
            if (arg_better(v, (int32_t)i, b.v, b.i)) { b.v = v; b.i = (int32_t)i; }
        }
        b = arg_wave_reduce(b);
        if (lane == 0) out[r] = b.i;
    } else {
        const int64_t r = blockIdx.x / chunks;
        const int c = blockIdx.x % chunks;
        const int64_t per = (n + chunks - 1) / chunks, lo = c * per, hi = lo + per < n ? lo + per : n;
        const float* p = x + r * n;
        for (int64_t i = lo + threadIdx.x; i < hi; i += 256) {
            const float v = p[i];
            if (arg_better(v, (int32_t)i, b.v, b.i)) { b.v = v; b.i = (int32_t)i; }
        }
        b = arg_wave_reduce(b);
        if (lane == 0) { sv[w] = b.v; si[w] = b.i; }
        __syncthreads();
        if (threadIdx.x == 0) {
#pragma unroll
            for (int k = 1; k < 4; ++k)
                if (arg_better(sv[k], si[k], b.v, b.i)) { b.v = sv[k]; b.i = si[k]; }
            if (chunks == 1) out[r] = b.i;
            else { pv[blockIdx.x] = b.v; pi[blockIdx.x] = b.i; }
        }
    }
}
__global__ __launch_bounds__(64) void arg_finish_kernel(const float* __restrict__ pv, const int32_t* __restrict__ pi,
                                                        int32_t* __restrict__ out, int chunks) {
    const int64_t r = blockIdx.x;
    ArgBest b{-INFINITY, 0x7fffffff};
    for (int c = threadIdx.x; c < chunks; c += 64) {
        const float v = pv[r * chunks + c];
        const int32_t i = pi[r * chunks + c];
        if (arg_better(v, i, b.v, b.i)) { b.v = v; b.i = i; }
    }
    b = arg_wave_reduce(b);
    if (threadIdx.x == 0) out[r] = b.i;
}
// inner > 1: one thread per output element (o, j), walking n with stride `inner` (coalesced across j)
__global__ __launch_bounds__(256) void argmax_strided_kernel(const float* __restrict__ x, int32_t* __restrict__ out,
                                                             int64_t outer, int64_t n, int64_t inner) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= outer * inner) return;
    const int64_t o = t / inner, j = t - o * inner;
    const float* p = x + o * n * inner + j;
    ArgBest b{-INFINITY, 0x7fffffff};
    for (int64_t i = 0; i < n; ++i) {
        const float v = p[i * inner];
        if (arg_better(v, (int32_t)i, b.v, b.i)) { b.v = v; b.i = (int32_t)i; }
    }
    out[t] = b.i;
}

}  // namespace nnhip

using namespace nnhip;

extern "C" int nnhipEmbeddingForward(float* out, const float* weight, const int32_t* ids, const float* pe,
                                     int64_t n_ids, int64_t dim, int64_t seq_len, int64_t vocab, float scale,
                                     nnhipStream_t s) {
    NNHIP_CHECK_ARG(n_ids >= 0 && dim >= 0 && vocab > 0 && seq_len > 0, NNHIP_EINVAL, "nnhipEmbeddingForward: bad sizes");
    if (n_ids == 0 || dim == 0) return 0;
    NNHIP_CHECK_ARG(out && weight && ids, NNHIP_EINVAL, "nnhipEmbeddingForward: null pointer");
    const bool vec = dim % 4 == 0 && aligned16(out) && aligned16(weight) && (!pe || aligned16(pe));
    const unsigned grid = (unsigned)ceil_div(n_ids, 4);
    if (vec) hipLaunchKernelGGL(embedding_fwd_kernel<true>, dim3(grid), dim3(256), 0, (hipStream_t)s, out, weight, ids, pe, n_ids, dim, seq_len, vocab, scale);
    else hipLaunchKernelGGL(embedding_fwd_kernel<false>, dim3(grid), dim3(256), 0, (hipStream_t)s, out, weight, ids, pe, n_ids, dim, seq_len, vocab, scale);
    NNHIP_LAUNCH_CHECK("embedding_fwd_kernel");
    return 0;
}

extern "C" int nnhipEmbeddingBackward(float* dW, const float* grad_out, const int32_t* ids, int64_t n_ids,
                                      int64_t dim, int64_t vocab, float scale, nnhipStream_t s) {
    NNHIP_CHECK_ARG(n_ids >= 0 && dim >= 0 && vocab > 0, NNHIP_EINVAL, "nnhipEmbeddingBackward: bad sizes");
    NNHIP_CHECK_ARG(n_ids < ((int64_t)1 << 31), NNHIP_EINVAL, "nnhipEmbeddingBackward: too many ids");
    if (dim == 0) return 0;
    NNHIP_CHECK_ARG(dW && (n_ids == 0 || (grad_out && ids)), NNHIP_EINVAL, "nnhipEmbeddingBackward: null pointer");
    hipStream_t st = (hipStream_t)s;
    int32_t* last = static_cast<int32_t*>(workspace((size_t)vocab * sizeof(int32_t)));
    NNHIP_CHECK_ARG(last != nullptr, NNHIP_ENOMEM, "nnhipEmbeddingBackward: workspace allocation failed");
    hipLaunchKernelGGL(fill_i32_kernel, dim3((unsigned)ceil_div(vocab, 256)), dim3(256), 0, st, last, vocab, -1);
    NNHIP_LAUNCH_CHECK("fill_i32_kernel");
    if (n_ids > 0) {
        hipLaunchKernelGGL(embedding_last_kernel, dim3((unsigned)ceil_div(n_ids, 256)), dim3(256), 0, st, ids, n_ids, vocab, last);
        NNHIP_LAUNCH_CHECK("embedding_last_kernel");
    }
    const bool vec = dim % 4 == 0 && aligned16(dW) && aligned16(grad_out);
    const unsigned grid = (unsigned)ceil_div(vocab, 4);
    if (vec) hipLaunchKernelGGL(embedding_bwd_kernel<true>, dim3(grid), dim3(256), 0, st, dW, grad_out, last, vocab, dim, scale);
    else hipLaunchKernelGGL(embedding_bwd_kernel<false>, dim3(grid), dim3(256), 0, st, dW, grad_out, last, vocab, dim, scale);
    NNHIP_LAUNCH_CHECK("embedding_bwd_kernel");
    return 0;
}

extern "C" int nnhipNotEqualInt32(int32_t* out, const int32_t* ids, int64_t n, int32_t value, nnhipStream_t s) {
    NNHIP_CHECK_ARG(n >= 0, NNHIP_EINVAL, "nnhipNotEqualInt32: negative size");
    if (n == 0) return 0;
    NNHIP_CHECK_ARG(out && ids, NNHIP_EINVAL, "nnhipNotEqualInt32: null pointer");
    hipLaunchKernelGGL(not_equal_i32_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, (hipStream_t)s, out, ids, n, value);
    NNHIP_LAUNCH_CHECK("not_equal_i32_kernel");
    return 0;
}

extern "C" int nnhipArgmaxF32(int32_t* out, const float* x, int64_t outer, int64_t n, int64_t inner, nnhipStream_t s) {
    NNHIP_CHECK_ARG(outer >= 0 && n >= 0 && inner >= 0, NNHIP_EINVAL, "nnhipArgmaxF32: negative size");
    NNHIP_CHECK_ARG(n > 0 || outer * inner == 0, NNHIP_EINVAL, "nnhipArgmaxF32: attempt to get argmax of an empty sequence");
    NNHIP_CHECK_ARG(n < ((int64_t)1 << 31), NNHIP_EINVAL, "nnhipArgmaxF32: axis longer than int32 indices can address");
    if (outer * inner == 0) return 0;
    NNHIP_CHECK_ARG(out && x, NNHIP_EINVAL, "nnhipArgmaxF32: null pointer");
    hipStream_t st = (hipStream_t)s;
    if (inner > 1) {
        hipLaunchKernelGGL(argmax_strided_kernel, dim3((unsigned)ceil_div(outer * inner, 256)), dim3(256), 0, st, x, out, outer, n, inner);
        NNHIP_LAUNCH_CHECK("argmax_strided_kernel");
        return 0;
    }
    if (n <= 4096 && outer >= 4) {
        hipLaunchKernelGGL(argmax_rows_kernel<4>, dim3((unsigned)ceil_div(outer, 4)), dim3(256), 0, st, x, out, outer, n, 1, nullptr, nullptr);
        NNHIP_LAUNCH_CHECK("argmax_rows_kernel");
        return 0;
    }
    // few long rows: cut each into chunks so that ~1024 blocks exist
    int chunks = 1;
    if (outer < 256 && n >= (1 << 16)) {
        chunks = (int)(1024 / outer);
        const int64_t cap = n / 4096;
        if (chunks > cap) chunks = (int)cap;
        if (chunks < 1) chunks = 1;
    }
    NNHIP_CHECK_ARG(outer * chunks < ((int64_t)1 << 31), NNHIP_EINVAL, "nnhipArgmaxF32: too many rows");
    float* pv = nullptr;
    int32_t* pi = nullptr;
    if (chunks > 1) {
        void* ws = workspace((size_t)outer * chunks * 8);
        NNHIP_CHECK_ARG(ws != nullptr, NNHIP_ENOMEM, "nnhipArgmaxF32: workspace allocation failed");
        pv = static_cast<float*>(ws);
        pi = reinterpret_cast<int32_t*>(pv + outer * chunks);
    }
    hipLaunchKernelGGL(argmax_rows_kernel<1>, dim3((unsigned)(outer * chunks)), dim3(256), 0, st, x, out, outer, n, chunks, pv, pi);
    NNHIP_LAUNCH_CHECK("argmax_rows_kernel");
    if (chunks > 1) {
        hipLaunchKernelGGL(arg_finish_kernel, dim3((unsigned)outer), dim3(64), 0, st, pv, pi, out, chunks);
        NNHIP_LAUNCH_CHECK("arg_finish_kernel");
    }
    return 0;
}
