// rowops.hip -- HBM-bound row kernels for gfx950: Softmax fwd/bwd, RMSNorm fwd/bwd (+dw/db),
// fused CrossEntropy fwd+bwd, column sums (Linear db).
//
// Shape of every row kernel: a row (the reduction axis) is owned by TPR threads -- one wave64
// (TPR=64, rows <= 1024 wide; a 256-thread block carries 4 rows) or a whole block (TPR=256 / 1024)
// -- and is read ONCE into registers (NV float4 per thread = 16-B coalesced loads), reduced with
// width-64 __shfl_xor (+ one LDS hop across waves), transformed and written once.  That is the
// algorithmic traffic of SURVEY 8(d): 8 B/elem forward, 12 B/elem backward, 8 B/elem fused CE.
// Rows wider than 16384 floats (or with a non-unit stride) take a looped fallback.
#include <math.h>

#include "common.h"

namespace nnhip {

// ---- row register tile -------------------------------------------------------------------------
// Thread t of a row (0 <= t < TPR) owns elements  c = 4*(t + TPR*v) + {0..3}, v < NV   (VEC), or
// c = t + TPR*e, e < 4*NV (scalar path for rows that are not 16-B aligned / cols % 4 != 0).
template <int TPR, int NV, bool VEC>
struct RowTile {
    static constexpr int NE = NV * 4;
    float x[NE];

    __device__ __forceinline__ static int64_t col(int t, int e) {
        if constexpr (VEC) return 4 * (int64_t)(t + TPR * (e >> 2)) + (e & 3);
        else return (int64_t)t + (int64_t)TPR * e;
    }
    __device__ __forceinline__ void load(const float* __restrict__ row, int64_t cols, int t, float fill) {
        if constexpr (VEC) {
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const int64_t c = 4 * (int64_t)(t + TPR * v);
                if (c < cols) {
                    const float4 q = *reinterpret_cast<const float4*>(row + c);
                    x[4 * v] = q.x; x[4 * v + 1] = q.y; x[4 * v + 2] = q.z; x[4 * v + 3] = q.w;
                } else {
                    x[4 * v] = x[4 * v + 1] = x[4 * v + 2] = x[4 * v + 3] = fill;
                }
            }
        } else {
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                const int64_t c = (int64_t)t + (int64_t)TPR * e;
                x[e] = c < cols ? row[c] : fill;
            }
        }
    }
    __device__ __forceinline__ void store(float* __restrict__ row, int64_t cols, int t) const {
        if constexpr (VEC) {
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const int64_t c = 4 * (int64_t)(t + TPR * v);
                if (c < cols)
                    *reinterpret_cast<float4*>(row + c) =
                        make_float4(x[4 * v], x[4 * v + 1], x[4 * v + 2], x[4 * v + 3]);
            }
        } else {
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                const int64_t c = (int64_t)t + (int64_t)TPR * e;
                if (c < cols) row[c] = x[e];
            }
        }
    }
};

template <int TPR>
__device__ __forceinline__ float row_sum(float v, float* red) {
    if constexpr (TPR == 64) return wave_sum(v);
    else return block_sum<TPR / 64>(v, red);
}
template <int TPR>
__device__ __forceinline__ float row_max(float v, float* red) {
    if constexpr (TPR == 64) return wave_max(v);
    else return block_max<TPR / 64>(v, red);
}

// block = max(TPR,256) threads; rows per block = blockDim/TPR
#define ROW_PROLOGUE(TPR)                                                     \
    __shared__ float red[16];                                                 \
    constexpr int RPB = (TPR >= 256) ? 1 : 256 / TPR;                         \
    const int t = threadIdx.x % TPR;                                          \
    const int64_t row = (int64_t)blockIdx.x * RPB + threadIdx.x / TPR;        \
    if (TPR == 64 && row >= rows) return; /* wave-uniform; no block barriers in the TPR=64 path */ \
    (void)red;

// =================================================================================================
// Softmax   (neunet/nn/activations.py:448-459 fwd, 437-446 bwd)
// =================================================================================================
template <int TPR, int NV, bool VEC>
__global__ __launch_bounds__((TPR >= 256) ? TPR : 256) void softmax_fwd_rows(
    float* __restrict__ out, const float* __restrict__ in, int64_t rows, int64_t cols) {
    ROW_PROLOGUE(TPR)
    RowTile<TPR, NV, VEC> r;
    r.load(in + row * cols, cols, t, -INFINITY);
    float m = r.x[0];
#pragma unroll
    for (int e = 1; e < r.NE; ++e) m = fmaxf(m, r.x[e]);
    m = row_max<TPR>(m, red);
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < r.NE; ++e) {
        r.x[e] = expf(r.x[e] - m);  // exp(-inf) = 0 for the padding lanes
        s += r.x[e];
    }
    s = row_sum<TPR>(s, red);
    const float inv = 1.0f / s;
#pragma unroll
    for (int e = 0; e < r.NE; ++e) r.x[e] *= inv;
    r.store(out + row * cols, cols, t);
}

template <int TPR, int NV, bool VEC>
__global__ __launch_bounds__((TPR >= 256) ? TPR : 256) void softmax_bwd_rows(
    float* __restrict__ dx, const float* __restrict__ dy, const float* __restrict__ y, int64_t rows,
    int64_t cols) {
    ROW_PROLOGUE(TPR)
    RowTile<TPR, NV, VEC> g, f;
    g.load(dy + row * cols, cols, t, 0.f);
    f.load(y + row * cols, cols, t, 0.f);
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < g.NE; ++e) s += g.x[e] * f.x[e];
    s = row_sum<TPR>(s, red);
#pragma unroll
    for (int e = 0; e < g.NE; ++e) g.x[e] = (g.x[e] - s) * f.x[e];
    g.store(dx + row * cols, cols, t);
}

// Attention-score softmax with the scale and the mask fused in (examples/gpt.ipynb cell 2:
// scores = QK^T / sqrt(d_model); scores = where(mask == 0, -1e9, scores); attn = Softmax(-1)(scores)).
// Row r <-> (b, h, i); column j is masked when key j of batch b is padding (key_valid[b,j] == 0) or, if
// `causal`, j > i + (cols - Tq).  Masked scores are REPLACED by -1e9 (not -inf), exactly like the reference.
template <int TPR, int NV, bool VEC>
__global__ __launch_bounds__((TPR >= 256) ? TPR : 256) void softmax_masked_fwd_rows(
    float* out, const float* in, const int32_t* __restrict__ key_valid,  // out may alias in
    int64_t rows, int64_t cols, int64_t HTq, int64_t Tq, float scale, int causal) {
    ROW_PROLOGUE(TPR)
    RowTile<TPR, NV, VEC> r;
    r.load(in + row * cols, cols, t, -INFINITY);
    const int64_t b = row / HTq;
    const int64_t i = row % Tq;
    const int64_t lim = causal ? i + (cols - Tq) : cols;  // last visible column
    const int32_t* kv = key_valid ? key_valid + b * cols : nullptr;
    float m = -INFINITY;
#pragma unroll
    for (int e = 0; e < r.NE; ++e) {
        const int64_t c = r.col(t, e);
        if (c < cols) {
            const bool masked = c > lim || (kv && kv[c] == 0);
            r.x[e] = masked ? -1e9f : r.x[e] * scale;
        }
        m = fmaxf(m, r.x[e]);
    }
    m = row_max<TPR>(m, red);
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < r.NE; ++e) {
        r.x[e] = expf(r.x[e] - m);
        s += r.x[e];
    }
    s = row_sum<TPR>(s, red);
    const float inv = 1.0f / s;
#pragma unroll
    for (int e = 0; e < r.NE; ++e) r.x[e] *= inv;
    r.store(out + row * cols, cols, t);
}

// d(raw scores) = where(mask, 0, (dy - sum(dy*y)) * y) * scale
template <int TPR, int NV, bool VEC>
__global__ __launch_bounds__((TPR >= 256) ? TPR : 256) void softmax_masked_bwd_rows(
    float* dx, const float* dy, const float* __restrict__ y,  // dx may alias dy
    const int32_t* __restrict__ key_valid, int64_t rows, int64_t cols, int64_t HTq, int64_t Tq,
    float scale, int causal) {
    ROW_PROLOGUE(TPR)
    RowTile<TPR, NV, VEC> g, f;
    g.load(dy + row * cols, cols, t, 0.f);
    f.load(y + row * cols, cols, t, 0.f);
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < g.NE; ++e) s += g.x[e] * f.x[e];
    s = row_sum<TPR>(s, red);
    const int64_t b = row / HTq;
    const int64_t i = row % Tq;
    const int64_t lim = causal ? i + (cols - Tq) : cols;
    const int32_t* kv = key_valid ? key_valid + b * cols : nullptr;
#pragma unroll
    for (int e = 0; e < g.NE; ++e) {
        const int64_t c = g.col(t, e);
        const bool masked = c < cols && (c > lim || (kv && kv[c] == 0));
        g.x[e] = masked ? 0.f : (g.x[e] - s) * f.x[e] * scale;
    }
    g.store(dx + row * cols, cols, t);
}

// Fallback: arbitrary slice length / stride.  One thread per slice when stride > 1 (adjacent slices
// are adjacent in memory -> coalesced across threads); one block per slice when stride == 1.
__global__ __launch_bounds__(256) void softmax_fwd_strided(float* __restrict__ out,
                                                           const float* __restrict__ in,
                                                           int64_t num_slices, int64_t n,
                                                           int64_t stride) {
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= num_slices) return;
    const int64_t base = (s / stride) * n * stride + (s % stride);
    float m = -INFINITY;
    for (int64_t i = 0; i < n; ++i) m = fmaxf(m, in[base + i * stride]);
    float d = 0.f;
    for (int64_t i = 0; i < n; ++i) d += expf(in[base + i * stride] - m);
    const float inv = 1.0f / d;
    for (int64_t i = 0; i < n; ++i) out[base + i * stride] = expf(in[base + i * stride] - m) * inv;
}
__global__ __launch_bounds__(256) void softmax_bwd_strided(float* __restrict__ dx,
                                                           const float* __restrict__ dy,
                                                           const float* __restrict__ y,
                                                           int64_t num_slices, int64_t n,
                                                           int64_t stride) {
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= num_slices) return;
    const int64_t base = (s / stride) * n * stride + (s % stride);
    float d = 0.f;
    for (int64_t i = 0; i < n; ++i) d += dy[base + i * stride] * y[base + i * stride];
    for (int64_t i = 0; i < n; ++i)
        dx[base + i * stride] = (dy[base + i * stride] - d) * y[base + i * stride];
}
__global__ __launch_bounds__(256) void softmax_fwd_looped(float* __restrict__ out,
                                                          const float* __restrict__ in, int64_t n) {
    __shared__ float red[4];
    const float* x = in + (int64_t)blockIdx.x * n;
    float* o = out + (int64_t)blockIdx.x * n;
    float m = -INFINITY;
    for (int64_t i = threadIdx.x; i < n; i += 256) m = fmaxf(m, x[i]);
    m = block_max<4>(m, red);
    float d = 0.f;
    for (int64_t i = threadIdx.x; i < n; i += 256) d += expf(x[i] - m);
    d = block_sum<4>(d, red);
    const float inv = 1.0f / d;
    for (int64_t i = threadIdx.x; i < n; i += 256) o[i] = expf(x[i] - m) * inv;
}
__global__ __launch_bounds__(256) void softmax_bwd_looped(float* __restrict__ dx,
                                                          const float* __restrict__ dy,
                                                          const float* __restrict__ y, int64_t n) {
    __shared__ float red[4];
    const int64_t b = (int64_t)blockIdx.x * n;
    float d = 0.f;
    for (int64_t i = threadIdx.x; i < n; i += 256) d += dy[b + i] * y[b + i];
    d = block_sum<4>(d, red);
    for (int64_t i = threadIdx.x; i < n; i += 256) dx[b + i] = (dy[b + i] - d) * y[b + i];
}

// =================================================================================================
// RMSNorm   (neunet/nn/layers/rmsnorm.py:84-94 fwd, 43-59 bwd)
// =================================================================================================
template <int TPR, int NV, bool VEC>
__global__ __launch_bounds__((TPR >= 256) ? TPR : 256) void rmsnorm_fwd_rows(
    const float* __restrict__ X, const float* __restrict__ w, const float* __restrict__ b,
    float* __restrict__ Y, float* __restrict__ Xstd, float* __restrict__ Xnorm, int64_t rows,
    int64_t cols, float eps) {
    ROW_PROLOGUE(TPR)
    RowTile<TPR, NV, VEC> r, wt;
    r.load(X + row * cols, cols, t, 0.f);
    wt.load(w, cols, t, 0.f);
    float ss = 0.f;
#pragma unroll
    for (int e = 0; e < r.NE; ++e) ss += r.x[e] * r.x[e];
    ss = row_sum<TPR>(ss, red);
    const float sd = sqrtf(ss / (float)cols + eps);
    const float inv = 1.0f / sd;
    if (t == 0) Xstd[row] = sd;
#pragma unroll
    for (int e = 0; e < r.NE; ++e) r.x[e] *= inv;
    if (Xnorm) r.store(Xnorm + row * cols, cols, t);
#pragma unroll
    for (int e = 0; e < r.NE; ++e) r.x[e] *= wt.x[e];
    if (b) {
        wt.load(b, cols, t, 0.f);
#pragma unroll
        for (int e = 0; e < r.NE; ++e) r.x[e] += wt.x[e];
    }
    r.store(Y + row * cols, cols, t);
}

// Backward: block b walks rows b*RPB+rslot + k*gridDim.x*RPB, TWO rows per iteration (both rows' loads
// are in flight together and their row reductions share one pair of barriers), keeps per-thread column
// partials of dw = sum dy*x/std and db = sum dy in registers (a thread always owns the same columns),
// and writes them once at the end to part[(b*RPB + rslot)][cols]; a column-sum pass finishes dw/db.
// dx needs one row reduction: S = sum(w dy x / std).
template <int TPR, int NV, bool VEC>
__global__ __launch_bounds__((TPR >= 256) ? TPR : 256) void rmsnorm_bwd_rows(
    const float* __restrict__ dY, const float* __restrict__ X, const float* __restrict__ w,
    const float* __restrict__ Xstd, float* __restrict__ dX, float* __restrict__ part_dw,
    float* __restrict__ part_db, int64_t rows, int64_t cols, const float* __restrict__ dXadd) {
    __shared__ float red[32];
    constexpr int RPB = (TPR >= 256) ? 1 : 256 / TPR;
    constexpr int NW = TPR / 64;
    const int t = threadIdx.x % TPR;
    const int rslot = threadIdx.x / TPR;
    RowTile<TPR, NV, VEC> wt, adw, adb;
    wt.load(w, cols, t, 0.f);
#pragma unroll
    for (int e = 0; e < wt.NE; ++e) adw.x[e] = adb.x[e] = 0.f;
    const float invN = 1.0f / (float)cols;
    const int64_t step = (int64_t)gridDim.x * RPB;
    // when TPR >= 256 every thread of the block runs the same trip count (block barriers inside)
    constexpr bool TWO = TPR < 1024;  // 1024-thread blocks have 128 VGPRs/lane: one row in flight, no spills
    if constexpr (TWO) {
        for (int64_t r0 = (int64_t)blockIdx.x * RPB + rslot; r0 < rows; r0 += 2 * step) {
            const int64_t r1 = r0 + step;
            const bool has1 = r1 < rows;
            RowTile<TPR, NV, VEC> x0, g0, x1, g1;
            x0.load(X + r0 * cols, cols, t, 0.f);
            g0.load(dY + r0 * cols, cols, t, 0.f);
            if (has1) {
                x1.load(X + r1 * cols, cols, t, 0.f);
                g1.load(dY + r1 * cols, cols, t, 0.f);
            } else {
#pragma unroll
                for (int e = 0; e < x1.NE; ++e) x1.x[e] = g1.x[e] = 0.f;
            }
            const float sd0 = Xstd[r0], sd1 = has1 ? Xstd[r1] : 1.f;
            const float i0 = 1.0f / sd0, i1 = 1.0f / sd1;
            float s0 = 0.f, s1 = 0.f;
#pragma unroll
            for (int e = 0; e < x0.NE; ++e) {
                adb.x[e] += g0.x[e] + g1.x[e];
                adw.x[e] += g0.x[e] * (x0.x[e] * i0) + g1.x[e] * (x1.x[e] * i1);
                g0.x[e] *= wt.x[e];  // dX_hat = w * dy
                g1.x[e] *= wt.x[e];
                s0 += g0.x[e] * x0.x[e] * i0;
                s1 += g1.x[e] * x1.x[e] * i1;
            }
            block_sum2<NW>(s0, s1, red);
            s0 *= invN;
            s1 *= invN;
            const float q0 = i0 * i0, q1 = i1 * i1;
#pragma unroll
            for (int e = 0; e < x0.NE; ++e) {
                g0.x[e] = (g0.x[e] * sd0 - x0.x[e] * s0) * q0;
                g1.x[e] = (g1.x[e] * sd1 - x1.x[e] * s1) * q1;
            }
            if (dXadd) {   // dX = rmsnorm gradient + an already accumulated gradient of X (x0/x1 are dead: reuse them)
                x0.load(dXadd + r0 * cols, cols, t, 0.f);
                if (has1) x1.load(dXadd + r1 * cols, cols, t, 0.f);
#pragma unroll
                for (int e = 0; e < x0.NE; ++e) { g0.x[e] += x0.x[e]; g1.x[e] += x1.x[e]; }
            }
            g0.store(dX + r0 * cols, cols, t);
            if (has1) g1.store(dX + r1 * cols, cols, t);
        }
    } else {
        for (int64_t r0 = (int64_t)blockIdx.x * RPB + rslot; r0 < rows; r0 += step) {
            RowTile<TPR, NV, VEC> x0, g0;
            x0.load(X + r0 * cols, cols, t, 0.f);
            g0.load(dY + r0 * cols, cols, t, 0.f);
            const float sd0 = Xstd[r0];
            const float i0 = 1.0f / sd0;
            float s0 = 0.f;
#pragma unroll
            for (int e = 0; e < x0.NE; ++e) {
                adb.x[e] += g0.x[e];
                adw.x[e] += g0.x[e] * (x0.x[e] * i0);
                g0.x[e] *= wt.x[e];
                s0 += g0.x[e] * x0.x[e] * i0;
            }
            s0 = block_sum<NW>(s0, red) * invN;
            const float q0 = i0 * i0;
#pragma unroll
            for (int e = 0; e < x0.NE; ++e) g0.x[e] = (g0.x[e] * sd0 - x0.x[e] * s0) * q0;
            if (dXadd) {
                x0.load(dXadd + r0 * cols, cols, t, 0.f);
#pragma unroll
                for (int e = 0; e < x0.NE; ++e) g0.x[e] += x0.x[e];
            }
            g0.store(dX + r0 * cols, cols, t);
        }
    }
    const int64_t prow = (int64_t)blockIdx.x * RPB + rslot;
    adw.store(part_dw + prow * cols, cols, t);
    if (part_db) adb.store(part_db + prow * cols, cols, t);
}

// =================================================================================================
// Fused CrossEntropy forward+backward
// (CPU semantics: LogSoftmax(axis=1) -> NLLLoss, neunet/nn/losses.py:59-126; fusion boundary of
//  cross_entropy.cu:18-229: one pass computes loss, lse and d(logits))
// =================================================================================================
template <int TPR, int NV, bool VEC>
__global__ __launch_bounds__((TPR >= 256) ? TPR : 256) void ce_fwd_bwd_rows(
    float* logits, float* dlogits, float* __restrict__ loss, float* __restrict__ lse_out,
    const int32_t* __restrict__ labels, int64_t ld, int32_t ignore_index, int64_t rows, int64_t cols,
    float scale_host, const int32_t* __restrict__ count_dev, int use_mean) {
    ROW_PROLOGUE(TPR)
    const int32_t label = labels[row];
    const bool valid = label != ignore_index && label >= 0 && label < cols;
    // read the label logit before anything is overwritten (in-place mode)
    const float xl = valid ? logits[row * ld + label] : 0.f;
    RowTile<TPR, NV, VEC> r;
    r.load(logits + row * ld, cols, t, -INFINITY);
    float m = r.x[0];
#pragma unroll
    for (int e = 1; e < r.NE; ++e) m = fmaxf(m, r.x[e]);
    m = row_max<TPR>(m, red);
    float d = 0.f;
#pragma unroll
    for (int e = 0; e < r.NE; ++e) {
        r.x[e] = expf(r.x[e] - m);   // keep exp(x - m): softmax = e / d, no second exp pass
        d += r.x[e];
    }
    d = row_sum<TPR>(d, red);
    const float lse = m + logf(d);
    float scale = scale_host;
    if (use_mean && count_dev) scale = 1.0f / (float)count_dev[0];
    if (t == 0) {
        lse_out[row] = lse;
        loss[row] = valid ? (lse - xl) : 0.f;
    }
    if (valid) {
        const float invd = 1.0f / d;
#pragma unroll
        for (int e = 0; e < r.NE; ++e)
            r.x[e] = (r.x[e] * invd - ((int64_t)label == r.col(t, e) ? 1.f : 0.f)) * scale;
    } else {
#pragma unroll
        for (int e = 0; e < r.NE; ++e) r.x[e] = 0.f;
    }
    r.store(dlogits + row * ld, cols, t);
}

// looped fallback (cols > 16384): 2 reads + 1 write
__global__ __launch_bounds__(1024) void ce_fwd_bwd_looped(float* logits, float* dlogits,
                                                          float* __restrict__ loss,
                                                          float* __restrict__ lse_out,
                                                          const int32_t* __restrict__ labels,
                                                          int64_t ld, int32_t ignore_index,
                                                          int64_t cols, float scale_host,
                                                          const int32_t* __restrict__ count_dev,
                                                          int use_mean) {
    __shared__ float red[16];
    const int64_t row = blockIdx.x;
    const int32_t label = labels[row];
    const bool valid = label != ignore_index && label >= 0 && label < cols;
    const float xl = valid ? logits[row * ld + label] : 0.f;
    const float* x = logits + row * ld;
    float m = -INFINITY;
    for (int64_t i = threadIdx.x; i < cols; i += 1024) m = fmaxf(m, x[i]);
    m = block_max<16>(m, red);
    float d = 0.f;
    for (int64_t i = threadIdx.x; i < cols; i += 1024) d += expf(x[i] - m);
    d = block_sum<16>(d, red);
    const float lse = m + logf(d);
    float scale = scale_host;
    if (use_mean && count_dev) scale = 1.0f / (float)count_dev[0];
    if (threadIdx.x == 0) {
        lse_out[row] = lse;
        loss[row] = valid ? (lse - xl) : 0.f;
    }
    __syncthreads();
    float* o = dlogits + row * ld;
    for (int64_t i = threadIdx.x; i < cols; i += 1024) {
        const float v = x[i];
        o[i] = valid ? (expf(v - lse) - (i == label ? 1.f : 0.f)) * scale : 0.f;
    }
}

__global__ __launch_bounds__(1024) void count_ne_kernel(const int32_t* __restrict__ labels, int64_t n,
                                                        int32_t ignore, int32_t* __restrict__ out) {
    __shared__ int ired[16];
    int ci = 0;
    for (int64_t i = threadIdx.x; i < n; i += 1024) ci += labels[i] != ignore ? 1 : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ci += __shfl_xor(ci, o, 64);
    if ((threadIdx.x & 63) == 0) ired[threadIdx.x >> 6] = ci;
    __syncthreads();
    if (threadIdx.x == 0) {
        int s = 0;
        for (int i = 0; i < 16; ++i) s += ired[i];
        out[0] = s;
    }
}

__global__ __launch_bounds__(1024) void reduce_loss_kernel(const float* __restrict__ loss, int64_t n,
                                                           int mean,
                                                           const int32_t* __restrict__ count_dev,
                                                           float* __restrict__ out) {
    __shared__ float red[16];
    float s = 0.f;
    for (int64_t i = threadIdx.x; i < n; i += 1024) s += loss[i];
    s = block_sum<16>(s, red);
    if (threadIdx.x == 0) out[0] = mean ? s / (float)count_dev[0] : s;
}

// Whole CrossEntropyLoss(mean|sum) of a SMALL problem in one launch (one 1024-thread block): count the non-ignored
// labels, per-row loss / lse / d(logits), reduce the loss.  Same arithmetic as count_ne + ce_fwd_bwd_rows + reduce_loss
// (the row sums run over a wave instead of a block, so the last bits of lse can differ by rounding); at MNIST-MLP scale
// (32 x 10) those were three ~4.6 us graph nodes.
__global__ __launch_bounds__(1024) void ce_small_kernel(const float* __restrict__ logits, float* __restrict__ dlogits,
                                                        float* __restrict__ loss_rows, float* __restrict__ lse_out,
                                                        const int32_t* __restrict__ labels, int64_t stride,
                                                        int32_t ignore, int64_t rows, int64_t cols, int mean,
                                                        float* __restrict__ loss_out, int32_t* __restrict__ count_out) {
    __shared__ int ired[16];
    __shared__ float fred[16];
    __shared__ int cnt_sh;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int ci = 0;
    for (int64_t i = threadIdx.x; i < rows; i += 1024) ci += labels[i] != ignore ? 1 : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ci += __shfl_xor(ci, o, 64);
    if (lane == 0) ired[wave] = ci;
    __syncthreads();
    if (threadIdx.x == 0) {
        int t = 0;
        for (int i = 0; i < 16; ++i) t += ired[i];
        cnt_sh = t;
        if (count_out) count_out[0] = t;
    }
    __syncthreads();
    const int count = cnt_sh;
    const float scale = mean ? (count > 0 ? 1.0f / (float)count : 0.0f) : 1.0f;
    float lsum = 0.f;
    for (int64_t r = wave; r < rows; r += 16) {
        const float* x = logits + r * stride;
        float* dx = dlogits + r * stride;
        const int32_t y = labels[r];
        float mx = -INFINITY;
        for (int64_t c = lane; c < cols; c += 64) mx = fmaxf(mx, x[c]);
        mx = wave_max(mx);
        float se = 0.f;
        for (int64_t c = lane; c < cols; c += 64) se += expf(x[c] - mx);
        se = wave_sum(se);
        const float lse = mx + logf(se);
        const bool live = y != ignore;
        const float xy = live ? x[y] : 0.f;
        const float inv = 1.0f / se;
        for (int64_t c = lane; c < cols; c += 64) {
            const float pr = expf(x[c] - mx) * inv;
            dx[c] = live ? (pr - (c == y ? 1.f : 0.f)) * scale : 0.f;
        }
        if (lane == 0) {
            const float l = live ? lse - xy : 0.f;
            loss_rows[r] = l;
            lse_out[r] = lse;
            lsum += l;
        }
    }
    if (lane == 0) fred[wave] = lsum;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int i = 0; i < 16; ++i) t += fred[i];
        loss_out[0] = mean ? t / (float)count : t;
    }
}

// =================================================================================================
// Column sums:  out[c] = sum_r X[r*ld + c]     (Linear db: neunet/nn/layers/linear.py:24; RMSNorm dw/db)
// Block = 4 waves; lane <-> one float4 column group (256 columns per block) or one column (scalar path,
// 64 columns per block); wave w sums rows w, w+4, ... of the block's row range (1 KiB coalesced per
// wave-load); the 4 waves meet in LDS.  grid (col_blocks, row_blocks); row_blocks > 1 writes partials
// [row_blocks][cols] that a second, single-row-block launch finishes.
// =================================================================================================
template <bool VEC>
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ X, int64_t rows,
                                                     int64_t cols, int64_t ld, int64_t rows_per_block,
                                                     float* __restrict__ out) {
    __shared__ float4 red[3][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t rbeg = (int64_t)blockIdx.y * rows_per_block;
    const int64_t rend = min(rows, rbeg + rows_per_block);
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    int64_t c;
    if constexpr (VEC) {
        c = ((int64_t)blockIdx.x * 64 + lane) * 4;
        if (c < cols) {
            int64_t r = rbeg + w;
            for (; r + 12 < rend; r += 16) {  // 4 loads in flight per lane
                const float4 v0 = *reinterpret_cast<const float4*>(X + r * ld + c);
                const float4 v1 = *reinterpret_cast<const float4*>(X + (r + 4) * ld + c);
                const float4 v2 = *reinterpret_cast<const float4*>(X + (r + 8) * ld + c);
                const float4 v3 = *reinterpret_cast<const float4*>(X + (r + 12) * ld + c);
                a.x += (v0.x + v1.x) + (v2.x + v3.x);
                a.y += (v0.y + v1.y) + (v2.y + v3.y);
                a.z += (v0.z + v1.z) + (v2.z + v3.z);
                a.w += (v0.w + v1.w) + (v2.w + v3.w);
            }
            for (; r < rend; r += 4) {
                const float4 v = *reinterpret_cast<const float4*>(X + r * ld + c);
                a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
            }
        }
    } else {
        c = (int64_t)blockIdx.x * 64 + lane;
        if (c < cols)
            for (int64_t r = rbeg + w; r < rend; r += 4) a.x += X[r * ld + c];
    }
    if (w > 0) red[w - 1][lane] = a;
    __syncthreads();
    if (w == 0 && c < cols) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const float4 v = red[i][lane];
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
        float* o = out + (int64_t)blockIdx.y * cols + c;
        if constexpr (VEC) *reinterpret_cast<float4*>(o) = a;
        else *o = a.x;
    }
}

// internal API ------------------------------------------------------------------------------------
struct ColsumPlan {
    bool vec;
    int64_t col_blocks, rows_per_block;
    int row_blocks;
    size_t scratch_floats;  // partials needed when row_blocks > 1
};

static ColsumPlan colsum_plan(const float* X, const float* out, int64_t rows, int64_t cols, int64_t ld) {
    ColsumPlan p;
    p.vec = aligned16(X) && aligned16(out) && (ld % 4 == 0) && (cols % 4 == 0);
    p.col_blocks = ceil_div(cols, p.vec ? 256 : 64);
    int64_t rb = 2048 / p.col_blocks;             // aim for ~2k blocks in stage 1
    if (rb > ceil_div(rows, 16)) rb = ceil_div(rows, 16);  // >= 16 rows per block
    if (rb > 512) rb = 512;                        // stage 2 sums <= 512 partial rows in one block row
    if (rb < 1) rb = 1;
    p.rows_per_block = ceil_div(rows > 0 ? rows : 1, rb);
    p.row_blocks = (int)ceil_div(rows > 0 ? rows : 1, p.rows_per_block);
    p.scratch_floats = p.row_blocks > 1 ? (size_t)p.row_blocks * cols : 0;
    return p;
}

// `scratch` must hold plan.scratch_floats floats (16-B aligned) when plan.row_blocks > 1.
static int colsum_run(const ColsumPlan& p, const float* X, int64_t rows, int64_t cols, int64_t ld, float* out,
                      float* scratch, hipStream_t st) {
    float* stage1 = p.row_blocks > 1 ? scratch : out;
    dim3 grid((unsigned)p.col_blocks, (unsigned)p.row_blocks);
    if (p.vec) hipLaunchKernelGGL(colsum_kernel<true>, grid, dim3(256), 0, st, X, rows, cols, ld, p.rows_per_block, stage1);
    else hipLaunchKernelGGL(colsum_kernel<false>, grid, dim3(256), 0, st, X, rows, cols, ld, p.rows_per_block, stage1);
    NNHIP_LAUNCH_CHECK("colsum_kernel");
    if (p.row_blocks > 1) {
        const bool v2 = p.vec && aligned16(stage1);
        dim3 g2((unsigned)ceil_div(cols, v2 ? 256 : 64), 1);
        if (v2) hipLaunchKernelGGL(colsum_kernel<true>, g2, dim3(256), 0, st, stage1, (int64_t)p.row_blocks, cols, cols, (int64_t)p.row_blocks, out);
        else hipLaunchKernelGGL(colsum_kernel<false>, g2, dim3(256), 0, st, stage1, (int64_t)p.row_blocks, cols, cols, (int64_t)p.row_blocks, out);
        NNHIP_LAUNCH_CHECK("colsum_kernel(stage2)");
    }
    return 0;
}

int colsum(const float* X, int64_t rows, int64_t cols, int64_t ld, float* out, hipStream_t st) {
    if (cols <= 0) return 0;
    const ColsumPlan p = colsum_plan(X, out, rows, cols, ld);
    float* scratch = nullptr;
    if (p.scratch_floats) {
        scratch = static_cast<float*>(workspace(p.scratch_floats * sizeof(float)));
        if (!scratch) { set_last_error("colsum workspace allocation failed"); return NNHIP_ENOMEM; }
    }
    return colsum_run(p, X, rows, cols, ld, out, scratch, st);
}

// Row-kernel dispatch: pick (TPR, NV) from the row width.
#define ROW_DISPATCH(KERNEL, cols, vec, rows, st, ...)                                              \
    do {                                                                                            \
        const int64_t _c = (cols);                                                                  \
        if (_c <= 256) { ROW_LAUNCH(KERNEL, 64, 1, vec, rows, st, __VA_ARGS__); }                   \
        else if (_c <= 1024) { ROW_LAUNCH(KERNEL, 64, 4, vec, rows, st, __VA_ARGS__); }             \
        else if (_c <= 4096) { ROW_LAUNCH(KERNEL, 256, 4, vec, rows, st, __VA_ARGS__); }            \
        else if (_c <= 8192) { ROW_LAUNCH(KERNEL, 256, 8, vec, rows, st, __VA_ARGS__); }            \
        else { ROW_LAUNCH(KERNEL, 1024, 4, vec, rows, st, __VA_ARGS__); }                           \
    } while (0)
#define ROW_LAUNCH(KERNEL, TPR, NV, vec, rows, st, ...)                                             \
    do {                                                                                            \
        constexpr int _rpb = (TPR >= 256) ? 1 : 256 / TPR;                                          \
        constexpr int _bs = (TPR >= 256) ? TPR : 256;                                               \
        const unsigned _g = (unsigned)ceil_div((rows), _rpb);                                       \
        if (vec) hipLaunchKernelGGL((KERNEL<TPR, NV, true>), dim3(_g), dim3(_bs), 0, st, __VA_ARGS__);  \
        else hipLaunchKernelGGL((KERNEL<TPR, NV, false>), dim3(_g), dim3(_bs), 0, st, __VA_ARGS__);     \
    } while (0)

constexpr int64_t kMaxRegRow = 16384;

}  // namespace nnhip

using namespace nnhip;

// ---- Softmax ------------------------------------------------------------------------------------
extern "C" int nnhipSoftmaxForward(float* out, const float* in, int64_t num_slices,
                                   int64_t slice_size, int64_t stride, nnhipStream_t s) {
    NNHIP_CHECK_ARG(num_slices >= 0 && slice_size >= 0, NNHIP_EINVAL, "nnhipSoftmaxForward: negative size");
    if (num_slices == 0 || slice_size == 0) return 0;
    NNHIP_CHECK_ARG(stride >= 1, NNHIP_EINVAL, "nnhipSoftmaxForward: stride must be >= 1");
    NNHIP_CHECK_ARG(out && in, NNHIP_EINVAL, "nnhipSoftmaxForward: null pointer");
    hipStream_t st = (hipStream_t)s;
    if (stride != 1) {
        hipLaunchKernelGGL(softmax_fwd_strided, dim3((unsigned)ceil_div(num_slices, 256)), dim3(256), 0, st,
                           out, in, num_slices, slice_size, stride);
    } else if (slice_size > kMaxRegRow) {
        hipLaunchKernelGGL(softmax_fwd_looped, dim3((unsigned)num_slices), dim3(256), 0, st, out, in, slice_size);
    } else {
        const bool vec = aligned16(out) && aligned16(in) && slice_size % 4 == 0;
        ROW_DISPATCH(softmax_fwd_rows, slice_size, vec, num_slices, st, out, in, num_slices, slice_size);
    }
    NNHIP_LAUNCH_CHECK("softmax_forward");
    return 0;
}

extern "C" int nnhipSoftmaxBackward(float* dX, const float* dY, const float* Y, int64_t num_slices,
                                    int64_t slice_size, int64_t stride, nnhipStream_t s) {
    NNHIP_CHECK_ARG(num_slices >= 0 && slice_size >= 0, NNHIP_EINVAL, "nnhipSoftmaxBackward: negative size");
    if (num_slices == 0 || slice_size == 0) return 0;
    NNHIP_CHECK_ARG(stride >= 1, NNHIP_EINVAL, "nnhipSoftmaxBackward: stride must be >= 1");
    NNHIP_CHECK_ARG(dX && dY && Y, NNHIP_EINVAL, "nnhipSoftmaxBackward: null pointer");
    hipStream_t st = (hipStream_t)s;
    if (stride != 1) {
        hipLaunchKernelGGL(softmax_bwd_strided, dim3((unsigned)ceil_div(num_slices, 256)), dim3(256), 0, st,
                           dX, dY, Y, num_slices, slice_size, stride);
    } else if (slice_size > kMaxRegRow) {
        hipLaunchKernelGGL(softmax_bwd_looped, dim3((unsigned)num_slices), dim3(256), 0, st, dX, dY, Y, slice_size);
    } else {
        const bool vec = aligned16(dX) && aligned16(dY) && aligned16(Y) && slice_size % 4 == 0;
        ROW_DISPATCH(softmax_bwd_rows, slice_size, vec, num_slices, st, dX, dY, Y, num_slices, slice_size);
    }
    NNHIP_LAUNCH_CHECK("softmax_backward");
    return 0;
}

// ---- RMSNorm ------------------------------------------------------------------------------------
extern "C" int nnhipRMSNormForward(const float* X, const float* weight, const float* bias, float* Y,
                                   float* X_std, float* X_norm, int64_t rows, int64_t cols, float eps,
                                   nnhipStream_t s) {
    NNHIP_CHECK_ARG(rows >= 0 && cols >= 0, NNHIP_EINVAL, "nnhipRMSNormForward: negative size");
    if (rows == 0 || cols == 0) return 0;
    NNHIP_CHECK_ARG(X && weight && Y && X_std, NNHIP_EINVAL, "nnhipRMSNormForward: null pointer");
    NNHIP_CHECK_ARG(cols <= kMaxRegRow, NNHIP_EINVAL,
                    "nnhipRMSNormForward: cols > 16384 not supported");
    hipStream_t st = (hipStream_t)s;
    const bool vec = aligned16(X) && aligned16(Y) && aligned16(weight) && (!bias || aligned16(bias)) &&
                     (!X_norm || aligned16(X_norm)) && cols % 4 == 0;
    ROW_DISPATCH(rmsnorm_fwd_rows, cols, vec, rows, st, X, weight, bias, Y, X_std, X_norm, rows, cols, eps);
    NNHIP_LAUNCH_CHECK("rmsnorm_forward");
    return 0;
}

extern "C" int nnhipRMSNormBackward(const float* dY, const float* X, const float* weight,
                                    const float* X_std, const float* X_norm_unused, float* dX,
                                    float* dW, float* db, int64_t rows, int64_t cols,
                                    nnhipStream_t s) {
    return nnhipRMSNormBackwardEx(dY, X, weight, X_std, X_norm_unused, nullptr, dX, dW, db, rows, cols, s);
}

extern "C" int nnhipRMSNormBackwardEx(const float* dY, const float* X, const float* weight,
                                      const float* X_std, const float* X_norm_unused, const float* dX_addend,
                                      float* dX, float* dW, float* db, int64_t rows, int64_t cols,
                                      nnhipStream_t s) {
    (void)X_norm_unused;
    NNHIP_CHECK_ARG(rows >= 0 && cols >= 0, NNHIP_EINVAL, "nnhipRMSNormBackward: negative size");
    if (cols == 0) return 0;
    NNHIP_CHECK_ARG(dY && X && weight && X_std && dX && dW, NNHIP_EINVAL,
                    "nnhipRMSNormBackward: null pointer");
    NNHIP_CHECK_ARG(cols <= kMaxRegRow, NNHIP_EINVAL,
                    "nnhipRMSNormBackward: cols > 16384 not supported");
    hipStream_t st = (hipStream_t)s;
    // persistent-ish grid: <= 1024 blocks (4 per CU), each accumulating dw/db partials over its rows,
    // two rows in flight per iteration
    const int rpb = cols <= 1024 ? 4 : 1;
    int64_t nblk = ceil_div(ceil_div(rows > 0 ? rows : 1, rpb), cols > 8192 ? 1 : 2);
    if (nblk > 1024) nblk = 1024;
    const int64_t prow = nblk * rpb;
    const size_t part_floats = ((size_t)prow * cols + 3) / 4 * 4;
    // One workspace block: [dw partials | db partials | column-sum scratch for dw | ... for db]
    const ColsumPlan cp = colsum_plan(nullptr, dW, prow, cols, cols);
    const size_t scr = (cp.scratch_floats + 3) / 4 * 4;
    const int nred = db ? 2 : 1;
    float* part = static_cast<float*>(workspace((part_floats + scr) * nred * sizeof(float)));
    NNHIP_CHECK_ARG(part != nullptr, NNHIP_ENOMEM, "nnhipRMSNormBackward: workspace allocation failed");
    float* part_dw = part;
    float* part_db = db ? part + part_floats : nullptr;
    float* scr_dw = part + part_floats * nred;
    float* scr_db = scr_dw + scr;
    const bool vec = aligned16(dY) && aligned16(X) && aligned16(weight) && aligned16(dX) && aligned16(dX_addend) && cols % 4 == 0;
    {
        const int64_t rows_ = rows;
        // ROW_LAUNCH computes its grid from `rows`; we want exactly nblk blocks -> pass nblk*rpb.
        ROW_DISPATCH(rmsnorm_bwd_rows, cols, vec, prow, st, dY, X, weight, X_std, dX, part_dw, part_db, rows_, cols, dX_addend);
    }
    NNHIP_LAUNCH_CHECK("rmsnorm_backward");
    // finish dw/db: column sums over the `prow` partial rows
    ColsumPlan cdw = colsum_plan(part_dw, dW, prow, cols, cols);
    if (int rc = colsum_run(cdw, part_dw, prow, cols, cols, dW, scr_dw, st)) return rc;
    if (db) {
        ColsumPlan cdb = colsum_plan(part_db, db, prow, cols, cols);
        if (int rc = colsum_run(cdb, part_db, prow, cols, cols, db, scr_db, st)) return rc;
    }
    return 0;
}

// ---- CrossEntropy -------------------------------------------------------------------------------
extern "C" int nnhipCrossEntropyForwardBackward(float* logits, float* loss, float* lse,
                                                const int32_t* labels, int64_t logits_stride,
                                                int32_t ignore_index, int64_t n_rows, int64_t n_cols,
                                                char reduction, int64_t n_non_ignore,
                                                const int32_t* n_non_ignore_dev, float* dlogits,
                                                nnhipStream_t s) {
    NNHIP_CHECK_ARG(n_rows >= 0 && n_cols >= 0 && logits_stride >= n_cols, NNHIP_EINVAL,
                    "nnhipCrossEntropyForwardBackward: bad sizes");
    NNHIP_CHECK_ARG(reduction == 'n' || reduction == 'm' || reduction == 's', NNHIP_EINVAL,
                    "nnhipCrossEntropyForwardBackward: reduction must be 'n', 'm' or 's'");
    if (n_rows == 0 || n_cols == 0) return 0;
    NNHIP_CHECK_ARG(logits && loss && lse && labels, NNHIP_EINVAL,
                    "nnhipCrossEntropyForwardBackward: null pointer");
    if (!dlogits) dlogits = logits;  // reference behaviour: overwrite logits with the gradient
    hipStream_t st = (hipStream_t)s;
    const int use_mean = reduction == 'm';
    float scale = 1.0f;
    if (use_mean && !n_non_ignore_dev) scale = n_non_ignore > 0 ? 1.0f / (float)n_non_ignore : 0.0f;
    if (n_cols > kMaxRegRow) {
        hipLaunchKernelGGL(ce_fwd_bwd_looped, dim3((unsigned)n_rows), dim3(1024), 0, st, logits, dlogits,
                           loss, lse, labels, logits_stride, ignore_index, n_cols, scale,
                           n_non_ignore_dev, use_mean);
    } else {
        const bool vec = aligned16(logits) && aligned16(dlogits) && n_cols % 4 == 0 && logits_stride % 4 == 0;
        ROW_DISPATCH(ce_fwd_bwd_rows, n_cols, vec, n_rows, st, logits, dlogits, loss, lse, labels,
                     logits_stride, ignore_index, n_rows, n_cols, scale, n_non_ignore_dev, use_mean);
    }
    NNHIP_LAUNCH_CHECK("cross_entropy_forward_backward");
    return 0;
}

extern "C" int nnhipCountNotEqual(const int32_t* labels, int64_t n, int32_t ignore_index,
                                  int32_t* out_count, nnhipStream_t s) {
    NNHIP_CHECK_ARG(n >= 0 && out_count && (labels || n == 0), NNHIP_EINVAL, "nnhipCountNotEqual: bad args");
    hipLaunchKernelGGL(count_ne_kernel, dim3(1), dim3(1024), 0, (hipStream_t)s, labels, n, ignore_index, out_count);
    NNHIP_LAUNCH_CHECK("count_ne_kernel");
    return 0;
}

extern "C" int nnhipReduceLoss(const float* loss_rows, int64_t n_rows, char reduction,
                               const int32_t* count_dev, float* out, nnhipStream_t s) {
    NNHIP_CHECK_ARG(n_rows >= 0 && out && (loss_rows || n_rows == 0), NNHIP_EINVAL, "nnhipReduceLoss: bad args");
    NNHIP_CHECK_ARG(reduction == 'm' || reduction == 's', NNHIP_EINVAL, "nnhipReduceLoss: reduction must be 'm' or 's'");
    NNHIP_CHECK_ARG(reduction != 'm' || count_dev, NNHIP_EINVAL, "nnhipReduceLoss: 'm' needs count_dev");
    hipLaunchKernelGGL(reduce_loss_kernel, dim3(1), dim3(1024), 0, (hipStream_t)s, loss_rows, n_rows,
                       reduction == 'm' ? 1 : 0, count_dev, out);
    NNHIP_LAUNCH_CHECK("reduce_loss_kernel");
    return 0;
}

// CrossEntropyLoss(reduction = 'm' | 's') in one call: d(logits), per-row loss, lse, the reduced loss and (for 'm') the
// non-ignored count.  Small problems (rows * cols <= 64 K, cols <= 4096) take one single-block launch; anything else
// runs nnhipCountNotEqual + nnhipCrossEntropyForwardBackward + nnhipReduceLoss with `count_out` as the device count.
extern "C" int nnhipCrossEntropyLoss(float* logits, float* dlogits_or_null, float* loss_rows, float* lse,
                                     const int32_t* labels, int64_t logits_stride, int32_t ignore_index,
                                     int64_t n_rows, int64_t n_cols, char reduction, float* loss_out,
                                     int32_t* count_out, nnhipStream_t s) {
    NNHIP_CHECK_ARG(n_rows >= 0 && n_cols >= 0 && logits_stride >= n_cols, NNHIP_EINVAL, "nnhipCrossEntropyLoss: bad sizes");
    NNHIP_CHECK_ARG(reduction == 'm' || reduction == 's', NNHIP_EINVAL, "nnhipCrossEntropyLoss: reduction must be 'm' or 's'");
    NNHIP_CHECK_ARG(loss_out && (reduction != 'm' || count_out), NNHIP_EINVAL, "nnhipCrossEntropyLoss: null loss_out / count_out");
    if (n_rows > 0 && n_cols > 0 && n_rows * n_cols <= 65536 && n_cols <= 4096) {
        NNHIP_CHECK_ARG(logits && loss_rows && lse && labels, NNHIP_EINVAL, "nnhipCrossEntropyLoss: null pointer");
        hipLaunchKernelGGL(ce_small_kernel, dim3(1), dim3(1024), 0, (hipStream_t)s, logits, dlogits_or_null ? dlogits_or_null : logits,
                           loss_rows, lse, labels, logits_stride, ignore_index, n_rows, n_cols, reduction == 'm' ? 1 : 0,
                           loss_out, count_out);
        NNHIP_LAUNCH_CHECK("ce_small_kernel");
        return 0;
    }
    if (reduction == 'm')
        if (int rc = nnhipCountNotEqual(labels, n_rows, ignore_index, count_out, s)) return rc;
    if (int rc = nnhipCrossEntropyForwardBackward(logits, loss_rows, lse, labels, logits_stride, ignore_index, n_rows, n_cols,
                                                  reduction, -1, reduction == 'm' ? count_out : nullptr, dlogits_or_null, s))
        return rc;
    return nnhipReduceLoss(loss_rows, n_rows, reduction, count_out, loss_out, s);
}

// ---- attention-score softmax (scale + pad/causal mask fused) --------------------------------------------
extern "C" int nnhipMaskedSoftmaxForward(float* out, const float* in, const int32_t* key_valid, int64_t B,
                                         int64_t H, int64_t Tq, int64_t Tk, float scale, int causal,
                                         nnhipStream_t s) {
    NNHIP_CHECK_ARG(B >= 0 && H >= 0 && Tq >= 0 && Tk >= 0, NNHIP_EINVAL, "nnhipMaskedSoftmaxForward: negative size");
    const int64_t rows = B * H * Tq;
    if (rows == 0 || Tk == 0) return 0;
    NNHIP_CHECK_ARG(out && in, NNHIP_EINVAL, "nnhipMaskedSoftmaxForward: null pointer");
    NNHIP_CHECK_ARG(Tk <= kMaxRegRow, NNHIP_EINVAL, "nnhipMaskedSoftmaxForward: Tk > 16384 not supported");
    hipStream_t st = (hipStream_t)s;
    const bool vec = aligned16(out) && aligned16(in) && Tk % 4 == 0;
    ROW_DISPATCH(softmax_masked_fwd_rows, Tk, vec, rows, st, out, in, key_valid, rows, Tk, H * Tq, Tq, scale, causal);
    NNHIP_LAUNCH_CHECK("softmax_masked_fwd_rows");
    return 0;
}

extern "C" int nnhipMaskedSoftmaxBackward(float* dX, const float* dY, const float* Y, const int32_t* key_valid,
                                          int64_t B, int64_t H, int64_t Tq, int64_t Tk, float scale, int causal,
                                          nnhipStream_t s) {
    NNHIP_CHECK_ARG(B >= 0 && H >= 0 && Tq >= 0 && Tk >= 0, NNHIP_EINVAL, "nnhipMaskedSoftmaxBackward: negative size");
    const int64_t rows = B * H * Tq;
    if (rows == 0 || Tk == 0) return 0;
    NNHIP_CHECK_ARG(dX && dY && Y, NNHIP_EINVAL, "nnhipMaskedSoftmaxBackward: null pointer");
    NNHIP_CHECK_ARG(Tk <= kMaxRegRow, NNHIP_EINVAL, "nnhipMaskedSoftmaxBackward: Tk > 16384 not supported");
    hipStream_t st = (hipStream_t)s;
    const bool vec = aligned16(dX) && aligned16(dY) && aligned16(Y) && Tk % 4 == 0;
    ROW_DISPATCH(softmax_masked_bwd_rows, Tk, vec, rows, st, dX, dY, Y, key_valid, rows, Tk, H * Tq, Tq, scale, causal);
    NNHIP_LAUNCH_CHECK("softmax_masked_bwd_rows");
    return 0;
}
