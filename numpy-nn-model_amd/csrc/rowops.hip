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
#include <stdlib.h>

#include <mutex>
#include <unordered_map>

#include "common.h"
#include "gemm_small.h"

namespace nnhip {

int fill_f32(float* p, float v, int64_t n, hipStream_t st);   // elementwise.hip

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
    // nt (wave-uniform): streaming (non-temporal) loads.  A tensor that cannot be cache-resident anyway is read without
    // allocating in L2, which leaves L2 to the write-back of the previous kernel's output: the cold C3 pass gained 10-30 %
    // per op (RMSNorm backward 0.59 -> 0.72 of the HBM spec, CE 0.46 -> 0.59, softmax backward 0.67 -> 0.80).  It LOSES on
    // tensors the producer left in L2 / MALL (RMSNorm at 16384x512: -20 %) and on rows whose pitch is not a multiple of the
    // 128-B line (CE at 16384x15000: the line two rows share is fetched twice, -14 %) -- row_streaming() decides.
    __device__ __forceinline__ void load(const float* __restrict__ row, int64_t cols, int t, float fill, bool nt = false) {
        if constexpr (VEC) {
            typedef float rt_f4 __attribute__((ext_vector_type(4)));
            if (nt) {
                // (a select between the two kinds of load is folded into ONE plain load: the hint is only metadata.  Two
                //  copies of the loop under a wave-uniform branch, kept apart by an empty asm)
                asm volatile("" ::: "memory");
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    const int64_t c = 4 * (int64_t)(t + TPR * v);
                    if (c < cols) {
                        const rt_f4 q = __builtin_nontemporal_load(reinterpret_cast<const rt_f4*>(row + c));
                        x[4 * v] = q.x; x[4 * v + 1] = q.y; x[4 * v + 2] = q.z; x[4 * v + 3] = q.w;
                    } else {
                        x[4 * v] = x[4 * v + 1] = x[4 * v + 2] = x[4 * v + 3] = fill;
                    }
                }
            } else {
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    const int64_t c = 4 * (int64_t)(t + TPR * v);
                    if (c < cols) {
                        const rt_f4 q = *reinterpret_cast<const rt_f4*>(row + c);
                        x[4 * v] = q.x; x[4 * v + 1] = q.y; x[4 * v + 2] = q.z; x[4 * v + 3] = q.w;
                    } else {
                        x[4 * v] = x[4 * v + 1] = x[4 * v + 2] = x[4 * v + 3] = fill;
                    }
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
    const int t = RPB > 1 ? threadIdx.x % TPR : threadIdx.x;                  \
    const int64_t row = (int64_t)blockIdx.x * RPB + (RPB > 1 ? threadIdx.x / TPR : 0); /* uniform when a block owns one row */ \
    if (TPR == 64 && row >= rows) return; /* wave-uniform; no block barriers in the TPR=64 path */ \
    (void)red;

// =================================================================================================
// Softmax   (neunet/nn/activations.py:448-459 fwd, 437-446 bwd)
// =================================================================================================
template <int TPR, int NV, bool VEC>
__global__ __launch_bounds__((TPR >= 256) ? TPR : 256) void softmax_fwd_rows(
    float* __restrict__ out, const float* __restrict__ in, int64_t rows, int64_t cols, int nt) {
    ROW_PROLOGUE(TPR)
    RowTile<TPR, NV, VEC> r;
    r.load(in + row * cols, cols, t, -INFINITY, nt);
    float m = r.x[0];
#pragma unroll
    for (int e = 1; e < r.NE; ++e) m = fmaxf(m, r.x[e]);
    m = row_max<TPR>(m, red);
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < r.NE; ++e) {
        r.x[e] = exp_fast_(r.x[e] - m);  // exp(-inf) = 0 for the padding lanes
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
    int64_t cols, int nt) {
    ROW_PROLOGUE(TPR)
    RowTile<TPR, NV, VEC> g, f;
    g.load(dy + row * cols, cols, t, 0.f, nt);
    f.load(y + row * cols, cols, t, 0.f, nt);
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
    int64_t rows, int64_t cols, int64_t HTq, int64_t Tq, float scale, int causal, const int32_t* __restrict__ dense) {
    ROW_PROLOGUE(TPR)
    RowTile<TPR, NV, VEC> r;
    r.load(in + row * cols, cols, t, -INFINITY);
    const int64_t b = row / HTq;
    const int64_t i = row % Tq;
    const int64_t lim = causal ? i + (cols - Tq) : cols;  // last visible column
    const int32_t* kv = key_valid ? key_valid + b * cols : nullptr;
    const int32_t* dm = dense ? dense + (b * Tq + i) * cols : nullptr;   // dense [B,Tq,Tk] mask row (0 = masked)
    float m = -INFINITY;
#pragma unroll
    for (int e = 0; e < r.NE; ++e) {
        const int64_t c = r.col(t, e);
        if (c < cols) {
            const bool masked = c > lim || (kv && kv[c] == 0) || (dm && dm[c] == 0);
            r.x[e] = masked ? -1e9f : r.x[e] * scale;
        }
        m = fmaxf(m, r.x[e]);
    }
    m = row_max<TPR>(m, red);
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < r.NE; ++e) {
        r.x[e] = exp_fast_(r.x[e] - m);
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
    float scale, int causal, const int32_t* __restrict__ dense) {
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
    const int32_t* dm = dense ? dense + (b * Tq + i) * cols : nullptr;
#pragma unroll
    for (int e = 0; e < g.NE; ++e) {
        const int64_t c = g.col(t, e);
        const bool masked = c < cols && (c > lim || (kv && kv[c] == 0) || (dm && dm[c] == 0));
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
    for (int64_t i = 0; i < n; ++i) d += exp_fast_(in[base + i * stride] - m);
    const float inv = 1.0f / d;
    for (int64_t i = 0; i < n; ++i) out[base + i * stride] = exp_fast_(in[base + i * stride] - m) * inv;
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
    for (int64_t i = threadIdx.x; i < n; i += 256) d += exp_fast_(x[i] - m);
    d = block_sum<4>(d, red);
    const float inv = 1.0f / d;
    for (int64_t i = threadIdx.x; i < n; i += 256) o[i] = exp_fast_(x[i] - m) * inv;
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
    int64_t cols, float eps, int nt) {
    ROW_PROLOGUE(TPR)
    RowTile<TPR, NV, VEC> r, wt;
    r.load(X + row * cols, cols, t, 0.f, nt);
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

// ---- column-sum of per-block partials ------------------------------------------------------------------------
// part[prow][cols] (dense) -> out[cols].  Work unit = a SLICE of SW float4 columns (VEC) or SW columns (scalar) over ALL
// partial rows: BS threads = SW column-threads x BS/SW row groups; a thread issues 16 row loads back to back (the
// partials come out of other XCDs' L2 / HBM: one ~2 us round trip instead of a chain of them), sums them, and the
// groups meet through wave shuffles + one LDS hop in a fixed order (deterministic).  Used two ways:
// by colsum_tall_kernel, one block per slice: the one-launch finish of RMSNorm dw/db and of the two-stage column sum.
template <int BS, int SW, bool VEC>
__device__ __forceinline__ void colsum_slices(const float* part, int64_t prow, int64_t cols, float* out, int f,
                                              int nfin, float4* lds) {
    static_assert(SW == 4 || SW == 8 || SW == 16, "slice width");
    constexpr int G = BS / SW;                       // row groups
    constexpr int GW = 64 / SW;                      // row groups inside one wave
    constexpr int NWV = BS / 64;
    const int c = threadIdx.x % SW, g = threadIdx.x / SW;
    const int wave = threadIdx.x >> 6;
    const int64_t units = VEC ? (cols >> 2) : cols;  // float4s or floats per partial row
    const int64_t nsl = (units + SW - 1) / SW;
    for (int64_t s = f; s < nsl; s += nfin) {
        const int64_t u = s * SW + c;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        if (u < units) {
            for (int64_t r0 = g; r0 < prow; r0 += 16 * G) {
                if constexpr (VEC) {
                    const float4* p4 = reinterpret_cast<const float4*>(part);
                    float4 v[16];
#pragma unroll
                    for (int j = 0; j < 16; ++j) {
                        const int64_t r = r0 + (int64_t)j * G;
                        v[j] = r < prow ? p4[r * units + u] : make_float4(0.f, 0.f, 0.f, 0.f);
                    }
#pragma unroll
                    for (int j = 0; j < 16; ++j) { a.x += v[j].x; a.y += v[j].y; a.z += v[j].z; a.w += v[j].w; }
                } else {
                    float v[16];
#pragma unroll
                    for (int j = 0; j < 16; ++j) {
                        const int64_t r = r0 + (int64_t)j * G;
                        v[j] = r < prow ? part[r * units + u] : 0.f;
                    }
#pragma unroll
                    for (int j = 0; j < 16; ++j) a.x += v[j];
                }
            }
        }
        // row groups of one wave: lanes c, c+SW, c+2SW, ... hold the same column
#pragma unroll
        for (int o = SW; o < 64; o <<= 1) {
            a.x += __shfl_xor(a.x, o, 64); a.y += __shfl_xor(a.y, o, 64);
            a.z += __shfl_xor(a.z, o, 64); a.w += __shfl_xor(a.w, o, 64);
        }
        (void)GW;
        if constexpr (NWV > 1) {
            if ((threadIdx.x & 63) < SW) lds[wave * SW + c] = a;
            __syncthreads();
            if (threadIdx.x < SW) {
                a = lds[c];
#pragma unroll
                for (int w = 1; w < NWV; ++w) {
                    const float4 v = lds[w * SW + c];
                    a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
                }
            }
        }
        if (threadIdx.x < SW && u < units) {
            if constexpr (VEC) reinterpret_cast<float4*>(out)[u] = a;
            else out[u] = a.x;
        }
        if constexpr (NWV > 1) __syncthreads();
    }
}
static inline int fin_sw(int64_t cols, bool vec) { return (vec ? (cols >> 2) : cols) >= 512 ? 8 : 4; }
// how many slices a [*, cols] column-sum has (= blocks of colsum_tall_kernel; the in-launch variant caps its finishers)
static inline int fin_slices(int64_t cols, bool vec) {
    const int64_t units = vec ? (cols >> 2) : cols;
    const int sw = fin_sw(cols, vec);
    const int64_t n = (units + sw - 1) / sw;
    return (int)(n < 1 ? 1 : n);
}

// One launch: up to two partial arrays (dw and db) -> their column sums.  grid (slices, narrays), 256 threads.
template <int SW, bool VEC>
__global__ __launch_bounds__(256) void colsum_tall_kernel(const float* __restrict__ part_a, float* __restrict__ out_a,
                                                          const float* __restrict__ part_b, float* __restrict__ out_b,
                                                          int64_t prow, int64_t cols) {
    __shared__ float4 lds[4 * SW];
    const float* part = blockIdx.y ? part_b : part_a;
    float* out = blockIdx.y ? out_b : out_a;
    colsum_slices<256, SW, VEC>(part, prow, cols, out, (int)blockIdx.x, (int)gridDim.x, lds);
}
static int colsum_tall(const float* part_a, float* out_a, const float* part_b, float* out_b, int64_t prow, int64_t cols,
                       bool vec, hipStream_t st) {
    const int sw = fin_sw(cols, vec);
    dim3 grid((unsigned)fin_slices(cols, vec), part_b ? 2u : 1u);
#define TALL(SW_, V_) hipLaunchKernelGGL((colsum_tall_kernel<SW_, V_>), grid, dim3(256), 0, st, part_a, out_a, part_b, out_b, prow, cols)
    if (vec) { if (sw == 8) TALL(8, true); else TALL(4, true); }
    else { if (sw == 8) TALL(8, false); else TALL(4, false); }
#undef TALL
    NNHIP_LAUNCH_CHECK("colsum_tall_kernel");
    return 0;
}

// ---- the column sums of several layers as ONE launch (deferred parameter gradients, common.h) ------------------------------------
// During Tensor.backward() (nnhipWeightGradDefer) the per-block dw / db partials of a RMSNorm backward go to an arena of their own and
// the finishing column sum is queued; nnhipWeightGradFlush launches the queue as one grid (GPT-tiny: 13 finishes of ~5 us each, every
// one a launch-latency-bound kernel of 16 blocks, become one launch per flush).  Same arithmetic and order per output column.
constexpr int CSQ_MAX = 16;
struct ColsumJob { const float* part; float* out; int64_t prow, cols; };
struct ColsumGroup { ColsumJob j[CSQ_MAX]; int start[CSQ_MAX + 1]; };
template <int SW, bool VEC>
__global__ __launch_bounds__(256) void colsum_group_kernel(const ColsumGroup grp) {
    __shared__ float4 lds[4 * SW];
    int k = 0;
#pragma unroll
    for (int i = 1; i < CSQ_MAX; ++i) k += (int)blockIdx.x >= grp.start[i] ? 1 : 0;
    const ColsumJob jb = grp.j[k];
    colsum_slices<256, SW, VEC>(jb.part, jb.prow, jb.cols, jb.out, (int)blockIdx.x - grp.start[k], grp.start[k + 1] - grp.start[k], lds);
}
static std::mutex g_csq_mu;
static ColsumJob g_csq[CSQ_MAX];
static int g_csq_n = 0, g_csq_sw = 0;
static bool g_csq_vec = false;
static hipStream_t g_csq_st = nullptr;
static float* g_csq_arena = nullptr;
static size_t g_csq_cap = 0, g_csq_used = 0;              // floats
// The arena is ONE per process.  A flush launched on stream A may still be reading it when a backward pass on stream B starts
// writing partials at offset 0 (advisor, round 5): every flush outside a graph capture records an event, and the first reservation
// on another stream waits for it.  (Inside a capture the step is single-stream by construction, neunet_hip/graph.py.)
static hipEvent_t g_csq_ev = nullptr;
static hipStream_t g_csq_flush_st = nullptr;
static bool g_csq_ev_live = false;
static bool csq_capturing(hipStream_t st) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess) { (void)hipGetLastError(); return false; }
    return cs != hipStreamCaptureStatusNone;
}
static int csq_flush_locked(hipStream_t st) {
    if (g_csq_n == 0) { g_csq_used = 0; return 0; }         // (a reservation whose launch failed leaves nothing queued: give its floats back)
    ColsumGroup grp;
    int blocks = 0;
    for (int i = 0; i < CSQ_MAX; ++i) {
        grp.start[i] = blocks;
        if (i < g_csq_n) { grp.j[i] = g_csq[i]; blocks += fin_slices(g_csq[i].cols, g_csq_vec); }
        else grp.j[i] = g_csq[0];
    }
    grp.start[CSQ_MAX] = blocks;
    g_csq_n = 0;
    g_csq_used = 0;
#define CSG(SW_, V_) hipLaunchKernelGGL((colsum_group_kernel<SW_, V_>), dim3((unsigned)blocks), dim3(256), 0, st, grp)
    if (g_csq_vec) { if (g_csq_sw == 8) CSG(8, true); else CSG(4, true); }
    else { if (g_csq_sw == 8) CSG(8, false); else CSG(4, false); }
#undef CSG
    NNHIP_LAUNCH_CHECK("colsum_group_kernel");
    if (!csq_capturing(st)) {
        if (!g_csq_ev && hipEventCreateWithFlags(&g_csq_ev, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); g_csq_ev = nullptr; }
        if (g_csq_ev && hipEventRecord(g_csq_ev, st) == hipSuccess) { g_csq_flush_st = st; g_csq_ev_live = true; }
        else (void)hipGetLastError();
    }
    return 0;
}
int colsum_flush(void* stream) {
    std::lock_guard<std::mutex> lk(g_csq_mu);
    return csq_flush_locked(g_csq_n ? g_csq_st : (hipStream_t)stream);
}
void colsum_cleanup() {
    std::lock_guard<std::mutex> lk(g_csq_mu);
    g_csq_n = 0;
    g_csq_used = g_csq_cap = 0;
    if (g_csq_arena) (void)hipFree(g_csq_arena);
    g_csq_arena = nullptr;
}
// Where a RMSNorm backward writes `floats` of partials whose column sums (`njobs` of them, slice width sw / vec) can wait for the
// flush: the arena (*deferred), else nullptr (the caller uses the shared workspace and finishes at once).
static float* colsum_partials(size_t floats, int njobs, int sw, bool vec, hipStream_t st, bool* deferred, int* rc) {
    static const bool on = []() { const char* e = getenv("NNHIP_COLSUM_DEFER"); return !e || atoi(e) != 0; }();
    *deferred = false;
    *rc = 0;
    if (!on || !wgrad_defer_on()) return nullptr;
    std::lock_guard<std::mutex> lk(g_csq_mu);
    if (g_csq_n && (g_csq_st != st || g_csq_n + njobs > CSQ_MAX || g_csq_used + floats > g_csq_cap || g_csq_sw != sw || g_csq_vec != vec))
        *rc = csq_flush_locked(g_csq_st);
    if (g_csq_ev_live && g_csq_flush_st != st && !csq_capturing(st)) {      // the last flush ran on another stream: order behind it
        if (hipStreamWaitEvent(st, g_csq_ev, 0) != hipSuccess) (void)hipGetLastError();
        g_csq_flush_st = st;                                                // (this stream is now ordered behind that flush)
    }
    if (floats > g_csq_cap && !workspace_locked()) {          // grow (nothing is queued here); never while a captured graph holds the address
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        (void)hipStreamIsCapturing(st, &cs);
        if (cs == hipStreamCaptureStatusNone) {
            if (g_csq_arena) (void)hipFree(g_csq_arena);      // (synchronises: earlier sums are done with it)
            g_csq_arena = nullptr;
            g_csq_cap = 0;
            const size_t want = floats * CSQ_MAX;              // room for a whole queue of the same size
            if (hipMalloc(&g_csq_arena, want * sizeof(float)) == hipSuccess) g_csq_cap = want;
            else g_csq_arena = nullptr;
        }
    }
    if (g_csq_arena && floats <= g_csq_cap - g_csq_used) {
        float* p = g_csq_arena + g_csq_used;
        g_csq_used += (floats + 63) & ~(size_t)63;
        g_csq_st = st; g_csq_sw = sw; g_csq_vec = vec;
        *deferred = true;
        return p;
    }
    return nullptr;
}
static void colsum_queue(const float* part, float* out, int64_t prow, int64_t cols) {
    std::lock_guard<std::mutex> lk(g_csq_mu);
    g_csq[g_csq_n++] = ColsumJob{part, out, prow, cols};
}

// Backward: block b walks rows b*RPB+rslot + k*gridDim.x*RPB, one row per iteration with the next row's loads already
// in flight, keeps per-thread column
// partials of dw = sum dy*x/std and db = sum dy in registers (a thread always owns the same columns),
// and writes them once at the end to part[b][cols] (wave-per-row blocks first add their 4 waves' partials in LDS).
// dw/db are finished by ONE more launch for both (colsum_tall_kernel; round 1 used two two-stage column sums = up to
// four launches).  Finishing them inside this launch -- arrival ticket after an agent-scope release, the last blocks
// to arrive column-sum the partials -- was built and measured in round 2: 107 us instead of 81 at 8192x4096, 39
// instead of 23 at 16384x512 (every block's release fence writes back its XCD's whole L2, which is full of dirty dX
// lines, and the finishers start from cold caches); a kernel boundary costs ~1.7 us (MI355X_MICROARCH.md).
// dx needs one row reduction: S = sum(w dy x / std).
template <int TPR, int NV, bool VEC>
__global__ __launch_bounds__((TPR >= 256) ? TPR : 256) void rmsnorm_bwd_rows(
    const float* __restrict__ dY, const float* __restrict__ X, const float* __restrict__ w,
    const float* __restrict__ Xstd, float* __restrict__ dX, float* __restrict__ part_dw, float* __restrict__ part_db,
    int64_t rows, int64_t cols, const float* __restrict__ dXadd, int nt) {
    constexpr int RPB = (TPR >= 256) ? 1 : 256 / TPR;
    constexpr int NW = TPR / 64;
    __shared__ float red[32];
    __shared__ float4 fin_lds[RPB > 1 ? 256 : 1];
    const int t = threadIdx.x % TPR;
    const int rslot = RPB > 1 ? threadIdx.x / TPR : 0;   // a compile-time 0 keeps the row pointers in scalar registers
    RowTile<TPR, NV, VEC> wt, adw, adb;
    wt.load(w, cols, t, 0.f);
#pragma unroll
    for (int e = 0; e < wt.NE; ++e) adw.x[e] = adb.x[e] = 0.f;
    const float invN = 1.0f / (float)cols;
    const int64_t step = (int64_t)gridDim.x * RPB;
    // when TPR >= 256 every thread of the block runs the same trip count (block barriers inside)
    // PRE: the NEXT row is loaded while the current one is reduced and stored (a second pair of row tiles).  Round 1 kept
    // two rows in flight per iteration without prefetch: 232 registers -> 2 blocks per CU, and nothing in flight during a
    // block's reduce / store phase (5.2 TB/s at 8192x4096).  One row + one prefetched row needs half the registers, so
    // twice the blocks are resident and every one of them always has a row in flight.
    constexpr bool PRE = TPR < 1024;   // 1024-thread blocks have 128 registers per lane: no room for a second pair
    RowTile<TPR, NV, VEC> x0, g0, nx, ng;
    const int64_t first = (int64_t)blockIdx.x * RPB + rslot;
    if constexpr (PRE) {
        if (first < rows) {
            x0.load(X + first * cols, cols, t, 0.f, nt);
            g0.load(dY + first * cols, cols, t, 0.f, nt);
        }
    }
    for (int64_t r0 = first; r0 < rows; r0 += step) {
        if constexpr (PRE) {
            const int64_t n0 = r0 + step;
            if (n0 < rows) {
                nx.load(X + n0 * cols, cols, t, 0.f, nt);
                ng.load(dY + n0 * cols, cols, t, 0.f, nt);
            }
        } else {
            x0.load(X + r0 * cols, cols, t, 0.f, nt);
            g0.load(dY + r0 * cols, cols, t, 0.f, nt);
        }
        const float sd0 = Xstd[r0];
        const float i0 = 1.0f / sd0;
        float s0 = 0.f;
#pragma unroll
        for (int e = 0; e < x0.NE; ++e) {
            adb.x[e] += g0.x[e];
            adw.x[e] += g0.x[e] * (x0.x[e] * i0);
            g0.x[e] *= wt.x[e];  // dX_hat = w * dy
            s0 += g0.x[e] * x0.x[e] * i0;
        }
        s0 = block_sum<NW>(s0, red) * invN;
        const float q0 = i0 * i0;
#pragma unroll
        for (int e = 0; e < x0.NE; ++e) g0.x[e] = (g0.x[e] * sd0 - x0.x[e] * s0) * q0;
        if (dXadd) {   // dX = rmsnorm gradient + an already accumulated gradient of X (x0 is dead: reuse it)
            x0.load(dXadd + r0 * cols, cols, t, 0.f);
#pragma unroll
            for (int e = 0; e < x0.NE; ++e) g0.x[e] += x0.x[e];
        }
        g0.store(dX + r0 * cols, cols, t);
        if constexpr (PRE) {
#pragma unroll
            for (int e = 0; e < x0.NE; ++e) { x0.x[e] = nx.x[e]; g0.x[e] = ng.x[e]; }
        }
    }
    // ---- this block's column partials -> part[blockIdx.x][cols] ---------------------------------------------------
    if constexpr (RPB > 1) {
        // 4 waves = 4 row slots with the same column ownership: add them in LDS (slots 1..3 deposit, slot 0 sums in
        // slot order), so a block publishes ONE partial row (4x less to re-read in the finish)
        float* xch = reinterpret_cast<float*>(fin_lds);      // 256 float4 = 1024 floats >= 3 slots x NE x 64 / pass
        constexpr int NE = NV * 4;
        constexpr int EPP = (1024 / (3 * 64)) < NE ? (1024 / (3 * 64)) : NE;   // elements per pass (5 of <= 16)
        auto fold = [&](RowTile<TPR, NV, VEC>& acc) {
#pragma unroll
            for (int e0 = 0; e0 < NE; e0 += EPP) {
                __syncthreads();
                if (rslot > 0) {
#pragma unroll
                    for (int e = 0; e < EPP; ++e)
                        if (e0 + e < NE) xch[((rslot - 1) * EPP + e) * 64 + t] = acc.x[e0 + e];
                }
                __syncthreads();
                if (rslot == 0) {
#pragma unroll
                    for (int e = 0; e < EPP; ++e)
                        if (e0 + e < NE)
                            acc.x[e0 + e] += (xch[(0 * EPP + e) * 64 + t] + xch[(1 * EPP + e) * 64 + t]) + xch[(2 * EPP + e) * 64 + t];
                }
            }
        };
        fold(adw);
        if (part_db) fold(adb);
        __syncthreads();
    }
    if (rslot == 0) {
        adw.store(part_dw + (int64_t)blockIdx.x * cols, cols, t);
        if (part_db) adb.store(part_db + (int64_t)blockIdx.x * cols, cols, t);
    }
    // dw/db = column sums of the partials: colsum_tall_kernel, the next launch.
}

// ---- RMSNorm for rows wider than the register tile (cols > 16384): looped, one block per row -------------------------
// (the reference's kernels loop over any width, rmsnorm.cu:17-113; this is the correctness path for such widths:
//  forward 12 B/elem, backward 20 B/elem + a column pass for dw/db)
__global__ __launch_bounds__(1024) void rmsnorm_fwd_looped(const float* __restrict__ X, const float* __restrict__ w,
                                                           const float* __restrict__ b, float* __restrict__ Y,
                                                           float* __restrict__ Xstd, float* __restrict__ Xnorm,
                                                           int64_t cols, float eps) {
    __shared__ float red[16];
    const int64_t row = blockIdx.x;
    const float* x = X + row * cols;
    float ss = 0.f;
    for (int64_t i = threadIdx.x; i < cols; i += 1024) ss += x[i] * x[i];
    ss = block_sum<16>(ss, red);
    const float sd = sqrtf(ss / (float)cols + eps);
    const float inv = 1.0f / sd;
    if (threadIdx.x == 0) Xstd[row] = sd;
    for (int64_t i = threadIdx.x; i < cols; i += 1024) {
        const float xn = x[i] * inv;
        if (Xnorm) Xnorm[row * cols + i] = xn;
        Y[row * cols + i] = b ? xn * w[i] + b[i] : xn * w[i];
    }
}
__global__ __launch_bounds__(1024) void rmsnorm_bwd_dx_looped(const float* __restrict__ dY, const float* __restrict__ X,
                                                              const float* __restrict__ w, const float* __restrict__ Xstd,
                                                              float* __restrict__ dX, int64_t cols,
                                                              const float* __restrict__ dXadd) {
    __shared__ float red[16];
    const int64_t row = blockIdx.x;
    const float* x = X + row * cols;
    const float* g = dY + row * cols;
    const float sd = Xstd[row], inv = 1.0f / sd;
    float s = 0.f;
    for (int64_t i = threadIdx.x; i < cols; i += 1024) s += (g[i] * w[i]) * x[i] * inv;
    s = block_sum<16>(s, red) / (float)cols;
    const float q = inv * inv;
    for (int64_t i = threadIdx.x; i < cols; i += 1024) {
        float v = ((g[i] * w[i]) * sd - x[i] * s) * q;
        if (dXadd) v += dXadd[row * cols + i];
        dX[row * cols + i] = v;
    }
}
// dw[c] = sum_r dy[r,c] x[r,c] / std[r], db[c] = sum_r dy[r,c]: thread per column (coalesced across threads), rows split
// over gridDim.y into partials part[y][cols] that a column sum finishes.
__global__ __launch_bounds__(256) void rmsnorm_bwd_dwdb_cols(const float* __restrict__ dY, const float* __restrict__ X,
                                                             const float* __restrict__ Xstd, float* __restrict__ part_dw,
                                                             float* __restrict__ part_db, int64_t rows, int64_t cols,
                                                             int64_t rows_per_block) {
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= cols) return;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
    const int64_t r1 = min(rows, r0 + rows_per_block);
    float aw = 0.f, ab = 0.f;
    for (int64_t r = r0; r < r1; ++r) {
        const float g = dY[r * cols + c];
        aw += g * (X[r * cols + c] / Xstd[r]);
        ab += g;
    }
    part_dw[(int64_t)blockIdx.y * cols + c] = aw;
    if (part_db) part_db[(int64_t)blockIdx.y * cols + c] = ab;
}

// =================================================================================================
// Fused CrossEntropy forward+backward
// (CPU semantics: LogSoftmax(axis=1) -> NLLLoss, neunet/nn/losses.py:59-126; fusion boundary of
//  cross_entropy.cu:18-229: one pass computes loss, lse and d(logits))
//
// ONE launch does everything CrossEntropyLoss(mean|sum|none) needs:
//   * the 'mean' denominator (count of labels != ignore_index, or sum of class_weight[label] over them,
//     losses.py:117-118) is computed by EVERY block from the label vector (rows*4 B out of L2 per block, in the shadow of
//     the block's first row load) in the same order, so all blocks scale by the identical value;
//   * a persistent grid walks the rows (row-in-register tiles as before: 8 B/elem);
//   * every block publishes the sum of its row losses and takes an arrival ticket; the last arrival adds the partials
//     in block order (deterministic) and writes the reduced loss.
// Round 1 ran count / rows / reduce as three launches (61.5 us for a 45 us kernel at 8192x4096).
// Labels may be int16 / int32 / int64 (losses.py:100); class weights are optional.
// A label outside [0, cols) that is not ignore_index contributes zero loss and zero gradient (the reference would
// raise IndexError / Python-wrap it: losses.py:104 TODO) and is left out of the 'mean' denominator: exactly as if the row
// carried ignore_index.
// =================================================================================================
struct CeArgs {
    float* logits;
    float* dlogits;
    float* loss_rows;
    float* lse;
    const void* labels;
    const float* cw;            // class weights [cols] or null
    const int32_t* count_dev;   // 'mean' with an externally supplied denominator (device int) or null
    const float* denom_dev;     // 'mean' with a denominator computed by ce_denominator_kernel (tall problems) or null
    float* partial;             // [gridDim.x] block loss sums
    unsigned* sync;
    float* loss_out;            // reduced loss (mean/sum) or null
    int32_t* count_out;         // #labels != ignore (written when the kernel counts) or null
    int64_t ld, ignore, rows, cols;
    float scale_host;           // gradient scale when nothing on the device supplies it
    int lbytes;                 // 2 / 4 / 8
    int mode;                   // 0 none, 1 mean, 2 sum
    int count_in_kernel;        // 1: every block derives the 'mean' denominator from the labels
    int nt;                     // 1: streaming loads of the logits (row_streaming())
};

__device__ __forceinline__ int64_t load_label(const void* labels, int64_t i, int lbytes) {
    if (lbytes == 4) return reinterpret_cast<const int32_t*>(labels)[i];
    if (lbytes == 8) return reinterpret_cast<const int64_t*>(labels)[i];
    return reinterpret_cast<const int16_t*>(labels)[i];
}

template <int NW>
__device__ __forceinline__ int block_sum_int(int v, int* red) {
    v = wave_sum_int(v);
    if constexpr (NW == 1) return v;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    int s = 0;
#pragma unroll
    for (int i = 0; i < NW; ++i) s += red[i];
    return s;
}

// Block-wide count of labels != ignore (and, with class weights, the sum of w[label] over them) with all loads in
// flight: a plain `for (i = tid; i < rows; i += BS)` is a chain of dependent L2 round trips (32 of them for 8192 labels
// and 256 threads ~ 20 us per block -- measured: it doubled the kernel).  T = label type.
template <int BS, class T>
__device__ __forceinline__ void count_labels(const T* __restrict__ lab, int64_t rows, int64_t ignore, const float* __restrict__ cw,
                                             int64_t cols, int& ci, float& ws) {
    constexpr int U = 8;
    auto one = [&](int64_t l) {
        // ONE predicate for "this row takes part", the rows kernels' `valid`: not ignore_index and inside [0, cols).  A row the
        // kernels treat as inert must not sit in the 'mean' denominator either (round-3 review).  cols <= 0: no class bound
        // (nnhipCountNotEqual, whose contract is its name).
        const bool live = l != ignore && (cols <= 0 || (l >= 0 && l < cols));
        ci += live ? 1 : 0;
        if (cw && live) ws += cw[l];
    };
    int64_t done = 0;
    if constexpr (sizeof(T) == 4) {
        if ((reinterpret_cast<uintptr_t>(lab) & 15u) == 0) {   // 4 labels per load: 8192 labels = 2 batches of 4 loads for 256 threads
            const int4* l4 = reinterpret_cast<const int4*>(lab);
            const int64_t n4 = rows >> 2;
            constexpr int U4 = 8;                                // 8192 labels = ONE batch of 8 loads for 256 threads (one L2 round trip)
            for (int64_t base = 0; base < n4; base += (int64_t)BS * U4) {
                int4 v[U4];
#pragma unroll
                for (int u = 0; u < U4; ++u) {
                    const int64_t i = base + (int64_t)u * BS + threadIdx.x;
                    v[u] = i < n4 ? l4[i] : make_int4((int)ignore, (int)ignore, (int)ignore, (int)ignore);
                }
#pragma unroll
                for (int u = 0; u < U4; ++u) {
                    const int64_t i = base + (int64_t)u * BS + threadIdx.x;
                    if (i < n4) { one(v[u].x); one(v[u].y); one(v[u].z); one(v[u].w); }
                }
            }
            done = n4 << 2;
        }
    }
    for (int64_t base = done; base < rows; base += (int64_t)BS * U) {
        T l[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = base + (int64_t)u * BS + threadIdx.x;
            l[u] = i < rows ? lab[i] : (T)0;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = base + (int64_t)u * BS + threadIdx.x;
            if (i < rows) one((int64_t)l[u]);
        }
    }
}

// 'mean' denominator and gradient scale of this launch; identical in every block.
template <int BS>
__device__ __forceinline__ void ce_prologue(const CeArgs& a, float* red, int* ired, float& scale, float& denom) {
    scale = a.scale_host;
    denom = 1.f;
    if (a.mode != 1) return;
    if (a.count_in_kernel) {
        int ci = 0;
        float ws = 0.f;
        if (a.lbytes == 4) count_labels<BS>(reinterpret_cast<const int32_t*>(a.labels), a.rows, a.ignore, a.cw, a.cols, ci, ws);
        else if (a.lbytes == 8) count_labels<BS>(reinterpret_cast<const int64_t*>(a.labels), a.rows, a.ignore, a.cw, a.cols, ci, ws);
        else count_labels<BS>(reinterpret_cast<const int16_t*>(a.labels), a.rows, a.ignore, a.cw, a.cols, ci, ws);
        ci = block_sum_int<BS / 64>(ci, ired);
        if (a.cw) ws = block_sum<BS / 64>(ws, red);
        denom = a.cw ? ws : (float)ci;
        if (a.count_out && blockIdx.x == 0 && threadIdx.x == 0) a.count_out[0] = ci;
        scale = denom > 0.f ? 1.0f / denom : 0.0f;
    } else if (a.denom_dev || a.count_dev) {
        // one plain (L1-cached) load per block, broadcast through LDS: the value was produced by an earlier launch or collective.
        // (An agent-scope load by every wave of every block is thousands of requests to one L2 channel at grid start.)
        if (threadIdx.x == 0) red[0] = a.denom_dev ? a.denom_dev[0] : (float)a.count_dev[0];
        __syncthreads();
        denom = red[0];
        __syncthreads();                                      // `red` is reused by the reductions that follow
        scale = a.denom_dev ? (denom > 0.f ? 1.0f / denom : 0.0f) : 1.0f / denom;
    } else {
        denom = a.scale_host > 0.f ? 1.0f / a.scale_host : 0.f;
    }
}

// Publish this block's loss sum, and let the last arrival reduce all of them.  `lsum` is valid on thread 0.
template <int BS>
__device__ __forceinline__ void ce_epilogue(const CeArgs& a, float lsum, float denom, float* red, int* ired) {
    if (!a.loss_out) return;
    const int nblk = (int)gridDim.x;
    if (nblk == 2) {
        // Two blocks (the README MLP's 32-row head): ONE exchange both publishes this block's sum and fetches the other's if it
        // is there -- one L2 round trip instead of three (store + drain, ticket, reload: ~3 us of an 8 us kernel).  Whoever comes
        // second adds the two sums in block order, the same bits as the general path below.
        if (threadIdx.x == 0) {
            unsigned long long* w = reinterpret_cast<unsigned long long*>(a.sync + 2);
            const unsigned long long mine = (1ull << 63) | ((unsigned long long)blockIdx.x << 32) | (unsigned long long)__float_as_uint(lsum);
            const unsigned long long old = atomicExch(w, mine);
            if (old >> 63) {
                const float other = __uint_as_float((unsigned)(old & 0xffffffffull));
                const float s = ((old >> 32) & 1ull) == 0 ? other + lsum : lsum + other;
                a.loss_out[0] = a.mode == 1 ? s / denom : s;
                __hip_atomic_store(w, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        return;
    }
    if (threadIdx.x == 0) {
        // one dword per block: agent-scope (write-through) store + drain, no L2-wide release needed for it
        __hip_atomic_store(&a.partial[blockIdx.x], lsum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ired[16] = (int)atomicAdd(&a.sync[0], 1u);
    }
    __syncthreads();
    if (ired[16] != nblk - 1) return;
    float s = 0.f;
    for (int base = 0; base < nblk; base += BS * 8) {        // all of a batch's loads in flight (nblk <= 8 * BS in practice)
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = base + u * BS + (int)threadIdx.x;
            v[u] = i < nblk ? __hip_atomic_load(&a.partial[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    s = block_sum<BS / 64>(s, red);
    if (threadIdx.x == 0) {
        a.loss_out[0] = a.mode == 1 ? s / denom : s;
        __hip_atomic_store(&a.sync[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

#ifndef NNHIP_CE_WAVES
#define NNHIP_CE_WAVES 4
#endif
template <int TPR, int NV, bool VEC>
__global__ __launch_bounds__((TPR >= 256) ? TPR : 256) __attribute__((amdgpu_waves_per_eu(NV <= 4 && VEC ? NNHIP_CE_WAVES : 4)))
void ce_rows_kernel(const CeArgs a) {
    constexpr int BS = (TPR >= 256) ? TPR : 256;
    constexpr int RPB = BS / TPR;
    __shared__ float red[32];
    __shared__ int ired[17];
    const int t = threadIdx.x % TPR;
    const int slot = RPB > 1 ? threadIdx.x / TPR : 0;    // a compile-time 0 keeps the row pointers in scalar registers
    const int64_t step = (int64_t)gridDim.x * RPB;
    const int64_t row0 = (int64_t)blockIdx.x * RPB + slot;
    // the first row's loads are issued BEFORE the denominator is counted: the count's label loads then travel in the
    // shadow of the row's HBM latency instead of in front of it.  PRE: while a row is reduced and stored, the block's NEXT
    // row is already being loaded into a second tile -- a persistent block otherwise has nothing in flight between its
    // rows (cold HBM: 4 resident blocks per CU x one 16 KB row is not enough to cover ~2 us of latency).
    constexpr bool PRE = NV <= 4;
    RowTile<TPR, NV, VEC> r, rn;
    if (row0 < a.rows) r.load(a.logits + row0 * a.ld, a.cols, t, -INFINITY, a.nt != 0);
    int64_t label = row0 < a.rows ? load_label(a.labels, row0, a.lbytes) : a.ignore;
    int64_t label_n = a.ignore;
    if constexpr (PRE) {                                      // ... and the second row too: two rows per block cross HBM while
        if (row0 + step < a.rows) {                           // the denominator is being counted out of L2
            label_n = load_label(a.labels, row0 + step, a.lbytes);
            rn.load(a.logits + (row0 + step) * a.ld, a.cols, t, -INFINITY, a.nt != 0);
        }
    }
    float scale, denom;
    ce_prologue<BS>(a, red, ired, scale, denom);
    float lsum = 0.f;                                         // meaningful on t == 0 of each row slot
    for (int64_t row = row0; row < a.rows; row += step) {
        // Nothing in an iteration waits on a load issued IN that iteration: the label arrived with the row's prefetch (an
        // iteration earlier), the label's logit is picked out of the register tile (no dependent gather), and the
        // optional class weight is requested before anything else.  Round 2's loop began with label load -> wait ->
        // gather -> prefetch: one L2 round trip per row in front of every prefetch.
        const bool valid = label != a.ignore && label >= 0 && label < a.cols;
        const float wy = valid ? (a.cw ? a.cw[label] : 1.f) : 0.f;
        const int lrel = valid ? (int)label - (VEC ? 4 * t : t) : -1;
        if constexpr (!PRE) {
            if (row != row0) r.load(a.logits + row * a.ld, a.cols, t, -INFINITY, a.nt != 0);
            label_n = row + step < a.rows ? load_label(a.labels, row + step, a.lbytes) : a.ignore;
        }
        float m = r.x[0];
#pragma unroll
        for (int e = 1; e < r.NE; ++e) m = fmaxf(m, r.x[e]);
        // the label's logit: exactly one thread of the row holds it (read before anything is overwritten: in-place mode)
        float xl = 0.f;
#pragma unroll
        for (int e = 0; e < r.NE; ++e) {
            const int off = VEC ? 4 * TPR * (e >> 2) + (e & 3) : TPR * e;
            xl = lrel == off ? r.x[e] : xl;
        }
        m = row_max<TPR>(m, red);
        float d = 0.f;
#pragma unroll
        for (int e = 0; e < r.NE; ++e) {
            r.x[e] = exp_fast_(r.x[e] - m);   // keep exp(x - m): softmax = e / d, no second exp pass
            d += r.x[e];
        }
        if constexpr (TPR == 64) {
            d = wave_sum(d);
            xl = wave_sum(xl);
        } else {
            block_sum2<TPR / 64>(d, xl, red);                 // one pair of barriers for both
        }
        const float lse = m + logf(d);
        if (t == 0) {
            const float l = valid ? (lse - xl) * wy : 0.f;
            a.lse[row] = lse;
            a.loss_rows[row] = l;
            lsum += l;
        }
        if (valid) {
            const float invd = 1.0f / d;
            const float gs = (a.mode == 1 ? scale : 1.f) * wy;
            // one-hot test against compile-time constants: label - (this thread's first column) is compared with the
            // element's fixed offset.  `label == r.col(t, e)` made the compiler keep NE 64-bit column indices alive
            // across the persistent loop (32 VGPRs at NV = 4 -- a quarter of the kernel's registers).
#pragma unroll
            for (int e = 0; e < r.NE; ++e) {
                const int off = VEC ? 4 * TPR * (e >> 2) + (e & 3) : TPR * e;
                r.x[e] = (r.x[e] * invd - (lrel == off ? 1.f : 0.f)) * gs;
            }
        } else {
#pragma unroll
            for (int e = 0; e < r.NE; ++e) r.x[e] = 0.f;
        }
        r.store(a.dlogits + row * a.ld, a.cols, t);
        label = label_n;
        if constexpr (PRE) {
#pragma unroll
            for (int e = 0; e < r.NE; ++e) r.x[e] = rn.x[e];
            if (row + 2 * step < a.rows) {                    // the row after next, into the tile just vacated
                label_n = load_label(a.labels, row + 2 * step, a.lbytes);
                rn.load(a.logits + (row + 2 * step) * a.ld, a.cols, t, -INFINITY, a.nt != 0);
            }
        }
    }
    if constexpr (RPB > 1) {                                  // row slots -> one block sum, in slot order
        if (a.loss_out) {
            __syncthreads();
            if (t == 0) red[slot] = lsum;
            __syncthreads();
            if (threadIdx.x == 0) {
                lsum = 0.f;
#pragma unroll
                for (int i = 0; i < RPB; ++i) lsum += red[i];
            }
        }
    }
    ce_epilogue<BS>(a, lsum, denom, red, ired);
}

// looped variant (cols > 16384): 2 reads + 1 write per element, one row per block iteration
__global__ __launch_bounds__(1024) void ce_looped_kernel(const CeArgs a) {
    __shared__ float red[16];
    __shared__ int ired[17];
    float scale, denom;
    ce_prologue<1024>(a, red, ired, scale, denom);
    float lsum = 0.f;
    for (int64_t row = blockIdx.x; row < a.rows; row += gridDim.x) {
        const int64_t label = load_label(a.labels, row, a.lbytes);
        const bool valid = label != a.ignore && label >= 0 && label < a.cols;
        const float wy = valid ? (a.cw ? a.cw[label] : 1.f) : 0.f;
        const float xl = valid ? a.logits[row * a.ld + label] : 0.f;
        const float* x = a.logits + row * a.ld;
        float m = -INFINITY;
        for (int64_t i = threadIdx.x; i < a.cols; i += 1024) m = fmaxf(m, x[i]);
        m = block_max<16>(m, red);
        float d = 0.f;
        for (int64_t i = threadIdx.x; i < a.cols; i += 1024) d += exp_fast_(x[i] - m);
        d = block_sum<16>(d, red);
        const float lse = m + logf(d);
        if (threadIdx.x == 0) {
            const float l = valid ? (lse - xl) * wy : 0.f;
            a.lse[row] = lse;
            a.loss_rows[row] = l;
            lsum += l;
        }
        __syncthreads();                                      // every thread has read xl before the row is overwritten
        float* o = a.dlogits + row * a.ld;
        const float gs = (a.mode == 1 ? scale : 1.f) * wy;
        const float invd = 1.0f / d;
        for (int64_t i = threadIdx.x; i < a.cols; i += 1024) {
            const float v = x[i];
            o[i] = valid ? (exp_fast_(v - m) * invd - (i == label ? 1.f : 0.f)) * gs : 0.f;
        }
    }
    ce_epilogue<1024>(a, lsum, denom, red, ired);
}

// Tall problems (rows * grid too large to count per block): the denominator as its own small launch.
// out_count[0] = #labels != ignore; out_denom[0] = that count, or the class-weight sum, as a float.
__global__ __launch_bounds__(1024) void ce_denominator_kernel(const void* __restrict__ labels, int lbytes, int64_t n,
                                                              int64_t ignore, const float* __restrict__ cw, int64_t cols,
                                                              int32_t* __restrict__ out_count, float* __restrict__ out_denom) {
    __shared__ int ired[16];
    __shared__ float red[16];
    int ci = 0;
    float ws = 0.f;
    if (lbytes == 4) count_labels<1024>(reinterpret_cast<const int32_t*>(labels), n, ignore, cw, cols, ci, ws);
    else if (lbytes == 8) count_labels<1024>(reinterpret_cast<const int64_t*>(labels), n, ignore, cw, cols, ci, ws);
    else count_labels<1024>(reinterpret_cast<const int16_t*>(labels), n, ignore, cw, cols, ci, ws);
    ci = block_sum_int<16>(ci, ired);
    if (cw) ws = block_sum<16>(ws, red);
    if (threadIdx.x == 0) {
        if (out_count) out_count[0] = ci;
        if (out_denom) out_denom[0] = cw ? ws : (float)ci;
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
    if (threadIdx.x == 0) out[0] = mean ? s / (float)ld_dev_i32(count_dev) : s;
}

// Whole CrossEntropyLoss of a SMALL problem in one single-block launch (one 1024-thread block, a wave per row): same
// arithmetic as the persistent kernel (the row sums run over a wave instead of a block, so the last bits of lse can
// differ by rounding); at MNIST-MLP scale (32 x 10) a 2048-block grid would be all launch overhead.
// The whole problem in one block of BS threads (one wave per row, waves striding the rows).
// Rows [r0, r1) of a small problem, one wave per row (waves striding the rows); xs / xld: where row r0 is READ and the pitch
// there (the logits themselves, or a copy of them in LDS).  Returns the wave's loss sum (meaningful on lane 0).
// ypre (optional): the labels of this wave's rows r0 + wave, + NWV, ... fetched by the caller ahead of time; then the range holds at
// most NPRE rows per wave (the loop has a compile-time trip count: ypre stays in registers).
template <int BS, int NPRE = 0>
__device__ __forceinline__ float ce_rows_range(const CeArgs& a, float scale, const float* xs, int64_t xld, int64_t r0, int64_t r1,
                                               const int64_t* ypre = nullptr) {
    constexpr int NWV = BS / 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float lsum = 0.f;
    auto one_row = [&](int64_t r, int64_t y) {
        const float* x = xs + (r - r0) * xld;
        float* dx = a.dlogits + r * a.ld;
        const bool live = y != a.ignore && y >= 0 && y < a.cols;   // same guard as the large kernels (cross_entropy.cu:176)
        const float wy = live ? (a.cw ? a.cw[y] : 1.f) : 0.f;
        float mx = -INFINITY;
        for (int64_t c = lane; c < a.cols; c += 64) mx = fmaxf(mx, x[c]);
        mx = wave_max(mx);
        float se = 0.f;
        for (int64_t c = lane; c < a.cols; c += 64) se += exp_fast_(x[c] - mx);
        se = wave_sum(se);
        const float lse = mx + logf(se);
        // x and dx may alias (in-place mode): every lane holds x[y] before any lane of this wave stores
        const float xy = live ? x[y] : 0.f;
        const float inv = 1.0f / se;
        const float gs = (a.mode == 1 ? scale : 1.f) * wy;
        __builtin_amdgcn_wave_barrier();
        for (int64_t c = lane; c < a.cols; c += 64) {
            const float pr = exp_fast_(x[c] - mx) * inv;
            dx[c] = live ? (pr - (c == y ? 1.f : 0.f)) * gs : 0.f;
        }
        if (lane == 0) {
            const float l = live ? (lse - xy) * wy : 0.f;
            a.loss_rows[r] = l;
            a.lse[r] = lse;
            lsum += l;
        }
    };
    if constexpr (NPRE > 0) {
#pragma unroll
        for (int j = 0; j < NPRE; ++j) {
            const int64_t r = r0 + wave + (int64_t)j * NWV;
            if (r < r1) one_row(r, ypre[j]);
        }
    } else {
        for (int64_t r = r0 + wave; r < r1; r += NWV) one_row(r, load_label(a.labels, r, a.lbytes));
    }
    return lsum;
}

// The whole problem in one block.
__global__ __launch_bounds__(1024) void ce_small_kernel(const CeArgs a) {
    __shared__ int ired[17];
    __shared__ float red[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float scale, denom;
    ce_prologue<1024>(a, red, ired, scale, denom);
    const float lsum = ce_rows_range<1024>(a, scale, a.logits, a.ld, 0, a.rows);
    if (!a.loss_out) return;
    __syncthreads();
    if (lane == 0) red[wave] = lsum;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int i = 0; i < 16; ++i) t += red[i];
        a.loss_out[0] = a.mode == 1 ? t / denom : t;
    }
}

// A small classifier head and its loss in ONE launch: block b computes rows [16b, 16b + 16) of logits = X W^T + b with
// gemm_small's tile body (NW waves, as gemm_small() picks them -- 4 for K < 113, else 8: bit-identical to
// nnhipLinearModuleForward's), keeps them in LDS and runs the loss rows on that copy; the label count (the 'mean' denominator)
// is taken first by every block -- its loads fly while the GEMM's do -- and the blocks' loss sums meet in ce_epilogue (ticket,
// fixed order).  At MNIST-MLP scale a launch is ~4.7 us of a 35 us step whatever it computes.  rows <= 256, classes <= 32
// (nnhipLinearCrossEntropyLoss).
template <bool VEC, int NW>
__global__ __launch_bounds__(NW * 64) void linear_ce_small_kernel(const SmallGemmParams p, const CeArgs a) {
    constexpr int BS = NW * 64;
    __shared__ float gred[NW][16 * 16];
    __shared__ float ared[NW][16];
    __shared__ float tile[16 * 32];
    __shared__ int ired[17];
    __shared__ float red[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int by = (int)blockIdx.x;
    const int64_t r0 = (int64_t)by * 16, r1 = min(r0 + 16, a.rows);
    // Every label this block needs is requested BEFORE the GEMM's operands: the label of row `tid` for the 'mean' denominator
    // (rows <= 256 < BS... or <= BS: one label per thread) and the labels of the rows this wave will finish.  Where they were
    // loaded at their point of use, the count and each row's label were dependent L2 round trips in front of / behind the GEMM.
    constexpr int NPRE = 16 / NW;
    const bool counting = a.mode == 1 && a.count_in_kernel && a.rows <= BS;
    int64_t lcount = a.ignore;
    if (counting) lcount = load_label(a.labels, (int64_t)threadIdx.x < a.rows ? (int64_t)threadIdx.x : a.rows - 1, a.lbytes);
    int64_t ypre[NPRE];
#pragma unroll
    for (int j = 0; j < NPRE; ++j) {
        const int64_t r = r0 + wave + (int64_t)j * NW;
        ypre[j] = load_label(a.labels, r < r1 ? r : r1 - 1, a.lbytes);
    }
    // (at most one k-group per wave -- a 128-wide hidden layer on 8 waves: the 1-group variant of the tile.  The 8-group body
    //  multiplies the seven groups past the end as zeros: 28 dependent MFMAs, ~0.4 us on the tail of the tile)
    const bool one_group = ((p.K + 15) >> 4) <= NW;
    for (int bx = 0; (int64_t)bx * 16 < p.N; ++bx) {
        if (one_group) sg_tile16<NW, true, true, VEC, 1>(p, bx, by, gred, ared, tile - r0 * 32);
        else sg_tile16<NW, true, true, VEC, 8>(p, bx, by, gred, ared, tile - r0 * 32);   // row r of C also lands at tile[(r - r0) * 32 + col]
        __syncthreads();                                   // gred is reused by the next tile; after the last one: tile is complete
    }
    float scale, denom;
    if (counting) {                                        // ce_prologue's count_in_kernel branch on the prefetched labels
        const bool cnt = (int64_t)threadIdx.x < a.rows && lcount != a.ignore && lcount >= 0 && lcount < a.cols;   // count_labels' predicate
        float ws = 0.f;
        if (a.cw && cnt) ws = a.cw[lcount];
        const int ci = block_sum_int<BS / 64>(cnt ? 1 : 0, ired);
        if (a.cw) ws = block_sum<BS / 64>(ws, red);
        denom = a.cw ? ws : (float)ci;
        if (a.count_out && blockIdx.x == 0 && threadIdx.x == 0) a.count_out[0] = ci;
        scale = denom > 0.f ? 1.0f / denom : 0.0f;
    } else {
        ce_prologue<BS>(a, red, ired, scale, denom);
    }
    const float lsum = ce_rows_range<BS, NPRE>(a, scale, tile, 32, r0, r1, ypre);
    if (!a.loss_out) return;
    __syncthreads();
    if (lane == 0) red[wave] = lsum;
    __syncthreads();
    float t = 0.f;
    if (threadIdx.x == 0)
        for (int i = 0; i < NW; ++i) t += red[i];
    __syncthreads();                                       // ce_epilogue reuses red / ired
    ce_epilogue<BS>(a, t, denom, red, ired);
}

// ---- BatchNorm2d (training) -> flatten -> Linear(C*HW -> N <= 16) -> Sigmoid -> MSELoss: the head kernel -------------------
// The tail of the reference's conv classifier (examples/convolutional_digits_classifier.ipynb cell 2).  No block waits for another
// (round 5, first attempt: one block per channel + a ticket hand-off of the logits' shares -- 22 us, EXPERIMENTS.md): the layer in
// front (nnhipConv2dLeakyMaxPoolForwardStats) has left per-block (mean, M2) pairs of its pooled output; EVERY block of this kernel
// combines them to the batch statistics (Chan et al., fixed order: the same bits in every block) while its GEMM operands are on
// their way, normalises the A operand between load and MFMA (sg_tile16's ATR hook) -- writing the BatchNorm output as it passes --
// and finishes 16 rows of prediction, d(loss)/dz and squared error; only the scalar loss meets through ce_epilogue's ticket.
struct BnHeadArgs {
    const float* stats;        // [nstat][C][2]: (mean, M2) of `count` values each
    const float* bn_w; const float* bn_b;
    float* Y; float* save_mean; float* save_inv; float* run_mean; float* run_var;
    const float* target; float* dz;
    int nstat, C, HW;
    float count, eps, momentum, invN, inv_hw;
};
struct BnApplyA {
    const float4* tab;         // LDS: per channel {mean, inv, weight, bias}
    float* Y;
    int HW, C;
    int64_t ld;
    float inv_hw;
    __device__ __forceinline__ float one(float x, int c) const {
        const float4 t = tab[c < C ? c : C - 1];
        return t.z * ((x - t.x) * t.y) + t.w;
    }
    __device__ __forceinline__ void operator()(float4& a, int64_t row, unsigned k0, bool ok) const {
        unsigned c0 = (unsigned)((float)k0 * inv_hw);
        if (c0 * (unsigned)HW > k0) --c0;
        if ((c0 + 1) * (unsigned)HW <= k0) ++c0;
        unsigned r = k0 - c0 * (unsigned)HW;
        float v[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (i) {
                ++r;
#pragma unroll
                for (int q = 0; q < 1; ++q)
                    if (r >= (unsigned)HW) { r -= (unsigned)HW; ++c0; }
            }
            v[i] = one(v[i], (int)c0);
        }
        a = make_float4(v[0], v[1], v[2], v[3]);
        if (ok) *reinterpret_cast<float4*>(Y + row * ld + k0) = a;
    }
};
template <int NW>
__global__ __launch_bounds__(NW * 64) void bn_linear_sigmoid_mse_kernel(const SmallGemmParams p, const BnHeadArgs h, const CeArgs a) {
    constexpr int BS = NW * 64, TPC = BS / 16;              // threads per channel in the statistics phase (C <= 16)
    __shared__ float gred[NW][16 * 16];
    __shared__ float ared[NW][16];
    __shared__ float tile[16 * 32];
    __shared__ __attribute__((aligned(16))) float4 tab[16];
    __shared__ int ired[17];
    __shared__ float red[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int by = (int)blockIdx.x;
    const int64_t r0 = (int64_t)by * 16;
    // the target of the element this thread finishes: requested first
    const int erow = tid >> 4, ecol = tid & 15;
    const bool emine = tid < 256 && r0 + erow < p.M && ecol < p.N;
    const float tv = emine ? h.target[(r0 + erow) * p.N + ecol] : 0.f;
    // ---- batch statistics from the producer's per-block pairs ----
    {
        const int c = tid / TPC, j = tid % TPC;
        const bool cok = c < h.C;
        float n = 0.f, m = 0.f, M2 = 0.f;
        auto merge = [&](float nb, float mb, float Mb) {
            if (nb == 0.f) return;
            if (n == 0.f) { n = nb; m = mb; M2 = Mb; return; }
            const float nt = n + nb, d = mb - m;
            m += d * (nb / nt);
            M2 += Mb + d * d * (n * nb / nt);
            n = nt;
        };
        for (int base = 0; base < h.nstat; base += 8 * TPC) {
            float mb[8], Mb[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int q = base + u * TPC + j;
                const float2 v = *reinterpret_cast<const float2*>(h.stats + ((int64_t)(q < h.nstat ? q : 0) * h.C + (cok ? c : 0)) * 2);
                mb[u] = v.x; Mb[u] = v.y;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (base + u * TPC + j < h.nstat) merge(h.count, mb[u], Mb[u]);
        }
#pragma unroll
        for (int off = TPC / 2; off >= 1; off >>= 1) {
            const float nb = __shfl_xor(n, off, 64), mb2 = __shfl_xor(m, off, 64), Mb2 = __shfl_xor(M2, off, 64);
            if ((j & off) == 0) merge(nb, mb2, Mb2);         // the lower lane of a pair keeps the merged triple (fixed order)
        }
        if (cok && j == 0) {
            const float var = M2 / n;
            const float inv = 1.0f / sqrtf(var + h.eps);
            tab[c] = make_float4(m, inv, h.bn_w ? h.bn_w[c] : 1.f, h.bn_w ? h.bn_b[c] : 0.f);
            if (by == 0) {
                h.save_mean[c] = m;
                h.save_inv[c] = inv;
                if (h.run_mean) {
                    h.run_mean[c] = h.momentum * h.run_mean[c] + (1.0f - h.momentum) * m;
                    h.run_var[c] = h.momentum * h.run_var[c] + (1.0f - h.momentum) * var;
                }
            }
        }
    }
    __syncthreads();
    BnApplyA atr{tab, h.Y, h.HW, h.C, p.lda, h.inv_hw};
    for (int bx = 0; (int64_t)bx * 16 < p.N; ++bx) {
        if (bx == 0) sg_tile16<NW, true, true, true, 8, SgNoPost, BnApplyA>(p, bx, by, gred, ared, tile - r0 * 32, SgNoPost(), atr);
        else sg_tile16<NW, true, true, true, 8>(p, bx, by, gred, ared, tile - r0 * 32);      // (N <= 16: never taken)
        __syncthreads();
    }
    float lsum = 0.f;
    if (emine) {
        const float pr = tile[erow * 32 + ecol];
        const float d = pr - tv;
        h.dz[(r0 + erow) * p.N + ecol] = (2.0f * d * h.invN) * pr * (1.0f - pr);
        lsum = d * d;
    }
    lsum = wave_sum(lsum);
    __syncthreads();
    if (lane == 0) red[wave] = lsum;
    __syncthreads();
    float t = 0.f;
    if (tid == 0)
        for (int i = 0; i < NW; ++i) t += red[i];
    __syncthreads();
    ce_epilogue<BS>(a, t, (float)((double)p.M * (double)p.N), red, ired);
}

// =================================================================================================
// Column sums:  out[c] = sum_r X[r*ld + c]     (Linear db: neunet/nn/layers/linear.py:24)
// Block = 4 waves; lane <-> one float4 column group (256 columns per block) or one column (scalar path,
// 64 columns per block); wave w sums rows w, w+4, ... of the block's row range (1 KiB coalesced per
// wave-load); the 4 waves meet in LDS.  grid (col_blocks, row_blocks); row_blocks > 1 writes partials
// [row_blocks][cols] that colsum_tall_kernel finishes (one slice of columns per block, all rows in flight).
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
    int64_t rb = 2048 / p.col_blocks;             // aim for ~2k blocks
    if (rb > ceil_div(rows, 16)) rb = ceil_div(rows, 16);  // >= 16 rows per block
    if (rb > 512) rb = 512;                        // the tall finish sums <= 512 partial rows per column
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
    if (p.row_blocks > 1)
        return colsum_tall(stage1, out, nullptr, nullptr, (int64_t)p.row_blocks, cols, p.vec && aligned16(stage1), st);
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

// ---- attention-score softmax for key counts beyond the register tile (Tk > 16384): looped, one block per row ------
__global__ __launch_bounds__(1024) void softmax_masked_fwd_looped(float* out, const float* in,
                                                                  const int32_t* __restrict__ key_valid, int64_t cols,
                                                                  int64_t HTq, int64_t Tq, float scale, int causal,
                                                                  const int32_t* __restrict__ dense) {
    __shared__ float red[16];
    const int64_t row = blockIdx.x;
    const int64_t b = row / HTq, i = row % Tq;
    const int64_t lim = causal ? i + (cols - Tq) : cols;
    const int32_t* kv = key_valid ? key_valid + b * cols : nullptr;
    const int32_t* dm = dense ? dense + (b * Tq + i) * cols : nullptr;
    const float* x = in + row * cols;
    float* o = out + row * cols;
    auto val = [&](int64_t c) { return (c > lim || (kv && kv[c] == 0) || (dm && dm[c] == 0)) ? -1e9f : x[c] * scale; };
    float m = -INFINITY;
    for (int64_t c = threadIdx.x; c < cols; c += 1024) m = fmaxf(m, val(c));
    m = block_max<16>(m, red);
    float d = 0.f;
    for (int64_t c = threadIdx.x; c < cols; c += 1024) d += exp_fast_(val(c) - m);
    d = block_sum<16>(d, red);
    const float inv = 1.0f / d;
    // out may alias in: element c is read and written by the same thread
    for (int64_t c = threadIdx.x; c < cols; c += 1024) o[c] = exp_fast_(val(c) - m) * inv;
}
__global__ __launch_bounds__(1024) void softmax_masked_bwd_looped(float* dx, const float* dy, const float* __restrict__ y,
                                                                  const int32_t* __restrict__ key_valid, int64_t cols,
                                                                  int64_t HTq, int64_t Tq, float scale, int causal,
                                                                  const int32_t* __restrict__ dense) {
    __shared__ float red[16];
    const int64_t row = blockIdx.x;
    const int64_t b = row / HTq, i = row % Tq;
    const int64_t lim = causal ? i + (cols - Tq) : cols;
    const int32_t* kv = key_valid ? key_valid + b * cols : nullptr;
    const int32_t* dm = dense ? dense + (b * Tq + i) * cols : nullptr;
    const float* g = dy + row * cols;
    const float* f = y + row * cols;
    float s = 0.f;
    for (int64_t c = threadIdx.x; c < cols; c += 1024) s += g[c] * f[c];
    s = block_sum<16>(s, red);
    for (int64_t c = threadIdx.x; c < cols; c += 1024) {
        const bool masked = c > lim || (kv && kv[c] == 0) || (dm && dm[c] == 0);
        dx[row * cols + c] = masked ? 0.f : (g[c] - s) * f[c] * scale;
    }
}

// Streaming loads for a [rows, cols] operand with row pitch ld?  Only when the tensor cannot be cache-resident anyway
// (>= 128 MB: the producer's output is long gone from the 32 MB of L2 and mostly from the 256 MB MALL) and its rows are
// whole 128-B lines (see RowTile::load).  NNHIP_ROW_NT=0/1 forces it off/on (developer switch).
static int row_streaming(int64_t rows, int64_t cols, int64_t ld) {
    static const int forced = []() { const char* e = getenv("NNHIP_ROW_NT"); return e ? atoi(e) : -1; }();
    if (forced >= 0) return forced;
    return (rows * cols * 4 >= ((int64_t)128 << 20) && ((ld * 4) & 127) == 0) ? 1 : 0;
}

// Row-kernel dispatch: pick (TPR, NV) from the row width.
#define ROW_DISPATCH(KERNEL, cols, vec, rows, st, ...)                                              \
    do {                                                                                            \
        const int64_t _c = (cols);                                                                  \
        if (_c <= 256) { ROW_LAUNCH(KERNEL, 64, 1, vec, rows, st, __VA_ARGS__); }                   \
        else if (_c <= 512) { ROW_LAUNCH(KERNEL, 64, 2, vec, rows, st, __VA_ARGS__); }              \
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

// Persistent variants: the caller fixes the number of blocks.
#define ROW_DISPATCH_GRID(KERNEL, cols, vec, nblk, st, ...)                                         \
    do {                                                                                            \
        const int64_t _c = (cols);                                                                  \
        if (_c <= 256) { ROW_LAUNCH_GRID(KERNEL, 64, 1, vec, nblk, st, __VA_ARGS__); }              \
        else if (_c <= 512) { ROW_LAUNCH_GRID(KERNEL, 64, 2, vec, nblk, st, __VA_ARGS__); }         \
        else if (_c <= 1024) { ROW_LAUNCH_GRID(KERNEL, 64, 4, vec, nblk, st, __VA_ARGS__); }        \
        else if (_c <= 4096) { ROW_LAUNCH_GRID(KERNEL, 256, 4, vec, nblk, st, __VA_ARGS__); }       \
        else if (_c <= 8192) { ROW_LAUNCH_GRID(KERNEL, 256, 8, vec, nblk, st, __VA_ARGS__); }       \
        else { ROW_LAUNCH_GRID(KERNEL, 1024, 4, vec, nblk, st, __VA_ARGS__); }                      \
    } while (0)
#define ROW_LAUNCH_GRID(KERNEL, TPR, NV, vec, nblk, st, ...)                                        \
    do {                                                                                            \
        constexpr int _bs = (TPR >= 256) ? TPR : 256;                                               \
        if (vec) hipLaunchKernelGGL((KERNEL<TPR, NV, true>), dim3((unsigned)(nblk)), dim3(_bs), 0, st, __VA_ARGS__);  \
        else hipLaunchKernelGGL((KERNEL<TPR, NV, false>), dim3((unsigned)(nblk)), dim3(_bs), 0, st, __VA_ARGS__);     \
    } while (0)

// Resident blocks of a kernel instantiation on the whole device (occupancy query, cached per instantiation): the
// persistent kernels size their grids in multiples of it so no block waits for a slot behind a full-length one.
static int resident_slots_impl(const void* kernel, int bs) {
    static std::mutex mu;
    static std::unordered_map<const void*, int> cache;
    static int cus = 0;
    std::lock_guard<std::mutex> lk(mu);
    auto it = cache.find(kernel);
    if (it != cache.end()) return it->second;
    if (cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        cus = 256;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
            cus = prop.multiProcessorCount;
    }
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, bs, 0) != hipSuccess || per_cu < 1) per_cu = 1;
    cache[kernel] = per_cu * cus;
    return per_cu * cus;
}
template <class K>
static int resident_slots(K kernel, int bs) { return resident_slots_impl(reinterpret_cast<const void*>(kernel), bs); }
#define ROW_DISPATCH_SLOTS(KERNEL, cols, vec, OUT)                                                  \
    do {                                                                                            \
        const int64_t _c = (cols);                                                                  \
        if (_c <= 256) { ROW_SLOTS(KERNEL, 64, 1, vec, OUT); }                                      \
        else if (_c <= 512) { ROW_SLOTS(KERNEL, 64, 2, vec, OUT); }                                 \
        else if (_c <= 1024) { ROW_SLOTS(KERNEL, 64, 4, vec, OUT); }                                \
        else if (_c <= 4096) { ROW_SLOTS(KERNEL, 256, 4, vec, OUT); }                               \
        else if (_c <= 8192) { ROW_SLOTS(KERNEL, 256, 8, vec, OUT); }                               \
        else { ROW_SLOTS(KERNEL, 1024, 4, vec, OUT); }                                              \
    } while (0)
#define ROW_SLOTS(KERNEL, TPR, NV, vec, OUT)                                                        \
    do {                                                                                            \
        constexpr int _bs = (TPR >= 256) ? TPR : 256;                                               \
        OUT = (vec) ? resident_slots(KERNEL<TPR, NV, true>, _bs) : resident_slots(KERNEL<TPR, NV, false>, _bs); \
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
        ROW_DISPATCH(softmax_fwd_rows, slice_size, vec, num_slices, st, out, in, num_slices, slice_size, row_streaming(num_slices, slice_size, slice_size));
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
        ROW_DISPATCH(softmax_bwd_rows, slice_size, vec, num_slices, st, dX, dY, Y, num_slices, slice_size, row_streaming(num_slices, slice_size, slice_size));
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
    hipStream_t st = (hipStream_t)s;
    if (cols > kMaxRegRow) {
        hipLaunchKernelGGL(rmsnorm_fwd_looped, dim3((unsigned)rows), dim3(1024), 0, st, X, weight, bias, Y, X_std, X_norm, cols, eps);
    } else {
        const bool vec = aligned16(X) && aligned16(Y) && aligned16(weight) && (!bias || aligned16(bias)) &&
                         (!X_norm || aligned16(X_norm)) && cols % 4 == 0;
        ROW_DISPATCH(rmsnorm_fwd_rows, cols, vec, rows, st, X, weight, bias, Y, X_std, X_norm, rows, cols, eps, row_streaming(rows, cols, cols));
    }
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
    hipStream_t st = (hipStream_t)s;
    if (cols > kMaxRegRow) {
        // wide rows: looped dX kernel + a column pass for dw/db (partials over row chunks, then the in-launch column sum)
        if (rows > 0) {
            hipLaunchKernelGGL(rmsnorm_bwd_dx_looped, dim3((unsigned)rows), dim3(1024), 0, st, dY, X, weight, X_std, dX, cols, dX_addend);
            NNHIP_LAUNCH_CHECK("rmsnorm_bwd_dx_looped");
        }
        const int64_t col_blocks = ceil_div(cols, 256);
        int64_t ry = 1024 / col_blocks;
        if (ry > ceil_div(rows > 0 ? rows : 1, 8)) ry = ceil_div(rows > 0 ? rows : 1, 8);
        if (ry < 1) ry = 1;
        const int64_t rpb = ceil_div(rows > 0 ? rows : 1, ry);
        ry = ceil_div(rows > 0 ? rows : 1, rpb);
        const size_t pf = ((size_t)ry * cols + 3) / 4 * 4;
        float* part = static_cast<float*>(workspace(pf * (db ? 2 : 1) * sizeof(float) + 2 * (size_t)cols * 64 * sizeof(float)));
        NNHIP_CHECK_ARG(part != nullptr, NNHIP_ENOMEM, "nnhipRMSNormBackward: workspace allocation failed");
        float* pdb = db ? part + pf : nullptr;
        hipLaunchKernelGGL(rmsnorm_bwd_dwdb_cols, dim3((unsigned)col_blocks, (unsigned)ry), dim3(256), 0, st, dY, X, X_std, part, pdb, rows, cols, rpb);
        NNHIP_LAUNCH_CHECK("rmsnorm_bwd_dwdb_cols");
        float* scr = part + pf * (db ? 2 : 1);
        {
            const ColsumPlan cp = colsum_plan(part, dW, ry, cols, cols);
            NNHIP_CHECK_ARG(cp.scratch_floats <= (size_t)cols * 64, NNHIP_ENOMEM, "nnhipRMSNormBackward: column-sum scratch too small");
            if (int rc = colsum_run(cp, part, ry, cols, cols, dW, scr, st)) return rc;
        }
        if (db) {
            const ColsumPlan cp = colsum_plan(pdb, db, ry, cols, cols);
            if (int rc = colsum_run(cp, pdb, ry, cols, cols, db, scr, st)) return rc;
        }
        return 0;
    }
    const bool vec = aligned16(dY) && aligned16(X) && aligned16(weight) && aligned16(dX) && aligned16(dX_addend) &&
                     aligned16(dW) && aligned16(db) && cols % 4 == 0;
    // persistent grid = the blocks that are resident at once (occupancy query), each accumulating dw/db partials over
    // its rows
    const int rpb = cols <= 1024 ? 4 : 1;
    int64_t nblk = ceil_div(rows > 0 ? rows : 1, rpb);
    int slots = 1024;
    ROW_DISPATCH_SLOTS(rmsnorm_bwd_rows, cols, vec, slots);
    {
        static const double mult = []() { const char* e = getenv("NNHIP_RMS_GRID_MULT"); return e ? atof(e) : 1.0; }();
        slots = (int)(slots * mult);
        if (slots < 1) slots = 1;
    }
    if (nblk > slots) nblk = slots;
    const size_t part_floats = ((size_t)nblk * cols + 63) / 64 * 64;
    // inside Tensor.backward() the finishing column sums wait for the flush (one launch for every RMSNorm of the pass)
    bool deferred = false;
    int rcq = 0;
    float* part = colsum_partials(part_floats * (db ? 2 : 1), db ? 2 : 1, fin_sw(cols, vec), vec, st, &deferred, &rcq);
    if (rcq) return rcq;
    if (!part) part = static_cast<float*>(workspace(part_floats * (db ? 2 : 1) * sizeof(float)));
    NNHIP_CHECK_ARG(part != nullptr, NNHIP_ENOMEM, "nnhipRMSNormBackward: workspace allocation failed");
    float* part_dw = part;
    float* part_db = db ? part + part_floats : nullptr;
    ROW_DISPATCH_GRID(rmsnorm_bwd_rows, cols, vec, nblk, st, dY, X, weight, X_std, dX, part_dw, part_db, rows, cols, dX_addend, row_streaming(rows, cols, cols));
    NNHIP_LAUNCH_CHECK("rmsnorm_backward");
    if (deferred) {
        colsum_queue(part_dw, dW, nblk, cols);
        if (db) colsum_queue(part_db, db, nblk, cols);
        return 0;
    }
    return colsum_tall(part_dw, dW, part_db, db, nblk, cols, vec, st);
}

// ---- CrossEntropy -------------------------------------------------------------------------------
namespace nnhip {
// One launch (two for tall-and-narrow 'mean' problems whose per-block label count would not be free).
static int ce_launch(CeArgs a, hipStream_t st) {
    unsigned* sync = sync_words();
    if (!sync) { set_last_error("cross entropy: sync words allocation failed"); return NNHIP_ENOMEM; }
    if (int rc = serialize_shared_state(st)) return rc;     // one ticket word / partial array per process: see runtime.hip
    const SharedStateUse in_use(st);
    a.sync = sync + SYNC_CE;
    a.nt = row_streaming(a.rows, a.cols, a.ld);
    float* denom_scratch = reinterpret_cast<float*>(sync + SYNC_CE + 4);
    const bool need_denom = a.mode == 1 && !a.count_dev && a.scale_host < 0.f;   // scale_host < 0: "derive it from the labels"
    if (a.rows * a.cols <= 65536 && a.cols <= 4096) {
        a.count_in_kernel = need_denom ? 1 : 0;
        hipLaunchKernelGGL(ce_small_kernel, dim3(1), dim3(1024), 0, st, a);
        NNHIP_LAUNCH_CHECK("ce_small_kernel");
        return 0;
    }
    const bool looped = a.cols > kMaxRegRow;
    const bool vec = aligned16(a.logits) && aligned16(a.dlogits) && a.cols % 4 == 0 && a.ld % 4 == 0;
    const int rpb = (!looped && a.cols <= 1024) ? 4 : 1;
    int64_t nblk = ceil_div(a.rows, rpb);
    // persistent grid = the blocks resident at once: the per-block label count and the final partial reduction stay
    // negligible
    int slots = 512;
    if (looped) slots = resident_slots(ce_looped_kernel, 1024);
    else ROW_DISPATCH_SLOTS(ce_rows_kernel, a.cols, vec, slots);
    if (nblk > (int64_t)slots) nblk = (int64_t)slots;       // one resident round: equal shares, one prologue per slot
    if (a.loss_out) {
        a.partial = static_cast<float*>(workspace((size_t)nblk * sizeof(float)));
        if (!a.partial) { set_last_error("cross entropy: workspace allocation failed"); return NNHIP_ENOMEM; }
    }
    a.count_in_kernel = 0;
    if (need_denom) {
        // every block re-reads the label vector out of L2: grid x rows x 4 B, ~7 us for 32 MB at 8192 rows x 1024 blocks.
        // (Tried in round 3, both slower: blocks counting one slice each and meeting in a device-wide counter -- one
        //  arrival word: 200 us for 1792 blocks, read-modify-writes on one address retire at ~25 M/s; 64 sharded 64-bit
        //  words polled by one lane each: +20 us, a grid barrier costs the launch ramp of the last block.)  The cost grows
        //  with the grid, so a counting launch caps its grid at four blocks per compute unit.
        //  Also slower: 2 / 8 / 32 "counter" blocks that publish {count, flag} while every other block polls the flag after issuing
        //  its first rows' loads (falls back to its own count): 56.1 / 56.1 / 58.6 us against 53.1 at 8192 x 4096.)
        const int64_t capped = nblk > 1024 ? 1024 : nblk;
        if ((double)capped * (double)a.rows * a.lbytes <= 64.0 * 1024 * 1024) {
            a.count_in_kernel = 1;
            nblk = capped;
        } else {
            hipLaunchKernelGGL(ce_denominator_kernel, dim3(1), dim3(1024), 0, st, a.labels, a.lbytes, a.rows, a.ignore, a.cw,
                               a.cols, a.count_out, denom_scratch);
            NNHIP_LAUNCH_CHECK("ce_denominator_kernel");
            a.denom_dev = denom_scratch;
        }
    }
    if (looped) {
        hipLaunchKernelGGL(ce_looped_kernel, dim3((unsigned)nblk), dim3(1024), 0, st, a);
    } else {
        ROW_DISPATCH_GRID(ce_rows_kernel, a.cols, vec, nblk, st, a);
    }
    NNHIP_LAUNCH_CHECK("cross_entropy");
    return 0;
}
}  // namespace nnhip

extern "C" int nnhipCrossEntropyLossEx(float* logits, float* dlogits_or_null, float* loss_rows, float* lse,
                                       const void* labels, int32_t label_bytes, const float* class_weight_or_null,
                                       int64_t logits_stride, int64_t ignore_index, int64_t n_rows, int64_t n_cols,
                                       char reduction, float* loss_out_or_null, int32_t* count_out_or_null,
                                       nnhipStream_t s) {
    NNHIP_CHECK_ARG(n_rows >= 0 && n_cols >= 0 && logits_stride >= n_cols, NNHIP_EINVAL, "nnhipCrossEntropyLossEx: bad sizes");
    NNHIP_CHECK_ARG(reduction == 'n' || reduction == 'm' || reduction == 's', NNHIP_EINVAL,
                    "nnhipCrossEntropyLossEx: reduction must be 'n', 'm' or 's'");
    NNHIP_CHECK_ARG(label_bytes == 2 || label_bytes == 4 || label_bytes == 8, NNHIP_EINVAL,
                    "nnhipCrossEntropyLossEx: labels must be int16, int32 or int64");
    NNHIP_CHECK_ARG(reduction == 'n' || loss_out_or_null, NNHIP_EINVAL, "nnhipCrossEntropyLossEx: 'm'/'s' need loss_out");
    if (n_rows == 0 || n_cols == 0) {
        // sum over nothing = 0; mean over nothing = 0/0 (NumPy gives nan): fill without reading anything
        if (loss_out_or_null) {
            // a fill kernel, not hipMemcpyAsync from this stack frame: the latter is not capturable into a hipGraph (the
            // node would keep a dangling host pointer)
            const float v = reduction == 'm' ? __builtin_nanf("") : 0.f;
            const int rc = fill_f32(loss_out_or_null, v, 1, (hipStream_t)s);
            if (rc) return rc;
        }
        return 0;
    }
    NNHIP_CHECK_ARG(logits && loss_rows && lse && labels, NNHIP_EINVAL, "nnhipCrossEntropyLossEx: null pointer");
    CeArgs a{};
    a.logits = logits;
    a.dlogits = dlogits_or_null ? dlogits_or_null : logits;   // reference behaviour: overwrite logits with the gradient
    a.loss_rows = loss_rows; a.lse = lse; a.labels = labels; a.cw = class_weight_or_null;
    a.ld = logits_stride; a.ignore = ignore_index; a.rows = n_rows; a.cols = n_cols;
    a.lbytes = label_bytes;
    a.mode = reduction == 'm' ? 1 : (reduction == 's' ? 2 : 0);
    a.scale_host = a.mode == 1 ? -1.f : 1.f;
    a.loss_out = a.mode ? loss_out_or_null : nullptr;
    a.count_out = count_out_or_null;
    return ce_launch(a, (hipStream_t)s);
}

// logits = X W^T + b (written to `logits`, [rows, classes] dense) followed by nnhipCrossEntropyLossEx on them, in ONE launch:
// Linear.forward (neunet/nn/layers/linear.py:48-58) + CrossEntropyLoss (losses.py:59-126) for a small classifier head.
// Limits: rows <= 256, 1 <= classes <= 32, in_features <= 2048; NNHIP_EINVAL outside them (the caller then runs the two
// entries separately -- same results).
extern "C" int nnhipLinearCrossEntropyLoss(const float* X, const float* W, const float* b, float* logits, float* dlogits,
                                           float* loss_rows, float* lse, const void* labels, int32_t label_bytes,
                                           const float* class_weight_or_null, int64_t ignore_index, int64_t rows,
                                           int64_t in_features, int64_t classes, char reduction, float* loss_out_or_null,
                                           int32_t* count_out_or_null, nnhipStream_t s) {
    NNHIP_CHECK_ARG(rows >= 1 && rows <= 256 && classes >= 1 && classes <= 32 && in_features >= 1 && in_features <= 2048, NNHIP_EINVAL,
                    "nnhipLinearCrossEntropyLoss: needs 1 <= rows <= 256, 1 <= classes <= 32, 1 <= in_features <= 2048");
    NNHIP_CHECK_ARG(reduction == 'n' || reduction == 'm' || reduction == 's', NNHIP_EINVAL,
                    "nnhipLinearCrossEntropyLoss: reduction must be 'n', 'm' or 's'");
    NNHIP_CHECK_ARG(label_bytes == 2 || label_bytes == 4 || label_bytes == 8, NNHIP_EINVAL,
                    "nnhipLinearCrossEntropyLoss: labels must be int16, int32 or int64");
    NNHIP_CHECK_ARG(reduction == 'n' || loss_out_or_null, NNHIP_EINVAL, "nnhipLinearCrossEntropyLoss: 'm'/'s' need loss_out");
    NNHIP_CHECK_ARG(X && W && logits && dlogits && logits != dlogits && loss_rows && lse && labels, NNHIP_EINVAL,
                    "nnhipLinearCrossEntropyLoss: null or aliased pointer");
    NNHIP_CHECK_ARG(aligned4(X) && aligned4(W) && aligned4(b) && aligned4(logits) && aligned4(dlogits), NNHIP_EALIGN,
                    "nnhipLinearCrossEntropyLoss: misaligned pointer");
    SmallGemmParams p{};
    p.A = X; p.B = W; p.C = logits; p.bias = b; p.M = rows; p.N = classes; p.K = in_features;
    p.lda = in_features; p.ldb = in_features; p.ldc = classes; p.alpha = 1.f; p.beta = 1.f; p.a_kmajor = 1; p.b_kmajor = 1;
    CeArgs a{};
    a.logits = logits; a.dlogits = dlogits; a.loss_rows = loss_rows; a.lse = lse; a.labels = labels; a.cw = class_weight_or_null;
    a.ld = classes; a.ignore = ignore_index; a.rows = rows; a.cols = classes; a.lbytes = label_bytes;
    a.mode = reduction == 'm' ? 1 : (reduction == 's' ? 2 : 0);
    a.scale_host = a.mode == 1 ? -1.f : 1.f;
    a.loss_out = a.mode ? loss_out_or_null : nullptr;
    a.count_out = count_out_or_null;
    a.count_in_kernel = a.mode == 1 ? 1 : 0;
    const unsigned nblk = (unsigned)ceil_div(rows, 16);    // one block per 16 rows
    if (a.loss_out) {
        unsigned* sync = sync_words();
        a.partial = static_cast<float*>(workspace((size_t)nblk * sizeof(float)));
        if (!sync || !a.partial) { set_last_error("nnhipLinearCrossEntropyLoss: workspace allocation failed"); return NNHIP_ENOMEM; }
        if (int rc = serialize_shared_state((hipStream_t)s)) return rc;
        a.sync = sync + SYNC_CE;
    }
    const SharedStateUse in_use((hipStream_t)s);      // (harmless when the loss is not reduced: an event behind the launch)
    const bool vec = (in_features & 3) == 0 && aligned16(X) && aligned16(W);
    const bool nw8 = ((in_features + 15) >> 4) >= 8;       // gemm_small()'s choice
    hipStream_t st = (hipStream_t)s;
    if (nw8) {
        if (vec) hipLaunchKernelGGL((linear_ce_small_kernel<true, 8>), dim3(nblk), dim3(512), 0, st, p, a);
        else hipLaunchKernelGGL((linear_ce_small_kernel<false, 8>), dim3(nblk), dim3(512), 0, st, p, a);
    } else {
        if (vec) hipLaunchKernelGGL((linear_ce_small_kernel<true, 4>), dim3(nblk), dim3(256), 0, st, p, a);
        else hipLaunchKernelGGL((linear_ce_small_kernel<false, 4>), dim3(nblk), dim3(256), 0, st, p, a);
    }
    NNHIP_LAUNCH_CHECK("linear_ce_small_kernel");
    return 0;
}

// BatchNorm2d(training) -> reshape(B, C*HW) -> Linear(W [N, C*HW], b) -> Sigmoid -> MSELoss(target [B, N]) as ONE launch
// (bn_linear_sigmoid_mse_kernel).  X [B,C,HW] is the pooled output of nnhipConv2dLeakyMaxPoolForwardStats and `stats` its
// [nstat][C][2] (mean, M2) pairs over `count` values each (nstat * count == B * HW).  Y = the BatchNorm output, save_mean / save_inv /
// running statistics as nnhipBatchNorm2dForward (the statistics are combined from the pairs instead of summed in two passes: equal
// to rounding); pred = the Sigmoid output; dz = d(loss)/d(Linear output); loss[0] = mean squared error.
extern "C" int nnhipBatchNorm2dLinearSigmoidMSEFits(int64_t B, int64_t C, int64_t HW, int64_t N) {
    return B >= 1 && B <= 4096 && C >= 1 && C <= 16 && HW >= 1 && N >= 1 && N <= 16 && ((C * HW) & 3) == 0 && C * HW <= 2048 ? 1 : 0;
}
extern "C" int nnhipBatchNorm2dLinearSigmoidMSE(const float* X, const float* stats, int64_t nstat, int64_t count, const float* bn_weight,
                                                const float* bn_bias, float* Y, float* save_mean, float* save_inv, float* running_mean,
                                                float* running_var, int64_t B, int64_t C, int64_t HW, float eps, float momentum,
                                                const float* W, const float* b, int64_t N, const float* target, float* pred, float* dz,
                                                float* loss, nnhipStream_t s) {
    NNHIP_CHECK_ARG(nnhipBatchNorm2dLinearSigmoidMSEFits(B, C, HW, N), NNHIP_EINVAL,
                    "nnhipBatchNorm2dLinearSigmoidMSE: needs C <= 16, N <= 16, C*HW a multiple of 4 and <= 2048, B <= 4096");
    NNHIP_CHECK_ARG(X && stats && Y && save_mean && save_inv && W && target && pred && dz && loss, NNHIP_EINVAL,
                    "nnhipBatchNorm2dLinearSigmoidMSE: null pointer");
    NNHIP_CHECK_ARG(nstat >= 1 && count >= 1 && nstat * count == B * HW, NNHIP_EINVAL,
                    "nnhipBatchNorm2dLinearSigmoidMSE: the statistics pairs must cover B*HW values per channel");
    NNHIP_CHECK_ARG((bn_weight == nullptr) == (bn_bias == nullptr) && (running_mean == nullptr) == (running_var == nullptr), NNHIP_EINVAL,
                    "nnhipBatchNorm2dLinearSigmoidMSE: weight / bias and running_mean / running_var go in pairs");
    NNHIP_CHECK_ARG(aligned16(X) && aligned16(W) && aligned16(Y) && (reinterpret_cast<uintptr_t>(stats) & 7u) == 0, NNHIP_EALIGN,
                    "nnhipBatchNorm2dLinearSigmoidMSE: X, W, Y must be 16-byte aligned");
    const int64_t K = C * HW;
    SmallGemmParams p{};
    p.A = X; p.B = W; p.C = pred; p.bias = b; p.M = B; p.N = N; p.K = K; p.lda = K; p.ldb = K; p.ldc = N;
    p.alpha = 1.f; p.beta = 1.f; p.a_kmajor = 1; p.b_kmajor = 1; p.act = SG_ACT_SIGMOID;
    BnHeadArgs h{};
    h.stats = stats; h.bn_w = bn_weight; h.bn_b = bn_bias; h.Y = Y; h.save_mean = save_mean; h.save_inv = save_inv;
    h.run_mean = running_mean; h.run_var = running_var; h.target = target; h.dz = dz; h.nstat = (int)nstat; h.C = (int)C; h.HW = (int)HW;
    h.count = (float)count; h.eps = eps; h.momentum = momentum; h.invN = 1.0f / (float)(B * N); h.inv_hw = 1.0f / (float)HW;
    CeArgs a{};
    a.mode = 1; a.loss_out = loss; a.rows = B; a.cols = N;
    const unsigned nblk = (unsigned)ceil_div(B, 16);
    unsigned* sync = sync_words();
    a.partial = static_cast<float*>(workspace((size_t)nblk * sizeof(float)));
    if (!sync || !a.partial) { set_last_error("nnhipBatchNorm2dLinearSigmoidMSE: workspace allocation failed"); return NNHIP_ENOMEM; }
    if (int rc = serialize_shared_state((hipStream_t)s)) return rc;
    const SharedStateUse in_use((hipStream_t)s);
    a.sync = sync + SYNC_CE;
    hipLaunchKernelGGL((bn_linear_sigmoid_mse_kernel<8>), dim3(nblk), dim3(512), 0, (hipStream_t)s, p, h, a);
    NNHIP_LAUNCH_CHECK("bn_linear_sigmoid_mse_kernel");
    return 0;
}

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
    CeArgs a{};
    a.logits = logits;
    a.dlogits = dlogits ? dlogits : logits;
    a.loss_rows = loss; a.lse = lse; a.labels = labels; a.lbytes = 4;
    a.ld = logits_stride; a.ignore = ignore_index; a.rows = n_rows; a.cols = n_cols;
    a.mode = reduction == 'm' ? 1 : (reduction == 's' ? 2 : 0);
    // the reference's contract: the caller supplies the 'mean' denominator (host int, or a device int here)
    a.scale_host = 1.0f;
    if (a.mode == 1) {
        a.count_dev = n_non_ignore_dev;
        if (!n_non_ignore_dev) a.scale_host = n_non_ignore > 0 ? 1.0f / (float)n_non_ignore : 0.0f;
    }
    return ce_launch(a, (hipStream_t)s);
}

extern "C" int nnhipCountNotEqual(const int32_t* labels, int64_t n, int32_t ignore_index,
                                  int32_t* out_count, nnhipStream_t s) {
    NNHIP_CHECK_ARG(n >= 0 && out_count && (labels || n == 0), NNHIP_EINVAL, "nnhipCountNotEqual: bad args");
    hipLaunchKernelGGL(ce_denominator_kernel, dim3(1), dim3(1024), 0, (hipStream_t)s, labels, 4, n, (int64_t)ignore_index,
                       (const float*)nullptr, (int64_t)0, out_count, (float*)nullptr);
    NNHIP_LAUNCH_CHECK("ce_denominator_kernel");
    return 0;
}

extern "C" int nnhipCrossEntropyDenominator(const void* labels, int32_t label_bytes, int64_t n, int64_t ignore_index,
                                            const float* class_weight_or_null, int64_t n_cols, int32_t* count_out_or_null,
                                            float* denom_out_or_null, nnhipStream_t s) {
    NNHIP_CHECK_ARG(n >= 0 && (labels || n == 0), NNHIP_EINVAL, "nnhipCrossEntropyDenominator: bad args");
    NNHIP_CHECK_ARG(label_bytes == 2 || label_bytes == 4 || label_bytes == 8, NNHIP_EINVAL,
                    "nnhipCrossEntropyDenominator: labels must be int16, int32 or int64");
    NNHIP_CHECK_ARG(count_out_or_null || denom_out_or_null, NNHIP_EINVAL, "nnhipCrossEntropyDenominator: no output");
    hipLaunchKernelGGL(ce_denominator_kernel, dim3(1), dim3(1024), 0, (hipStream_t)s, labels, (int)label_bytes, n, ignore_index,
                       class_weight_or_null, n_cols, count_out_or_null, denom_out_or_null);
    NNHIP_LAUNCH_CHECK("ce_denominator_kernel");
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

// CrossEntropyLoss(reduction = 'm' | 's') with int32 labels and no class weights: nnhipCrossEntropyLossEx.
extern "C" int nnhipCrossEntropyLoss(float* logits, float* dlogits_or_null, float* loss_rows, float* lse,
                                     const int32_t* labels, int64_t logits_stride, int32_t ignore_index,
                                     int64_t n_rows, int64_t n_cols, char reduction, float* loss_out,
                                     int32_t* count_out, nnhipStream_t s) {
    NNHIP_CHECK_ARG(reduction == 'm' || reduction == 's', NNHIP_EINVAL, "nnhipCrossEntropyLoss: reduction must be 'm' or 's'");
    NNHIP_CHECK_ARG(loss_out && (reduction != 'm' || count_out), NNHIP_EINVAL, "nnhipCrossEntropyLoss: null loss_out / count_out");
    return nnhipCrossEntropyLossEx(logits, dlogits_or_null, loss_rows, lse, labels, 4, nullptr, logits_stride, ignore_index,
                                   n_rows, n_cols, reduction, loss_out, count_out, s);
}

// ---- attention-score softmax (scale + pad/causal mask fused) --------------------------------------------
extern "C" int nnhipMaskedSoftmaxForwardEx(float* out, const float* in, const int32_t* key_valid, const int32_t* dense_mask,
                                           int64_t B, int64_t H, int64_t Tq, int64_t Tk, float scale, int causal,
                                           nnhipStream_t s) {
    NNHIP_CHECK_ARG(B >= 0 && H >= 0 && Tq >= 0 && Tk >= 0, NNHIP_EINVAL, "nnhipMaskedSoftmaxForward: negative size");
    const int64_t rows = B * H * Tq;
    if (rows == 0 || Tk == 0) return 0;
    NNHIP_CHECK_ARG(out && in, NNHIP_EINVAL, "nnhipMaskedSoftmaxForward: null pointer");
    hipStream_t st = (hipStream_t)s;
    if (Tk > kMaxRegRow) {
        hipLaunchKernelGGL(softmax_masked_fwd_looped, dim3((unsigned)rows), dim3(1024), 0, st, out, in, key_valid, Tk, H * Tq, Tq, scale, causal, dense_mask);
    } else {
        const bool vec = aligned16(out) && aligned16(in) && Tk % 4 == 0;
        ROW_DISPATCH(softmax_masked_fwd_rows, Tk, vec, rows, st, out, in, key_valid, rows, Tk, H * Tq, Tq, scale, causal, dense_mask);
    }
    NNHIP_LAUNCH_CHECK("softmax_masked_fwd_rows");
    return 0;
}

extern "C" int nnhipMaskedSoftmaxForward(float* out, const float* in, const int32_t* key_valid, int64_t B,
                                         int64_t H, int64_t Tq, int64_t Tk, float scale, int causal,
                                         nnhipStream_t s) {
    return nnhipMaskedSoftmaxForwardEx(out, in, key_valid, nullptr, B, H, Tq, Tk, scale, causal, s);
}

extern "C" int nnhipMaskedSoftmaxBackwardEx(float* dX, const float* dY, const float* Y, const int32_t* key_valid,
                                            const int32_t* dense_mask, int64_t B, int64_t H, int64_t Tq, int64_t Tk,
                                            float scale, int causal, nnhipStream_t s) {
    NNHIP_CHECK_ARG(B >= 0 && H >= 0 && Tq >= 0 && Tk >= 0, NNHIP_EINVAL, "nnhipMaskedSoftmaxBackward: negative size");
    const int64_t rows = B * H * Tq;
    if (rows == 0 || Tk == 0) return 0;
    NNHIP_CHECK_ARG(dX && dY && Y, NNHIP_EINVAL, "nnhipMaskedSoftmaxBackward: null pointer");
    hipStream_t st = (hipStream_t)s;
    if (Tk > kMaxRegRow) {
        hipLaunchKernelGGL(softmax_masked_bwd_looped, dim3((unsigned)rows), dim3(1024), 0, st, dX, dY, Y, key_valid, Tk, H * Tq, Tq, scale, causal, dense_mask);
    } else {
        const bool vec = aligned16(dX) && aligned16(dY) && aligned16(Y) && Tk % 4 == 0;
        ROW_DISPATCH(softmax_masked_bwd_rows, Tk, vec, rows, st, dX, dY, Y, key_valid, rows, Tk, H * Tq, Tq, scale, causal, dense_mask);
    }
    NNHIP_LAUNCH_CHECK("softmax_masked_bwd_rows");
    return 0;
}

extern "C" int nnhipMaskedSoftmaxBackward(float* dX, const float* dY, const float* Y, const int32_t* key_valid,
                                          int64_t B, int64_t H, int64_t Tq, int64_t Tk, float scale, int causal,
                                          nnhipStream_t s) {
    return nnhipMaskedSoftmaxBackwardEx(dX, dY, Y, key_valid, nullptr, B, H, Tq, Tk, scale, causal, s);
}
