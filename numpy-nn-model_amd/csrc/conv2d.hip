// conv2d.hip -- Conv2d entry points (nnhipConv2dForward / Backward and the fused small-layer variants), the direct kernels of the
// <= 16-channel 3x3 layers (the conv classifier, C5) and the partial-sum reduces.  Layers above 16 channels run as implicit GEMM on
// the fp32 MFMA: conv_mfma.hip.
// CPU semantics: neunet/nn/layers/conv2d.py:297-355 (forward: zero-pad, dilate W, 6-D strided view,
// einsum "bihwkl,oikl->bohw") and :16-115 (backward: dW / db / dX einsums).  No im2col buffer ever
// exists in HBM.
//
//   forward : O[b,co,ho,wo]  = sum_{ci,r,s} W[co,ci,r,s] * X[b,ci, ho*sh-pu+r*dh, wo*sw-pl+s*dw] + bias[co]
//   dgrad   : dX[b,ci,h,w]   = sum_{co,r,s} W[co,ci,r,s] * dO[b,co,(h+pu-r*dh)/sh,(w+pl-s*dw)/sw]
//             (terms exist only where the divisions are exact and in range)
//   wgrad   : dW[co,(ci,r,s)] = sum_{b,ho,wo} dO[b,co,ho,wo] * X[b,ci,ho*sh-pu+r*dh, wo*sw-pl+s*dw];  db[co] = sum dO
// At the C5 shapes (K = 9/72, Cout = 8/16) the work is HBM/latency bound, not MFMA bound.
#include <stdlib.h>

#include <mutex>

#include "conv_common.h"

namespace nnhip {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// The MFMA implicit-GEMM kernels (any layer above 16 channels) live in conv_mfma.hip; this file holds the entry points, the
// direct kernels of the small-channel 3x3 layers (C5) and the partial-sum reduces they share.

// dW[m][n] = sum_c part[c][m][n] (n < Nw);  db[m] = sum_c part[c][m][Nw].  One wave per output element:
// the 64 lanes stride over the chunks and meet in a shuffle reduction (fixed order: deterministic).
__global__ __launch_bounds__(256) void conv_wgrad_reduce_kernel(const float* __restrict__ part,
                                                                float* __restrict__ dW,
                                                                float* __restrict__ db, int chunks,
                                                                int Cout, int Nw, int ncols) {
    const int idx = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int total = Cout * ncols;
    if (idx >= total) return;
    const int m = idx / ncols, n = idx - m * ncols;
    float s = 0.f;
    for (int c = lane; c < chunks; c += 64) s += part[(int64_t)c * total + idx];
    s = wave_sum(s);
    if (lane == 0) {
        if (n < Nw) {
            if (dW) dW[(int64_t)m * Nw + n] = s;
        } else if (db) {
            db[m] = s;
        }
    }
}

// ---- the reduces of several layers as one launch (deferred parameter gradients, common.h) ----------------------------------
// During Tensor.backward() the per-image partials of a small-channel conv go to an arena of their own and the reduce is queued;
// the queue is launched as ONE grid when it is full, at nnhipWeightGradFlush (before anybody reads a gradient) or when deferral
// ends.  C5: the two conv layers' reduces, 2 x 4.8 us, become one launch.  Same arithmetic, same order per output.
constexpr int CRQ_MAX = 4;
struct ConvReduceJob { const float* part; float* dW; float* db; int chunks, Cout, Nw, ncols; };
struct ConvReduceGroup { ConvReduceJob j[CRQ_MAX]; int start[CRQ_MAX + 1]; };
__global__ __launch_bounds__(256) void conv_wgrad_reduce_group_kernel(const ConvReduceGroup grp) {
    int k = 0;
#pragma unroll
    for (int i = 1; i < CRQ_MAX; ++i) k += (int)blockIdx.x >= grp.start[i] ? 1 : 0;
    const float* __restrict__ part = k == 0 ? grp.j[0].part : k == 1 ? grp.j[1].part : k == 2 ? grp.j[2].part : grp.j[3].part;
    float* __restrict__ dW = k == 0 ? grp.j[0].dW : k == 1 ? grp.j[1].dW : k == 2 ? grp.j[2].dW : grp.j[3].dW;
    float* __restrict__ db = k == 0 ? grp.j[0].db : k == 1 ? grp.j[1].db : k == 2 ? grp.j[2].db : grp.j[3].db;
    const int chunks = k == 0 ? grp.j[0].chunks : k == 1 ? grp.j[1].chunks : k == 2 ? grp.j[2].chunks : grp.j[3].chunks;
    const int Cout = k == 0 ? grp.j[0].Cout : k == 1 ? grp.j[1].Cout : k == 2 ? grp.j[2].Cout : grp.j[3].Cout;
    const int Nw = k == 0 ? grp.j[0].Nw : k == 1 ? grp.j[1].Nw : k == 2 ? grp.j[2].Nw : grp.j[3].Nw;
    const int ncols = k == 0 ? grp.j[0].ncols : k == 1 ? grp.j[1].ncols : k == 2 ? grp.j[2].ncols : grp.j[3].ncols;
    const int first = k == 0 ? 0 : k == 1 ? grp.start[1] : k == 2 ? grp.start[2] : grp.start[3];
    const int idx = ((int)blockIdx.x - first) * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int total = Cout * ncols;
    if (idx >= total) return;
    const int m = idx / ncols, n = idx - m * ncols;
    float s = 0.f;
    for (int c = lane; c < chunks; c += 64) s += part[(int64_t)c * total + idx];
    s = wave_sum(s);
    if (lane == 0) {
        if (n < Nw) {
            if (dW) dW[(int64_t)m * Nw + n] = s;
        } else if (db) {
            db[m] = s;
        }
    }
}

static std::mutex g_crq_mu;
static ConvReduceJob g_crq[CRQ_MAX];
static int g_crq_n = 0;
static hipStream_t g_crq_st = nullptr;
static float* g_crq_arena = nullptr;
static size_t g_crq_cap = 0, g_crq_used = 0;              // floats

static int crq_flush_locked(hipStream_t st) {
    if (g_crq_n == 0) return 0;
    ConvReduceGroup grp;
    int blocks = 0;
    for (int i = 0; i < CRQ_MAX; ++i) {
        grp.start[i] = blocks;
        if (i < g_crq_n) { grp.j[i] = g_crq[i]; blocks += (int)ceil_div((int64_t)g_crq[i].Cout * g_crq[i].ncols, 4); }
        else grp.j[i] = g_crq[0];
    }
    grp.start[CRQ_MAX] = blocks;
    for (int i = g_crq_n; i < CRQ_MAX; ++i) grp.start[i] = blocks;      // no block maps to an unused slot
    const int n = g_crq_n;
    g_crq_n = 0;
    g_crq_used = 0;
    if (n == 1)
        hipLaunchKernelGGL(conv_wgrad_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, st, grp.j[0].part, grp.j[0].dW, grp.j[0].db,
                           grp.j[0].chunks, grp.j[0].Cout, grp.j[0].Nw, grp.j[0].ncols);
    else
        hipLaunchKernelGGL(conv_wgrad_reduce_group_kernel, dim3((unsigned)blocks), dim3(256), 0, st, grp);
    NNHIP_LAUNCH_CHECK("conv_wgrad_reduce_group_kernel");
    return 0;
}
int conv_reduce_flush(void* stream) {
    std::lock_guard<std::mutex> lk(g_crq_mu);
    return crq_flush_locked(g_crq_n ? g_crq_st : (hipStream_t)stream);
}
void conv_reduce_cleanup() {
    std::lock_guard<std::mutex> lk(g_crq_mu);
    g_crq_n = 0;
    g_crq_used = g_crq_cap = 0;
    if (g_crq_arena) (void)hipFree(g_crq_arena);
    g_crq_arena = nullptr;
}
// Where a weight-gradient kernel writes its `floats` partials: the arena when its reduce can wait for the flush (*deferred),
// else the shared workspace (reduce launched at once).  nullptr: out of memory.
static float* conv_partials(size_t floats, hipStream_t st, bool* deferred, int* rc) {
    static const bool on = []() { const char* e = getenv("NNHIP_CONV_DEFER_REDUCE"); return !e || atoi(e) != 0; }();
    *deferred = false;
    *rc = 0;
    if (on && wgrad_defer_on()) {
        std::lock_guard<std::mutex> lk(g_crq_mu);
        if (g_crq_n && (g_crq_st != st || g_crq_n == CRQ_MAX || g_crq_used + floats > g_crq_cap)) *rc = crq_flush_locked(g_crq_st);
        if (floats > g_crq_cap && !workspace_locked()) {      // grow (nothing is queued here); never while a captured graph holds the address
            hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
            (void)hipStreamIsCapturing(st, &cs);
            if (cs == hipStreamCaptureStatusNone) {
                if (g_crq_arena) (void)hipFree(g_crq_arena);  // (synchronises: earlier reduces are done with it)
                g_crq_arena = nullptr;
                g_crq_cap = 0;
                const size_t want = floats + (floats >> 1);
                if (hipMalloc(&g_crq_arena, want * sizeof(float)) == hipSuccess) g_crq_cap = want;
                else g_crq_arena = nullptr;
            }
        }
        if (floats <= g_crq_cap - g_crq_used) {
            float* p = g_crq_arena + g_crq_used;
            g_crq_used += (floats + 63) & ~(size_t)63;
            g_crq_st = st;
            *deferred = true;
            return p;
        }
    }
    return static_cast<float*>(workspace(floats * sizeof(float)));
}
static int conv_reduce(const float* part, float* dW, float* db, int chunks, int Cout, int Nw, int ncols, hipStream_t st, bool deferred) {
    if (deferred) {
        std::lock_guard<std::mutex> lk(g_crq_mu);
        g_crq[g_crq_n++] = ConvReduceJob{part, dW, db, chunks, Cout, Nw, ncols};
        return 0;
    }
    hipLaunchKernelGGL(conv_wgrad_reduce_kernel, dim3((unsigned)ceil_div((int64_t)Cout * ncols, 4)), dim3(256), 0, st, part, dW, db, chunks,
                       Cout, Nw, ncols);
    NNHIP_LAUNCH_CHECK("conv_wgrad_reduce_kernel");
    return 0;
}

// =================================================================================================
// Direct kernels for SMALL channel counts (Cin, Cout <= 16, kh*kw <= 25): the implicit GEMM above pads an 8- or
// 16-row problem to 32-row MFMA tiles and spends its time gathering (C5: dgrad 60 us, wgrad 38 us, fwd 29 us for
// 58 MFLOP layers).  Here one thread owns one output (input) pixel and all CO (CI) channels of it in registers; the
// weights sit in LDS laid out so that one ds_read_b128 hands a thread 4 channels' weights for the tap it is on (all
// threads read the same address: broadcast).  Same formulas, same summation order over (ci, r, s).
// =================================================================================================
// global -> LDS staging by a 256-thread block with the loads BATCHED: 8 loads per thread are in flight before the first LDS
// store (`for (i = tid; i < n; i += 256) S[i] = f(i)` is one memory round trip per iteration).  `src_index(i)` maps a staged
// element to its source index or -1 (zero fill).
template <typename F>
__device__ __forceinline__ void stage_to_lds(float* __restrict__ dst, const float* __restrict__ src, int n, int tid, F src_index) {
    constexpr int D = 8;
    for (int base = 0; base < n; base += 256 * D) {
        float t[D];
#pragma unroll
        for (int j = 0; j < D; ++j) {
            const int i = base + tid + 256 * j;
            const int64_t si = i < n ? src_index(i) : -1;
            const float v = src[si >= 0 ? si : 0];
            t[j] = si >= 0 ? v : 0.f;
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < D; ++j) {
            const int i = base + tid + 256 * j;
            if (i < n) dst[i] = t[j];
        }
    }
}

constexpr unsigned CD_OOB = 0xFFFFFFF0u;                   // a byte offset no descriptor below covers
__device__ __forceinline__ __amdgpu_buffer_rsrc_t cd_rsrc(const float* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, (int)bytes, 0x00020000);
}
constexpr int CD_MAXC = 16;
constexpr int CD_MAXTAPS = 25;

// forward: thread <-> (b, ho, wo); acc[co].   Wl[(ci*khkw + rs) * CO + co]
template <int CO>
__global__ __launch_bounds__(256) void conv_direct_fwd_kernel(const float* __restrict__ Wt, const float* __restrict__ X,
                                                              const float* __restrict__ bias, float* __restrict__ O,
                                                              const ConvGeom g) {
    __shared__ __attribute__((aligned(16))) float Wl[CD_MAXC * CD_MAXTAPS * CO];
    const int khkw = g.kh * g.kw, K = g.Cin * khkw;
    stage_to_lds(Wl, Wt, K * CO, threadIdx.x, [&](int i) -> int64_t {
        const int co = i % CO, k = i / CO;
        return co < g.Cout ? (int64_t)co * K + k : -1;
    });
    __syncthreads();
    const int64_t HWo = (int64_t)g.Ho * g.Wo, N = (int64_t)g.B * HWo;
    const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const int b = (int)(n / HWo), p = (int)(n - (int64_t)b * HWo);
    const int ho = p / g.Wo, wo = p - ho * g.Wo;
    float acc[CO];
#pragma unroll
    for (int c = 0; c < CO; ++c) acc[c] = 0.f;
    const float* xb = X + (int64_t)b * g.Cin * g.H * g.W;
    if (khkw == 9 && g.kw == 3) {
        // 3x3: the 9 tap offsets are computed once, then every channel issues its 9 loads back to back (independent,
        // all in flight together) before the FMAs -- the generic loop below has one dependent load per tap
        int off[9];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int hi = ho * g.sh - g.pu + r * g.dh, wi = wo * g.sw - g.pl + q * g.dw;
                off[r * 3 + q] = (hi >= 0 && hi < g.H && wi >= 0 && wi < g.W) ? hi * g.W + wi : -1;
            }
        // The loads go through ONE buffer descriptor of the whole tensor: the tap's byte offset sits in a VGPR (computed once; a
        // tap outside the image gets an offset past num_records and reads 0), the channel's plane is the scalar offset -- no
        // vector instruction per load (with 64-bit pointers and `off >= 0 ? x[..] : 0` the addressing and selects were 3/4 of
        // the kernel's vector instructions: SQ counters, 2300 VALU per wave for 576 packed FMAs).  The descriptor must be
        // wave-uniform: with a per-image descriptor a wave that straddles two images turns every load into a readfirstlane loop
        // (measured: 16 -> 22 us).  Channels go in groups of CG with the 9*CG loads of a group in flight before its first FMA.
        // (Measured and dropped: ALL taps up front, before the weights are staged -- 144 live values cost the occupancy that hides
        // the rest: conv1's forward 5.8 -> 10.9 us.)
        const int HWi = g.H * g.W;
        const __amdgpu_buffer_rsrc_t rx = cd_rsrc(X, (unsigned)((int64_t)g.B * g.Cin * HWi) * 4u);
        unsigned vo[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) vo[t] = off[t] >= 0 ? (unsigned)(b * g.Cin * HWi + off[t]) * 4u : CD_OOB;
        constexpr int CG = 8;
        for (int c0 = 0; c0 < g.Cin; c0 += CG) {
            float x[CG][9];
#pragma unroll
            for (int j = 0; j < CG; ++j) {
                const unsigned so = (unsigned)((c0 + j) * HWi) * 4u;        // a channel past Cin: past num_records, reads 0
#pragma unroll
                for (int t = 0; t < 9; ++t) x[j][t] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rx, vo[t], so, 0));
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < CG; ++j) {
                if (c0 + j >= g.Cin) break;
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const float4* w4 = reinterpret_cast<const float4*>(&Wl[((c0 + j) * 9 + t) * CO]);
#pragma unroll
                    for (int c4 = 0; c4 < CO / 4; ++c4) {
                        const float4 w = w4[c4];
                        acc[4 * c4] += x[j][t] * w.x; acc[4 * c4 + 1] += x[j][t] * w.y; acc[4 * c4 + 2] += x[j][t] * w.z; acc[4 * c4 + 3] += x[j][t] * w.w;
                    }
                }
            }
        }
    } else
    for (int ci = 0; ci < g.Cin; ++ci)
        for (int r = 0; r < g.kh; ++r) {
            const int hi = ho * g.sh - g.pu + r * g.dh;
            if (hi < 0 || hi >= g.H) continue;
            for (int q = 0; q < g.kw; ++q) {
                const int wi = wo * g.sw - g.pl + q * g.dw;
                if (wi < 0 || wi >= g.W) continue;
                const float x = xb[((int64_t)ci * g.H + hi) * g.W + wi];
                const float4* w4 = reinterpret_cast<const float4*>(&Wl[(ci * khkw + r * g.kw + q) * CO]);
#pragma unroll
                for (int c4 = 0; c4 < CO / 4; ++c4) {
                    const float4 w = w4[c4];
                    acc[4 * c4] += x * w.x; acc[4 * c4 + 1] += x * w.y; acc[4 * c4 + 2] += x * w.z; acc[4 * c4 + 3] += x * w.w;
                }
            }
        }
#pragma unroll
    for (int c = 0; c < CO; ++c)
        if (c < g.Cout) O[((int64_t)b * g.Cout + c) * HWo + p] = acc[c] + (bias ? bias[c] : 0.f);
}

// forward of  MaxPool2d(2, 2)(LeakyReLU(Conv2d(X); alpha))  in one launch (alpha = 1: no activation): thread <-> one pool WINDOW, i.e.
// the 2x2 conv outputs under it -- a 4x4 input patch per channel (16 loads where four separate positions take 36), every
// weight read from LDS once for the four positions.  Per output the products run in conv_direct_fwd_kernel's order (ci, then taps;
// values agree with the three-module chain to an ulp or two -- the compiler contracts the two kernels' multiply-adds differently);
// the conv output itself is never written (the backward needs only the arg-max and the pooled output: the activation's slope is its
// sign).  3x3, unit stride and dilation, windows
// tiling the conv output exactly.
template <int CO>
__global__ __launch_bounds__(256) void conv_pool_fwd_kernel(const float* __restrict__ Wt, const float* __restrict__ X,
                                                            const float* __restrict__ bias, float* __restrict__ P,
                                                            int32_t* __restrict__ arg, const ConvGeom g, float alpha) {
    __shared__ __attribute__((aligned(16))) float Wl[CD_MAXC * 9 * CO];
    const int K = g.Cin * 9;
    stage_to_lds(Wl, Wt, K * CO, threadIdx.x, [&](int i) -> int64_t {
        const int co = i % CO, k = i / CO;
        return co < g.Cout ? (int64_t)co * K + k : -1;
    });
    __syncthreads();
    const int Hq = g.Ho >> 1, Wq = g.Wo >> 1, HWq = Hq * Wq;
    const int64_t N = (int64_t)g.B * HWq;
    const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const int b = (int)(n / HWq), pq = (int)(n - (int64_t)b * HWq);
    const int hq = pq / Wq, wq = pq - hq * Wq;
    const int HWi = g.H * g.W;
    const __amdgpu_buffer_rsrc_t rx = cd_rsrc(X, (unsigned)((int64_t)g.B * g.Cin * HWi) * 4u);
    unsigned vo[16];                                        // the 4x4 patch: rows 2hq - pu + {0..3}, columns 2wq - pl + {0..3}
#pragma unroll
    for (int y = 0; y < 4; ++y)
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            const int hi = 2 * hq - g.pu + y, wi = 2 * wq - g.pl + x;
            vo[y * 4 + x] = (hi >= 0 && hi < g.H && wi >= 0 && wi < g.W) ? (unsigned)(b * g.Cin * HWi + hi * g.W + wi) * 4u : CD_OOB;
        }
    float acc[4][CO];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int c = 0; c < CO; ++c) acc[k][c] = 0.f;
    for (int ci = 0; ci < g.Cin; ++ci) {
        const unsigned so = (unsigned)(ci * HWi) * 4u;
        float x[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) x[t] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rx, vo[t], so, 0));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const float4* w4 = reinterpret_cast<const float4*>(&Wl[(ci * 9 + r * 3 + q) * CO]);
#pragma unroll
                for (int c4 = 0; c4 < CO / 4; ++c4) {
                    const float4 w = w4[c4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float xv = x[((k >> 1) + r) * 4 + (k & 1) + q];
                        acc[k][4 * c4] += xv * w.x; acc[k][4 * c4 + 1] += xv * w.y; acc[k][4 * c4 + 2] += xv * w.z; acc[k][4 * c4 + 3] += xv * w.w;
                    }
                }
            }
    }
#pragma unroll
    for (int c = 0; c < CO; ++c) {
        if (c < g.Cout) {
            const float bc = bias ? bias[c] : 0.f;
            float best = -INFINITY;
            int bi = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {                   // MaxPool2d's scan order and first-max rule (maxpool_fwd_kernel)
                float v = acc[k][c] + bc;
                if (alpha != 1.0f) v = v <= 0.f ? alpha * v : v;
                if (v > best) { best = v; bi = k; }
            }
            const int64_t o = ((int64_t)b * g.Cout + c) * HWq + pq;
            P[o] = best;
            arg[o] = bi;
        }
    }
}

// dgrad: thread <-> (b, h, w); acc[ci].   Wl[(co*khkw + rs) * CI + ci]
template <int CI>
__global__ __launch_bounds__(256) void conv_direct_dgrad_kernel(const float* __restrict__ Wt, const float* __restrict__ dO,
                                                                float* __restrict__ dX, const ConvGeom g) {
    __shared__ __attribute__((aligned(16))) float Wl[CD_MAXC * CD_MAXTAPS * CI];
    const int khkw = g.kh * g.kw;
    stage_to_lds(Wl, Wt, g.Cout * khkw * CI, threadIdx.x, [&](int i) -> int64_t {
        const int ci = i % CI, k = i / CI, co = k / khkw, rs = k - co * khkw;
        return ci < g.Cin ? ((int64_t)co * g.Cin + ci) * khkw + rs : -1;
    });
    __syncthreads();
    const int64_t HW = (int64_t)g.H * g.W, HWo = (int64_t)g.Ho * g.Wo, N = (int64_t)g.B * HW;
    const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const int b = (int)(n / HW), p = (int)(n - (int64_t)b * HW);
    const int h = p / g.W, w = p - h * g.W;
    float acc[CI];
#pragma unroll
    for (int c = 0; c < CI; ++c) acc[c] = 0.f;
    const float* gb = dO + (int64_t)b * g.Cout * HWo;
    if (khkw == 9 && g.kw == 3) {
        int off[9];                                     // as in the forward kernel: offsets once, 9 loads in flight per channel
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int th = h + g.pu - r * g.dh, tw = w + g.pl - q * g.dw;
                const bool ok = th >= 0 && tw >= 0 && th % g.sh == 0 && tw % g.sw == 0 && th / g.sh < g.Ho && tw / g.sw < g.Wo;
                off[r * 3 + q] = ok ? (th / g.sh) * g.Wo + tw / g.sw : -1;
            }
        const __amdgpu_buffer_rsrc_t rg = cd_rsrc(dO, (unsigned)((int64_t)g.B * g.Cout * HWo) * 4u);      // as in the forward kernel
        unsigned vo[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) vo[t] = off[t] >= 0 ? (unsigned)(b * g.Cout * (int)HWo + off[t]) * 4u : CD_OOB;
        constexpr int CG = 8;
        for (int c0 = 0; c0 < g.Cout; c0 += CG) {
            float v[CG][9];
#pragma unroll
            for (int j = 0; j < CG; ++j) {
                const unsigned so = (unsigned)((c0 + j) * (int)HWo) * 4u;
#pragma unroll
                for (int t = 0; t < 9; ++t) v[j][t] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rg, vo[t], so, 0));
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < CG; ++j) {
                if (c0 + j >= g.Cout) break;
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const float4* w4 = reinterpret_cast<const float4*>(&Wl[((c0 + j) * 9 + t) * CI]);
#pragma unroll
                    for (int c4 = 0; c4 < CI / 4; ++c4) {
                        const float4 ww = w4[c4];
                        acc[4 * c4] += v[j][t] * ww.x; acc[4 * c4 + 1] += v[j][t] * ww.y; acc[4 * c4 + 2] += v[j][t] * ww.z; acc[4 * c4 + 3] += v[j][t] * ww.w;
                    }
                }
            }
        }
    } else
    for (int co = 0; co < g.Cout; ++co)
        for (int r = 0; r < g.kh; ++r) {
            const int th = h + g.pu - r * g.dh;
            if (th < 0 || th % g.sh != 0) continue;
            const int ho = th / g.sh;
            if (ho >= g.Ho) continue;
            for (int q = 0; q < g.kw; ++q) {
                const int tw = w + g.pl - q * g.dw;
                if (tw < 0 || tw % g.sw != 0) continue;
                const int wo = tw / g.sw;
                if (wo >= g.Wo) continue;
                const float v = gb[(int64_t)co * HWo + (int64_t)ho * g.Wo + wo];
                const float4* w4 = reinterpret_cast<const float4*>(&Wl[(co * khkw + r * g.kw + q) * CI]);
#pragma unroll
                for (int c4 = 0; c4 < CI / 4; ++c4) {
                    const float4 ww = w4[c4];
                    acc[4 * c4] += v * ww.x; acc[4 * c4 + 1] += v * ww.y; acc[4 * c4 + 2] += v * ww.z; acc[4 * c4 + 3] += v * ww.w;
                }
            }
        }
#pragma unroll
    for (int c = 0; c < CI; ++c)
        if (c < g.Cin) dX[((int64_t)b * g.Cin + c) * HW + p] = acc[c];
}

// ---- "quad" variants of the two direct kernels, 3x3 only, for grids that leave most SIMDs with less than one wave ------------
// (C5's second conv layer: 256 x 14 x 14 = 50176 positions = 196 blocks on 256 CUs, each thread a serial chain of 144 loads and
// 1152 FMAs).  Four adjacent lanes share one position and split the REDUCTION channels (lane s takes channels s, s + 4, ...):
// four times the waves, a quarter of the chain each; the partial sums meet in two DPP quad-permute steps (x + swap1(x), then
// + swap2: every lane of the quad ends with the same bits) and each lane stores a quarter of the position's outputs.
__device__ __forceinline__ float quad_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
    return v;
}

// forward: Wl[ci * CSTR + tap * CO + co]; CSTR = 9 CO (+ 8 for CO = 16: the four lanes of a quad read four different ci at once,
// and a stride of 144 floats would put two of them on the same LDS banks)
template <int CO>
__global__ __launch_bounds__(256) void conv_direct_fwd_quad_kernel(const float* __restrict__ Wt, const float* __restrict__ X,
                                                                   const float* __restrict__ bias, float* __restrict__ O,
                                                                   const ConvGeom g) {
    constexpr int CSTR = 9 * CO + (CO == 16 ? 8 : 0), CPL = CD_MAXC / 4;      // channels per lane
    __shared__ __attribute__((aligned(16))) float Wl[CD_MAXC * CSTR];
    const int K = g.Cin * 9;
    stage_to_lds(Wl, Wt, g.Cin * CSTR, threadIdx.x, [&](int i) -> int64_t {
        const int ci = i / CSTR, r = i - ci * CSTR, t = r / CO, co = r - t * CO;
        return (t < 9 && co < g.Cout) ? (int64_t)co * K + ci * 9 + t : -1;
    });
    __syncthreads();
    const int64_t HWo = (int64_t)g.Ho * g.Wo, N = (int64_t)g.B * HWo;
    const int64_t n = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 2;
    const int s = threadIdx.x & 3;
    if (n >= N) return;                                   // whole quads leave together
    const int b = (int)(n / HWo), p = (int)(n - (int64_t)b * HWo);
    const int ho = p / g.Wo, wo = p - ho * g.Wo;
    float acc[CO];
#pragma unroll
    for (int c = 0; c < CO; ++c) acc[c] = 0.f;
    const int HWi = g.H * g.W;
    const __amdgpu_buffer_rsrc_t rx = cd_rsrc(X, (unsigned)((int64_t)g.B * g.Cin * HWi) * 4u);
    unsigned vo[9];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int hi = ho * g.sh - g.pu + r * g.dh, wi = wo * g.sw - g.pl + q * g.dw;
            vo[r * 3 + q] = (hi >= 0 && hi < g.H && wi >= 0 && wi < g.W) ? (unsigned)(b * g.Cin * HWi + hi * g.W + wi) * 4u : CD_OOB;
        }
    float x[CPL][9];
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
        const int ci = 4 * j + s;
        const bool live = ci < g.Cin;
        const unsigned chan = (unsigned)(ci * HWi) * 4u;                          // per-lane channel: a vector offset here
#pragma unroll
        for (int t = 0; t < 9; ++t)
            x[j][t] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rx, (live && vo[t] != CD_OOB) ? vo[t] + chan : CD_OOB, 0, 0));
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
        if (4 * j >= g.Cin) break;                          // uniform
        const int ci = min(4 * j + s, g.Cin - 1);           // (a lane past Cin multiplies zeros by staged weights)
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const float4* w4 = reinterpret_cast<const float4*>(&Wl[ci * CSTR + t * CO]);
#pragma unroll
            for (int c4 = 0; c4 < CO / 4; ++c4) {
                const float4 w = w4[c4];
                acc[4 * c4] += x[j][t] * w.x; acc[4 * c4 + 1] += x[j][t] * w.y; acc[4 * c4 + 2] += x[j][t] * w.z; acc[4 * c4 + 3] += x[j][t] * w.w;
            }
        }
    }
#pragma unroll
    for (int c = 0; c < CO; ++c) acc[c] = quad_sum(acc[c]);
#pragma unroll
    for (int c = 0; c < CO / 4; ++c) {                      // lane s stores outputs s CO/4 ... (selects: no dynamic register index)
        float a0 = acc[c], a1 = acc[CO / 4 + c], a2 = acc[2 * (CO / 4) + c], a3 = acc[3 * (CO / 4) + c];
        asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));      // (or the selects become acc[f(s)]: a scratch array)
        const float v = s == 0 ? a0 : s == 1 ? a1 : s == 2 ? a2 : a3;
        const int co = s * (CO / 4) + c;
        if (co < g.Cout) O[((int64_t)b * g.Cout + co) * HWo + p] = v + (bias ? bias[co] : 0.f);
    }
}

// dgrad: Wl[(co * 9 + tap) * CI + ci] as in conv_direct_dgrad_kernel (CI <= 8: the quad's four co sit 9 CI floats apart, on
// different banks); lane s takes output channels s, s + 4, ...
template <int CI>
__global__ __launch_bounds__(256) void conv_direct_dgrad_quad_kernel(const float* __restrict__ Wt, const float* __restrict__ dO,
                                                                     float* __restrict__ dX, const ConvGeom g) {
    constexpr int CPL = CD_MAXC / 4;
    __shared__ __attribute__((aligned(16))) float Wl[CD_MAXC * 9 * CI];
    stage_to_lds(Wl, Wt, g.Cout * 9 * CI, threadIdx.x, [&](int i) -> int64_t {
        const int ci = i % CI, k = i / CI, co = k / 9, rs = k - co * 9;
        return ci < g.Cin ? ((int64_t)co * g.Cin + ci) * 9 + rs : -1;
    });
    __syncthreads();
    const int64_t HW = (int64_t)g.H * g.W, N = (int64_t)g.B * HW;
    const int HWo = g.Ho * g.Wo;
    const int64_t n = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 2;
    const int s = threadIdx.x & 3;
    if (n >= N) return;
    const int b = (int)(n / HW), p = (int)(n - (int64_t)b * HW);
    const int h = p / g.W, w = p - h * g.W;
    float acc[CI];
#pragma unroll
    for (int c = 0; c < CI; ++c) acc[c] = 0.f;
    const __amdgpu_buffer_rsrc_t rg = cd_rsrc(dO, (unsigned)((int64_t)g.B * g.Cout * HWo) * 4u);
    int off[9];
    if (g.sh == 1 && g.sw == 1) {                          // unit stride: no divisions (18 div + 18 mod per thread otherwise -- more
#pragma unroll                                             // instructions than the FMAs)
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int th = h + g.pu - r * g.dh, tw = w + g.pl - q * g.dw;
                off[r * 3 + q] = (th >= 0 && tw >= 0 && th < g.Ho && tw < g.Wo) ? th * g.Wo + tw : -1;
            }
    } else {
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int th = h + g.pu - r * g.dh, tw = w + g.pl - q * g.dw;
                const bool ok = th >= 0 && tw >= 0 && th % g.sh == 0 && tw % g.sw == 0 && th / g.sh < g.Ho && tw / g.sw < g.Wo;
                off[r * 3 + q] = ok ? (th / g.sh) * g.Wo + tw / g.sw : -1;
            }
    }
    unsigned vo[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) vo[t] = off[t] >= 0 ? (unsigned)(b * g.Cout * HWo + off[t]) * 4u : CD_OOB;
    float v[CPL][9];
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
        const int co = 4 * j + s;
        const bool live = co < g.Cout;
        const unsigned chan = (unsigned)(co * HWo) * 4u;
#pragma unroll
        for (int t = 0; t < 9; ++t)
            v[j][t] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rg, (live && vo[t] != CD_OOB) ? vo[t] + chan : CD_OOB, 0, 0));
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
        // (no early exit for Cout < 16: a channel past Cout multiplies zeros by staged weights)
        const int co = min(4 * j + s, g.Cout - 1);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const float4* w4 = reinterpret_cast<const float4*>(&Wl[(co * 9 + t) * CI]);
#pragma unroll
            for (int c4 = 0; c4 < CI / 4; ++c4) {
                const float4 ww = w4[c4];
                acc[4 * c4] += v[j][t] * ww.x; acc[4 * c4 + 1] += v[j][t] * ww.y; acc[4 * c4 + 2] += v[j][t] * ww.z; acc[4 * c4 + 3] += v[j][t] * ww.w;
            }
        }
    }
#pragma unroll
    for (int c = 0; c < CI; ++c) acc[c] = quad_sum(acc[c]);
#pragma unroll
    for (int c = 0; c < CI / 4; ++c) {
        float a0 = acc[c], a1 = acc[CI / 4 + c], a2 = acc[2 * (CI / 4) + c], a3 = acc[3 * (CI / 4) + c];
        asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));      // (or the selects become acc[f(s)]: a scratch array)
        const float o = s == 0 ? a0 : s == 1 ? a1 : s == 2 ? a2 : a3;
        const int ci = s * (CI / 4) + c;
        if (ci < g.Cin) dX[((int64_t)b * g.Cin + ci) * HW + p] = o;
    }
}

// dgrad, unit stride and dilation, four lanes per 2x2 BLOCK of input positions (each lane a quarter of the output channels for all
// four positions): the four positions' taps reach a 4x4 patch of dO per channel (16 loads where four separate positions take
// 36) and a weight read from LDS serves four positions -- the position-per-quad kernel above re-reads its 72 ds_read_b128 per
// position and is LDS-bandwidth-bound at C5's second layer.  Same products in the same order per output (co = s, s + 4, ...; taps
// in order; quad tree).
template <int CI>
__global__ __launch_bounds__(256) void conv_direct_dgrad_quad2_kernel(const float* __restrict__ Wt, const float* __restrict__ dO,
                                                                      float* __restrict__ dX, const ConvGeom g) {
    constexpr int CPL = CD_MAXC / 4;
    __shared__ __attribute__((aligned(16))) float Wl[CD_MAXC * 9 * CI];
    stage_to_lds(Wl, Wt, g.Cout * 9 * CI, threadIdx.x, [&](int i) -> int64_t {
        const int ci = i % CI, k = i / CI, co = k / 9, rs = k - co * 9;
        return ci < g.Cin ? ((int64_t)co * g.Cin + ci) * 9 + rs : -1;
    });
    __syncthreads();
    const int Hb = (g.H + 1) >> 1, Wb = (g.W + 1) >> 1, HWb = Hb * Wb;
    const int64_t N = (int64_t)g.B * HWb, HW = (int64_t)g.H * g.W;
    const int HWo = g.Ho * g.Wo;
    const int64_t n = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 2;
    const int s = threadIdx.x & 3;
    if (n >= N) return;
    const int b = (int)(n / HWb), pb = (int)(n - (int64_t)b * HWb);
    const int h0 = 2 * (pb / Wb), w0 = 2 * (pb % Wb);
    const __amdgpu_buffer_rsrc_t rg = cd_rsrc(dO, (unsigned)((int64_t)g.B * g.Cout * HWo) * 4u);
    unsigned vo[16];                                        // patch rows h0 + pu - 2 + {0..3}, columns w0 + pl - 2 + {0..3} of dO
#pragma unroll
    for (int y = 0; y < 4; ++y)
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            const int th = h0 + g.pu - 2 + y, tw = w0 + g.pl - 2 + x;
            vo[y * 4 + x] = (th >= 0 && th < g.Ho && tw >= 0 && tw < g.Wo) ? (unsigned)(b * g.Cout * HWo + th * g.Wo + tw) * 4u : CD_OOB;
        }
    float acc[4][CI];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int c = 0; c < CI; ++c) acc[k][c] = 0.f;
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
        if (4 * j >= g.Cout) break;                         // uniform
        const bool live = 4 * j + s < g.Cout;
        const int co = min(4 * j + s, g.Cout - 1);          // (a lane past Cout multiplies zeros by staged weights)
        const unsigned chan = (unsigned)(co * HWo) * 4u;
        float v[16];
#pragma unroll
        for (int t = 0; t < 16; ++t)
            v[t] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rg, (live && vo[t] != CD_OOB) ? vo[t] + chan : CD_OOB, 0, 0));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const float4* w4 = reinterpret_cast<const float4*>(&Wl[(co * 9 + r * 3 + q) * CI]);
#pragma unroll
                for (int c4 = 0; c4 < CI / 4; ++c4) {
                    const float4 ww = w4[c4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {             // position (k >> 1, k & 1) of the block: dO row (k >> 1) + 2 - r of the patch
                        const float xv = v[((k >> 1) + 2 - r) * 4 + (k & 1) + 2 - q];
                        acc[k][4 * c4] += xv * ww.x; acc[k][4 * c4 + 1] += xv * ww.y; acc[k][4 * c4 + 2] += xv * ww.z; acc[k][4 * c4 + 3] += xv * ww.w;
                    }
                }
            }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int c = 0; c < CI; ++c) acc[k][c] = quad_sum(acc[k][c]);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int h = h0 + (k >> 1), w = w0 + (k & 1);
        if (h < g.H && w < g.W) {
#pragma unroll
            for (int c = 0; c < CI / 4; ++c) {
                float a0 = acc[k][c], a1 = acc[k][CI / 4 + c], a2 = acc[k][2 * (CI / 4) + c], a3 = acc[k][3 * (CI / 4) + c];
                asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));      // (or the selects become acc[f(s)]: a scratch array)
                const float o = s == 0 ? a0 : s == 1 ? a1 : s == 2 ? a2 : a3;
                const int ci = s * (CI / 4) + c;
                if (ci < g.Cin) dX[((int64_t)b * g.Cin + ci) * HW + (int64_t)h * g.W + w] = o;
            }
        }
    }
}

// the same fused forward for layers with more input channels and too few windows for one thread each (C5's second layer: 12 544
// windows): FOUR lanes per window, each a quarter of the input channels (ci = s, s + 4, ...) for all four positions -- a weight read
// from LDS serves four positions (a position-per-quad variant read it per position: 72 ds_read_b128 per
// position made it LDS-bandwidth-bound, 10 us against 8.9) -- then 4 x CO quad sums, and lane s finishes output channels
// s CO/4 ...: bias, activation, first maximum over the four positions, pooled value + arg-max.  Unit stride and dilation.
// STATS (the launch has whole blocks only: B Hq Wq a multiple of 64): the block also leaves (mean, M2 = sum of squared deviations)
// of its 64 pooled values per output channel in stats[block][Cout][2] -- the partial batch statistics a BatchNorm2d behind this
// layer needs; the consumer combines the blocks' pairs (Chan et al.), so nothing waits for anything here.
template <int CO, bool STATS = false>
__global__ __launch_bounds__(256) void conv_pool_fwd_quad_kernel(const float* __restrict__ Wt, const float* __restrict__ X,
                                                                 const float* __restrict__ bias, float* __restrict__ P,
                                                                 int32_t* __restrict__ arg, const ConvGeom g, float alpha,
                                                                 float* __restrict__ stats = nullptr) {
    constexpr int CSTR = 9 * CO + (CO == 16 ? 8 : 0), CPL = CD_MAXC / 4;
    __shared__ __attribute__((aligned(16))) float Wl[CD_MAXC * CSTR];
    __shared__ float sred[STATS ? 4 * CO : 1];
    const int K = g.Cin * 9;
    stage_to_lds(Wl, Wt, g.Cin * CSTR, threadIdx.x, [&](int i) -> int64_t {
        const int ci = i / CSTR, r = i - ci * CSTR, t = r / CO, co = r - t * CO;
        return (t < 9 && co < g.Cout) ? (int64_t)co * K + ci * 9 + t : -1;
    });
    __syncthreads();
    const int Hq = g.Ho >> 1, Wq = g.Wo >> 1, HWq = Hq * Wq;
    const int64_t N = (int64_t)g.B * HWq;
    const int64_t n = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 2;
    const int s = threadIdx.x & 3;
    if (n >= N) return;                                   // whole quads leave together
    const int b = (int)(n / HWq), pq = (int)(n - (int64_t)b * HWq);
    const int hq = pq / Wq, wq = pq - hq * Wq;
    const int HWi = g.H * g.W;
    const __amdgpu_buffer_rsrc_t rx = cd_rsrc(X, (unsigned)((int64_t)g.B * g.Cin * HWi) * 4u);
    unsigned vo[16];
#pragma unroll
    for (int y = 0; y < 4; ++y)
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            const int hi = 2 * hq - g.pu + y, wi = 2 * wq - g.pl + x;
            vo[y * 4 + x] = (hi >= 0 && hi < g.H && wi >= 0 && wi < g.W) ? (unsigned)(b * g.Cin * HWi + hi * g.W + wi) * 4u : CD_OOB;
        }
    float acc[4][CO];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int c = 0; c < CO; ++c) acc[k][c] = 0.f;
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
        if (4 * j >= g.Cin) break;                          // uniform
        const bool live = 4 * j + s < g.Cin;
        const int ci = min(4 * j + s, g.Cin - 1);           // (a lane past Cin multiplies zeros by staged weights)
        const unsigned chan = (unsigned)(ci * HWi) * 4u;
        float x[16];
#pragma unroll
        for (int t = 0; t < 16; ++t)
            x[t] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rx, (live && vo[t] != CD_OOB) ? vo[t] + chan : CD_OOB, 0, 0));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const float4* w4 = reinterpret_cast<const float4*>(&Wl[ci * CSTR + (r * 3 + q) * CO]);
#pragma unroll
                for (int c4 = 0; c4 < CO / 4; ++c4) {
                    const float4 w = w4[c4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float xv = x[((k >> 1) + r) * 4 + (k & 1) + q];
                        acc[k][4 * c4] += xv * w.x; acc[k][4 * c4 + 1] += xv * w.y; acc[k][4 * c4 + 2] += xv * w.z; acc[k][4 * c4 + 3] += xv * w.w;
                    }
                }
            }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int c = 0; c < CO; ++c) acc[k][c] = quad_sum(acc[k][c]);
    float pooled[CO / 4];
#pragma unroll
    for (int c = 0; c < CO / 4; ++c) {                      // lane s finishes outputs s CO/4 ...
        const int co = s * (CO / 4) + c;
        const float bc = (bias && co < g.Cout) ? bias[co] : 0.f;
        float best = -INFINITY;
        int bi = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float a0 = acc[k][c], a1 = acc[k][CO / 4 + c], a2 = acc[k][2 * (CO / 4) + c], a3 = acc[k][3 * (CO / 4) + c];
            asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));      // (or the selects become acc[f(s)]: a scratch array)
            float v = (s == 0 ? a0 : s == 1 ? a1 : s == 2 ? a2 : a3) + bc;
            if (alpha != 1.0f) v = v <= 0.f ? alpha * v : v;
            if (v > best) { best = v; bi = k; }
        }
        pooled[c] = best;
        if (co < g.Cout) {
            const int64_t o = ((int64_t)b * g.Cout + co) * HWq + pq;
            P[o] = best;
            arg[o] = bi;
        }
    }
    if constexpr (STATS) {
        // channel s CO/4 + c lives on the 16 lanes of a wave with this s (lane bits 2..5 = the window): xor tree over those
        // bits, the four waves meet in LDS.  Two rounds: the block's mean, then the squared deviations from it.
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        float mean_b[CO / 4];
#pragma unroll
        for (int round = 0; round < 2; ++round) {
            float t[CO / 4];
#pragma unroll
            for (int c = 0; c < CO / 4; ++c) {
                const float d = pooled[c] - (round ? mean_b[c] : 0.f);
                t[c] = round ? d * d : d;
#pragma unroll
                for (int m = 4; m < 64; m <<= 1) t[c] += __shfl_xor(t[c], m, 64);
            }
            if (round) __syncthreads();                      // sred is read below by everybody: the next round overwrites it
            if ((lane >> 2) == 0)
#pragma unroll
                for (int c = 0; c < CO / 4; ++c) sred[wave * CO + s * (CO / 4) + c] = t[c];
            __syncthreads();
#pragma unroll
            for (int c = 0; c < CO / 4; ++c) {
                const int co = s * (CO / 4) + c;
                const float tot = (sred[co] + sred[CO + co]) + (sred[2 * CO + co] + sred[3 * CO + co]);
                if (round == 0) mean_b[c] = tot * (1.0f / 64.0f);
                else if (threadIdx.x < 4 && co < g.Cout) {
                    stats[((int64_t)blockIdx.x * g.Cout + co) * 2] = mean_b[c];
                    stats[((int64_t)blockIdx.x * g.Cout + co) * 2 + 1] = tot;
                }
            }
        }
    }
}

// wgrad, 3x3 kernels, <= 16 channels: one block per image (grid-strided) with X[b] and dO[b] staged in LDS.  Thread <->
// (input channel ci, pixel slice): it owns dW[0..CO)[ci][3x3] for its pixels -- CO*9 accumulators, the 9 taps of a pixel
// loaded once and used for every co, dO[co][p] a broadcast read shared by the CIP threads of the slice.  The slices lie
// along the lanes, so the per-image partial sums meet in a shuffle tree + one LDS hop; per-block partials go to
// conv_wgrad_reduce_kernel.  (Thread <-> output element with a serial loop over the pixels was tried first: 1168 outputs
// x 196 pixels per image left most lanes idle, 70-88 us; the MFMA split-K wgrad above takes 38 us at C5 -- it gathers
// every X element 9 times.)
template <int CO, int CIP>
__global__ __launch_bounds__(256) void conv_direct_wgrad_kernel(const float* __restrict__ X, const float* __restrict__ dO,
                                                                float* __restrict__ part, const ConvGeom g, int ncols) {
    extern __shared__ __attribute__((aligned(16))) float wsm[];
    constexpr int S = 256 / CIP;                       // pixel slices per block
    constexpr int SW = 64 / CIP;                       // slices per wave (CIP <= 16 -> >= 4)
    const int HW = g.H * g.W, HWo = g.Ho * g.Wo;
    float* Xs = wsm;                                   // [Cin][H][W]
    float* Gs = wsm + g.Cin * HW;                      // [Cout][Ho][Wo]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ci = lane % CIP, slice = wave * SW + lane / CIP;
    const bool ci_ok = ci < g.Cin;
    float acc[CO][9], accb[CO];
#pragma unroll
    for (int c = 0; c < CO; ++c) {
        accb[c] = 0.f;
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[c][t] = 0.f;
    }
    for (int b = blockIdx.x; b < g.B; b += gridDim.x) {
        __syncthreads();
        stage_to_lds(Xs, X + (int64_t)b * g.Cin * HW, g.Cin * HW, tid, [](int i) -> int64_t { return i; });
        stage_to_lds(Gs, dO + (int64_t)b * g.Cout * HWo, g.Cout * HWo, tid, [](int i) -> int64_t { return i; });
        __syncthreads();
        for (int p = slice; p < HWo; p += S) {
            const int ho = p / g.Wo, wo = p - ho * g.Wo;
            float x[9];
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const int hi = ho * g.sh - g.pu + r * g.dh, wi = wo * g.sw - g.pl + q * g.dw;
                    const bool ok = ci_ok && hi >= 0 && hi < g.H && wi >= 0 && wi < g.W;
                    x[r * 3 + q] = ok ? Xs[(ci * g.H + hi) * g.W + wi] : 0.f;
                }
#pragma unroll
            for (int c = 0; c < CO; ++c) {
                const float gv = c < g.Cout ? Gs[c * HWo + p] : 0.f;
                accb[c] += gv;
#pragma unroll
                for (int t = 0; t < 9; ++t) acc[c][t] += gv * x[t];
            }
        }
    }
    // ---- sum over the pixel slices: lanes (stride CIP) within the wave, then the 4 waves through LDS ----------------
#pragma unroll
    for (int c = 0; c < CO; ++c) {
#pragma unroll
        for (int o = CIP; o < 64; o <<= 1) {
            accb[c] += __shfl_xor(accb[c], o, 64);
#pragma unroll
            for (int t = 0; t < 9; ++t) acc[c][t] += __shfl_xor(acc[c][t], o, 64);
        }
    }
    __syncthreads();                                   // Xs / Gs are dead: reuse as [4 waves][CO][CIP*9 + 1]
    constexpr int ROW = CIP * 9 + 1;
    float* R = wsm;
    if (lane < CIP) {
#pragma unroll
        for (int c = 0; c < CO; ++c) {
#pragma unroll
            for (int t = 0; t < 9; ++t) R[(wave * CO + c) * ROW + ci * 9 + t] = acc[c][t];
            if (ci == 0) R[(wave * CO + c) * ROW + CIP * 9] = accb[c];
        }
    }
    __syncthreads();
    const int Nw = g.Cin * 9, total = g.Cout * ncols;
    for (int idx = tid; idx < total; idx += 256) {
        const int co = idx / ncols, n = idx - co * ncols;
        const int col = n == Nw ? CIP * 9 : n;         // the db column sits last in R's rows
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) v += R[(w * CO + co) * ROW + col];
        part[(int64_t)blockIdx.x * total + idx] = v;
    }
}

template <int CO, int CIP>
static int launch_direct_wgrad(const float* X, const float* dO, float* part, const ConvGeom& g, int ncols, int blocks,
                               size_t lds, hipStream_t st) {
    auto kern = conv_direct_wgrad_kernel<CO, CIP>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
        if (e != hipSuccess) return hip_status(e, "hipFuncSetAttribute(conv_direct_wgrad_kernel)");
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), lds, st, X, dO, part, g, ncols);
    NNHIP_LAUNCH_CHECK("conv_direct_wgrad_kernel");
    return 0;
}

// wgrad for small channel counts on the 16x16x4 MFMA: per image a [Cout <= 16] x [Cin*9 + 1] x [Ho*Wo] GEMM
//   dW[co][ci,r,q] = sum_p dO[co][p] * X[ci][ho(p)*sh - pu + r*dh][wo(p)*sw - pl + q*dw],   db[co] = sum_p dO[co][p] * 1
// with the image zero-padded in LDS (no bounds logic in the loop) and dO[b] next to it.  Lane (l16, kq): A operand = dO[co = l16]
// [pixel 4s + kq], B operand of column tile t = the padded image at (column n = l16 + 16t: its (ci, r, q) offset, fixed per
// lane) + the pixel's offset; the db column reads a cell that holds 1.0.  The four waves interleave the 4-pixel steps and keep
// their 16 x 16NT accumulators across the images a block walks; they meet once, in LDS, in wave order (deterministic), and the
// per-block partials go to conv_wgrad_reduce_kernel as before.  The direct kernel this replaces (thread <-> (ci, pixel slice),
// CO*9 accumulators per thread, a 3-stage shuffle tree over 160 values) spent its time in LDS latency and that tree: C5 conv2
// 18.5 us, conv1 13.1.
//
// POOL: the conv's output went through [LeakyReLU ->] MaxPool2d with windows that tile it exactly, and NOTHING else reads its
// gradient (the layer's input needs none: C5's first layer).  dO[b] is then never written to memory: the block builds it in LDS
// from the pool's gradient, arg-max and (for the LeakyReLU factor) pooled output -- a quarter of the bytes, and the pool's
// backward launch disappears (nnhipConv2dWeightGradPooled).
struct PoolGrad {
    const float* dP;          // [B, Cout, Hq, Wq] gradient of the pooled output
    const int32_t* arg;       // [B, Cout, Hq, Wq] window-local arg-max (r * kw + s)
    const float* P;           // pooled output of MaxPool(LeakyReLU(.)) or null (no activation in between)
    float alpha;
    int Hq, Wq, kh, kw;
};
typedef float cw_f32x4 __attribute__((ext_vector_type(4)));
template <int NT, bool POOL>
__global__ __launch_bounds__(256) void conv_mfma_wgrad_kernel(const float* __restrict__ X, const float* __restrict__ dO,
                                                              float* __restrict__ part, const ConvGeom g, int ncols,
                                                              const PoolGrad pg) {
    extern __shared__ __attribute__((aligned(16))) float wsm[];
    const int HW = g.H * g.W, HWo = g.Ho * g.Wo;
    const int Hp = (g.Ho - 1) * g.sh + 2 * g.dh + 1, Wp = (g.Wo - 1) * g.sw + 2 * g.dw + 1;   // padded extent the taps reach
    const int S = (HWo + 3) >> 2, Lg = 4 * S;             // 4-pixel steps; dO rows padded with zeros to whole steps
    const int nXp = g.Cin * Hp * Wp;
    float* Xp = wsm;                                       // [Cin][Hp][Wp], then Xp[nXp] = 1.0 (the db column)
    float* Gs = wsm + nXp + 4;                             // [Cout][Lg]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l16 = lane & 15, kq = lane >> 4;
    const int Nw = g.Cin * 9;
    int nbase[NT], nmul[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int n = l16 + 16 * t;
        if (n < Nw) {
            const int ci = n / 9, rq = n - ci * 9, r = rq / 3, q = rq - r * 3;
            nbase[t] = (ci * Hp + r * g.dh) * Wp + q * g.dw;
            nmul[t] = 1;
        } else {
            nbase[t] = nXp;                                // 1.0: the db column (columns past it are never stored)
            nmul[t] = 0;
        }
    }
    const bool co_ok = l16 < g.Cout;
    const int abase = (co_ok ? l16 : 0) * Lg;
    const float inv_wo = 1.0f / (float)g.Wo;
    cw_f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = cw_f32x4{0.f, 0.f, 0.f, 0.f};
    for (int b = blockIdx.x; b < g.B; b += gridDim.x) {
        __syncthreads();                                   // the previous image's readers are done
        const float* xb = X + (int64_t)b * g.Cin * HW;
        stage_to_lds(Xp, xb, nXp, tid, [&](int i) -> int64_t {
            const int c = i / (Hp * Wp), rem = i - c * (Hp * Wp), hp = rem / Wp, wp = rem - hp * Wp;
            const int h = hp - g.pu, w = wp - g.pl;
            return (h >= 0 && h < g.H && w >= 0 && w < g.W) ? (int64_t)c * HW + h * g.W + w : -1;
        });
        if (tid == 0) Xp[nXp] = 1.0f;
        if constexpr (POOL) {
            const int HWq = pg.Hq * pg.Wq, nq = g.Cout * HWq, tail = Lg - HWo;
            const int64_t qb = (int64_t)b * nq;
            for (int i = tid; i < g.Cout * tail; i += 256) Gs[(i / tail) * Lg + HWo + i % tail] = 0.f;      // the rows' padding
            if (pg.kh == 2 && pg.kw == 2 && ((nXp | g.Wo) & 1) == 0) {
                // 2x2 windows (the usual case): eight windows per thread in flight, index arithmetic through reciprocals (exact
                // for the sizes that fit LDS, as below), the window's two rows as two 8-byte stores
                constexpr int D2 = 8;
                const float inv_hwq = 1.0f / (float)HWq, inv_wq = 1.0f / (float)pg.Wq;
                for (int base = 0; base < nq; base += 256 * D2) {
                    float gv[D2], pv[D2];
                    int av[D2];
#pragma unroll
                    for (int j = 0; j < D2; ++j) {
                        const int i = min(base + tid + 256 * j, nq - 1);
                        gv[j] = pg.dP[qb + i];
                        av[j] = pg.arg[qb + i];
                        pv[j] = pg.P ? pg.P[qb + i] : 1.f;
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < D2; ++j) {
                        const int i = base + tid + 256 * j;
                        if (i < nq) {
                            const int c = (int)(((float)i + 0.5f) * inv_hwq), rem = i - c * HWq;
                            const int hq = (int)(((float)rem + 0.5f) * inv_wq), wq = rem - hq * pg.Wq;
                            const float gg = pv[j] <= 0.f ? gv[j] * pg.alpha : gv[j];
                            float* cell = Gs + c * Lg + 2 * hq * g.Wo + 2 * wq;
                            *reinterpret_cast<float2*>(cell) = make_float2(av[j] == 0 ? gg : 0.f, av[j] == 1 ? gg : 0.f);
                            *reinterpret_cast<float2*>(cell + g.Wo) = make_float2(av[j] == 2 ? gg : 0.f, av[j] == 3 ? gg : 0.f);
                        }
                    }
                }
            } else {
            constexpr int D = 4;                            // one thread per pool window: all its loads first, then kh x kw cells
            for (int base = 0; base < nq; base += 256 * D) {
                float gv[D], pv[D];
                int av[D];
#pragma unroll
                for (int j = 0; j < D; ++j) {
                    const int i = base + tid + 256 * j;
                    const bool ok = i < nq;
                    gv[j] = pg.dP[qb + (ok ? i : 0)];
                    av[j] = pg.arg[qb + (ok ? i : 0)];
                    pv[j] = pg.P ? pg.P[qb + (ok ? i : 0)] : 1.f;
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < D; ++j) {
                    const int i = base + tid + 256 * j;
                    if (i >= nq) break;
                    const int c = i / HWq, rem = i - c * HWq, hq = rem / pg.Wq, wq = rem - hq * pg.Wq;
                    const float gg = pv[j] <= 0.f ? gv[j] * pg.alpha : gv[j];        // LeakyReLU factor read off the pooled output
                    float* cell = Gs + c * Lg + hq * pg.kh * g.Wo + wq * pg.kw;
                    for (int r = 0; r < pg.kh; ++r)
                        for (int q = 0; q < pg.kw; ++q) cell[r * g.Wo + q] = (av[j] == r * pg.kw + q) ? gg : 0.f;
                }
            }
            }
        } else {
            const float* gb = dO + (int64_t)b * g.Cout * HWo;
            stage_to_lds(Gs, gb, g.Cout * Lg, tid, [&](int i) -> int64_t {
                const int c = i / Lg, pz = i - c * Lg;
                return pz < HWo ? (int64_t)c * HWo + pz : -1;
            });
        }
        __syncthreads();
        // four steps per iteration, all their LDS reads issued before the first MFMA (one step at a time is address -> LDS
        // latency -> MFMA, 49 times over for a 28x28 image); a step past the last one multiplies a = 0
        for (int s0 = wave; s0 < S; s0 += 16) {
            float av[4], bv[4][NT];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int sx = s0 + 4 * u;
                const bool live = sx < S;
                const int p = 4 * (live ? sx : S - 1) + kq, pc = min(p, HWo - 1);   // a pixel past the image multiplies dO = 0
                // p / Wo through the reciprocal: exact for the pixel counts that fit LDS (|error| << 0.5 / Wo)
                const int ho = (int)(((float)pc + 0.5f) * inv_wo), wo = pc - ho * g.Wo;
                const int poff = ho * g.sh * Wp + wo * g.sw;
                const float gv = Gs[abase + p];
                av[u] = (co_ok && live) ? gv : 0.f;
#pragma unroll
                for (int t = 0; t < NT; ++t) bv[u][t] = Xp[nbase[t] + nmul[t] * poff];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u][t], acc[t], 0, 0, 0);
        }
    }
    // ---- the four waves meet: accumulator v of tile t is D[co = 4 kq + v][n = l16 + 16 t] -------------------------------------
    __syncthreads();
    constexpr int RW = 16 * NT;
    float* R = wsm;                                        // [4 waves][16][RW]
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int v = 0; v < 4; ++v) R[(wave * 16 + 4 * kq + v) * RW + 16 * t + l16] = acc[t][v];
    __syncthreads();
    const int total = g.Cout * ncols;
    for (int idx = tid; idx < total; idx += 256) {
        const int co = idx / ncols, n = idx - co * ncols;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) v += R[(w * 16 + co) * RW + n];
        part[(int64_t)blockIdx.x * total + idx] = v;
    }
}

template <int NT, bool POOL = false>
static int launch_mfma_wgrad(const float* X, const float* dO, float* part, const ConvGeom& g, int ncols, int blocks, size_t lds,
                             hipStream_t st, const PoolGrad pg = PoolGrad{}) {
    auto kern = conv_mfma_wgrad_kernel<NT, POOL>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
        if (e != hipSuccess) return hip_status(e, "hipFuncSetAttribute(conv_mfma_wgrad_kernel)");
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), lds, st, X, dO, part, g, ncols, pg);
    NNHIP_LAUNCH_CHECK("conv_mfma_wgrad_kernel");
    return 0;
}

static bool conv_quad_on() {      // NNHIP_CONV_QUAD=0: the one-thread-per-position kernels everywhere (A/B switch)
    static const bool on = []() { const char* e = getenv("NNHIP_CONV_QUAD"); return !e || atoi(e) != 0; }();
    return on;
}
static bool conv_direct_ok(const ConvGeom& g) {
    static const bool off = []() { const char* e = getenv("NNHIP_CONV_DIRECT"); return e && atoi(e) == 0; }();
    // (both activation tensors are addressed with 32-bit byte offsets below CD_OOB)
    const int64_t lim = (int64_t)1 << 29;
    return !off && g.Cin <= CD_MAXC && g.Cout <= CD_MAXC && g.kh * g.kw <= CD_MAXTAPS && (int64_t)g.B * g.Cin * g.H * g.W < lim &&
           (int64_t)g.B * g.Cout * g.Ho * g.Wo < lim;
}

static int make_geom(const nnhipConv2dDesc* d, ConvGeom& g) {
    NNHIP_CHECK_ARG(d != nullptr, NNHIP_EINVAL, "conv2d: null descriptor");
    NNHIP_CHECK_ARG(d->B >= 0 && d->Cin > 0 && d->H > 0 && d->W > 0 && d->Cout > 0 && d->kh > 0 && d->kw > 0 &&
                        d->sh > 0 && d->sw > 0 && d->dh > 0 && d->dw > 0 && d->pu >= 0 && d->pd >= 0 &&
                        d->pl >= 0 && d->pr >= 0,
                    NNHIP_EINVAL, "conv2d: bad descriptor");
    const int64_t Ho = (d->H + d->pu + d->pd - d->dh * (d->kh - 1) - 1) / d->sh + 1;  // conv2d.py:245-258
    const int64_t Wo = (d->W + d->pl + d->pr - d->dw * (d->kw - 1) - 1) / d->sw + 1;
    NNHIP_CHECK_ARG(Ho > 0 && Wo > 0, NNHIP_EINVAL, "conv2d: empty output");
    const int64_t lim = (int64_t)1 << 31;
    NNHIP_CHECK_ARG(d->B * d->Cin * d->H * d->W < lim * 8 && d->H * d->W < lim && Ho * Wo < lim &&
                        d->Cin * d->kh * d->kw < lim && d->Cout * d->kh * d->kw < lim,
                    NNHIP_EINVAL, "conv2d: dimension too large");
    g.B = (int)d->B; g.Cin = (int)d->Cin; g.H = (int)d->H; g.W = (int)d->W; g.Cout = (int)d->Cout;
    g.kh = (int)d->kh; g.kw = (int)d->kw; g.sh = (int)d->sh; g.sw = (int)d->sw; g.dh = (int)d->dh; g.dw = (int)d->dw;
    g.pu = (int)d->pu; g.pl = (int)d->pl; g.Ho = (int)Ho; g.Wo = (int)Wo;
    return 0;
}

}  // namespace nnhip

using namespace nnhip;

extern "C" int nnhipConv2dForward(const float* X, const float* W, const float* bias, float* O,
                                  const nnhipConv2dDesc* d, nnhipStream_t s) {
    ConvGeom g;
    if (int rc = make_geom(d, g)) return rc;
    if (g.B == 0) return 0;
    NNHIP_CHECK_ARG(X && W && O, NNHIP_EINVAL, "nnhipConv2dForward: null pointer");
    const int64_t N = (int64_t)g.B * g.Ho * g.Wo;
    if (conv_direct_ok(g)) {
        const dim3 dgrid((unsigned)ceil_div(N, 256));
        // under two blocks per CU and a reduction worth splitting: four lanes per position (conv_direct_fwd_quad_kernel)
        if (conv_quad_on() && g.kh == 3 && g.kw == 3 && N <= 512 * 256 && g.Cin >= 4 && g.Cout > 4) {
            const dim3 qgrid((unsigned)ceil_div(4 * N, 256));
            if (g.Cout <= 8) hipLaunchKernelGGL(conv_direct_fwd_quad_kernel<8>, qgrid, dim3(256), 0, (hipStream_t)s, W, X, bias, O, g);
            else hipLaunchKernelGGL(conv_direct_fwd_quad_kernel<16>, qgrid, dim3(256), 0, (hipStream_t)s, W, X, bias, O, g);
            NNHIP_LAUNCH_CHECK("conv_direct_fwd_quad_kernel");
            return 0;
        }
        if (g.Cout <= 4) hipLaunchKernelGGL(conv_direct_fwd_kernel<4>, dgrid, dim3(256), 0, (hipStream_t)s, W, X, bias, O, g);
        else if (g.Cout <= 8) hipLaunchKernelGGL(conv_direct_fwd_kernel<8>, dgrid, dim3(256), 0, (hipStream_t)s, W, X, bias, O, g);
        else hipLaunchKernelGGL(conv_direct_fwd_kernel<16>, dgrid, dim3(256), 0, (hipStream_t)s, W, X, bias, O, g);
        NNHIP_LAUNCH_CHECK("conv_direct_fwd_kernel");
        return 0;
    }
    return conv_mfma_forward(X, W, bias, O, g, (hipStream_t)s);
}

// the LDS plan of conv_mfma_wgrad_kernel for geometry g: true iff that kernel can run it
static bool mfma_wgrad_plan(const ConvGeom& g, int& ncols, int& nt, size_t& m_lds) {
    static const bool mfma_on = []() { const char* e = getenv("NNHIP_CONV_WGRAD_MFMA"); return !e || atoi(e) != 0; }();
    if (!mfma_on || !conv_direct_ok(g) || g.kh != 3 || g.kw != 3) return false;
    ncols = g.Cin * 9 + 1;
    nt = (ncols + 15) / 16;
    const int Hp = (g.Ho - 1) * g.sh + 2 * g.dh + 1, Wp = (g.Wo - 1) * g.sw + 2 * g.dw + 1;
    const int Lg = 4 * ((g.Ho * g.Wo + 3) / 4);
    const int nt_inst = nt <= 3 ? nt : nt <= 5 ? 5 : 10;
    m_lds = ((size_t)g.Cin * Hp * Wp + 4 + (size_t)g.Cout * Lg) * sizeof(float);
    const size_t m_red = (size_t)4 * 16 * 16 * nt_inst * sizeof(float);
    if (m_lds < m_red) m_lds = m_red;
    return nt <= 10 && m_lds <= 60 * 1024;
}
static bool pooled_wgrad_ok(const ConvGeom& g, const nnhipPool2dDesc* pd) {
    static const bool on = []() { const char* e = getenv("NNHIP_CONV_POOLED_WGRAD"); return !e || atoi(e) != 0; }();
    int ncols, nt; size_t lds;
    return on && pd && pd->kh == pd->sh && pd->kw == pd->sw && pd->pu + pd->pd + pd->pl + pd->pr == 0 && pd->dh <= 1 && pd->dw <= 1 &&
           pd->kh >= 1 && pd->kw >= 1 && pd->kh * pd->kw <= 16 && pd->B == g.B && pd->C == g.Cout && pd->H == g.Ho && pd->W == g.Wo &&
           g.Ho % pd->kh == 0 && g.Wo % pd->kw == 0 && mfma_wgrad_plan(g, ncols, nt, lds);
}

extern "C" int nnhipConv2dWeightGradPooledOk(const nnhipConv2dDesc* d, const nnhipPool2dDesc* pd) {
    ConvGeom g;
    if (!d || !pd || make_geom(d, g) || g.B == 0) return 0;
    return pooled_wgrad_ok(g, pd) ? 1 : 0;
}

extern "C" int nnhipConv2dWeightGradPooled(const float* X, const float* dP, const int32_t* argmax, const float* pooled, float alpha,
                                           float* dW, float* db, const nnhipConv2dDesc* d, const nnhipPool2dDesc* pd,
                                           nnhipStream_t s) {
    ConvGeom g;
    if (int rc = make_geom(d, g)) return rc;
    if (g.B == 0) return 0;
    NNHIP_CHECK_ARG(X && dP && argmax && (dW || db), NNHIP_EINVAL, "nnhipConv2dWeightGradPooled: null pointer");
    NNHIP_CHECK_ARG(pooled_wgrad_ok(g, pd), NNHIP_EINVAL,
                    "nnhipConv2dWeightGradPooled: unsupported geometry (ask nnhipConv2dWeightGradPooledOk first)");
    hipStream_t st = (hipStream_t)s;
    int ncols, nt; size_t m_lds;
    mfma_wgrad_plan(g, ncols, nt, m_lds);
    const int Nw = g.Cin * 9, total = g.Cout * ncols, blocks = g.B < 512 ? g.B : 512;
    bool deferred;
    int frc;
    float* part = conv_partials((size_t)blocks * total, st, &deferred, &frc);
    if (frc) return frc;
    NNHIP_CHECK_ARG(part != nullptr, NNHIP_ENOMEM, "nnhipConv2dWeightGradPooled: workspace allocation failed");
    PoolGrad pg;
    pg.dP = dP; pg.arg = argmax; pg.P = pooled; pg.alpha = alpha;
    pg.Hq = g.Ho / (int)pd->kh; pg.Wq = g.Wo / (int)pd->kw; pg.kh = (int)pd->kh; pg.kw = (int)pd->kw;
    int rc = nt <= 1 ? launch_mfma_wgrad<1, true>(X, nullptr, part, g, ncols, blocks, m_lds, st, pg)
           : nt <= 2 ? launch_mfma_wgrad<2, true>(X, nullptr, part, g, ncols, blocks, m_lds, st, pg)
           : nt <= 3 ? launch_mfma_wgrad<3, true>(X, nullptr, part, g, ncols, blocks, m_lds, st, pg)
           : nt <= 5 ? launch_mfma_wgrad<5, true>(X, nullptr, part, g, ncols, blocks, m_lds, st, pg)
                     : launch_mfma_wgrad<10, true>(X, nullptr, part, g, ncols, blocks, m_lds, st, pg);
    if (rc) return rc;
    return conv_reduce(part, dW, db, blocks, g.Cout, Nw, ncols, st, deferred);
}

// one thread per window (conv_pool_fwd_kernel): few input channels, and windows enough to make a grid
static bool conv_pool_window_ok(const ConvGeom& g) {
    return g.sh == 1 && g.sw == 1 && g.dh == 1 && g.dw == 1 && g.Cout > 4 && g.Cin <= 4 && (int64_t)g.B * (g.Ho / 2) * (g.Wo / 2) >= 128 * 256;
}
// four lanes per window (conv_pool_fwd_quad_kernel): where the plain forward would take the quad kernel
static bool conv_pool_quad_ok(const ConvGeom& g) {
    return conv_quad_on() && (int64_t)g.B * g.Ho * g.Wo <= 512 * 256 && g.Cin >= 4 && g.Cout > 4 && g.sh == 1 && g.sw == 1 && g.dh == 1 &&
           g.dw == 1;
}
static bool conv_pool_fwd_ok(const ConvGeom& g, const nnhipPool2dDesc* pd) {
    static const bool on = []() { const char* e = getenv("NNHIP_CONV_POOL_FWD"); return !e || atoi(e) != 0; }();
    // one thread per window: worth it while the windows still make a grid (the second C5 layer's 49 blocks do not)
    if (!(on && pd && pd->kh == 2 && pd->kw == 2 && pd->sh == 2 && pd->sw == 2 && pd->pu + pd->pd + pd->pl + pd->pr == 0 && pd->dh <= 1 &&
          pd->dw <= 1 && pd->B == g.B && pd->C == g.Cout && pd->H == g.Ho && pd->W == g.Wo && g.Ho % 2 == 0 && g.Wo % 2 == 0 &&
          conv_direct_ok(g) && g.kh == 3 && g.kw == 3))
        return false;
    return conv_pool_window_ok(g) || conv_pool_quad_ok(g);
}
extern "C" int nnhipConv2dLeakyMaxPoolForwardOk(const nnhipConv2dDesc* d, const nnhipPool2dDesc* pd) {
    ConvGeom g;
    if (!d || !pd || make_geom(d, g) || g.B == 0) return 0;
    return conv_pool_fwd_ok(g, pd) ? 1 : 0;
}
// Number of 64-window blocks whose per-channel (mean, M2) pairs nnhipConv2dLeakyMaxPoolForwardStats leaves in `stats`
// [blocks][Cout][2]; 0: this geometry has no statistics variant (not the four-lanes-per-window kernel, or a ragged last block).
extern "C" int nnhipConv2dLeakyMaxPoolStatsBlocks(const nnhipConv2dDesc* d, const nnhipPool2dDesc* pd) {
    ConvGeom g;
    if (!d || !pd || make_geom(d, g) || g.B == 0) return 0;
    if (!conv_pool_fwd_ok(g, pd) || conv_pool_window_ok(g)) return 0;
    const int64_t N = (int64_t)g.B * (g.Ho / 2) * (g.Wo / 2);
    return (N % 64 == 0 && N / 64 <= 65535) ? (int)(N / 64) : 0;
}
static int conv_pool_forward(const float* X, const float* W, const float* bias, float alpha, float* P, int32_t* argmax,
                             const nnhipConv2dDesc* d, const nnhipPool2dDesc* pd, float* stats, nnhipStream_t s);
extern "C" int nnhipConv2dLeakyMaxPoolForward(const float* X, const float* W, const float* bias, float alpha, float* P, int32_t* argmax,
                                              const nnhipConv2dDesc* d, const nnhipPool2dDesc* pd, nnhipStream_t s) {
    return conv_pool_forward(X, W, bias, alpha, P, argmax, d, pd, nullptr, s);
}
// The same launch, which also leaves the partial batch statistics of the pooled output (see ...StatsBlocks): what a BatchNorm2d
// behind this layer needs, produced where the values are still in registers.  ABI 209
extern "C" int nnhipConv2dLeakyMaxPoolForwardStats(const float* X, const float* W, const float* bias, float alpha, float* P,
                                                   int32_t* argmax, const nnhipConv2dDesc* d, const nnhipPool2dDesc* pd, float* stats,
                                                   nnhipStream_t s) {
    NNHIP_CHECK_ARG(stats != nullptr && nnhipConv2dLeakyMaxPoolStatsBlocks(d, pd) > 0, NNHIP_EINVAL,
                    "nnhipConv2dLeakyMaxPoolForwardStats: no statistics variant for this geometry (ask nnhipConv2dLeakyMaxPoolStatsBlocks)");
    return conv_pool_forward(X, W, bias, alpha, P, argmax, d, pd, stats, s);
}
static int conv_pool_forward(const float* X, const float* W, const float* bias, float alpha, float* P, int32_t* argmax,
                             const nnhipConv2dDesc* d, const nnhipPool2dDesc* pd, float* stats, nnhipStream_t s) {
    ConvGeom g;
    if (int rc = make_geom(d, g)) return rc;
    if (g.B == 0) return 0;
    NNHIP_CHECK_ARG(X && W && P && argmax, NNHIP_EINVAL, "nnhipConv2dLeakyMaxPoolForward: null pointer");
    NNHIP_CHECK_ARG(alpha > 0.f, NNHIP_EINVAL, "nnhipConv2dLeakyMaxPoolForward: alpha must be > 0 (1 = no activation)");
    NNHIP_CHECK_ARG(conv_pool_fwd_ok(g, pd), NNHIP_EINVAL,
                    "nnhipConv2dLeakyMaxPoolForward: unsupported geometry (ask nnhipConv2dLeakyMaxPoolForwardOk first)");
    if (!conv_pool_window_ok(g)) {                           // four lanes per window, each all four positions
        const dim3 wgrid((unsigned)ceil_div(4 * (int64_t)g.B * (g.Ho / 2) * (g.Wo / 2), 256));
        if (stats) {
            if (g.Cout <= 8) hipLaunchKernelGGL((conv_pool_fwd_quad_kernel<8, true>), wgrid, dim3(256), 0, (hipStream_t)s, W, X, bias, P, argmax, g, alpha, stats);
            else hipLaunchKernelGGL((conv_pool_fwd_quad_kernel<16, true>), wgrid, dim3(256), 0, (hipStream_t)s, W, X, bias, P, argmax, g, alpha, stats);
        } else if (g.Cout <= 8) hipLaunchKernelGGL((conv_pool_fwd_quad_kernel<8, false>), wgrid, dim3(256), 0, (hipStream_t)s, W, X, bias, P, argmax, g, alpha, nullptr);
        else hipLaunchKernelGGL((conv_pool_fwd_quad_kernel<16, false>), wgrid, dim3(256), 0, (hipStream_t)s, W, X, bias, P, argmax, g, alpha, nullptr);
        NNHIP_LAUNCH_CHECK("conv_pool_fwd_quad_kernel");
        return 0;
    }
    const int64_t N = (int64_t)g.B * (g.Ho / 2) * (g.Wo / 2);
    const dim3 grid((unsigned)ceil_div(N, 256));
    if (g.Cout <= 8) hipLaunchKernelGGL(conv_pool_fwd_kernel<8>, grid, dim3(256), 0, (hipStream_t)s, W, X, bias, P, argmax, g, alpha);
    else hipLaunchKernelGGL(conv_pool_fwd_kernel<16>, grid, dim3(256), 0, (hipStream_t)s, W, X, bias, P, argmax, g, alpha);
    NNHIP_LAUNCH_CHECK("conv_pool_fwd_kernel");
    return 0;
}

extern "C" int nnhipConv2dBackward(const float* X, const float* W, const float* dO, float* dX, float* dW,
                                   float* db, const nnhipConv2dDesc* d, nnhipStream_t s) {
    ConvGeom g;
    if (int rc = make_geom(d, g)) return rc;
    if (g.B == 0) return 0;
    NNHIP_CHECK_ARG(X && W && dO, NNHIP_EINVAL, "nnhipConv2dBackward: null pointer");
    hipStream_t st = (hipStream_t)s;
    const bool direct = conv_direct_ok(g);
    if (dX && direct) {
        const dim3 dgrid((unsigned)ceil_div((int64_t)g.B * g.H * g.W, 256));
        const int64_t Nd = (int64_t)g.B * g.H * g.W;
        if (conv_quad_on() && g.kh == 3 && g.kw == 3 && Nd <= 512 * 256 && g.Cout >= 4 && g.Cin > 2 && g.Cin <= 8) {
            static const bool quad2 = []() { const char* e = getenv("NNHIP_CONV_DGRAD_QUAD2"); return !e || atoi(e) != 0; }();
            if (quad2 && g.sh == 1 && g.sw == 1 && g.dh == 1 && g.dw == 1) {      // 2x2 positions per quad of lanes
                const dim3 bgrid((unsigned)ceil_div(4 * (int64_t)g.B * ((g.H + 1) / 2) * ((g.W + 1) / 2), 256));
                if (g.Cin <= 4) hipLaunchKernelGGL(conv_direct_dgrad_quad2_kernel<4>, bgrid, dim3(256), 0, st, W, dO, dX, g);
                else hipLaunchKernelGGL(conv_direct_dgrad_quad2_kernel<8>, bgrid, dim3(256), 0, st, W, dO, dX, g);
                NNHIP_LAUNCH_CHECK("conv_direct_dgrad_quad2_kernel");
            } else {
            const dim3 qgrid((unsigned)ceil_div(4 * Nd, 256));
            if (g.Cin <= 4) hipLaunchKernelGGL(conv_direct_dgrad_quad_kernel<4>, qgrid, dim3(256), 0, st, W, dO, dX, g);
            else hipLaunchKernelGGL(conv_direct_dgrad_quad_kernel<8>, qgrid, dim3(256), 0, st, W, dO, dX, g);
            NNHIP_LAUNCH_CHECK("conv_direct_dgrad_quad_kernel");
            }
        } else
        if (g.Cin <= 4) hipLaunchKernelGGL(conv_direct_dgrad_kernel<4>, dgrid, dim3(256), 0, st, W, dO, dX, g);
        else if (g.Cin <= 8) hipLaunchKernelGGL(conv_direct_dgrad_kernel<8>, dgrid, dim3(256), 0, st, W, dO, dX, g);
        else hipLaunchKernelGGL(conv_direct_dgrad_kernel<16>, dgrid, dim3(256), 0, st, W, dO, dX, g);
        NNHIP_LAUNCH_CHECK("conv_direct_dgrad_kernel");
    } else if (dX) {
        if (int rc = conv_mfma_dgrad(dO, W, dX, g, st)) return rc;
    }
    size_t wg_lds = ((size_t)g.Cin * g.H * g.W + (size_t)g.Cout * g.Ho * g.Wo) * sizeof(float);
    if ((dW || db) && direct && g.kh == 3 && g.kw == 3 && wg_lds <= 60 * 1024) {
        const int Nw = g.Cin * 9, ncols = Nw + 1, total = g.Cout * ncols;
        const int blocks = g.B < 512 ? g.B : 512;
        bool deferred;
        int frc;
        float* part = conv_partials((size_t)blocks * total, st, &deferred, &frc);     // arena (reduce queued) or shared workspace
        if (frc) return frc;
        NNHIP_CHECK_ARG(part != nullptr, NNHIP_ENOMEM, "nnhipConv2dBackward: workspace allocation failed");
        // the 16x16x4-MFMA kernel when its padded image + zero-padded dO fit LDS (NNHIP_CONV_WGRAD_MFMA=0: the direct kernel)
        static const bool mfma_on = []() { const char* e = getenv("NNHIP_CONV_WGRAD_MFMA"); return !e || atoi(e) != 0; }();
        const int Hp = (g.Ho - 1) * g.sh + 2 * g.dh + 1, Wp = (g.Wo - 1) * g.sw + 2 * g.dw + 1;
        const int Lg = 4 * ((g.Ho * g.Wo + 3) / 4), nt = (ncols + 15) / 16;
        const int nt_inst = nt <= 3 ? nt : nt <= 5 ? 5 : 10;   // the instantiation that will run (its LDS meeting area is 16*NT wide)
        size_t m_lds = ((size_t)g.Cin * Hp * Wp + 4 + (size_t)g.Cout * Lg) * sizeof(float);
        const size_t m_red = (size_t)4 * 16 * 16 * nt_inst * sizeof(float);
        if (m_lds < m_red) m_lds = m_red;
        int rc;
        if (mfma_on && nt <= 10 && m_lds <= 60 * 1024) {
            rc = nt <= 1 ? launch_mfma_wgrad<1>(X, dO, part, g, ncols, blocks, m_lds, st)
               : nt <= 2 ? launch_mfma_wgrad<2>(X, dO, part, g, ncols, blocks, m_lds, st)
               : nt <= 3 ? launch_mfma_wgrad<3>(X, dO, part, g, ncols, blocks, m_lds, st)
               : nt <= 5 ? launch_mfma_wgrad<5>(X, dO, part, g, ncols, blocks, m_lds, st)
                         : launch_mfma_wgrad<10>(X, dO, part, g, ncols, blocks, m_lds, st);
            if (rc) return rc;
            return conv_reduce(part, dW, db, blocks, g.Cout, Nw, ncols, st, deferred);
        }
        const int cip = g.Cin <= 1 ? 1 : g.Cin <= 2 ? 2 : g.Cin <= 4 ? 4 : g.Cin <= 8 ? 8 : 16;
        const size_t red = (size_t)4 * (g.Cout <= 8 ? 8 : 16) * (cip * 9 + 1) * sizeof(float);
        if (wg_lds < red) wg_lds = red;
#define NNHIP_WG(CO_)                                                                                                        \
        (cip == 1 ? launch_direct_wgrad<CO_, 1>(X, dO, part, g, ncols, blocks, wg_lds, st)                                      \
         : cip == 2 ? launch_direct_wgrad<CO_, 2>(X, dO, part, g, ncols, blocks, wg_lds, st)                                    \
         : cip == 4 ? launch_direct_wgrad<CO_, 4>(X, dO, part, g, ncols, blocks, wg_lds, st)                                    \
         : cip == 8 ? launch_direct_wgrad<CO_, 8>(X, dO, part, g, ncols, blocks, wg_lds, st)                                    \
                    : launch_direct_wgrad<CO_, 16>(X, dO, part, g, ncols, blocks, wg_lds, st))
        rc = g.Cout <= 8 ? NNHIP_WG(8) : NNHIP_WG(16);
#undef NNHIP_WG
        if (rc) return rc;
        return conv_reduce(part, dW, db, blocks, g.Cout, Nw, ncols, st, deferred);
    } else if (dW || db) {
        if (int rc = conv_mfma_wgrad(X, dO, dW, db, g, st)) return rc;
    }
    return 0;
}
