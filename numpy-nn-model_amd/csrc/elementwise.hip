// elementwise.hip -- bandwidth-bound elementwise kernels: ReLU, Swish, SwiGLU gate, scale, add.
// All are float4 grid-stride loops (16 B/lane = 1 KiB per wave instruction), capped at 8 blocks/CU,
// with a scalar tail.  sigmoid = v_rcp_f32(1 + v_exp_f32(-x log2 e)): the two hardware transcendentals are
// 1 ulp each, so x*sigmoid(x) stays within ~3 ulp of NumPy's float32 result (tests/test_hip_parity.py sweeps
// [-88, 88]); the library expf + IEEE divide were ~30 VALU instructions per element and made the *simplest*
// kernel of the C3 pass (Swish forward) slower than Softmax forward on the same 268 MB (round-1 VERDICT).
#include <stdlib.h>

#include "common.h"

namespace nnhip {

constexpr int EW_THREADS = 256;
constexpr int EW_MAX_BLOCKS = 1 << 20;  // one 4-float4 span per thread at any realistic size; the loop is a backstop

inline int ew_blocks(int64_t n_items) {
    int64_t b = ceil_div(n_items > 0 ? n_items : 1, EW_THREADS);
    return (int)(b < EW_MAX_BLOCKS ? b : EW_MAX_BLOCKS);
}

// Generic unary/binary maps over float4 with scalar tail -----------------------------------------
// Each block walks contiguous 256*EW_U-float4 spans; the EW_U loads of a span are issued together
// (EW_U x 16 B in flight per lane) before any math, then EW_U stores.
constexpr int EW_U = 4;

typedef float ew_f4 __attribute__((ext_vector_type(4)));
// Streaming accesses.  Non-temporal LOADS (bit 0, the default for tensors of >= 128 MB) read the inputs without allocating
// in L2, which leaves L2 to the write-back of the previous kernel's output: cold C3 pass Swish forward 0.58-0.64 -> 0.74-0.78
// of the HBM spec, backward 0.62 -> 0.69-0.77; back-to-back on hot buffers forward -4 %, backward +22 %.  Non-temporal
// STORES (bit 1) win back-to-back (backward 6.8 TB/s) but lose cold (forward 0.58 -> 0.51) -- the consumer of an output is
// usually the next kernel.  NNHIP_EW_NT overrides (developer switch).
__device__ __forceinline__ float4 ew_load(const float4* p, int nt) {
    if (nt & 1) { const ew_f4 t = __builtin_nontemporal_load(reinterpret_cast<const ew_f4*>(p)); return make_float4(t.x, t.y, t.z, t.w); }
    return *p;
}
__device__ __forceinline__ void ew_store(float4* p, float4 v, int nt) {
    if (nt & 2) { ew_f4 t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w; __builtin_nontemporal_store(t, reinterpret_cast<ew_f4*>(p)); }
    else *p = v;
}
static int ew_nt(int64_t n) {
    static const int forced = []() { const char* e = getenv("NNHIP_EW_NT"); return e ? atoi(e) : -1; }();
    if (forced >= 0) return forced;
    return n * 4 >= ((int64_t)128 << 20) ? 1 : 0;
}
template <class F>
__global__ __launch_bounds__(EW_THREADS) void map1_kernel(float* out, const float* a, int64_t n,
                                                          bool vec, F f, int nt) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    if (vec) {
        const int64_t nv = n >> 2;
        const float4* a4 = reinterpret_cast<const float4*>(a);
        float4* o4 = reinterpret_cast<float4*>(out);
        for (int64_t base = (int64_t)blockIdx.x * (EW_THREADS * EW_U); base < nv; base += gsz * EW_U) {
            float4 x[EW_U];
#pragma unroll
            for (int u = 0; u < EW_U; ++u) {
                const int64_t i = base + u * EW_THREADS + threadIdx.x;
                if (i < nv) x[u] = ew_load(a4 + i, nt);
            }
#pragma unroll
            for (int u = 0; u < EW_U; ++u) {
                const int64_t i = base + u * EW_THREADS + threadIdx.x;
                if (i < nv) {
                    float4 y;
                    y.x = f(x[u].x); y.y = f(x[u].y); y.z = f(x[u].z); y.w = f(x[u].w);
                    ew_store(o4 + i, y, nt);
                }
            }
        }
        for (int64_t i = (nv << 2) + gid; i < n; i += gsz) out[i] = f(a[i]);
    } else {
        for (int64_t i = gid; i < n; i += gsz) out[i] = f(a[i]);
    }
}

template <class F>
__global__ __launch_bounds__(EW_THREADS) void map2_kernel(float* out, const float* a, const float* b, int64_t n,
                                                          bool vec, F f, int nt) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    if (vec) {
        const int64_t nv = n >> 2;
        const float4* a4 = reinterpret_cast<const float4*>(a);
        const float4* b4 = reinterpret_cast<const float4*>(b);
        float4* o4 = reinterpret_cast<float4*>(out);
        for (int64_t base = (int64_t)blockIdx.x * (EW_THREADS * EW_U); base < nv; base += gsz * EW_U) {
            float4 x[EW_U], z[EW_U];
#pragma unroll
            for (int u = 0; u < EW_U; ++u) {
                const int64_t i = base + u * EW_THREADS + threadIdx.x;
                if (i < nv) { x[u] = ew_load(a4 + i, nt); z[u] = ew_load(b4 + i, nt); }
            }
#pragma unroll
            for (int u = 0; u < EW_U; ++u) {
                const int64_t i = base + u * EW_THREADS + threadIdx.x;
                if (i < nv) {
                    float4 y;
                    y.x = f(x[u].x, z[u].x); y.y = f(x[u].y, z[u].y); y.z = f(x[u].z, z[u].z); y.w = f(x[u].w, z[u].w);
                    ew_store(o4 + i, y, nt);
                }
            }
        }
        for (int64_t i = (nv << 2) + gid; i < n; i += gsz) out[i] = f(a[i], b[i]);
    } else {
        for (int64_t i = gid; i < n; i += gsz) out[i] = f(a[i], b[i]);
    }
}

struct ReluF { __device__ float operator()(float x) const { return fmaxf(x, 0.f); } };
// dIn = dOut * (out > 0)      (neunet/nn/activations.py:44-45)
struct ReluB { __device__ float operator()(float dy, float y) const { return y > 0.f ? dy : 0.f; } };
// f = x * sigmoid(beta x)     (neunet/nn/activations.py:225-230)
struct SwishF {
    float beta;
    __device__ float operator()(float x) const { return x * sigmoid_fast_(beta * x); }
};
// dx = dy * (beta f + s (1 - beta f)), s = sigmoid(beta x)   (neunet/nn/activations.py:212-216)
struct SwishB {
    float beta;
    __device__ float operator()(float dy, float x) const {
        const float s = sigmoid_fast_(beta * x);
        const float f = x * s;
        return dy * (beta * f + s * (1.f - beta * f));
    }
};
struct ScaleF {
    float alpha;
    __device__ float operator()(float x) const { return alpha * x; }
};
struct AddF { __device__ float operator()(float a, float b) const { return a + b; } };
struct MulF { __device__ float operator()(float a, float b) const { return a * b; } };

template <class F>
static int launch_map1(float* out, const float* a, int64_t n, F f, hipStream_t st, const char* nm) {
    if (n == 0) return 0;
    const bool vec = aligned16(out) && aligned16(a);
    hipLaunchKernelGGL(map1_kernel<F>, dim3(ew_blocks(vec ? ceil_div(n >> 2, EW_U) : n)), dim3(EW_THREADS), 0, st,
                       out, a, n, vec, f, ew_nt(n));
    NNHIP_LAUNCH_CHECK(nm);
    return 0;
}
template <class F>
static int launch_map2(float* out, const float* a, const float* b, int64_t n, F f, hipStream_t st,
                       const char* nm) {
    if (n == 0) return 0;
    const bool vec = aligned16(out) && aligned16(a) && aligned16(b);
    hipLaunchKernelGGL(map2_kernel<F>, dim3(ew_blocks(vec ? ceil_div(n >> 2, EW_U) : n)), dim3(EW_THREADS), 0, st,
                       out, a, b, n, vec, f, ew_nt(n));
    NNHIP_LAUNCH_CHECK(nm);
    return 0;
}

// SwiGLU gate.  in rows = [gate(h) | up(h)] (width 2h), out rows width h.
// (fused_swish_and_mul.cu:13-57 semantics; oracle = Swish(gate)*up on the reference tape)
template <bool VEC>
__global__ __launch_bounds__(EW_THREADS) void swiglu_fwd_kernel(float* __restrict__ out,
                                                                const float* __restrict__ in,
                                                                float beta, int64_t h, int64_t size) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    if constexpr (VEC) {  // h % 4 == 0
        const int64_t hv = h >> 2, nv = size >> 2;
        for (int64_t i = gid; i < nv; i += gsz) {
            const int64_t row = i / hv, c = i - row * hv;
            const float4 g = reinterpret_cast<const float4*>(in)[row * 2 * hv + c];
            const float4 u = reinterpret_cast<const float4*>(in)[row * 2 * hv + hv + c];
            float4 y;
            y.x = g.x * sigmoid_fast_(beta * g.x) * u.x;
            y.y = g.y * sigmoid_fast_(beta * g.y) * u.y;
            y.z = g.z * sigmoid_fast_(beta * g.z) * u.z;
            y.w = g.w * sigmoid_fast_(beta * g.w) * u.w;
            reinterpret_cast<float4*>(out)[i] = y;
        }
    } else {
        for (int64_t i = gid; i < size; i += gsz) {
            const int64_t row = i / h, c = i - row * h;
            const float g = in[row * 2 * h + c], u = in[row * 2 * h + h + c];
            out[i] = g * sigmoid_fast_(beta * g) * u;
        }
    }
}

__device__ __forceinline__ void swiglu_bwd1(float dy, float g, float u, float beta, float& dg,
                                            float& du) {
    const float s = sigmoid_fast_(beta * g);
    const float f = g * s;
    du = dy * f;
    dg = dy * u * (beta * f + s * (1.f - beta * f));
}

template <bool VEC>
__global__ __launch_bounds__(EW_THREADS) void swiglu_bwd_kernel(float* __restrict__ din,
                                                                const float* __restrict__ dout,
                                                                const float* __restrict__ in,
                                                                float beta, int64_t h, int64_t size) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x;
    if constexpr (VEC) {
        const int64_t hv = h >> 2, nv = size >> 2;
        for (int64_t i = gid; i < nv; i += gsz) {
            const int64_t row = i / hv, c = i - row * hv;
            const float4 g = reinterpret_cast<const float4*>(in)[row * 2 * hv + c];
            const float4 u = reinterpret_cast<const float4*>(in)[row * 2 * hv + hv + c];
            const float4 dy = reinterpret_cast<const float4*>(dout)[i];
            float4 dg, du;
            swiglu_bwd1(dy.x, g.x, u.x, beta, dg.x, du.x);
            swiglu_bwd1(dy.y, g.y, u.y, beta, dg.y, du.y);
            swiglu_bwd1(dy.z, g.z, u.z, beta, dg.z, du.z);
            swiglu_bwd1(dy.w, g.w, u.w, beta, dg.w, du.w);
            reinterpret_cast<float4*>(din)[row * 2 * hv + c] = dg;
            reinterpret_cast<float4*>(din)[row * 2 * hv + hv + c] = du;
        }
    } else {
        for (int64_t i = gid; i < size; i += gsz) {
            const int64_t row = i / h, c = i - row * h;
            float dg, du;
            swiglu_bwd1(dout[i], in[row * 2 * h + c], in[row * 2 * h + h + c], beta, dg, du);
            din[row * 2 * h + c] = dg;
            din[row * 2 * h + h + c] = du;
        }
    }
}

struct FillF {
    float v;
    __device__ float operator()(float) const { return v; }
};
int fill_f32(float* p, float v, int64_t n, hipStream_t st) {
    if (n <= 0) return 0;
    if (!p) { set_last_error("fill_f32: null pointer"); return NNHIP_EINVAL; }
    return launch_map1(p, p, n, FillF{v}, st, "fill");
}

// internal: used by linear.hip for dZ = dY * swish'(z), in place over z
int mul_f32(float* out, const float* a, const float* b, int64_t n, hipStream_t st) { return launch_map2(out, a, b, n, MulF{}, st, "mul"); }
int swish_backward_inplace(float* z_inout, const float* dY, float beta, int64_t n, hipStream_t st) {
    return launch_map2(z_inout, dY, z_inout, n, SwishB{beta}, st, "swish_backward(inplace)");
}

}  // namespace nnhip

using namespace nnhip;

#define NNHIP_PTRS(fn, ...)                                                        \
    do {                                                                           \
        const void* _ps[] = {__VA_ARGS__};                                         \
        for (const void* _p : _ps) {                                               \
            NNHIP_CHECK_ARG(_p != nullptr, NNHIP_EINVAL, fn ": null pointer");      \
            NNHIP_CHECK_ARG(aligned4(_p), NNHIP_EALIGN, fn ": misaligned pointer"); \
        }                                                                          \
    } while (0)

extern "C" int nnhipReLUForward(float* out, const float* in, int64_t size, nnhipStream_t s) {
    NNHIP_CHECK_ARG(size >= 0, NNHIP_EINVAL, "nnhipReLUForward: negative size");
    if (size == 0) return 0;
    NNHIP_PTRS("nnhipReLUForward", out, in);
    return launch_map1(out, in, size, ReluF{}, (hipStream_t)s, "relu_forward");
}
extern "C" int nnhipReLUBackward(float* dIn, const float* dOut, const float* out, int64_t size,
                                 nnhipStream_t s) {
    NNHIP_CHECK_ARG(size >= 0, NNHIP_EINVAL, "nnhipReLUBackward: negative size");
    if (size == 0) return 0;
    NNHIP_PTRS("nnhipReLUBackward", dIn, dOut, out);
    return launch_map2(dIn, dOut, out, size, ReluB{}, (hipStream_t)s, "relu_backward");
}
extern "C" int nnhipSwishForward(float* out, const float* in, float beta, int64_t size,
                                 nnhipStream_t s) {
    NNHIP_CHECK_ARG(size >= 0, NNHIP_EINVAL, "nnhipSwishForward: negative size");
    if (size == 0) return 0;
    NNHIP_PTRS("nnhipSwishForward", out, in);
    return launch_map1(out, in, size, SwishF{beta}, (hipStream_t)s, "swish_forward");
}
extern "C" int nnhipSwishBackward(float* dIn, const float* dOut, const float* in, float beta,
                                  int64_t size, nnhipStream_t s) {
    NNHIP_CHECK_ARG(size >= 0, NNHIP_EINVAL, "nnhipSwishBackward: negative size");
    if (size == 0) return 0;
    NNHIP_PTRS("nnhipSwishBackward", dIn, dOut, in);
    return launch_map2(dIn, dOut, in, size, SwishB{beta}, (hipStream_t)s, "swish_backward");
}
extern "C" int nnhipFusedSwishAndMul(float* out, const float* in, float beta, int64_t hidden,
                                     int64_t size, nnhipStream_t s) {
    NNHIP_CHECK_ARG(size >= 0 && hidden > 0 && size % hidden == 0, NNHIP_EINVAL,
                    "nnhipFusedSwishAndMul: size must be a non-negative multiple of hidden > 0");
    if (size == 0) return 0;
    NNHIP_PTRS("nnhipFusedSwishAndMul", out, in);
    const bool vec = (hidden % 4 == 0) && aligned16(out) && aligned16(in);
    if (vec)
        hipLaunchKernelGGL(swiglu_fwd_kernel<true>, dim3(ew_blocks(size >> 2)), dim3(EW_THREADS), 0,
                           (hipStream_t)s, out, in, beta, hidden, size);
    else
        hipLaunchKernelGGL(swiglu_fwd_kernel<false>, dim3(ew_blocks(size)), dim3(EW_THREADS), 0,
                           (hipStream_t)s, out, in, beta, hidden, size);
    NNHIP_LAUNCH_CHECK("swiglu_fwd_kernel");
    return 0;
}
extern "C" int nnhipFusedSwishAndMulBackward(float* dIn, const float* dOut, const float* in,
                                             float beta, int64_t hidden, int64_t size,
                                             nnhipStream_t s) {
    NNHIP_CHECK_ARG(size >= 0 && hidden > 0 && size % hidden == 0, NNHIP_EINVAL,
                    "nnhipFusedSwishAndMulBackward: size must be a non-negative multiple of hidden > 0");
    if (size == 0) return 0;
    NNHIP_PTRS("nnhipFusedSwishAndMulBackward", dIn, dOut, in);
    const bool vec = (hidden % 4 == 0) && aligned16(dIn) && aligned16(dOut) && aligned16(in);
    if (vec)
        hipLaunchKernelGGL(swiglu_bwd_kernel<true>, dim3(ew_blocks(size >> 2)), dim3(EW_THREADS), 0,
                           (hipStream_t)s, dIn, dOut, in, beta, hidden, size);
    else
        hipLaunchKernelGGL(swiglu_bwd_kernel<false>, dim3(ew_blocks(size)), dim3(EW_THREADS), 0,
                           (hipStream_t)s, dIn, dOut, in, beta, hidden, size);
    NNHIP_LAUNCH_CHECK("swiglu_bwd_kernel");
    return 0;
}
// ---- Dropout (neunet/nn/layers/dropout.py:17-37): out = in * mask, mask = Bernoulli(1 - p) / (1 - p) -------------------------
// The mask is a counter-based hash of (seed [+ a device word], element index) -- the same construction as the fused
// attention's dropout (attention.hip: at_hash) -- so it is never stored: the backward pass calls the same entry with the
// upstream gradient as `in` and gets dX = dY * mask.  `seed_dev` (optional device uint32, e.g. the optimizer's step
// counter) is added to the seed inside the kernel: a captured hipGraph then draws a fresh mask on every replay.
// The reference draws its mask with the host NumPy RNG: streams cannot match, parity is tested with injected masks
// (nnhipMul) and statistically (keep rate, scale, forward/backward consistency).
__global__ __launch_bounds__(256) void dropout_hash_kernel(float* __restrict__ out, const float* __restrict__ in, int64_t n,
                                                           unsigned seed, const unsigned* __restrict__ seed_dev, unsigned threshold,
                                                           float scale, bool vec) {
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x, gsz = (int64_t)gridDim.x * 256;
    const unsigned sd = (seed + (seed_dev ? __hip_atomic_load(seed_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u)) * 0x85EBCA6Bu + 0x9E3779B9u;
    auto keep = [&](int64_t i) {
        unsigned x = sd ^ ((unsigned)i * 0x9E3779B1u) ^ ((unsigned)(i >> 32) * 0xC2B2AE35u);
        x ^= x >> 16; x *= 0x7FEB352Du;
        x ^= x >> 15; x *= 0x846CA68Bu;
        x ^= x >> 16;
        return x >= threshold ? scale : 0.f;
    };
    if (vec) {
        const int64_t nv = n >> 2;
        for (int64_t i = gid; i < nv; i += gsz) {
            const float4 x = reinterpret_cast<const float4*>(in)[i];
            reinterpret_cast<float4*>(out)[i] = make_float4(x.x * keep(4 * i), x.y * keep(4 * i + 1), x.z * keep(4 * i + 2), x.w * keep(4 * i + 3));
        }
        for (int64_t i = (nv << 2) + gid; i < n; i += gsz) out[i] = in[i] * keep(i);
    } else {
        for (int64_t i = gid; i < n; i += gsz) out[i] = in[i] * keep(i);
    }
}
extern "C" int nnhipDropout(float* out, const float* in, int64_t n, float p, uint32_t seed, const uint32_t* seed_dev,
                            nnhipStream_t s) {
    NNHIP_CHECK_ARG(n >= 0 && p >= 0.f && p <= 1.f, NNHIP_EINVAL, "nnhipDropout: need n >= 0 and 0 <= p <= 1");
    if (n == 0) return 0;
    NNHIP_PTRS("nnhipDropout", out, in);
    const double th = (double)p * 4294967296.0;
    const unsigned threshold = (unsigned)(th < 4294967295.0 ? th : 4294967295.0);
    const float scale = p < 1.f ? 1.0f / (1.0f - p) : 0.f;
    const bool vec = aligned16(out) && aligned16(in);
    hipLaunchKernelGGL(dropout_hash_kernel, dim3(ew_blocks(vec ? (n >> 2) + 1 : n)), dim3(EW_THREADS), 0, (hipStream_t)s, out, in, n,
                       (unsigned)seed, reinterpret_cast<const unsigned*>(seed_dev), threshold, scale, vec);
    NNHIP_LAUNCH_CHECK("dropout_hash_kernel");
    return 0;
}
__global__ void increment_u32_kernel(unsigned* p, unsigned by) { *p += by; }
extern "C" int nnhipIncrementU32(uint32_t* word, uint32_t by, nnhipStream_t s) {
    NNHIP_CHECK_ARG(word != nullptr, NNHIP_EINVAL, "nnhipIncrementU32: null pointer");
    hipLaunchKernelGGL(increment_u32_kernel, dim3(1), dim3(1), 0, (hipStream_t)s, reinterpret_cast<unsigned*>(word), (unsigned)by);
    NNHIP_LAUNCH_CHECK("increment_u32_kernel");
    return 0;
}
extern "C" int nnhipScale(float* x, float alpha, int64_t n, nnhipStream_t s) {
    NNHIP_CHECK_ARG(n >= 0, NNHIP_EINVAL, "nnhipScale: negative size");
    if (n == 0) return 0;
    NNHIP_PTRS("nnhipScale", x);
    return launch_map1(x, x, n, ScaleF{alpha}, (hipStream_t)s, "scale");
}
// out[r, c] = in[r, c] * scale[r * scale_stride]: the product with an upstream gradient that a loss node's backward forms
// (cross_entropy.py:111-114 `grad_y_pred * grad`, losses.py:9-22): scale_stride = 0 -- one DEVICE scalar (loss.backward(g) on a
// reduced loss), 1 -- one factor per row (reduction 'none').  One thread per element.
__global__ __launch_bounds__(256) void scale_rows_kernel(float* __restrict__ out, const float* __restrict__ in, const float* __restrict__ scale,
                                                         int64_t rows, int64_t cols, int64_t scale_stride) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * cols) return;
    out[i] = in[i] * scale[(i / cols) * scale_stride];
}
extern "C" int nnhipScaleRows(float* out, const float* in, const float* scale, int64_t rows, int64_t cols, int64_t scale_stride, nnhipStream_t s) {
    NNHIP_CHECK_ARG(rows >= 0 && cols >= 0 && (scale_stride == 0 || scale_stride == 1), NNHIP_EINVAL, "nnhipScaleRows: bad sizes");
    if (rows * cols == 0) return 0;
    NNHIP_PTRS("nnhipScaleRows", out, in, scale);
    hipLaunchKernelGGL(scale_rows_kernel, dim3((unsigned)ceil_div(rows * cols, 256)), dim3(256), 0, (hipStream_t)s, out, in, scale, rows, cols,
                       scale_stride);
    NNHIP_LAUNCH_CHECK("scale_rows_kernel");
    return 0;
}
extern "C" int nnhipAdd(float* out, const float* a, const float* b, int64_t n, nnhipStream_t s) {
    NNHIP_CHECK_ARG(n >= 0, NNHIP_EINVAL, "nnhipAdd: negative size");
    if (n == 0) return 0;
    NNHIP_PTRS("nnhipAdd", out, a, b);
    return launch_map2(out, a, b, n, AddF{}, (hipStream_t)s, "add");
}
extern "C" int nnhipMul(float* out, const float* a, const float* b, int64_t n, nnhipStream_t s) {
    NNHIP_CHECK_ARG(n >= 0, NNHIP_EINVAL, "nnhipMul: negative size");
    if (n == 0) return 0;
    NNHIP_PTRS("nnhipMul", out, a, b);
    return launch_map2(out, a, b, n, MulF{}, (hipStream_t)s, "mul");
}
