// pool_norm.hip -- the remaining ops of the conv-classifier step (BASELINE config 5;
// examples/convolutional_digits_classifier.ipynb cell 2): LeakyReLU, Sigmoid, MaxPool2d, BatchNorm2d, MSELoss.
// All are small, HBM/latency-bound kernels on NCHW fp32.
#include <math.h>

#include "common.h"

namespace nnhip {

// ---- generic float4 maps (same shape as elementwise.hip) -----------------------------------------------------
template <class F>
__global__ __launch_bounds__(256) void pn_map1(float* out, const float* a, int64_t n, bool vec, F f) {
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x, gsz = (int64_t)gridDim.x * 256;
    if (vec) {
        const int64_t nv = n >> 2;
        for (int64_t i = gid; i < nv; i += gsz) {
            const float4 x = reinterpret_cast<const float4*>(a)[i];
            reinterpret_cast<float4*>(out)[i] = make_float4(f(x.x), f(x.y), f(x.z), f(x.w));
        }
        for (int64_t i = (nv << 2) + gid; i < n; i += gsz) out[i] = f(a[i]);
    } else {
        for (int64_t i = gid; i < n; i += gsz) out[i] = f(a[i]);
    }
}
template <class F>
__global__ __launch_bounds__(256) void pn_map2(float* out, const float* a, const float* b, int64_t n, bool vec, F f) {
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x, gsz = (int64_t)gridDim.x * 256;
    if (vec) {
        const int64_t nv = n >> 2;
        for (int64_t i = gid; i < nv; i += gsz) {
            const float4 x = reinterpret_cast<const float4*>(a)[i], z = reinterpret_cast<const float4*>(b)[i];
            reinterpret_cast<float4*>(out)[i] = make_float4(f(x.x, z.x), f(x.y, z.y), f(x.z, z.z), f(x.w, z.w));
        }
        for (int64_t i = (nv << 2) + gid; i < n; i += gsz) out[i] = f(a[i], b[i]);
    } else {
        for (int64_t i = gid; i < n; i += gsz) out[i] = f(a[i], b[i]);
    }
}
static inline unsigned pn_blocks(int64_t n) {
    int64_t b = ceil_div(n > 0 ? n : 1, 1024);
    return (unsigned)(b < 65535 ? b : 65535);
}

// f = x <= 0 ? alpha x : x                       (neunet/nn/activations.py:79-81)
struct LeakyF { float alpha; __device__ float operator()(float x) const { return x <= 0.f ? alpha * x : x; } };
// dx = dy * (f <= 0 ? alpha : 1)                 (activations.py:64-68)
struct LeakyB { float alpha; __device__ float operator()(float dy, float f) const { return f <= 0.f ? dy * alpha : dy; } };
// f = 1 / (1 + exp(-x))                          (activations.py:24-25)
struct SigmoidF { __device__ float operator()(float x) const { return 1.0f / (1.0f + expf(-x)); } };
// dx = dy * f * (1 - f)                          (activations.py:12-13)
struct SigmoidB { __device__ float operator()(float dy, float f) const { return dy * f * (1.0f - f); } };

// ---- MaxPool2d (neunet/nn/layers/maxpool2d.py:85-249; dilation: taps at r*dh, s*dw -- the reference multiplies the
// dilated window by a kernel that is NaN between the taps and takes nanmax / nanargmax, :187-220) ----------------------------------------------
// Forward: one thread per output; windows read -inf outside the padded input; the FIRST maximum in (r, s)
// row-major order is remembered (np.nanargmax, maxpool2d.py:37).  Backward is a gather over the windows that
// cover an input pixel (deterministic, also correct for overlapping windows).
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(float* __restrict__ out, int32_t* __restrict__ arg,
                                                          const float* __restrict__ x, int64_t BC, int H, int W,
                                                          int Ho, int Wo, int kh, int kw, int sh, int sw, int pu,
                                                          int pl, int dh, int dw, float pre_alpha) {
    // pre_alpha != 1: the window is taken over LeakyReLU(x; pre_alpha), evaluated on the fly (nnhipMaxPool2dLeakyForward)
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total = BC * Ho * Wo;
    if (i >= total) return;
    int wo, ho;
    int64_t bc;
    if (total < ((int64_t)1 << 31)) {     // 32-bit index arithmetic: the 64-bit divisions were most of this kernel's time
        const unsigned u = (unsigned)i, hw = (unsigned)(Ho * Wo), b32 = u / hw, rem = u - b32 * hw;
        ho = (int)(rem / (unsigned)Wo); wo = (int)(rem - (unsigned)ho * (unsigned)Wo); bc = b32;
    } else {
        wo = (int)(i % Wo); ho = (int)((i / Wo) % Ho); bc = i / ((int64_t)Ho * Wo);
    }
    const float* p = x + bc * H * W;
    float best = -INFINITY;
    int bi = 0;
    for (int r = 0; r < kh; ++r)
        for (int s = 0; s < kw; ++s) {
            const int y = ho * sh - pu + r * dh, xx = wo * sw - pl + s * dw;
            float v = -INFINITY;
            if (y >= 0 && y < H && xx >= 0 && xx < W) {
                v = p[(int64_t)y * W + xx];
                if (pre_alpha != 1.0f) v = v <= 0.f ? pre_alpha * v : v;       // LeakyF
            }
            if (v > best) { best = v; bi = r * kw + s; }
        }
    out[i] = best;
    arg[i] = bi;
}

__global__ __launch_bounds__(256) void maxpool_bwd_kernel(float* __restrict__ dx, const float* __restrict__ dy,
                                                          const int32_t* __restrict__ arg, int64_t BC, int H, int W,
                                                          int Ho, int Wo, int kh, int kw, int sh, int sw, int pu,
                                                          int pl, int dh, int dw, const float* __restrict__ pooled, float alpha) {
    // pooled != null: dx is the gradient of the LeakyReLU's INPUT -- the routed gradient times LeakyB's factor, read off
    // the pooled output (= the LeakyReLU output at the arg-max: f <= 0 ? alpha : 1)
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total = BC * H * W;
    if (i >= total) return;
    int xx, y;
    int64_t bc;
    if (total < ((int64_t)1 << 31)) {
        const unsigned u = (unsigned)i, hw = (unsigned)(H * W), b32 = u / hw, rem = u - b32 * hw;
        y = (int)(rem / (unsigned)W); xx = (int)(rem - (unsigned)y * (unsigned)W); bc = b32;
    } else {
        xx = (int)(i % W); y = (int)((i / W) % H); bc = i / ((int64_t)H * W);
    }
    float g = 0.f;
    for (int r = 0; r < kh; ++r) {
        const int ty = y + pu - r * dh;
        if (ty < 0 || ty % sh) continue;
        const int ho = ty / sh;
        if (ho >= Ho) continue;
        for (int s = 0; s < kw; ++s) {
            const int tx = xx + pl - s * dw;
            if (tx < 0 || tx % sw) continue;
            const int wo = tx / sw;
            if (wo >= Wo) continue;
            const int64_t o = (bc * Ho + ho) * Wo + wo;
            if (arg[o] == r * kw + s) g += (pooled && pooled[o] <= 0.f) ? dy[o] * alpha : dy[o];
        }
    }
    dx[i] = g;
}

// Non-overlapping windows that tile the input exactly (kernel == stride, no padding, H = Ho*kh, W = Wo*kw: the conv
// classifier's 2x2/2 pools): one thread per WINDOW writes its kh x kw input gradients -- the routed one and zeros -- with
// no per-pixel division / modulo (the gather kernel above spends 15 us on 1.6 M pixels, this one ~4).
template <int KW>
__global__ __launch_bounds__(256) void maxpool_bwd_tiles_kernel(float* __restrict__ dx, const float* __restrict__ dy,
                                                                const int32_t* __restrict__ arg, int64_t total_out, int W,
                                                                int Ho, int Wo, int kh, int kw_rt,
                                                                const float* __restrict__ pooled, float alpha) {
    const int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (o >= total_out) return;
    const int kw = KW ? KW : kw_rt;
    const int64_t row_o = o / Wo;                       // bc * Ho + ho
    const int wo = (int)(o - row_o * Wo);
    float g = dy[o];
    if (pooled && pooled[o] <= 0.f) g *= alpha;
    const int a = arg[o];
    float* __restrict__ base = dx + row_o * kh * (int64_t)W + (int64_t)wo * kw;     // H = Ho * kh: row (bc*Ho + ho)*kh of [BC*H, W]
    for (int r = 0; r < kh; ++r) {
        if constexpr (KW == 2) {
            *reinterpret_cast<float2*>(base + (int64_t)r * W) = make_float2(a == 2 * r ? g : 0.f, a == 2 * r + 1 ? g : 0.f);
        } else {
            for (int s = 0; s < kw; ++s) base[(int64_t)r * W + s] = (a == r * kw + s) ? g : 0.f;
        }
    }
}

// ---- BatchNorm2d (neunet/nn/layers/batchnorm2d.py:57-115 fwd, 11-54 bwd) --------------------------------------
// Statistics: two passes (mean, then mean of squared deviations = np.var, biased).
// stats[c] = {mean, var};  running = momentum*running + (1-momentum)*stat  (the reference's convention, :84-85).
// One 1024-thread block per channel: wave w walks images w, w+16, ..., its lanes the HW contiguous floats of one image
// -- no per-element index division (the 256-thread / divide-per-element version took 33 us for 256x16x7x7).
__global__ __launch_bounds__(1024) void bn_stats_kernel(const float* __restrict__ x, float* __restrict__ mean_out,
                                                        float* __restrict__ inv_out, float* __restrict__ run_mean,
                                                        float* __restrict__ run_var, int B, int C, int HW, float eps,
                                                        float momentum) {
    __shared__ float red[16];
    const int c = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float n = (float)((int64_t)B * HW);
    float s = 0.f;
    for (int b = wave; b < B; b += 16) {
        const float* xb = x + ((int64_t)b * C + c) * HW;
        for (int i = lane; i < HW; i += 64) s += xb[i];
    }
    const float mean = block_sum<16>(s, red) / n;
    float q = 0.f;
    for (int b = wave; b < B; b += 16) {
        const float* xb = x + ((int64_t)b * C + c) * HW;
        for (int i = lane; i < HW; i += 64) {
            const float d = xb[i] - mean;
            q += d * d;
        }
    }
    const float var = block_sum<16>(q, red) / n;
    if (threadIdx.x == 0) {
        mean_out[c] = mean;
        inv_out[c] = 1.0f / sqrtf(var + eps);
        if (run_mean) {
            run_mean[c] = momentum * run_mean[c] + (1.0f - momentum) * mean;
            run_var[c] = momentum * run_var[c] + (1.0f - momentum) * var;
        }
    }
}
// eval mode: mean = running_mean, inv = 1/sqrt(running_var + eps)
__global__ void bn_eval_stats_kernel(const float* __restrict__ run_mean, const float* __restrict__ run_var,
                                     float* __restrict__ mean_out, float* __restrict__ inv_out, int C, float eps) {
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c < C) { mean_out[c] = run_mean[c]; inv_out[c] = 1.0f / sqrtf(run_var[c] + eps); }
}
// y = (x - mean[c]) * inv[c] * w[c] + b[c]
__global__ __launch_bounds__(256) void bn_apply_kernel(float* __restrict__ y, const float* __restrict__ x,
                                                       const float* __restrict__ mean, const float* __restrict__ inv,
                                                       const float* __restrict__ w, const float* __restrict__ b,
                                                       int64_t total, int C, int HW) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c = (int)((i / HW) % C);
    float v = (x[i] - mean[c]) * inv[c];
    if (w) v = w[c] * v + b[c];
    y[i] = v;
}
// per-channel sums for the backward: sums[c] = {sum dxh*xc, sum dxh, sum g*xhat, sum g},  dxh = w*g
__global__ __launch_bounds__(1024) void bn_bwd_stats_kernel(const float* __restrict__ g, const float* __restrict__ x,
                                                            const float* __restrict__ mean,
                                                            const float* __restrict__ inv, const float* __restrict__ w,
                                                            float* __restrict__ sums, int B, int C, int HW) {
    __shared__ float red[32];
    const int c = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float m = mean[c], iv = inv[c], wc = w ? w[c] : 1.f;
    float s1 = 0.f, s2 = 0.f, s3 = 0.f, s4 = 0.f;
    for (int b = wave; b < B; b += 16) {
        const int64_t o0 = ((int64_t)b * C + c) * HW;
        for (int i = lane; i < HW; i += 64) {
            const float gg = g[o0 + i], xc = x[o0 + i] - m;
            s1 += wc * gg * xc;
            s2 += wc * gg;
            s3 += gg * (xc * iv);
            s4 += gg;
        }
    }
    block_sum2<16>(s1, s2, red);
    block_sum2<16>(s3, s4, red);
    if (threadIdx.x == 0) { sums[4 * c] = s1; sums[4 * c + 1] = s2; sums[4 * c + 2] = s3; sums[4 * c + 3] = s4; }
}
// grad_X = dxh*inv + dvar + dmean ;  dstd_inv = -0.5 inv^3 s1 ; dvar = dstd_inv*2*xc/N ; dmean = -(s2*inv)/N
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(float* __restrict__ dx, const float* __restrict__ g,
                                                           const float* __restrict__ x, const float* __restrict__ mean,
                                                           const float* __restrict__ inv, const float* __restrict__ w,
                                                           const float* __restrict__ sums, float* __restrict__ dw,
                                                           float* __restrict__ db, int64_t total, int C, int HW,
                                                           float invN) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < C && dw) { dw[i] = sums[4 * i + 2]; db[i] = sums[4 * i + 3]; }
    if (i >= total) return;
    const int c = (int)((i / HW) % C);
    const float iv = inv[c], wc = w ? w[c] : 1.f;
    const float xc = x[i] - mean[c];
    const float dstd = -0.5f * (iv * iv * iv) * sums[4 * c];
    dx[i] = wc * g[i] * iv + dstd * 2.0f * xc * invN + sums[4 * c + 1] * iv * (-1.0f) * invN;
}

// ---- one launch per direction when a channel fits the registers of one block ------------------------------------------
// A 1024-thread block holds up to NE = 16 elements per thread: ceil(B/16) * ceil(HW/64) <= 16 (the conv classifier's
// 256 x 16 x 7 x 7 is exactly 16 x 1).  Same element -> thread map and the same reduction order as bn_stats_kernel /
// bn_bwd_stats_kernel, so the statistics are bit-identical to the two-launch path; the point is the launch it saves
// (C5 is bound by its ~30 graph nodes, not by bytes).
constexpr int BN_NE = 16;
// element e of this thread: image wave + 16 (e / nh), position lane + 64 (e % nh); off[e] = its offset from the channel's
// first element, or -1.  Computed once (the runtime division by nh is not something to repeat in three loops: the first
// version of the backward kernel spilled 1.4 KB per lane and took 316 us).
// loads through a buffer descriptor (channel base in SGPRs + a 32-bit offset per element): 32 plain global loads in flight
// would hold 32 64-bit address pairs -- that alone spilled the backward kernel
__device__ __forceinline__ __amdgpu_buffer_rsrc_t bn_rsrc(const float* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, 0xFFFFFFFF, 0x00020000);
}
__device__ __forceinline__ float bn_ld(__amdgpu_buffer_rsrc_t rs, int off) {     // off < 0: no element -> 0
    const float v = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (off >= 0 ? off : 0) * 4, 0, 0));
    return off >= 0 ? v : 0.f;
}
__device__ __forceinline__ void bn_offsets(int (&off)[BN_NE], int B, int C, int HW) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nh = (HW + 63) >> 6;
    int eb = 0, eh = 0;                                     // e / nh, e % nh carried incrementally
#pragma unroll
    for (int e = 0; e < BN_NE; ++e) {
        const int b = wave + 16 * eb, i = lane + 64 * eh;
        off[e] = (b < B && i < HW) ? b * C * HW + i : -1;
        if (++eh == nh) { eh = 0; ++eb; }
    }
}
__global__ __launch_bounds__(1024) void bn_fwd_fused_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                            const float* __restrict__ w, const float* __restrict__ bias,
                                                            float* __restrict__ mean_out, float* __restrict__ inv_out,
                                                            float* __restrict__ run_mean, float* __restrict__ run_var, int B,
                                                            int C, int HW, float eps, float momentum) {
    __shared__ float red[16];
    const int c = blockIdx.x;
    const float n = (float)((int64_t)B * HW);
    const __amdgpu_buffer_rsrc_t rx = bn_rsrc(x + (int64_t)c * HW);
    float* __restrict__ yc_ = y + (int64_t)c * HW;
    int off[BN_NE];
    bn_offsets(off, B, C, HW);
    float v[BN_NE];
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < BN_NE; ++e) {
        v[e] = bn_ld(rx, off[e]);
        s += v[e];
    }
    const float mean = block_sum<16>(s, red) / n;
    float q = 0.f;
#pragma unroll
    for (int e = 0; e < BN_NE; ++e)
        if (off[e] >= 0) { const float d = v[e] - mean; q += d * d; }
    const float var = block_sum<16>(q, red) / n;
    const float inv = 1.0f / sqrtf(var + eps);
    if (threadIdx.x == 0) {
        mean_out[c] = mean;
        inv_out[c] = inv;
        if (run_mean) {
            run_mean[c] = momentum * run_mean[c] + (1.0f - momentum) * mean;
            run_var[c] = momentum * run_var[c] + (1.0f - momentum) * var;
        }
    }
    const float wc = w ? w[c] : 1.f, bc = w ? bias[c] : 0.f;
#pragma unroll
    for (int e = 0; e < BN_NE; ++e)
        if (off[e] >= 0) {
            float t = (v[e] - mean) * inv;
            if (w) t = wc * t + bc;
            yc_[off[e]] = t;
        }
}
__global__ __launch_bounds__(1024) void bn_bwd_fused_kernel(const float* __restrict__ g, const float* __restrict__ x,
                                                            const float* __restrict__ mean, const float* __restrict__ inv,
                                                            const float* __restrict__ w, float* __restrict__ dx,
                                                            float* __restrict__ dw, float* __restrict__ db, int B, int C, int HW,
                                                            float invN) {
    __shared__ float red[32];
    __shared__ float tot[2];
    const int c = blockIdx.x;
    const float m = mean[c], iv = inv[c], wc = w ? w[c] : 1.f;
    const __amdgpu_buffer_rsrc_t rg = bn_rsrc(g + (int64_t)c * HW), rx = bn_rsrc(x + (int64_t)c * HW);
    float* __restrict__ dc_ = dx + (int64_t)c * HW;
    int off[BN_NE];
    bn_offsets(off, B, C, HW);
    float gv[BN_NE], xc[BN_NE];
    float s1 = 0.f, s2 = 0.f, s3 = 0.f, s4 = 0.f;
    // every load is issued before the first value is used (with `off >= 0 ? bn_ld(..) - m : 0` the x load sat under a branch
    // that ended in s_waitcnt vmcnt(0): 16 memory round trips per thread)
#pragma unroll
    for (int e = 0; e < BN_NE; ++e) {
        gv[e] = bn_ld(rg, off[e]);                            // 0 for a slot without an element: all four sums take 0
        xc[e] = bn_ld(rx, off[e]);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int e = 0; e < BN_NE; ++e) {
        xc[e] = off[e] >= 0 ? xc[e] - m : 0.f;
        s1 += wc * gv[e] * xc[e];
        s2 += wc * gv[e];
        s3 += gv[e] * (xc[e] * iv);
        s4 += gv[e];
    }
    block_sum2<16>(s1, s2, red);
    block_sum2<16>(s3, s4, red);
    if (threadIdx.x == 0) {
        tot[0] = s1; tot[1] = s2;
        if (dw) { dw[c] = s3; db[c] = s4; }
    }
    __syncthreads();
    const float dstd = -0.5f * (iv * iv * iv) * tot[0];
    const float t2 = tot[1];
#pragma unroll
    for (int e = 0; e < BN_NE; ++e)
        if (off[e] >= 0) dc_[off[e]] = wc * gv[e] * iv + dstd * 2.0f * xc[e] * invN + t2 * iv * (-1.0f) * invN;
}
// (also guarantees that every offset fits an int: B * HW <= 16384 elements per channel, C * that < 2^31 for C < 131072)
static inline bool bn_fits_fused(int64_t B, int64_t C, int64_t HW) { return ceil_div(B, 16) * ceil_div(HW, 64) <= BN_NE && C < 131072; }

// ---- MSELoss (neunet/nn/losses.py:9-22): loss = sum((p - t)^2) / N ; dp = 2 (p - t) / N -------------------------
__global__ __launch_bounds__(256) void mse_kernel(const float* __restrict__ p, const float* __restrict__ t,
                                                  float* __restrict__ dp, float* __restrict__ part, int64_t n,
                                                  float invN, int sig) {
    __shared__ float red[4];
    float s = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float d = p[i] - t[i];
        s += d * d;
        if (dp) dp[i] = sig ? (2.0f * d * invN) * p[i] * (1.0f - p[i]) : 2.0f * d * invN;   // sig: SigmoidB folded in
    }
    s = block_sum<4>(s, red);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}
__global__ __launch_bounds__(256) void mse_final_kernel(const float* __restrict__ part, int nparts, float invN,
                                                        float* __restrict__ loss) {
    __shared__ float red[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < nparts; i += 256) s += part[i];
    s = block_sum<4>(s, red);
    if (threadIdx.x == 0) loss[0] = s * invN;
}

// small problems (the conv classifier's 256 x 10): one block, one launch
__global__ __launch_bounds__(1024) void mse_small_kernel(const float* __restrict__ p, const float* __restrict__ t,
                                                         float* __restrict__ dp, int64_t n, float invN, float* __restrict__ loss,
                                                         int sig) {
    __shared__ float red[16];
    float s = 0.f;
    for (int64_t i = threadIdx.x; i < n; i += 1024) {
        const float d = p[i] - t[i];
        s += d * d;
        if (dp) dp[i] = sig ? (2.0f * d * invN) * p[i] * (1.0f - p[i]) : 2.0f * d * invN;
    }
    s = block_sum<16>(s, red);
    if (threadIdx.x == 0) loss[0] = s * invN;
}

}  // namespace nnhip

using namespace nnhip;

#define PN_MAP1(name, F, fobj)                                                                         \
    extern "C" int name(float* out, const float* in, int64_t size, nnhipStream_t s) {                   \
        NNHIP_CHECK_ARG(size >= 0, NNHIP_EINVAL, #name ": negative size");                              \
        if (size == 0) return 0;                                                                        \
        NNHIP_CHECK_ARG(out && in, NNHIP_EINVAL, #name ": null pointer");                               \
        const bool vec = aligned16(out) && aligned16(in);                                               \
        hipLaunchKernelGGL((pn_map1<F>), dim3(pn_blocks(size)), dim3(256), 0, (hipStream_t)s, out, in, size, vec, fobj); \
        NNHIP_LAUNCH_CHECK(#name);                                                                      \
        return 0;                                                                                       \
    }

extern "C" int nnhipLeakyReLUForward(float* out, const float* in, float alpha, int64_t size, nnhipStream_t s) {
    NNHIP_CHECK_ARG(size >= 0, NNHIP_EINVAL, "nnhipLeakyReLUForward: negative size");
    if (size == 0) return 0;
    NNHIP_CHECK_ARG(out && in, NNHIP_EINVAL, "nnhipLeakyReLUForward: null pointer");
    const bool vec = aligned16(out) && aligned16(in);
    hipLaunchKernelGGL((pn_map1<LeakyF>), dim3(pn_blocks(size)), dim3(256), 0, (hipStream_t)s, out, in, size, vec, LeakyF{alpha});
    NNHIP_LAUNCH_CHECK("leaky_relu_forward");
    return 0;
}
extern "C" int nnhipLeakyReLUBackward(float* dIn, const float* dOut, const float* out, float alpha, int64_t size,
                                      nnhipStream_t s) {
    NNHIP_CHECK_ARG(size >= 0, NNHIP_EINVAL, "nnhipLeakyReLUBackward: negative size");
    if (size == 0) return 0;
    NNHIP_CHECK_ARG(dIn && dOut && out, NNHIP_EINVAL, "nnhipLeakyReLUBackward: null pointer");
    const bool vec = aligned16(dIn) && aligned16(dOut) && aligned16(out);
    hipLaunchKernelGGL((pn_map2<LeakyB>), dim3(pn_blocks(size)), dim3(256), 0, (hipStream_t)s, dIn, dOut, out, size, vec, LeakyB{alpha});
    NNHIP_LAUNCH_CHECK("leaky_relu_backward");
    return 0;
}
PN_MAP1(nnhipSigmoidForward, SigmoidF, SigmoidF{})
extern "C" int nnhipSigmoidBackward(float* dIn, const float* dOut, const float* out, int64_t size, nnhipStream_t s) {
    NNHIP_CHECK_ARG(size >= 0, NNHIP_EINVAL, "nnhipSigmoidBackward: negative size");
    if (size == 0) return 0;
    NNHIP_CHECK_ARG(dIn && dOut && out, NNHIP_EINVAL, "nnhipSigmoidBackward: null pointer");
    const bool vec = aligned16(dIn) && aligned16(dOut) && aligned16(out);
    hipLaunchKernelGGL((pn_map2<SigmoidB>), dim3(pn_blocks(size)), dim3(256), 0, (hipStream_t)s, dIn, dOut, out, size, vec, SigmoidB{});
    NNHIP_LAUNCH_CHECK("sigmoid_backward");
    return 0;
}

static inline int64_t pool_dil(int64_t v) { return v > 0 ? v : 1; }
static int pool_check(const nnhipPool2dDesc* d, int& Ho, int& Wo) {
    NNHIP_CHECK_ARG(d != nullptr, NNHIP_EINVAL, "maxpool2d: null descriptor");
    NNHIP_CHECK_ARG(d->dh >= 0 && d->dw >= 0, NNHIP_EINVAL, "maxpool2d: negative dilation");
    NNHIP_CHECK_ARG(d->B >= 0 && d->C > 0 && d->H > 0 && d->W > 0 && d->kh > 0 && d->kw > 0 && d->sh > 0 && d->sw > 0 &&
                        d->pu >= 0 && d->pd >= 0 && d->pl >= 0 && d->pr >= 0,
                    NNHIP_EINVAL, "maxpool2d: bad descriptor");
    const int64_t ho = (d->H + d->pu + d->pd - pool_dil(d->dh) * (d->kh - 1) - 1) / d->sh + 1;   // maxpool2d.py:170-183
    const int64_t wo = (d->W + d->pl + d->pr - pool_dil(d->dw) * (d->kw - 1) - 1) / d->sw + 1;
    NNHIP_CHECK_ARG(d->H + d->pu + d->pd >= pool_dil(d->dh) * (d->kh - 1) + 1 && d->W + d->pl + d->pr >= pool_dil(d->dw) * (d->kw - 1) + 1,
                    NNHIP_EINVAL, "maxpool2d: the (dilated) window is larger than the padded input");
    NNHIP_CHECK_ARG(ho > 0 && wo > 0 && d->H * d->W < ((int64_t)1 << 31), NNHIP_EINVAL, "maxpool2d: bad geometry");
    Ho = (int)ho; Wo = (int)wo;
    return 0;
}

static int maxpool_forward(const char* fn, float* out, int32_t* argmax, const float* X, const nnhipPool2dDesc* d, float pre_alpha,
                           nnhipStream_t s) {
    int Ho, Wo;
    if (int rc = pool_check(d, Ho, Wo)) return rc;
    const int64_t total = d->B * d->C * Ho * Wo;
    if (total == 0) return 0;
    NNHIP_CHECK_ARG(out && argmax && X, NNHIP_EINVAL, "%s: null pointer", fn);
    hipLaunchKernelGGL(maxpool_fwd_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, (hipStream_t)s, out, argmax, X,
                       d->B * d->C, (int)d->H, (int)d->W, Ho, Wo, (int)d->kh, (int)d->kw, (int)d->sh, (int)d->sw,
                       (int)d->pu, (int)d->pl, (int)pool_dil(d->dh), (int)pool_dil(d->dw), pre_alpha);
    NNHIP_LAUNCH_CHECK("maxpool_fwd_kernel");
    return 0;
}
static int maxpool_backward(const char* fn, float* dX, const float* dY, const int32_t* argmax, const float* pooled, float alpha,
                            const nnhipPool2dDesc* d, nnhipStream_t s) {
    int Ho, Wo;
    if (int rc = pool_check(d, Ho, Wo)) return rc;
    const int64_t total = d->B * d->C * d->H * d->W;
    if (total == 0) return 0;
    NNHIP_CHECK_ARG(dX && dY && argmax, NNHIP_EINVAL, "%s: null pointer", fn);
    if (d->kh == d->sh && d->kw == d->sw && d->pu + d->pd + d->pl + d->pr == 0 && d->H == (int64_t)Ho * d->kh &&
        d->W == (int64_t)Wo * d->kw && pool_dil(d->dh) == 1 && pool_dil(d->dw) == 1) {
        const int64_t nout = d->B * d->C * Ho * Wo;
        const bool two = d->kw == 2 && (d->W & 1) == 0 && (reinterpret_cast<uintptr_t>(dX) & 7u) == 0;
        if (two)
            hipLaunchKernelGGL(maxpool_bwd_tiles_kernel<2>, dim3((unsigned)ceil_div(nout, 256)), dim3(256), 0, (hipStream_t)s, dX, dY,
                               argmax, nout, (int)d->W, Ho, Wo, (int)d->kh, (int)d->kw, pooled, alpha);
        else
            hipLaunchKernelGGL(maxpool_bwd_tiles_kernel<0>, dim3((unsigned)ceil_div(nout, 256)), dim3(256), 0, (hipStream_t)s, dX, dY,
                               argmax, nout, (int)d->W, Ho, Wo, (int)d->kh, (int)d->kw, pooled, alpha);
        NNHIP_LAUNCH_CHECK("maxpool_bwd_tiles_kernel");
        return 0;
    }
    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, (hipStream_t)s, dX, dY, argmax,
                       d->B * d->C, (int)d->H, (int)d->W, Ho, Wo, (int)d->kh, (int)d->kw, (int)d->sh, (int)d->sw,
                       (int)d->pu, (int)d->pl, (int)pool_dil(d->dh), (int)pool_dil(d->dw), pooled, alpha);
    NNHIP_LAUNCH_CHECK("maxpool_bwd_kernel");
    return 0;
}
extern "C" int nnhipMaxPool2dForward(float* out, int32_t* argmax, const float* X, const nnhipPool2dDesc* d,
                                     nnhipStream_t s) {
    return maxpool_forward("nnhipMaxPool2dForward", out, argmax, X, d, 1.0f, s);
}
extern "C" int nnhipMaxPool2dBackward(float* dX, const float* dY, const int32_t* argmax, const nnhipPool2dDesc* d,
                                      nnhipStream_t s) {
    return maxpool_backward("nnhipMaxPool2dBackward", dX, dY, argmax, nullptr, 1.0f, d, s);
}
// MaxPool2d(LeakyReLU(X; alpha)) and its backward as one launch each (alpha > 0; the conv classifier's
// conv -> LeakyReLU -> MaxPool chain, examples/convolutional_digits_classifier.ipynb): the activation is evaluated inside the
// pooling window, its full-resolution output is never written; the backward routes dY through the arg-max and applies the
// LeakyReLU gradient factor of that element (read off the pooled output) -- dX is the gradient of the LeakyReLU's input.
extern "C" int nnhipMaxPool2dLeakyForward(float* out, int32_t* argmax, const float* X, float alpha, const nnhipPool2dDesc* d,
                                          nnhipStream_t s) {
    NNHIP_CHECK_ARG(alpha > 0.f, NNHIP_EINVAL, "nnhipMaxPool2dLeakyForward: alpha must be > 0 (a strictly increasing activation)");
    return maxpool_forward("nnhipMaxPool2dLeakyForward", out, argmax, X, d, alpha, s);
}
extern "C" int nnhipMaxPool2dLeakyBackward(float* dX, const float* dY, const int32_t* argmax, const float* pooled, float alpha,
                                           const nnhipPool2dDesc* d, nnhipStream_t s) {
    NNHIP_CHECK_ARG(pooled != nullptr, NNHIP_EINVAL, "nnhipMaxPool2dLeakyBackward: null pointer");
    return maxpool_backward("nnhipMaxPool2dLeakyBackward", dX, dY, argmax, pooled, alpha, d, s);
}

extern "C" int nnhipBatchNorm2dForward(const float* X, const float* weight, const float* bias, float* Y,
                                       float* save_mean, float* save_inv, float* running_mean, float* running_var,
                                       int64_t B, int64_t C, int64_t HW, float eps, float momentum, int training,
                                       nnhipStream_t s) {
    NNHIP_CHECK_ARG(B >= 0 && C > 0 && HW > 0 && B * HW < ((int64_t)1 << 31), NNHIP_EINVAL, "nnhipBatchNorm2dForward: bad sizes");
    if (B == 0) return 0;
    NNHIP_CHECK_ARG(X && Y && save_mean && save_inv, NNHIP_EINVAL, "nnhipBatchNorm2dForward: null pointer");
    NNHIP_CHECK_ARG((weight == nullptr) == (bias == nullptr), NNHIP_EINVAL, "nnhipBatchNorm2dForward: weight and bias go together");
    NNHIP_CHECK_ARG(training || (running_mean && running_var), NNHIP_EINVAL, "nnhipBatchNorm2dForward: eval needs running stats");
    hipStream_t st = (hipStream_t)s;
    if (training && bn_fits_fused(B, C, HW)) {
        hipLaunchKernelGGL(bn_fwd_fused_kernel, dim3((unsigned)C), dim3(1024), 0, st, X, Y, weight, bias, save_mean, save_inv,
                           running_mean, running_var, (int)B, (int)C, (int)HW, eps, momentum);
        NNHIP_LAUNCH_CHECK("bn_fwd_fused_kernel");
        return 0;
    }
    if (training)
        hipLaunchKernelGGL(bn_stats_kernel, dim3((unsigned)C), dim3(1024), 0, st, X, save_mean, save_inv, running_mean, running_var,
                           (int)B, (int)C, (int)HW, eps, momentum);
    else
        hipLaunchKernelGGL(bn_eval_stats_kernel, dim3((unsigned)ceil_div(C, 64)), dim3(64), 0, st, running_mean, running_var,
                           save_mean, save_inv, (int)C, eps);
    NNHIP_LAUNCH_CHECK("bn_stats");
    const int64_t total = B * C * HW;
    hipLaunchKernelGGL(bn_apply_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, st, Y, X, save_mean, save_inv, weight,
                       bias, total, (int)C, (int)HW);
    NNHIP_LAUNCH_CHECK("bn_apply_kernel");
    return 0;
}
extern "C" int nnhipBatchNorm2dBackward(const float* dY, const float* X, const float* weight, const float* save_mean,
                                        const float* save_inv, float* dX, float* dW, float* db, int64_t B, int64_t C,
                                        int64_t HW, nnhipStream_t s) {
    NNHIP_CHECK_ARG(B >= 0 && C > 0 && HW > 0 && B * HW < ((int64_t)1 << 31), NNHIP_EINVAL, "nnhipBatchNorm2dBackward: bad sizes");
    if (B == 0) return 0;
    NNHIP_CHECK_ARG(dY && X && save_mean && save_inv && dX, NNHIP_EINVAL, "nnhipBatchNorm2dBackward: null pointer");
    NNHIP_CHECK_ARG((dW == nullptr) == (db == nullptr), NNHIP_EINVAL, "nnhipBatchNorm2dBackward: dW and db go together");
    hipStream_t st = (hipStream_t)s;
    if (bn_fits_fused(B, C, HW)) {
        hipLaunchKernelGGL(bn_bwd_fused_kernel, dim3((unsigned)C), dim3(1024), 0, st, dY, X, save_mean, save_inv, weight, dX, dW, db,
                           (int)B, (int)C, (int)HW, 1.0f / (float)(B * HW));
        NNHIP_LAUNCH_CHECK("bn_bwd_fused_kernel");
        return 0;
    }
    float* sums = static_cast<float*>(workspace((size_t)C * 4 * sizeof(float)));
    NNHIP_CHECK_ARG(sums != nullptr, NNHIP_ENOMEM, "nnhipBatchNorm2dBackward: workspace allocation failed");
    hipLaunchKernelGGL(bn_bwd_stats_kernel, dim3((unsigned)C), dim3(1024), 0, st, dY, X, save_mean, save_inv, weight, sums, (int)B,
                       (int)C, (int)HW);
    NNHIP_LAUNCH_CHECK("bn_bwd_stats_kernel");
    const int64_t total = B * C * HW;
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3((unsigned)ceil_div(total > C ? total : C, 256)), dim3(256), 0, st, dX, dY, X,
                       save_mean, save_inv, weight, sums, dW, db, total, (int)C, (int)HW, 1.0f / (float)(B * HW));
    NNHIP_LAUNCH_CHECK("bn_bwd_apply_kernel");
    return 0;
}

static int mse_launch(const char* fn, const float* pred, const float* target, float* loss, float* dpred, int64_t n, int sig,
                      nnhipStream_t s) {
    NNHIP_CHECK_ARG(n > 0, NNHIP_EINVAL, "%s: n must be > 0", fn);
    NNHIP_CHECK_ARG(pred && target && loss, NNHIP_EINVAL, "%s: null pointer", fn);
    hipStream_t st = (hipStream_t)s;
    if (n <= 16384) {
        hipLaunchKernelGGL(mse_small_kernel, dim3(1), dim3(1024), 0, st, pred, target, dpred, n, 1.0f / (float)n, loss, sig);
        NNHIP_LAUNCH_CHECK("mse_small_kernel");
        return 0;
    }
    int64_t blocks = ceil_div(n, 1024);
    if (blocks > 1024) blocks = 1024;
    float* part = static_cast<float*>(workspace((size_t)blocks * sizeof(float)));
    NNHIP_CHECK_ARG(part != nullptr, NNHIP_ENOMEM, "%s: workspace allocation failed", fn);
    const float invN = 1.0f / (float)n;
    hipLaunchKernelGGL(mse_kernel, dim3((unsigned)blocks), dim3(256), 0, st, pred, target, dpred, part, n, invN, sig);
    NNHIP_LAUNCH_CHECK("mse_kernel");
    hipLaunchKernelGGL(mse_final_kernel, dim3(1), dim3(256), 0, st, part, (int)blocks, invN, loss);
    NNHIP_LAUNCH_CHECK("mse_final_kernel");
    return 0;
}
extern "C" int nnhipMSELossForwardBackward(const float* pred, const float* target, float* loss, float* dpred,
                                           int64_t n, nnhipStream_t s) {
    return mse_launch("nnhipMSELossForwardBackward", pred, target, loss, dpred, n, 0, s);
}
// MSELoss(Sigmoid(z), target): `pred` is the sigmoid OUTPUT; dz_out = d(loss)/dz = 2 (pred - target) / n * pred (1 - pred)
// -- the Sigmoid backward (neunet/nn/activations.py:12-13) folded into the loss kernel.
extern "C" int nnhipMSELossSigmoidForwardBackward(const float* pred, const float* target, float* loss, float* dz_out,
                                                  int64_t n, nnhipStream_t s) {
    return mse_launch("nnhipMSELossSigmoidForwardBackward", pred, target, loss, dz_out, n, 1, s);
}
