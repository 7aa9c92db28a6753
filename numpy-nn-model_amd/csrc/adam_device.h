// adam_device.h -- the Adam / AdamW element update and the device-side stepping protocol, shared by the optimizer kernels
// (optim.hip) and by kernels that apply the update in their own epilogue (gemm_small.hip: the README-MLP backward).
#pragma once
#include <math.h>

#include "common.h"

namespace nnhip {

struct AdamHyper {
    float lr, b1, b2, one_m_b1, one_m_b2, eps, wd, bc1, bc2, grad_scale;
    int decay_mode;  // 0 decoupled (AdamW), 1 L2-on-grad (Adam)
};

inline AdamHyper make_hyper(double lr, double b1, double b2, double eps, double wd, int step, int mode,
                            float grad_scale) {
    AdamHyper h;
    // NumPy weak-scalar promotion: every python-double hyper-parameter is rounded to fp32 at the point
    // it meets an fp32 array -- (1 - beta) is formed in double FIRST, then rounded.
    h.lr = (float)lr; h.b1 = (float)b1; h.b2 = (float)b2; h.eps = (float)eps; h.wd = (float)wd;
    h.grad_scale = grad_scale;
    h.one_m_b1 = (float)(1.0 - b1);
    h.one_m_b2 = (float)(1.0 - b2);
    // bias corrections in double on the host, as the CPU path's python floats (optim.py:30-31);
    // the reference kernel used powf in-kernel (fused_adamw_multitensor.cu:145-146)
    h.bc1 = (float)(1.0 - pow(b1, (double)step));
    h.bc2 = (float)(1.0 - pow(b2, (double)step));
    h.decay_mode = mode;
    return h;
}

// No floating-point contraction in here: the update is inlined into several kernels (optim.hip, gemm_small.hip) and must give
// the same bits in each of them -- left to the compiler, `b1*m + (1-b1)*g` became an fma in one kernel and mul + add in another
// (1-ulp differences in 10 % of the elements between the fused and the separate optimizer step).
__device__ __forceinline__ void adam1(float& p, float g, float& m, float& v, const AdamHyper& h) {
#pragma clang fp contract(off)
    g *= h.grad_scale;
    if (h.decay_mode == 1) {
        g = g + h.wd * p;                 // optim.py:24-25
    } else if (h.wd != 0.f) {
        p = p - h.lr * h.wd * p;          // optim.py:59-60
    }
    m = h.b1 * m + h.one_m_b1 * g;        // optim.py:63
    v = h.b2 * v + h.one_m_b2 * (g * g);  // optim.py:64
    const float mh = m / h.bc1;
    const float vh = v / h.bc2;
    p = p - h.lr * mh / (sqrtf(vh) + h.eps);  // optim.py:69
}

// ---- device-driven stepping (hipGraph replay) -------------------------------------------------------------------------------
// dev_state: {step (int bits), lr, grad_scale, finished-block ticket (int bits), weight_decay, bc_step (int bits), bc1, bc2,
// beta1, beta2}.  Every block reads it when it starts (thread 0, plain L1-cached loads, broadcast through `sh`: the state only
// changes in the last block's epilogue, after every block has read it) and derives this step's hyper-parameters; the LAST block
// to finish advances the step and leaves the next step's bias corrections.  grad_div (device float or null): gradients are
// additionally divided by grad_div[0].  `sh`: 5 floats of LDS.  Contains block barriers: call it from every thread.
// Split in two so that a kernel can put its own prefetch loads between them: adam_dev_issue (thread 0's loads of the state leave)
// ... the kernel's other loads ... adam_dev_resolve (thread 0 derives the step's hyper-parameters, one barrier, everyone reads).
struct AdamDevRaw { float4 s0, s1; float2 s2; float gd; };
__device__ __forceinline__ void adam_dev_issue(AdamDevRaw& r, const float* __restrict__ dev_state, const float* __restrict__ grad_div) {
    r.s0 = r.s1 = make_float4(0.f, 0.f, 0.f, 0.f); r.s2 = make_float2(0.f, 0.f); r.gd = 1.f;
    if (threadIdx.x != 0) return;
    if (dev_state) {
        r.s0 = *reinterpret_cast<const float4*>(dev_state);        // step, lr, grad_scale, ticket
        r.s1 = *reinterpret_cast<const float4*>(dev_state + 4);    // wd, bc_step, bc1, bc2
        r.s2 = *reinterpret_cast<const float2*>(dev_state + 8);    // the betas bc1 / bc2 were computed for
    }
    if (grad_div) r.gd = grad_div[0];
}
__device__ __forceinline__ void adam_dev_resolve(AdamHyper& h, const AdamDevRaw& r, const float* __restrict__ dev_state, double b1,
                                                 double b2, const float* __restrict__ grad_div, float* sh) {
    if (!dev_state && !grad_div) return;
    if (threadIdx.x == 0) {
        float gs = h.grad_scale;
        if (dev_state) {
            const int step = __float_as_int(r.s0.x) + 1;
            const bool cached = __float_as_int(r.s1.y) == step && r.s2.x == (float)b1 && r.s2.y == (float)b2;
            sh[0] = cached ? r.s1.z : (float)(1.0 - pow(b1, (double)step));
            sh[1] = cached ? r.s1.w : (float)(1.0 - pow(b2, (double)step));
            sh[2] = r.s0.y;
            sh[4] = r.s1.x;
            gs = r.s0.z;
        }
        if (grad_div) gs = gs / r.gd;
        sh[3] = gs;
    }
    __syncthreads();
    if (dev_state) { h.bc1 = sh[0]; h.bc2 = sh[1]; h.lr = sh[2]; h.wd = sh[4]; }
    h.grad_scale = sh[3];
}
__device__ __forceinline__ void adam_dev_begin(AdamHyper& h, const float* __restrict__ dev_state, double b1, double b2,
                                               const float* __restrict__ grad_div, float* sh) {
    if (!dev_state && !grad_div) return;
    AdamDevRaw r;
    adam_dev_issue(r, dev_state, grad_div);
    adam_dev_resolve(h, r, dev_state, b1, b2, grad_div, sh);
}
// The last block's part: advance the step, leave the next step's bias corrections (one thread).  In two halves for callers that
// can compute the (slow: two double-precision pow) corrections while other blocks still read the state, and store them afterwards.
struct AdamDevNext { int next; float bc1, bc2; };
__device__ __forceinline__ AdamDevNext adam_dev_next(const float* __restrict__ dev_state, double b1, double b2) {
    AdamDevNext r;
    r.next = reinterpret_cast<const int*>(dev_state)[0] + 2;       // the step after the one that now ends
    r.bc1 = (float)(1.0 - pow(b1, (double)r.next));
    r.bc2 = (float)(1.0 - pow(b2, (double)r.next));
    return r;
}
__device__ __forceinline__ void adam_dev_commit(float* __restrict__ dev_state, const AdamDevNext& r, double b1, double b2) {
    int* si = reinterpret_cast<int*>(dev_state);
    si[3] = 0;
    si[0] = r.next - 1;
    dev_state[6] = r.bc1;
    dev_state[7] = r.bc2;
    dev_state[8] = (float)b1;
    dev_state[9] = (float)b2;
    si[5] = r.next;
}
__device__ __forceinline__ void adam_dev_advance(float* __restrict__ dev_state, double b1, double b2) {
    adam_dev_commit(dev_state, adam_dev_next(dev_state, b1, b2), b1, b2);
}
// Called by every block of the launch when its updates are done (contains a block barrier).
__device__ __forceinline__ void adam_dev_finish(float* __restrict__ dev_state, int nblk, double b1, double b2) {
    if (!dev_state) return;
    __syncthreads();                                       // this block's reads of the step are long done
    if (threadIdx.x == 0) {
        int* si = reinterpret_cast<int*>(dev_state);
        if (atomicAdd(&si[3], 1) == nblk - 1) adam_dev_advance(dev_state, b1, b2);   // last block to finish
    }
}

}  // namespace nnhip
