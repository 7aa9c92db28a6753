// optim.hip -- fused Adam / AdamW for gfx950: per-tensor launch and one-launch multi-tensor.
// CPU semantics: neunet/optim.py:17-33 (Adam, L2 decay on the gradient) and :52-69 (AdamW,
// decoupled decay).  28 B/elem of HBM traffic (read p,g,m,v; write p,m,v) -- pure bandwidth.
#include <string.h>

#include <vector>

#include "adam_device.h"
#include "common.h"

namespace nnhip {

// ADAM_UNR float4 rounds of all four streams in flight per thread (developer A/B: -DADAM_UNR=1|2|4, -DADAM_NT_G=1 reads the gradient --
// dead after this pass -- past L2 allocation, -DADAM_NT_ST=1 stores p / m / v non-temporally)
#ifndef ADAM_UNR
#define ADAM_UNR 2
#endif
#ifndef ADAM_NT_G
#define ADAM_NT_G 1
#endif
#ifndef ADAM_NT_ST
#define ADAM_NT_ST 0
#endif
typedef float adam_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void adam4(adam_f4& P, const adam_f4& G, adam_f4& M, adam_f4& V, const AdamHyper& h) {
    float p0 = P.x, p1 = P.y, p2 = P.z, p3 = P.w, m0 = M.x, m1 = M.y, m2 = M.z, m3 = M.w, v0 = V.x, v1 = V.y, v2 = V.z, v3 = V.w;
    adam1(p0, G.x, m0, v0, h); adam1(p1, G.y, m1, v1, h); adam1(p2, G.z, m2, v2, h); adam1(p3, G.w, m3, v3, h);
    P.x = p0; P.y = p1; P.z = p2; P.w = p3; M.x = m0; M.y = m1; M.z = m2; M.w = m3; V.x = v0; V.y = v1; V.z = v2; V.w = v3;
}
__device__ __forceinline__ void adam_span(float* __restrict__ p, const float* __restrict__ g,
                                          float* __restrict__ m, float* __restrict__ v, int64_t n,
                                          int64_t first, int64_t stride_threads, bool vec,
                                          const AdamHyper& h) {
    if (vec) {
        const int64_t nv = n >> 2;
        adam_f4* __restrict__ p4 = reinterpret_cast<adam_f4*>(p);
        const adam_f4* __restrict__ g4 = reinterpret_cast<const adam_f4*>(g);
        adam_f4* __restrict__ m4 = reinterpret_cast<adam_f4*>(m);
        adam_f4* __restrict__ v4 = reinterpret_cast<adam_f4*>(v);
        int64_t i = first;
        for (; i + (ADAM_UNR - 1) * stride_threads < nv; i += ADAM_UNR * stride_threads) {
            adam_f4 P[ADAM_UNR], G[ADAM_UNR], M[ADAM_UNR], V[ADAM_UNR];
#pragma unroll
            for (int u = 0; u < ADAM_UNR; ++u) {
                const int64_t k = i + u * stride_threads;
                P[u] = p4[k];
                G[u] = ADAM_NT_G ? __builtin_nontemporal_load(&g4[k]) : g4[k];
                M[u] = m4[k];
                V[u] = v4[k];
            }
#pragma unroll
            for (int u = 0; u < ADAM_UNR; ++u) {
                const int64_t k = i + u * stride_threads;
                adam4(P[u], G[u], M[u], V[u], h);
                if (ADAM_NT_ST) {
                    __builtin_nontemporal_store(P[u], &p4[k]); __builtin_nontemporal_store(M[u], &m4[k]); __builtin_nontemporal_store(V[u], &v4[k]);
                } else {
                    p4[k] = P[u]; m4[k] = M[u]; v4[k] = V[u];
                }
            }
        }
        for (; i < nv; i += stride_threads) {
            adam_f4 P = p4[i], G = g4[i], M = m4[i], V = v4[i];
            adam4(P, G, M, V, h);
            p4[i] = P; m4[i] = M; v4[i] = V;
        }
        for (int64_t j = (nv << 2) + first; j < n; j += stride_threads) adam1(p[j], g[j], m[j], v[j], h);
    } else {
        for (int64_t i = first; i < n; i += stride_threads) adam1(p[i], g[i], m[i], v[i], h);
    }
}

__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v,
                                                    int64_t n, bool vec, const AdamHyper h) {
    adam_span(p, g, m, v, n, (int64_t)blockIdx.x * 256 + threadIdx.x, (int64_t)gridDim.x * 256, vec, h);
}

// ---- multi-tensor --------------------------------------------------------------------------------
constexpr int64_t MT_CHUNK = 16384;       // elements per block (large models: 64 per thread, measured 5.7 TB/s)
constexpr int64_t MT_CHUNK_SMALL = 2048;  // below 4 M parameters: 8 per thread -- a 100 k-parameter MLP ran 7 blocks of 16
                                          // dependent float4 rounds (14 us, pure latency); 50 blocks of 2 rounds take ~5

// Device blob layout (one upload): [p*][g*][m*][v*] (n pointers each) [sizes int64 n]
// [blk_tensor int32 nblk][blk_chunk int32 nblk]
// Device-side state for hipGraph replay: state[0] = step (int bits), state[1] = lr, state[2] = grad_scale,
// state[3] = finished-block ticket (int bits), state[4] = weight_decay.  A captured launch freezes its by-value kernel
// arguments, so everything a training loop changes between steps (the step count, an LR schedule, the DP gradient
// scale) is read from here instead (nnhipFusedOptimizerSetStep / SetHyper write it).
// Every block reads the step when it starts and derives the bias corrections itself; the LAST block to finish (ticket)
// advances the counter -- no block can still have to read it then -- so a replayed step needs no separate
// "advance" launch (it was one more ~4.6 us graph node per step at MNIST-MLP scale).
// grad_div (device float or null): gradients are additionally divided by grad_div[0] -- the all-reduced count of
// non-ignored targets when every rank back-propagated a 'sum' loss (no host read of the count, no scale pass).
__global__ __launch_bounds__(256) void adamw_multi_kernel(const unsigned char* __restrict__ blob, int n,
                                                          int nblk, AdamHyper h, float* __restrict__ dev_state,
                                                          double b1, double b2, int chunk,
                                                          const float* __restrict__ grad_div) {
    // Device-driven stepping: thread 0 reads the optimizer's device state with PLAIN (L1-cached) loads and hands it to the
    // block through LDS.  Every thread used to read it with agent-scope atomic loads: L1-bypassing requests of 2100 blocks x
    // 4 waves x 5 values that all land on ONE L2 channel the moment the grid starts, and the last wave to be served sets the
    // kernel's length -- the C4 step's launch took 237-270 us in graph replay against 152-169 us with host-side stepping.
    // Plain loads are safe: the state only changes in the last block's epilogue below, after every block has read it, and the
    // next launch starts with clean caches.  The bias corrections of this step were left in dev_state by the previous step's
    // last block; only the first step after SetStep, or a change of the betas, computes the two double-precision pow()s here.
    __shared__ float sh[5];
    AdamDevRaw raw;
    adam_dev_issue(raw, dev_state, grad_div);              // thread 0's state loads leave now; the plan lookups below (a chain of
                                                           // dependent scalar loads: block -> tensor -> pointers) run behind them
    float* const* P = reinterpret_cast<float* const*>(blob);
    const float* const* G = reinterpret_cast<const float* const*>(blob + sizeof(void*) * n);
    float* const* M = reinterpret_cast<float* const*>(blob + sizeof(void*) * 2 * n);
    float* const* V = reinterpret_cast<float* const*>(blob + sizeof(void*) * 3 * n);
    const int64_t* sizes = reinterpret_cast<const int64_t*>(blob + sizeof(void*) * 4 * n);
    const int32_t* blk_tensor = reinterpret_cast<const int32_t*>(blob + sizeof(void*) * 4 * n + sizeof(int64_t) * n);
    const int32_t* blk_chunk = blk_tensor + nblk;
    const int ti = blk_tensor[blockIdx.x];
    const int64_t off = (int64_t)blk_chunk[blockIdx.x] * chunk;
    int64_t cnt = sizes[ti] - off;
    if (cnt > chunk) cnt = chunk;
    float* p = P[ti] + off;
    const float* g = G[ti] + off;
    float* m = M[ti] + off;
    float* v = V[ti] + off;
    // chunk offsets are multiples of 2048 elements, so 16-B alignment of the chunk == of the tensor
    const bool vec = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) |
                       reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(v)) & 15u) == 0;
    adam_dev_resolve(h, raw, dev_state, b1, b2, grad_div, sh);
    adam_span(p, g, m, v, cnt, threadIdx.x, 256, vec, h);
    adam_dev_finish(dev_state, nblk, b1, b2);
}

// Host object behind CreateFusedOptimizer (replaces the reference's C++ FusedOptimizer,
// fused_adamw_multitensor.cu:239-302, whose plan cache was keyed on the tensor COUNT only and
// re-uploaded 4 pointer tables every step).  Here the whole plan is one blob, compared bytewise
// with the last upload and re-sent only when something changed, through a small ring of pinned
// staging buffers so the upload is asynchronous.
struct FusedOptimizer {
    static constexpr int kRing = 4;
    std::vector<unsigned char> last;  // last uploaded blob (host copy)
    unsigned char* dev = nullptr;
    size_t dev_cap = 0;
    unsigned char* pinned[kRing] = {nullptr, nullptr, nullptr, nullptr};
    size_t pinned_cap[kRing] = {0, 0, 0, 0};
    hipEvent_t ev[kRing] = {nullptr, nullptr, nullptr, nullptr};
    bool ev_pending[kRing] = {false, false, false, false};
    int ring = 0;
    int nblk = 0;
    float* dev_state = nullptr;  // {step, lr, grad_scale, ticket, weight_decay, bc_step, bc1, bc2, beta1, beta2} for device-driven
                                 // stepping (graph replay); bc*: the bias corrections of step bc_step, left by the previous step
    bool hyper_set = false;
    const float* grad_div = nullptr;  // device float: divide gradients by it (all-reduced target count), or null

    ~FusedOptimizer() {
        (void)hipDeviceSynchronize();
        if (dev) (void)hipFree(dev);
        if (dev_state) (void)hipFree(dev_state);
        for (int i = 0; i < kRing; ++i) {
            if (pinned[i]) (void)hipHostFree(pinned[i]);
            if (ev[i]) (void)hipEventDestroy(ev[i]);
        }
    }
};

}  // namespace nnhip

using namespace nnhip;

extern "C" int nnhipFusedAdamWStep(float* p, const float* g, float* m, float* v, double lr, double beta1,
                                   double beta2, double eps, double weight_decay, int32_t step, int64_t n,
                                   int32_t decay_mode, float grad_scale, nnhipStream_t s) {
    NNHIP_CHECK_ARG(n >= 0 && step >= 1, NNHIP_EINVAL, "nnhipFusedAdamWStep: n >= 0 and step >= 1 required");
    NNHIP_CHECK_ARG(decay_mode == 0 || decay_mode == 1, NNHIP_EINVAL, "nnhipFusedAdamWStep: decay_mode must be 0 or 1");
    if (n == 0) return 0;
    NNHIP_CHECK_ARG(p && g && m && v, NNHIP_EINVAL, "nnhipFusedAdamWStep: null pointer");
    const AdamHyper h = make_hyper(lr, beta1, beta2, eps, weight_decay, step, decay_mode, grad_scale);
    const bool vec = aligned16(p) && aligned16(g) && aligned16(m) && aligned16(v);
    int64_t blocks = ceil_div(vec ? (n >> 2) + 1 : n, 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)s, p, g, m, v, n, vec, h);
    NNHIP_LAUNCH_CHECK("adamw_kernel");
    return 0;
}

namespace nnhip {
// For kernels that apply the update in their own epilogue: the device state / gradient divisor of an optimizer handle.
// step == 0 = device-driven stepping (needs SetStep + SetHyper first).  Returns 0 or an NNHIP_E* code.
int fused_optimizer_state(void* opt, int step, float** dev_state, const float** grad_div) {
    *dev_state = nullptr; *grad_div = nullptr;
    if (!opt) return step == 0 ? NNHIP_EINVAL : 0;
    FusedOptimizer* fo = static_cast<FusedOptimizer*>(opt);
    *grad_div = fo->grad_div;
    if (step == 0) {
        if (!fo->dev_state || !fo->hyper_set) { set_last_error("fused optimizer: step == 0 needs nnhipFusedOptimizerSetStep and SetHyper first"); return NNHIP_EINVAL; }
        *dev_state = fo->dev_state;
    }
    return 0;
}
}  // namespace nnhip

extern "C" void* nnhipCreateFusedOptimizer(void) { return new (std::nothrow) FusedOptimizer(); }

extern "C" int nnhipDestroyFusedOptimizer(void* opt) {
    delete static_cast<FusedOptimizer*>(opt);
    return 0;
}

extern "C" int nnhipFusedAdamWMultiTensorStep(void* opt, int32_t n_tensors, float* const* p,
                                              const float* const* g, float* const* m, float* const* v,
                                              const int64_t* sizes, double lr, double beta1, double beta2,
                                              double eps, double weight_decay, int32_t step,
                                              int32_t decay_mode, float grad_scale, nnhipStream_t s) {
    NNHIP_CHECK_ARG(opt != nullptr, NNHIP_EINVAL, "nnhipFusedAdamWMultiTensorStep: null optimizer handle");
    if (int rc = device_error_status("nnhipFusedAdamWMultiTensorStep")) return rc;     // an earlier kernel raised the device error word
    NNHIP_CHECK_ARG(n_tensors >= 0 && step >= 0, NNHIP_EINVAL, "nnhipFusedAdamWMultiTensorStep: bad n_tensors/step");
    NNHIP_CHECK_ARG(decay_mode == 0 || decay_mode == 1, NNHIP_EINVAL, "nnhipFusedAdamWMultiTensorStep: decay_mode must be 0 or 1");
    if (n_tensors == 0) return 0;
    NNHIP_CHECK_ARG(p && g && m && v && sizes, NNHIP_EINVAL, "nnhipFusedAdamWMultiTensorStep: null table");
    FusedOptimizer* fo = static_cast<FusedOptimizer*>(opt);
    hipStream_t st = (hipStream_t)s;
    const int n = n_tensors;

    // ---- build the plan blob ------------------------------------------------------------------
    int64_t total = 0;
    for (int i = 0; i < n; ++i) total += sizes[i] > 0 ? sizes[i] : 0;
    const int64_t chunk = total >= ((int64_t)1 << 22) ? MT_CHUNK : MT_CHUNK_SMALL;
    int64_t nblk = 0;
    for (int i = 0; i < n; ++i) {
        NNHIP_CHECK_ARG(sizes[i] >= 0, NNHIP_EINVAL, "nnhipFusedAdamWMultiTensorStep: negative size");
        NNHIP_CHECK_ARG(sizes[i] == 0 || (p[i] && g[i] && m[i] && v[i]), NNHIP_EINVAL,
                        "nnhipFusedAdamWMultiTensorStep: null tensor pointer");
        nblk += ceil_div(sizes[i], chunk);
    }
    if (nblk == 0) return 0;
    const size_t bytes = sizeof(void*) * 4 * n + sizeof(int64_t) * n + sizeof(int32_t) * 2 * (size_t)nblk;
    std::vector<unsigned char> blob(bytes);
    memcpy(blob.data(), p, sizeof(void*) * n);
    memcpy(blob.data() + sizeof(void*) * n, g, sizeof(void*) * n);
    memcpy(blob.data() + sizeof(void*) * 2 * n, m, sizeof(void*) * n);
    memcpy(blob.data() + sizeof(void*) * 3 * n, v, sizeof(void*) * n);
    memcpy(blob.data() + sizeof(void*) * 4 * n, sizes, sizeof(int64_t) * n);
    int32_t* bt = reinterpret_cast<int32_t*>(blob.data() + sizeof(void*) * 4 * n + sizeof(int64_t) * n);
    int32_t* bc = bt + nblk;
    int64_t b = 0;
    for (int i = 0; i < n; ++i) {
        const int64_t nc = ceil_div(sizes[i], chunk);
        for (int64_t c = 0; c < nc; ++c, ++b) { bt[b] = i; bc[b] = (int32_t)c; }
    }

    // ---- upload only if it changed ------------------------------------------------------------
    if (blob != fo->last) {
        if (bytes > fo->dev_cap) {
            if (fo->dev) { (void)hipDeviceSynchronize(); (void)hipFree(fo->dev); fo->dev = nullptr; }
            const size_t cap = bytes * 2;
            hipError_t e = hipMalloc(reinterpret_cast<void**>(&fo->dev), cap);
            if (e != hipSuccess) { fo->dev_cap = 0; return hip_status(e, "hipMalloc(optimizer plan)"); }
            fo->dev_cap = cap;
        }
        const int r = fo->ring;
        fo->ring = (fo->ring + 1) % FusedOptimizer::kRing;
        if (fo->ev_pending[r]) { (void)hipEventSynchronize(fo->ev[r]); fo->ev_pending[r] = false; }
        if (bytes > fo->pinned_cap[r]) {
            if (fo->pinned[r]) (void)hipHostFree(fo->pinned[r]);
            hipError_t e = hipHostMalloc(reinterpret_cast<void**>(&fo->pinned[r]), bytes * 2, hipHostMallocDefault);
            if (e != hipSuccess) { fo->pinned[r] = nullptr; fo->pinned_cap[r] = 0; return hip_status(e, "hipHostMalloc(optimizer plan)"); }
            fo->pinned_cap[r] = bytes * 2;
        }
        if (!fo->ev[r]) {
            hipError_t e = hipEventCreateWithFlags(&fo->ev[r], hipEventDisableTiming);
            if (e != hipSuccess) return hip_status(e, "hipEventCreate(optimizer plan)");
        }
        memcpy(fo->pinned[r], blob.data(), bytes);
        hipError_t e = hipMemcpyAsync(fo->dev, fo->pinned[r], bytes, hipMemcpyHostToDevice, st);
        if (e != hipSuccess) return hip_status(e, "hipMemcpyAsync(optimizer plan)");
        e = hipEventRecord(fo->ev[r], st);
        if (e != hipSuccess) return hip_status(e, "hipEventRecord(optimizer plan)");
        fo->ev_pending[r] = true;
        fo->last.swap(blob);
        fo->nblk = (int)nblk;
    }

    const AdamHyper h = make_hyper(lr, beta1, beta2, eps, weight_decay, step > 0 ? step : 1, decay_mode, grad_scale);
    float* dstate = nullptr;
    if (step == 0) {  // device-driven step counter (set with nnhipFusedOptimizerSetStep): graph-replay safe
        NNHIP_CHECK_ARG(fo->dev_state != nullptr && fo->hyper_set, NNHIP_EINVAL,
                        "nnhipFusedAdamWMultiTensorStep: step == 0 needs nnhipFusedOptimizerSetStep and SetHyper first");
        dstate = fo->dev_state;
    }
    hipLaunchKernelGGL(adamw_multi_kernel, dim3((unsigned)nblk), dim3(256), 0, st, fo->dev, n, (int)nblk, h, dstate, beta1,
                       beta2, (int)chunk, fo->grad_div);
    NNHIP_LAUNCH_CHECK("adamw_multi_kernel");
    return 0;
}

namespace nnhip {
__global__ void set_step_kernel(float* st, int step) {
    reinterpret_cast<int*>(st)[0] = step;
    reinterpret_cast<int*>(st)[3] = 0;
}
__global__ void set_hyper_kernel(float* st, float lr, float grad_scale, float wd) {
    st[1] = lr; st[2] = grad_scale; st[4] = wd;
}
static int ensure_dev_state(FusedOptimizer* fo) {
    if (fo->dev_state) return 0;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&fo->dev_state), 16 * sizeof(float));
    if (e != hipSuccess) { fo->dev_state = nullptr; return hip_status(e, "hipMalloc(optimizer device state)"); }
    e = hipMemset(fo->dev_state, 0, 16 * sizeof(float));
    return hip_status(e, "hipMemset(optimizer device state)");
}
}  // namespace nnhip

// Both setters are ordinary stream-ordered launches (values travel as kernel arguments): no host memory to keep alive,
// no synchronisation, legal between two replays of a captured step.
extern "C" int nnhipFusedOptimizerSetStep(void* opt, int32_t step, nnhipStream_t s) {
    NNHIP_CHECK_ARG(opt != nullptr && step >= 0, NNHIP_EINVAL, "nnhipFusedOptimizerSetStep: bad arguments");
    FusedOptimizer* fo = static_cast<FusedOptimizer*>(opt);
    if (int rc = ensure_dev_state(fo)) return rc;
    hipLaunchKernelGGL(set_step_kernel, dim3(1), dim3(1), 0, (hipStream_t)s, fo->dev_state, (int)step);
    NNHIP_LAUNCH_CHECK("set_step_kernel");
    return 0;
}

extern "C" int nnhipFusedOptimizerSetHyper(void* opt, double lr, double weight_decay, float grad_scale, nnhipStream_t s) {
    NNHIP_CHECK_ARG(opt != nullptr, NNHIP_EINVAL, "nnhipFusedOptimizerSetHyper: null optimizer handle");
    FusedOptimizer* fo = static_cast<FusedOptimizer*>(opt);
    if (int rc = ensure_dev_state(fo)) return rc;
    hipLaunchKernelGGL(set_hyper_kernel, dim3(1), dim3(1), 0, (hipStream_t)s, fo->dev_state, (float)lr, grad_scale,
                       (float)weight_decay);
    NNHIP_LAUNCH_CHECK("set_hyper_kernel");
    fo->hyper_set = true;
    return 0;
}

extern "C" int nnhipFusedOptimizerSetGradDivisor(void* opt, const float* divisor_dev_or_null) {
    NNHIP_CHECK_ARG(opt != nullptr, NNHIP_EINVAL, "nnhipFusedOptimizerSetGradDivisor: null optimizer handle");
    NNHIP_CHECK_ARG(aligned4(divisor_dev_or_null), NNHIP_EALIGN, "nnhipFusedOptimizerSetGradDivisor: misaligned pointer");
    static_cast<FusedOptimizer*>(opt)->grad_div = divisor_dev_or_null;
    return 0;
}
