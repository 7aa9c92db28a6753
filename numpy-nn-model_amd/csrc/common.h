// common.h -- shared host/device helpers for libneunet_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/neunet_hip.h"

namespace nnhip {

constexpr int kWave = 64;  // CDNA wavefront

// ---- status plumbing ---------------------------------------------------------------------------
void set_last_error(const char* fmt, ...);
int hip_status(hipError_t e, const char* what);

#define NNHIP_CHECK_ARG(cond, code, ...)      \
    do {                                       \
        if (!(cond)) {                         \
            nnhip::set_last_error(__VA_ARGS__); \
            return (code);                     \
        }                                      \
    } while (0)

// Launch check: catches bad launch configuration without synchronising.
#define NNHIP_LAUNCH_CHECK(name)                          \
    do {                                                  \
        hipError_t _e = hipGetLastError();                \
        if (_e != hipSuccess) return nnhip::hip_status(_e, name); \
    } while (0)

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
inline bool aligned4(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 3u) == 0; }
inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Grow-only per-process device workspace (split-K slabs, column-sum partials).  Freed by
// nnhipCleanup().  Growing it synchronises the device (hipFree) -- it happens at most a few times.
void* workspace(size_t bytes);
void* workspace_arena(int which, size_t bytes);   // 0 = the general block (== workspace), 1 = the grouped dW launch's slabs
bool workspace_locked();               // nnhipWorkspaceLock(1): a captured hipGraph holds library-owned addresses -- nothing may move
// Deferred parameter gradients (nnhipWeightGradDefer, linear.hip): on while Tensor.backward() walks the tape.  conv2d.hip queues the
// REDUCE of a small-channel conv's per-image partial weight gradients behind it (the partials sit in an arena of their own, not
// in the shared workspace) and launches the queued reduces as one grid at the flush.
bool wgrad_defer_on();
int conv_reduce_flush(void* stream);
void conv_reduce_cleanup();            // frees the partials arena (nnhipCleanup)
int colsum_flush(void* stream);        // rowops.hip: the queued RMSNorm dw / db column sums of a backward pass, as one launch
void colsum_cleanup();
// one deferred parameter-gradient GEMM (gemm.hip: gemm_f32_wgrad_group): C[M,N] = A^T B with A [K, M] and B [K, N] dense
struct WgradJob {
    const float* A; const float* B; float* C; float* asum;
    int64_t M, N, K;
};

// 256 bytes of device zeros, allocated once per process (never freed): where out-of-range GEMM lanes load from.
const float* zero_block();
// Library-owned, zero-initialised device words for in-launch arrival tickets (kSyncWords unsigned ints, never freed).
// Every kernel that uses a slot leaves it zeroed for the next launch.  Slots are per kernel family (the library's
// workspace already makes it one-stream-at-a-time per process).
constexpr int kSyncWords = 1024;
enum SyncSlot { SYNC_CE = 0, SYNC_MLP = 24 };
unsigned* sync_words();
// The library's device error word: 4 bytes of pinned, device-mapped host memory.  A kernel that detects a broken device state
// (today: the optimizer-in-backward arrival barrier timing out, gemm_small.hip) stores a NNHIP_DEVERR_* code there with a
// system-scope store and carries on WITHOUT its side effect; the host reads the word with a plain load -- no synchronisation --
// and turns it into the sticky status NNHIP_EDEVICE.  nullptr when the allocation failed (kernels then just skip the store).
enum DeviceErrorCode { NNHIP_DEVERR_NONE = 0, NNHIP_DEVERR_MLP_BARRIER = 1 };
unsigned* device_error_word();                   // device-side address
int device_error_status(const char* who);        // 0, or NNHIP_EDEVICE with the message set
// Order this launch behind the previous user of the sync words / ticket partials when it arrives on another stream (runtime.hip).
int serialize_shared_state(hipStream_t st);
void shared_state_done(hipStream_t st);        // the user's launches are enqueued: leave an event for the next stream to wait on
struct SharedStateUse {                         // RAII: construct after serialize_shared_state(), every return path reports the launches
    hipStream_t st;
    explicit SharedStateUse(hipStream_t s) : st(s) {}
    ~SharedStateUse() { shared_state_done(st); }
    SharedStateUse(const SharedStateUse&) = delete;
    SharedStateUse& operator=(const SharedStateUse&) = delete;
};

// ---- device helpers ----------------------------------------------------------------------------
// Wave64 all-reduce on the DPP lanes, no LDS traffic and no address registers: four row-local steps (quad_perm xor 1,
// xor 2, row_half_mirror, row_mirror -- each one VALU instruction with the lane permutation folded into its operand
// fetch) leave every lane of a 16-lane row with the row's total; v_readlane of lanes 0/16/32/48 then combines the four
// rows into a wave-uniform (scalar-register) result.  The __shfl_xor form this replaces is six ds_bpermute_b32 round
// trips through the LDS pipe plus six VGPRs of lane addresses the compiler kept alive across a persistent row loop
// (round 3: ce_rows_kernel 100 -> 6x VGPRs together with the constant one-hot test).  Must be called wave-converged.
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
template <int CTRL>
__device__ __forceinline__ int dpp_i32(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, false); }
constexpr int kDppXor1 = 0xB1, kDppXor2 = 0x4E, kDppHalfMirror = 0x141, kDppMirror = 0x140;
__device__ __forceinline__ float readlane_f32(float v, int lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}

__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_f32<kDppXor1>(v);
    v += dpp_f32<kDppXor2>(v);
    v += dpp_f32<kDppHalfMirror>(v);
    v += dpp_f32<kDppMirror>(v);
    return (readlane_f32(v, 0) + readlane_f32(v, 16)) + (readlane_f32(v, 32) + readlane_f32(v, 48));
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_f32<kDppXor1>(v));
    v = fmaxf(v, dpp_f32<kDppXor2>(v));
    v = fmaxf(v, dpp_f32<kDppHalfMirror>(v));
    v = fmaxf(v, dpp_f32<kDppMirror>(v));
    return fmaxf(fmaxf(readlane_f32(v, 0), readlane_f32(v, 16)), fmaxf(readlane_f32(v, 32), readlane_f32(v, 48)));
}
__device__ __forceinline__ int wave_sum_int(int v) {
    v += dpp_i32<kDppXor1>(v);
    v += dpp_i32<kDppXor2>(v);
    v += dpp_i32<kDppHalfMirror>(v);
    v += dpp_i32<kDppMirror>(v);
    return (__builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16)) +
           (__builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48));
}

// Block-wide sum for blockDim.x = NW*64 threads.  `red` is NW floats of LDS.  All threads get the sum.
template <int NW>
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    if constexpr (NW == 1) return v;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();  // protect `red` against a previous use
    if (lane == 0) red[w] = v;
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NW; ++i) s += red[i];
    return s;
}
// Two sums with one pair of barriers.  `red` is 2*NW floats.
template <int NW>
__device__ __forceinline__ void block_sum2(float& a, float& b, float* red) {
    a = wave_sum(a);
    b = wave_sum(b);
    if constexpr (NW == 1) return;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) { red[w] = a; red[NW + w] = b; }
    __syncthreads();
    float sa = 0.f, sb = 0.f;
#pragma unroll
    for (int i = 0; i < NW; ++i) { sa += red[i]; sb += red[NW + i]; }
    a = sa;
    b = sb;
}
template <int NW>
__device__ __forceinline__ float block_max(float v, float* red) {
    v = wave_max(v);
    if constexpr (NW == 1) return v;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    float s = red[0];
#pragma unroll
    for (int i = 1; i < NW; ++i) s = fmaxf(s, red[i]);
    return s;
}

// A device WORD that other kernels rewrite between launches (step counters, counts, scales) and that every thread of a
// block reads: take it with an agent-scope load.  A plain uniform load becomes a scalar load, and a scalar-cache line of
// such a word was observed stale across a kernel boundary on some CUs (round 2, the attention dropout seed offset).
__device__ __forceinline__ float ld_dev_f32(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int ld_dev_i32(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

constexpr float kLog2e = 1.4426950408889634f;
// exp(x) for x <= ~0 as one v_exp_f32 (1 ulp) after a multiply: softmax / log-sum-exp terms exp(x - max).
__device__ __forceinline__ float exp_fast_(float x) { return __builtin_amdgcn_exp2f(x * kLog2e); }
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
// d/dz [z * sigmoid(beta z)] in the reference's form (neunet/nn/activations.py:223-232): beta f + s (1 - beta f), f = z s
// The GEMM epilogues use the hardware exp2 / rcp (1 ulp each; the library expf + IEEE divide cost ~30 VALU
// instructions per element and made the epilogue of a K=512 GEMM as expensive as a separate pass over the tensor).
__device__ __forceinline__ float sigmoid_fast_(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
// h = z sigmoid(beta z) and d = dh/dz = s + beta h (1 - s) from ONE sigmoid: the forward epilogue that saves d instead of z
// (ACT_SWISH_D) leaves the backward epilogue a single multiply -- in an fp32 MFMA kernel every VALU instruction is paid in
// matrix-pipe time, and the sigmoid's two transcendentals are 32 of the ~64 cycles per element swish_grad_ costs.
__device__ __forceinline__ void swish_fwd_d_(float z, float beta, float& h, float& d) {
    const float s = sigmoid_fast_(beta * z);
    h = z * s;
    d = fmaf(beta * h, 1.0f - s, s);
}
__device__ __forceinline__ float swish_grad_(float z, float beta) {
    const float s = sigmoid_fast_(beta * z);
    const float f = z * s;
    return beta * f + s * (1.f - beta * f);
}

}  // namespace nnhip
