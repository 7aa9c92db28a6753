// attention.h -- structures and small device helpers shared by the fused attention kernels (attention.hip: the general
// tiled kernels; attention_sb.hip: the balanced "super-block" kernels for causal self-attention at T = 256, head dim 64).
#pragma once
#include <math.h>

#include "common.h"

namespace nnhip {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int AT_BK = 64;    // keys (queries in the dK/dV kernel) per tile
constexpr float AT_MASKED = -1e9f;
constexpr float AT_LOG2E = 1.4426950408889634f;
constexpr float AT_MASKED2 = AT_MASKED * AT_LOG2E;   // the -1e9 fill in log2 units (scores are carried as s*log2(e))

// Developer instrumentation (tools/attn_prof.py builds this file a second time with -DAT_PROF): per-wave cycle counts of
// the kernel's phases, written to p.prof[(block * 4 + wave) * 8 + phase].  Compiled out of the library.
#ifdef AT_PROF
#define AT_PROF_DECL long long pacc_[8] = {0, 0, 0, 0, 0, 0, 0, 0}; long long tlast_ = clock64()
#define AT_T(i) do { const long long t_ = clock64(); pacc_[i] += t_ - tlast_; tlast_ = t_; } while (0)
#define AT_PROF_STORE(P) do { if ((P) && (threadIdx.x & 63) == 0) { for (int i_ = 0; i_ < 8; ++i_) (P)[((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 + i_] = pacc_[i_]; } } while (0)
#else
#define AT_PROF_DECL do {} while (0)
#define AT_T(i) do {} while (0)
#define AT_PROF_STORE(P) do {} while (0)
#endif

#define AT_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
#define AT_SCHED_MFMA(n) __builtin_amdgcn_sched_group_barrier(0x008, (n), 0)
#define AT_SCHED_DSRD(n) __builtin_amdgcn_sched_group_barrier(0x100, (n), 0)

struct AttnExtra {                                    // GEN = true only
    const unsigned long long* mask_bits;              // [B, Tq, ceil(Tk/64)]: bit j of word w = key 64w+j visible to this query; or null
    const unsigned long long* mask_bitsT;             // [B, Tk, ceil(Tq/64)]: the same mask, bits running over queries; or null
    const unsigned char* row_any;                     // [B, Tq]: 1 iff the query sees at least one key (tile skipping); or null
    const float* drop_mask;                           // [B, H, Tq, Tk] multipliers (0 or 1/(1-p)); or null
    const unsigned* seed_dev;                         // or null: a device word ADDED to drop_seed (e.g. a step counter, so that a
                                                      // captured hipGraph draws a fresh mask on every replay)
    unsigned drop_seed, drop_threshold;               // hash dropout: keep iff hash >= threshold (0 = no dropout)
    float drop_scale;                                 // 1/(1-p)
};

struct AttnParams {
    const float* Q; const float* K; const float* V;   // [B, T, D]
    float* O;                                         // [B, Tq, D]
    float* LSE;                                       // [B, H, Tq, 2] = (max, log2 sum), log2 units
    const int32_t* key_valid;                         // [B, Tk] or null
    int B, H, Tq, Tk;
    int64_t D;                                        // row stride of O (floats): H * DH
    int64_t LQ;                                       // row stride of Q, K, V (>= H * DH: 3*H*DH for a fused q|k|v buffer)
    float scale;                                      // multiplies QK^T (1/sqrt(d_model))
    int causal;
    int pair;                                         // forward: 1 = a block runs TWO query blocks, the heaviest left and its light complement
    AttnExtra x;
#ifdef AT_PROF
    long long* prof;
#endif
};

struct AttnBwdParams {
    const float* Q; const float* K; const float* V; const float* dO;   // [B, T, D]
    const float* LSE; float* Dsum;                                     // [B, H, Tq, 2], [B, H, Tq] (written by the dQ kernel)
    const float* O;                                                    // [B, Tq, D] forward output
    float* dQ; float* dK; float* dV;                                   // [B, T, D]
    const int32_t* key_valid;
    int B, H, Tq, Tk;
    int64_t D;                                                         // row stride of dO / O
    int64_t LQ;                                                        // row stride of Q, K, V, dQ, dK, dV
    float scale;
    int causal;
    AttnExtra x;
#ifdef AT_PROF
    long long* prof;
#endif
};

// row (within a 32-row tile) carried by accumulator register e, for half-wave lh
__device__ __forceinline__ int acc_row(int e, int lh) { return (e & 3) + 8 * (e >> 2) + 4 * lh; }

// ---- dropout multiplier of element (row = (b*H + h)*Tq + q, key) ------------------------------------------------------
// lowbias32-style integer hash of (seed, row, key): cheap enough (~10 VALU) to re-evaluate in all three kernels.
__device__ __forceinline__ unsigned at_rowkey(unsigned seed, unsigned row) { return (seed * 0x85EBCA6Bu + 0x9E3779B9u) ^ (row * 0xC2B2AE35u); }
__device__ __forceinline__ unsigned at_hash(unsigned rowkey, unsigned key) {
    unsigned x = rowkey ^ (key * 0x9E3779B1u);
    x ^= x >> 16; x *= 0x7FEB352Du;
    x ^= x >> 15; x *= 0x846CA68Bu;
    x ^= x >> 16;
    return x;
}
// the device-side seed offset: an agent-scope (L2-served) load, NOT a scalar load -- the word is rewritten between launches
// by other kernels and a scalar-cache line of it was observed stale on some CUs
__device__ __forceinline__ unsigned at_seed_offset(const AttnExtra& x) {
    return x.seed_dev ? __hip_atomic_load(x.seed_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
}
__device__ __forceinline__ float at_drop_mult(const AttnExtra& x, unsigned rowkey, int64_t row, int Tk, int key) {
    if (x.drop_mask) return x.drop_mask[row * Tk + key];
    return at_hash(rowkey, (unsigned)key) >= x.drop_threshold ? x.drop_scale : 0.f;
}

// attention_sb.hip: the balanced 8-wave kernels for causal self-attention at T = 256, head dim 64 (no dense mask, no dropout)
bool attn_sb_applicable(int64_t B, int64_t H, int64_t Tq, int64_t Tk, int64_t head_dim, int causal, bool gen);
int attn_sb_forward(const AttnParams& p, hipStream_t st);
int attn_sb_backward(const AttnBwdParams& p, hipStream_t st);

}  // namespace nnhip
