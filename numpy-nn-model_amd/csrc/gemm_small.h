// gemm_small.h -- the tile body of the small-problem GEMM (gemm_small.hip), shared with the fused Linear -> CrossEntropy kernel of
// rowops.hip.  See gemm_small.hip for the design.
#pragma once
#include "common.h"

namespace nnhip {

typedef float f32x16 __attribute__((ext_vector_type(16)));
enum { SG_ACT_NONE = 0, SG_ACT_SWISH = 1, SG_ACT_RELU = 2, SG_ACT_SIGMOID = 3, SG_ACT_SWISH_D = 4 };

struct SmallGemmParams {
    const float* A;
    const float* B;
    float* C;
    const float* bias;
    float* preact;
    const float* addend;
    const float* dact_arg;   // activation-gradient operand (see gemm.hip: dswish / dact)
    float* asum;
    int64_t M, N, K, lda, ldb, ldc;
    float alpha, beta;
    int act, dact, a_kmajor, b_kmajor;
};

// 4 consecutive k (k0 .. k0+3) of row r of an operand.  k-major: P[r*ld + k]; outer-major: P[k*ld + r].
// Buffer loads with a per-lane byte offset; a lane with nothing to fetch (row past the operand, k past K, k-group past the
// last) gets an offset beyond the descriptor's num_records and reads 0 -- no branch, no select on the loaded value.  (With
// `if (in range) v = *p` every load sat in its own control-flow region that ended in `s_waitcnt vmcnt(0)`: the 32 loads of a
// wave of the 32x784x128 forward were 32 memory round trips, 12 us for 6 MFLOP.)
typedef unsigned sg_u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned SG_OOB = 0xFFFFFFF0u;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t sg_rsrc(const float* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, (int)bytes, 0x00020000);
}

// row_off: byte offset of the lane's row (k-major: r*ld*4; outer-major: r*4), SG_OOB-safe only through `ok`
template <bool KMAJOR, bool VEC>
__device__ __forceinline__ float4 sg_fetch(__amdgpu_buffer_rsrc_t rs, unsigned ld4, unsigned row_off, bool ok, unsigned k0, unsigned K) {
    float4 v;
    if constexpr (KMAJOR && VEC) {                             // K % 4 == 0 here: a float4 never straddles K
        const unsigned off = (ok && k0 < K) ? row_off + k0 * 4u : SG_OOB;
        const sg_u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
        v.x = __uint_as_float(t.x); v.y = __uint_as_float(t.y); v.z = __uint_as_float(t.z); v.w = __uint_as_float(t.w);
    } else {
        const unsigned step = KMAJOR ? 4u : ld4;
        const unsigned base = KMAJOR ? row_off + k0 * 4u : k0 * ld4 + row_off;
        v.x = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (ok && k0 < K) ? base : SG_OOB, 0, 0));
        v.y = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (ok && k0 + 1 < K) ? base + step : SG_OOB, 0, 0));
        v.z = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (ok && k0 + 2 < K) ? base + 2 * step : SG_OOB, 0, 0));
        v.w = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (ok && k0 + 3 < K) ? base + 3 * step : SG_OOB, 0, 0));
    }
    return v;
}

// One 16x16 output tile (bx, by) of problem p on the 16x16x4 MFMA; `red` / `ared`: NW x 256 and NW x 16 floats of LDS.
// The block's NW waves split K (wave w takes k-groups w, w + NW, ...; a k-group = 16 consecutive k = 4 MFMAs); lane (l16, kq)
// holds A[m0 + l16][16g + 4kq .. +3] and the same of B, and step j of a group multiplies component j of both: k = 16g + 4kq + j
// on both sides (any permutation of k will do).  The four kq lanes of a row read 64 CONTIGUOUS bytes, so a wave-load touches 16
// cache lines.  (Round 2 first had 32x32 tiles on the 32x32x2 MFMA with lane <-> row, 16 B per lane: 64 lines per wave-load,
// and the vector L1 looks lines up one per clock -- the 32x784x128 forward spent ~16 k of its 17 k cycles there, 8.4 us
// against 4.5 now: tools/probes/small_gemm_phases.hip keeps both layouts side by side.)
// U k-groups (2U float4 of operands) are in flight per wave before the first MFMA.
typedef float sg_f32x4 __attribute__((ext_vector_type(4)));
// POST: called as post(c_index, value) after every C store and post.asum(row, value) after every asum store -- how a caller
// applies an optimizer update to the element it has just produced (gemm_small.hip: the README-MLP backward with Adam inside).
struct SgNoPost {
    __device__ __forceinline__ void operator()(int64_t, float) const {}
    __device__ __forceinline__ void asum(int64_t, float) const {}
};
// ATR: called as atr(a, row, k0, ok) on every A operand quad (k0 .. k0 + 3 of row `row`; ok: the quad lies inside the operand) after
// the loads of a batch have been issued and before its MFMAs -- how a caller transforms A on its way into the matrix pipe (rowops.hip:
// BatchNorm applied to the classifier head's input).  Every A element passes exactly once per (bx, by) tile.  k-major A only.
struct SgNoATransform {
    __device__ __forceinline__ void operator()(float4&, int64_t, unsigned, bool) const {}
};
template <int NW, bool AKM, bool BKM, bool VEC, int U, class POST = SgNoPost, class ATR = SgNoATransform>
__device__ __forceinline__ void sg_tile16(const SmallGemmParams& p, int bx, int by, float (*red)[16 * 16], float (*ared)[16],
                                          float* lds_copy = nullptr, POST post = POST(), ATR atr = ATR()) {   // lds_copy: also keep C[row][col] at lds_copy[row * 32 + col]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l16 = lane & 15, kq = lane >> 4;
    const int64_t m0 = (int64_t)by * 16, n0 = (int64_t)bx * 16;
    const unsigned groups = (unsigned)((p.K + 15) >> 4), K = (unsigned)p.K;
    const unsigned la4 = (unsigned)p.lda * 4u, lb4 = (unsigned)p.ldb * 4u;
    const __amdgpu_buffer_rsrc_t rsa = sg_rsrc(p.A, (unsigned)((AKM ? (p.M - 1) * p.lda + p.K : (p.K - 1) * p.lda + p.M) * 4));
    const __amdgpu_buffer_rsrc_t rsb = sg_rsrc(p.B, (unsigned)((BKM ? (p.N - 1) * p.ldb + p.K : (p.K - 1) * p.ldb + p.N) * 4));
    const bool a_ok = m0 + l16 < p.M, b_ok = n0 + l16 < p.N;
    const unsigned a_row = (unsigned)(m0 + l16) * (AKM ? la4 : 4u), b_row = (unsigned)(n0 + l16) * (BKM ? lb4 : 4u);
    sg_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    float asum = 0.f;
    // the epilogue's operands of the element this thread will finish (bias, addend, activation-gradient argument): fetched NOW,
    // under the operand loads -- loaded where they are used they were one more dependent L2 round trip (~1 us) on the tail of
    // every small GEMM.  Clamped index: unconditional loads.
    const int64_t erow = m0 + ((tid & 255) >> 4), ecol = n0 + (tid & 15);
    const bool emine = tid < 256 && erow < p.M && ecol < p.N;
    const int64_t eidx = emine ? erow * p.ldc + ecol : 0;
    float e_bias = 0.f, e_add = 0.f, e_dact = 0.f;
    if (p.bias) e_bias = p.bias[emine ? ecol : 0];
    if (p.addend) e_add = p.addend[eidx];
    if (p.dact_arg) e_dact = p.dact_arg[eidx];
    for (unsigned gb = wave; gb < groups; gb += (unsigned)NW * U) {
        float4 a[U], b[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const unsigned k0 = 16u * (gb + (unsigned)u * NW) + 4u * kq;     // a k-group past the last has k0 >= K: reads 0
            a[u] = sg_fetch<AKM, VEC>(rsa, la4, a_row, a_ok, k0, K);
            b[u] = sg_fetch<BKM, VEC>(rsb, lb4, b_row, b_ok, k0, K);
        }
        __builtin_amdgcn_sched_barrier(0);                 // all 2U loads in flight before the first MFMA (see sg_tile)
        if constexpr (!__is_same(ATR, SgNoATransform)) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const unsigned k0 = 16u * (gb + (unsigned)u * NW) + 4u * kq;
                atr(a[u], m0 + l16, k0, a_ok && k0 < K);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].x, b[u].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].y, b[u].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].z, b[u].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].w, b[u].w, acc, 0, 0, 0);
            asum += (a[u].x + a[u].y) + (a[u].z + a[u].w);
        }
    }
    // accumulator register v holds row 4 kq + v, column l16
#pragma unroll
    for (int v = 0; v < 4; ++v) red[wave][(4 * kq + v) * 16 + l16] = acc[v];
    if (p.asum) {
        asum += __shfl_xor(asum, 16, 64);
        asum += __shfl_xor(asum, 32, 64);
        if (kq == 0) ared[wave][l16] = asum;
    }
    __syncthreads();
    if (tid < 256) {
        const int o = tid;
        float v = red[0][o];
#pragma unroll
        for (int w = 1; w < NW; ++w) v += red[w][o];
        const int64_t row = m0 + (o >> 4), col = n0 + (o & 15);
        if (row < p.M && col < p.N) {
            v = p.alpha * v + e_bias;
            if (p.addend) v += e_add;
            if (p.dact_arg) {
                const float x = e_dact;
                v = p.dact == 2 ? (x > 0.f ? v : 0.f) : p.dact == 3 ? v * x : v * swish_grad_(x, p.beta);
            }
            if (p.act == SG_ACT_SWISH) {
                if (p.preact) p.preact[row * p.ldc + col] = v;
                v = v * sigmoid_fast_(p.beta * v);
            } else if (p.act == SG_ACT_SWISH_D) {
                float d;
                swish_fwd_d_(v, p.beta, v, d);
                if (p.preact) p.preact[row * p.ldc + col] = d;
            } else if (p.act == SG_ACT_RELU) {
                v = fmaxf(v, 0.f);
            } else if (p.act == SG_ACT_SIGMOID) {
                v = sigmoid_fast_(v);
            }
            p.C[row * p.ldc + col] = v;
            post(row * p.ldc + col, v);
            if (lds_copy) lds_copy[row * 32 + col] = v;
        }
    }
    if (p.asum && bx == 0 && tid < 16) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) s += ared[w][tid];
        if (m0 + tid < p.M) { p.asum[m0 + tid] = s; post.asum(m0 + tid, s); }
    }
}

}  // namespace nnhip
