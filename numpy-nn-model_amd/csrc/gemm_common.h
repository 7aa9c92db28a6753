// gemm_common.h -- what the GEMM kernels of gemm.hip (exact fp32 MFMA) and gemm_bf3.hip (split-bf16 MFMA) share:
// the launch parameters and the epilogue (alpha, bias, addend, activation-gradient mask, activation, pre-activation
// store, row sums of A, split-K slabs) -- both accumulate 128x128 tiles as 2x2 32x32 MFMA accumulators per wave, so
// everything after the K loop is the same code.
#pragma once
#include "common.h"

namespace nnhip {

typedef float f32x16 __attribute__((ext_vector_type(16)));

enum { ACT_NONE = 0, ACT_SWISH = 1, ACT_RELU = 2, ACT_SIGMOID = 3,
       ACT_SWISH_D = 4 };   // swish, and `preact` receives swish'(z) instead of z (the backward pass then only multiplies)

struct GemmParams {
    const float* A;
    const float* B;
    float* C;
    const float* bias;   // [N] or null
    float* preact;       // [M, ldc] or null: pre-activation (z) output for ACT_SWISH
    int64_t M, N, K, lda, ldb, ldc, sA, sB, sC;
    int64_t sA2, sB2, sC2;  // inner batch level: batch index z -> (z / batch2, z % batch2)
    int batch2;
    float alpha;            // C = alpha * (A B) + bias
    int tiles_m, tiles_n, splitk;
    int lockstep = 0;     // 1: the two blocks of a CU keep step through the k-loop (gemm.hip, lock-step mode)
#ifdef GEMM_PROF
    long long* prof;      // developer instrumentation (tools/gemm_prof.py): per-block cycle stamps
#endif
    int64_t k_per_split;  // multiple of BK
    float* slab;          // split-K partials [splitk][M][N] (dense)
    const float* zeros;   // >= 16 bytes of zeros: the load target of out-of-range lanes
    int cvec;             // 1: C/bias/preact rows are 16-B aligned and N % 4 == 0 -> float4 epilogue
    int act;
    float beta;
    const float* dswish;  // [M, ldc] or null: C = (alpha*AB + bias + addend) * act'(dswish[m,n])  (may alias C); act' per `dact`
    int dact;             // 1: swish'(z; beta), dswish = z;  2: relu'(f) = [f > 0], dswish = the forward OUTPUT f;  3: a plain multiplier (the saved swish'(z))
    const float* addend;  // [M, ldc] or null: C = act(alpha*AB + bias + addend)  (residual / gradient accumulation)
    float* asum;          // CS variants: asum[m] = sum_k A[m,k] (Linear: db = column sums of dO, fused into dW = dO^T X)
    float* asum_slab;     // split-K partials [splitk][M]
};

constexpr int BM = 128, BN = 128, NT = 256;

// global -> registers.  R = extent of the operand's outer dim, r0/k0 = tile origin.
// Branch-free: an out-of-range lane loads from a 16-byte block of zeros (`Z`, library-owned) instead of
// being masked or zeroed afterwards, so the whole K-loop body is one basic block, the loads can be
// interleaved with MFMAs, and nothing consumes a loaded value before the LDS store at the end of the tile.
template <int BK, bool KC, bool VEC>
__device__ __forceinline__ void g2r(float4 (&r)[BK / 8], const float* __restrict__ P, int64_t ld,
                                    int64_t R, int64_t Kend, int64_t r0, int64_t k0, int tid, bool live,
                                    const float* __restrict__ Z) {
#pragma unroll
    for (int p = 0; p < BK / 8; ++p) {
        const int idx = tid + NT * p;
        int64_t gr, gk;
        const float* src;
        if constexpr (KC) {
            const int rr = idx / (BK / 4), k4 = (idx % (BK / 4)) * 4;
            gr = r0 + rr; gk = k0 + k4;
            src = P + gr * ld + gk;
        } else {
            const int kk = idx / 32, r4 = (idx % 32) * 4;
            gk = k0 + kk; gr = r0 + r4;
            src = P + gk * ld + gr;
        }
        if constexpr (VEC) {
            const bool ok = live && gr < R && gk < Kend;
            r[p] = *reinterpret_cast<const float4*>(ok ? src : Z);
        } else {
            // scalar path: element e steps along the contiguous dim (k for k-major, outer otherwise)
            float t[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const bool ok = live && (KC ? (gr < R && gk + e < Kend) : (gk < Kend && gr + e < R));
                t[e] = *(ok ? src + e : Z);
            }
            r[p] = make_float4(t[0], t[1], t[2], t[3]);
        }
    }
}

// ---- LDS staging shared by gemm.hip and gemm_pst.hip ---------------------------------------------------------------------
template <int BK, bool KC>
struct Tile {
    static constexpr int LD = KC ? (BK + 4) : 128;
    static constexpr int SIZE = KC ? 128 * (BK + 4) : BK * 128;  // floats
    static constexpr int NV = BK / 8;                             // float4 per thread per tile
};

// registers -> LDS stage
template <int BK, bool KC>
__device__ __forceinline__ void r2s(const float4 (&r)[BK / 8], float* __restrict__ S, int tid) {
#pragma unroll
    for (int p = 0; p < BK / 8; ++p) {
        const int idx = tid + NT * p;
        if constexpr (KC) {
            const int rr = idx / (BK / 4), k4 = (idx % (BK / 4)) * 4;
            *reinterpret_cast<float4*>(&S[rr * (BK + 4) + k4]) = r[p];
        } else {
            const int kk = idx / 32, r4 = (idx % 32) * 4;
            *reinterpret_cast<float4*>(&S[kk * 128 + r4]) = r[p];
        }
    }
}

// Read the 4 MFMA operands (k = 8g+j [+4 for the upper half-wave], j = 0..3) of one 32-row subtile.
template <int BK, bool KC>
__device__ __forceinline__ void frag(float (&f)[4], const float* __restrict__ S, int row, int g,
                                     int lh) {
    if constexpr (KC) {
        const float4 v = *reinterpret_cast<const float4*>(&S[row * (BK + 4) + g * 8 + lh * 4]);
        f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) f[j] = S[(g * 8 + j + 4 * lh) * 128 + row];
    }
}

// ---- fast operand fetch (vectorised variants of gemm.hip) ---------------------------------------------------------------
// The fp32 MFMA of gfx950 runs on the SAME lanes as the vector ALU (peak 157.3 TFLOP/s either way; a probe,
// tools/probes/mfma_valu_probe.hip, shows VALU instructions add their ~3.3 cycles to the MFMA time instead of hiding under
// it).  Round 1's g2r formed a 64-bit address and an out-of-range select per load: 43 VALU instructions per k-tile of 64
// MFMAs.  Here a load is `buffer_load_dwordx4 v, v_off, s[rsrc], s_koff offen`: the buffer descriptor holds the operand's
// row origin for this block, the k position is a scalar byte offset advanced by scalar adds, and the per-thread 32-bit
// byte offset is computed ONCE -- no vector instruction in the loop besides the MFMAs and the LDS traffic.
//   * rows past the operand's extent are CLAMPED to its last rows, not zeroed: row m of A (n of B) only ever reaches
//     row m (column n) of C, which the epilogue does not store;
//   * a partial last k-tile is fetched as the LAST BK columns of the k range (offset shifted back, all in bounds) and
//     only its new k-groups are multiplied (gemm.hip, tail_mma) -- no lane ever needs zeros.
// Host-side conditions (gemm_f32_ex): 16-B aligned rows, K % 8 == 0, K >= BK, byte offsets within a block < 2^32.
typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t operand_rsrc(const float* origin) {
    // raw buffer (stride 0), no bounds clamp wanted (num_records = 2^32 - 1), gfx9 dword 3 (32-bit data format)
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(origin), 0, 0xFFFFFFFF, 0x00020000);
}
template <int BK, bool KC>
__device__ __forceinline__ void op_offsets(unsigned (&off)[BK / 8], int64_t ld, int64_t R, int64_t r0, int tid) {
#pragma unroll
    for (int p = 0; p < BK / 8; ++p) {
        const int idx = tid + NT * p;
        if constexpr (KC) {
            const int rr = idx / (BK / 4), k4 = (idx % (BK / 4)) * 4;
            const int64_t rc = min((int64_t)rr, R - 1 - r0);
            off[p] = (unsigned)((rc * ld + k4) * 4);
        } else {
            const int kk = idx / 32, r4 = (idx % 32) * 4;
            const int64_t rc = min((int64_t)r4, R - 4 - r0);
            off[p] = (unsigned)(((int64_t)kk * ld + rc) * 4);
        }
    }
}
template <int BK>
__device__ __forceinline__ void g2r_fast(float4 (&r)[BK / 8], __amdgpu_buffer_rsrc_t rs, unsigned koff, const unsigned (&off)[BK / 8]) {
#pragma unroll
    for (int p = 0; p < BK / 8; ++p) {
        // (component-wise: `r[p] = *(float4*)...` is a struct copy that hipcc turns into memcpys through a stack slot)
        const u32x4_ t = __builtin_amdgcn_raw_buffer_load_b128(rs, off[p], koff, 0);
        r[p].x = __uint_as_float(t.x); r[p].y = __uint_as_float(t.y); r[p].z = __uint_as_float(t.z); r[p].w = __uint_as_float(t.w);
    }
}

// `cs`: this thread's row sums of the A elements it staged (CS variants only).  The K loop has ended with a barrier, so
// the whole dynamic LDS block `smem` is free.
template <bool CS>
__device__ __forceinline__ void gemm_epilogue(f32x16 (&acc)[2][2], const GemmParams& p, float* __restrict__ smem, float4 cs,
                                              int tid, int wave, int lane, int wm, int wn, int l31, int lh, int64_t m0,
                                              int64_t n0, int tn, int split, int64_t c_off, long long* ets = nullptr) {
    // ---- epilogue ------------------------------------------------------------------------------------------
    // acc register e of a 32x32 MFMA tile holds row (e&3) + 8*(e>>2) + 4*lh, column l31: a lane owns a strided
    // COLUMN, so direct stores are 64 dword stores per lane (2 rows x 128 B per instruction) and the tail is
    // store-issue bound (~15 us per generation of tiles, measured by a K sweep).  Instead each wave transposes
    // its tile through its own LDS region, 32 rows at a time, and stores float4 rows: 4x fewer instructions,
    // 4 rows x 256 B each.  (Scalar path kept for unaligned / N % 4 != 0 outputs.)
    const bool to_slab = p.splitk > 1;
    if constexpr (CS) {
        if (p.asum && tn == 0) {                             // the K loop ended with a barrier: LDS is free
            *reinterpret_cast<float4*>(&smem[(tid >> 5) * 128 + (tid & 31) * 4]) = cs;
            __syncthreads();
            if (tid < 128) {
                float t = 0.f;
#pragma unroll
                for (int r = 0; r < 8; ++r) t += smem[r * 128 + tid];
                if (m0 + tid < p.M) (to_slab ? p.asum_slab + (int64_t)split * p.M : p.asum)[m0 + tid] = t;
            }
            __syncthreads();
        }
    }
    float* __restrict__ C = to_slab ? p.slab + (int64_t)split * p.M * p.N : p.C + c_off;
    const int64_t ldc = to_slab ? p.N : p.ldc;
    if (p.cvec) {
        constexpr int ELD = 68;                              // 64 + 4 floats: rows stay 16-B aligned
        float* E = smem + wave * (32 * ELD);                 // the K loop ended with a barrier: LDS is free
        const int er = lane >> 4, ec = (lane & 15) * 4;      // read side: row-in-group, column of the float4
        const int64_t col = n0 + wn * 64 + ec;
        const bool col_ok = col < p.N;                       // N % 4 == 0 on this path
        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (!to_slab && p.bias && col_ok) bv = *reinterpret_cast<const float4*>(p.bias + col);
        // the one extra epilogue operand (addend or swish' argument; the dispatcher rejects both at once) is fetched one
        // 32-row group ahead: issued before the LDS transposition of the group that uses it, so its latency hides under
        // that and under the previous group's stores (loading it inside the store loop cost ~4 us per tile).
        const float* __restrict__ extra = to_slab ? nullptr : (p.addend ? p.addend : p.dswish);
        const bool is_add = p.addend != nullptr;
        float4 ex[8];
        auto fetch_extra = [&](int i, int it) {
            const int64_t row = m0 + wm * 64 + i * 32 + it * 4 + er;
            ex[it] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < p.M && col_ok) ex[it] = *reinterpret_cast<const float4*>(extra + c_off + row * ldc + col);
        };
        if (extra) {
#pragma unroll
            for (int it = 0; it < 8; ++it) fetch_extra(0, it);
        }
#ifdef GEMM_PROF
        if (ets) ets[0] = clock64();
#endif
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    E[((e & 3) + 8 * (e >> 2) + 4 * lh) * ELD + n * 32 + l31] = acc[i][n][e];
            __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0); E is private to the wave: no block barrier needed
            __builtin_amdgcn_wave_barrier();
#ifdef GEMM_PROF
            if (ets) ets[1 + 2 * i] = clock64();
#endif
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int rl = it * 4 + er;
                const int64_t row = m0 + wm * 64 + i * 32 + rl;
                float4 v = *reinterpret_cast<const float4*>(&E[rl * ELD + ec]);
                const float4 x = ex[it];
                if (extra && i == 0) fetch_extra(1, it);
                if (row < p.M && col_ok) {
                    if (!to_slab) {
                        v.x = p.alpha * v.x + bv.x; v.y = p.alpha * v.y + bv.y;
                        v.z = p.alpha * v.z + bv.z; v.w = p.alpha * v.w + bv.w;
                        if (extra) {
                            if (is_add) {
                                v.x += x.x; v.y += x.y; v.z += x.z; v.w += x.w;
                            } else if (p.dact == 2) {   // gradient through h = relu(z): mask by the forward output
                                v.x = x.x > 0.f ? v.x : 0.f; v.y = x.y > 0.f ? v.y : 0.f;
                                v.z = x.z > 0.f ? v.z : 0.f; v.w = x.w > 0.f ? v.w : 0.f;
                            } else if (p.dact == 3) {   // the forward pass saved swish'(z) itself (ACT_SWISH_D)
                                v.x *= x.x; v.y *= x.y; v.z *= x.z; v.w *= x.w;
                            } else {   // gradient through h = swish(z): the dX GEMM of the NEXT layer hands back dz
                                v.x *= swish_grad_(x.x, p.beta); v.y *= swish_grad_(x.y, p.beta);
                                v.z *= swish_grad_(x.z, p.beta); v.w *= swish_grad_(x.w, p.beta);
                            }
                        }
                        if (p.act == ACT_SWISH) {
                            if (p.preact) *reinterpret_cast<float4*>(p.preact + c_off + row * ldc + col) = v;
                            v.x *= sigmoid_fast_(p.beta * v.x); v.y *= sigmoid_fast_(p.beta * v.y);
                            v.z *= sigmoid_fast_(p.beta * v.z); v.w *= sigmoid_fast_(p.beta * v.w);
                        } else if (p.act == ACT_SWISH_D) {
                            float4 d;
                            swish_fwd_d_(v.x, p.beta, v.x, d.x); swish_fwd_d_(v.y, p.beta, v.y, d.y);
                            swish_fwd_d_(v.z, p.beta, v.z, d.z); swish_fwd_d_(v.w, p.beta, v.w, d.w);
                            if (p.preact) *reinterpret_cast<float4*>(p.preact + c_off + row * ldc + col) = d;
                        } else if (p.act == ACT_RELU) {
                            v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                        } else if (p.act == ACT_SIGMOID) {
                            v.x = sigmoid_fast_(v.x); v.y = sigmoid_fast_(v.y); v.z = sigmoid_fast_(v.z); v.w = sigmoid_fast_(v.w);
                        }
                    }
                    *reinterpret_cast<float4*>(C + row * ldc + col) = v;
                }
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);   // this group's LDS reads are done before the next group overwrites E
            __builtin_amdgcn_wave_barrier();
#ifdef GEMM_PROF
            if (ets) ets[2 + 2 * i] = clock64();
#endif
        }
        return;
    }
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int64_t col = n0 + wn * 64 + n * 32 + l31;
        if (col >= p.N) continue;
        const float bv = (!to_slab && p.bias) ? p.bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int64_t rbase = m0 + wm * 64 + i * 32 + 4 * lh;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int64_t row = rbase + (e & 3) + 8 * (e >> 2);
                if (row >= p.M) continue;
                float v = to_slab ? acc[i][n][e] : p.alpha * acc[i][n][e] + bv;
                if (!to_slab) {
                    if (p.addend) v += p.addend[c_off + row * ldc + col];
                    if (p.dswish) {
                        const float x = p.dswish[c_off + row * ldc + col];
                        v = p.dact == 2 ? (x > 0.f ? v : 0.f) : p.dact == 3 ? v * x : v * swish_grad_(x, p.beta);
                    }
                    if (p.act == ACT_SWISH) {
                        if (p.preact) p.preact[c_off + row * ldc + col] = v;
                        v = v * sigmoid_fast_(p.beta * v);
                    } else if (p.act == ACT_SWISH_D) {
                        float d;
                        swish_fwd_d_(v, p.beta, v, d);
                        if (p.preact) p.preact[c_off + row * ldc + col] = d;
                    } else if (p.act == ACT_RELU) {
                        v = fmaxf(v, 0.f);
                    } else if (p.act == ACT_SIGMOID) {
                        v = sigmoid_fast_(v);
                    }
                }
                C[row * ldc + col] = v;
            }
        }
    }
}

// block id -> logical id: XCD-aware order.  Block b runs on XCD b % 8; each XCD gets a contiguous run of logical ids.  Bijective.
__device__ __forceinline__ int xcd_order(int b, int nwg) {
    const int xcd = b & 7, q = nwg >> 3, r = nwg & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
}


// a group of independent parameter-gradient GEMMs launched as one grid (gemm.hip: gemm_f32_group_kernel, gemm_bf3.hip)
constexpr int GEMM_GROUP_MAX = 8;
struct GemmGroup {
    GemmParams p[GEMM_GROUP_MAX];
    int start[GEMM_GROUP_MAX + 1];
    int n;
    int by_job;      // 0: one XCD-aware order over the whole grid (an XCD works on one or two jobs);  1: jobs in DISPATCH order, the
                     // XCD-aware order inside each (all of job 0's blocks are handed out before any of job 1's)
};

}  // namespace nnhip
