// gemm_bf3.hip -- fp32 GEMM on the bf16 matrix cores: every fp32 operand is split EXACTLY into three bf16 pieces
// (x = hi + mid + lo, 8 + 8 + 8 significand bits, by truncation: no rounding anywhere in the split) and the product is
// formed from the six largest of the nine piece products,
//     a*b ~= hi*hi + hi*mid + mid*hi + mid*mid + hi*lo + lo*hi          (dropped: mid*lo, lo*mid, lo*lo <= 2^-23 |a b|)
// each an exact bf16 x bf16 product accumulated in fp32 by v_mfma_f32_32x32x16_bf16.  The result carries a relative
// error of <= ~2^-23 per product -- the same order as fp32 rounding itself (an fp32 MFMA / sgemm dot product of length K
// has a worst-case bound of K 2^-24) -- at 6 bf16 MFMAs per 16 k instead of 8 fp32 MFMAs of 4x the issue time:
// 6 x 32 = 192 matrix-pipe cycles per 32x32x16 block instead of 8 x 64 = 512, i.e. a ceiling of 2.67x the fp32 MFMA
// rate (~419 TFLOP/s fp32-equivalent).  gfx950 has no xf32/TF32 (MI355X_MICROARCH.md); this is the MI355X-native way
// to put an fp32 GEMM on the fast matrix path.  The reference's own fused CUDA path multiplies in TF32 (10-bit
// significand, 1e-3 test tolerance: linear_swish_cutlass_evt_full.cu:440); this path keeps 24 bits.
//
// OPT-IN (nnhipSetGemmMode(1) / NNHIP_GEMM_MODE=bf16x3): the default stays the exact-fp32 MFMA kernel of gemm.hip, and
// bench.py's headline numbers are measured with the default.  Parity tests run both.
//
// Structure = gemm.hip's: 256 threads = 2x2 waves, 128x128 tile, wave tile 64x64 = 2x2 32x32 accumulators, 2-stage LDS
// ring with one barrier per k-tile, the shared epilogue of gemm_common.h.  Differences:
//   * BK = 16 (one MFMA k-step per tile); a stage holds the three bf16 planes of both operands:
//     plane[128 rows][16 k] with a 48-byte row stride (conflict-free ds_read_b128 / ds_write_b128) = 36 KB per stage;
//   * a k-major operand is fetched as float4 (4 k of a row), split, and stored as three 8-byte pieces; an outer-major
//     operand is fetched as 8 coalesced dwords (8 k of one row: lanes <-> rows), split and stored as three 16-byte
//     pieces -- the transposition to k-major happens in the fetch pattern, not in LDS;
//   * the MFMA operand of lane (l31, lh) is 8 consecutive k of row l31: one ds_read_b128 per plane.
#include <stdlib.h>

#include "gemm_common.h"

namespace nnhip {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int B3_BK = 16;
constexpr int B3_RS = 48;                               // bytes per LDS row: 16 bf16 + 16 bytes of padding
constexpr int B3_PLANE = 128 * B3_RS;                   // one plane of one operand
constexpr int B3_OPERAND = 3 * B3_PLANE;
constexpr int B3_STAGE = 2 * B3_OPERAND;                // 36864 bytes

// x = hi + mid + lo exactly; each piece has its low 16 bits clear (a bf16 in the high half)
__device__ __forceinline__ void split3(float x, unsigned& hi, unsigned& mid, unsigned& lo) {
    hi = __float_as_uint(x) & 0xFFFF0000u;
    const float r1 = x - __uint_as_float(hi);
    mid = __float_as_uint(r1) & 0xFFFF0000u;
    lo = __float_as_uint(r1 - __uint_as_float(mid));   // <= 8 significant bits left: already a bf16
}
// {a.high16, b.high16} -> one dword (a in the low half)
__device__ __forceinline__ unsigned pack_hi16(unsigned a, unsigned b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }

// ---- operand staging ---------------------------------------------------------------------------------------------
// k-major operand: 2 float4 per thread (rows idx/4, k = 4 (idx%4) .. +3) -- g2r<16, true, VEC> of gemm_common.h.
// outer-major operand: thread (row = tid & 127, kh = tid >> 7) fetches k = 8 kh .. +7 of its row: 8 dwords, each
// wave-load 256 contiguous bytes.
struct StageRegs {
    float4 v[2];       // k-major: 2 x (4 k of a row);  outer-major: 8 k of one row
};

template <bool KC, bool VEC>
__device__ __forceinline__ void b3_fetch(StageRegs& r, const float* __restrict__ P, int64_t ld, int64_t R, int64_t Kend,
                                         int64_t r0, int64_t k0, int tid, bool live, const float* __restrict__ Z) {
    if constexpr (KC) {
        g2r<B3_BK, true, VEC>(r.v, P, ld, R, Kend, r0, k0, tid, live, Z);
    } else {
        const int64_t gr = r0 + (tid & 127);
        const int64_t kb = k0 + 8 * (tid >> 7);
        float t[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const bool ok = live && gr < R && kb + j < Kend;
            t[j] = *(ok ? P + (kb + j) * ld + gr : Z);
        }
        r.v[0] = make_float4(t[0], t[1], t[2], t[3]);
        r.v[1] = make_float4(t[4], t[5], t[6], t[7]);
    }
}

// ---- fast operand fetch (vectorised variants; see gemm_common.h "fast operand fetch") -----------------------------------------
// Loads go through a buffer descriptor (the operand's rows of this block at k = 0) + a scalar k offset + per-thread byte
// offsets computed once, rows clamped instead of zero-filled.  A partial last k-tile (K % 16 == 8) is fetched as the last
// 16 columns of the k range and the half that repeats the previous tile is zeroed on its way into LDS (`zero_lo`).
struct B3Offs { unsigned o[2]; };
template <bool KC>
__device__ __forceinline__ void b3_offsets(B3Offs& t, int64_t ld, int64_t R, int64_t r0, int tid) {
    if constexpr (KC) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int idx = tid + NT * p;
            const int64_t rc = min((int64_t)(idx >> 2), R - 1 - r0);
            t.o[p] = (unsigned)((rc * ld + (idx & 3) * 4) * 4);
        }
    } else {
        const int64_t rc = min((int64_t)(tid & 127), R - 1 - r0);
        t.o[0] = (unsigned)(((int64_t)(8 * (tid >> 7)) * ld + rc) * 4);
        t.o[1] = 0u;
    }
}
// koff = byte offset of the tile's first k (4k for a k-major operand, 4k*ld for an outer-major one); ldb4 = 4*ld
template <bool KC>
__device__ __forceinline__ void b3_fetch_fast(StageRegs& r, __amdgpu_buffer_rsrc_t rs, unsigned koff, unsigned ldb4, const B3Offs& t) {
    if constexpr (KC) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const u32x4_ v = __builtin_amdgcn_raw_buffer_load_b128(rs, t.o[p], koff, 0);
            r.v[p].x = __uint_as_float(v.x); r.v[p].y = __uint_as_float(v.y); r.v[p].z = __uint_as_float(v.z); r.v[p].w = __uint_as_float(v.w);
        }
    } else {
        float f[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, t.o[0], koff + (unsigned)j * ldb4, 0));
        r.v[0].x = f[0]; r.v[0].y = f[1]; r.v[0].z = f[2]; r.v[0].w = f[3];
        r.v[1].x = f[4]; r.v[1].y = f[5]; r.v[1].z = f[6]; r.v[1].w = f[7];
    }
}
// zero the k-half [0, 8) of a staged tile (the shifted partial last tile repeats it from the tile before)
template <bool KC>
__device__ __forceinline__ void b3_zero_lo(StageRegs& r, int tid) {
    if constexpr (KC) {
        if ((tid & 3) < 2) { r.v[0] = make_float4(0.f, 0.f, 0.f, 0.f); r.v[1] = make_float4(0.f, 0.f, 0.f, 0.f); }
    } else {
        if ((tid >> 7) == 0) { r.v[0] = make_float4(0.f, 0.f, 0.f, 0.f); r.v[1] = make_float4(0.f, 0.f, 0.f, 0.f); }
    }
}

template <bool KC>
__device__ __forceinline__ void b3_commit(const StageRegs& r, unsigned char* __restrict__ S, int tid) {
    if constexpr (KC) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int idx = tid + NT * p;
            const int rr = idx >> 2, k4 = (idx & 3) * 4;
            unsigned h[4], m[4], l[4];
            split3(r.v[p].x, h[0], m[0], l[0]);
            split3(r.v[p].y, h[1], m[1], l[1]);
            split3(r.v[p].z, h[2], m[2], l[2]);
            split3(r.v[p].w, h[3], m[3], l[3]);
            unsigned char* dst = S + rr * B3_RS + k4 * 2;
            *reinterpret_cast<uint2*>(dst) = make_uint2(pack_hi16(h[0], h[1]), pack_hi16(h[2], h[3]));
            *reinterpret_cast<uint2*>(dst + B3_PLANE) = make_uint2(pack_hi16(m[0], m[1]), pack_hi16(m[2], m[3]));
            *reinterpret_cast<uint2*>(dst + 2 * B3_PLANE) = make_uint2(pack_hi16(l[0], l[1]), pack_hi16(l[2], l[3]));
        }
    } else {
        const float t[8] = {r.v[0].x, r.v[0].y, r.v[0].z, r.v[0].w, r.v[1].x, r.v[1].y, r.v[1].z, r.v[1].w};
        unsigned h[8], m[8], l[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) split3(t[j], h[j], m[j], l[j]);
        unsigned char* dst = S + (tid & 127) * B3_RS + (tid >> 7) * 16;
        *reinterpret_cast<uint4*>(dst) = make_uint4(pack_hi16(h[0], h[1]), pack_hi16(h[2], h[3]), pack_hi16(h[4], h[5]), pack_hi16(h[6], h[7]));
        *reinterpret_cast<uint4*>(dst + B3_PLANE) = make_uint4(pack_hi16(m[0], m[1]), pack_hi16(m[2], m[3]), pack_hi16(m[4], m[5]), pack_hi16(m[6], m[7]));
        *reinterpret_cast<uint4*>(dst + 2 * B3_PLANE) = make_uint4(pack_hi16(l[0], l[1]), pack_hi16(l[2], l[3]), pack_hi16(l[4], l[5]), pack_hi16(l[6], l[7]));
    }
}

__device__ __forceinline__ bf16x8 b3_frag(const unsigned char* __restrict__ S, int row, int lh) {
    const uint4 v = *reinterpret_cast<const uint4*>(S + row * B3_RS + lh * 16);
    return __builtin_bit_cast(bf16x8, v);
}

#define B3_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

// CS (outer-major A only): every thread also sums the A elements it stages (8 k of ONE row) -> asum[m] = sum_k A[m,k].
template <bool AKC, bool BKC, bool VEC, bool CS>
__global__ __launch_bounds__(NT, 2) void gemm_bf3_kernel(const GemmParams p) {
    static_assert(!CS || !AKC, "row sums of A are only implemented for an outer-major A");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, lh = lane >> 5;

    // ---- block id -> (split, tile_m, tile_n): the XCD-aware grouped order of gemm.hip -------------------------------
    const int nwg = gridDim.x;
    int L;
    {
        const int b = blockIdx.x, xcd = b & 7, q = nwg >> 3, r = nwg & 7;
        L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
    }
    const int tiles = p.tiles_m * p.tiles_n;
    const int split = L / tiles;
    const int t = L - split * tiles;
    constexpr int GM = 8;
    const int in_group = GM * p.tiles_n;
    const int grp = t / in_group;
    const int first_m = grp * GM;
    const int gsz = min(p.tiles_m - first_m, GM);
    const int tm = first_m + (t % in_group) % gsz;
    const int tn = (t % in_group) / gsz;

    const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;
    const int64_t kbeg = (int64_t)split * p.k_per_split;
    const int64_t kend = min(p.K, kbeg + p.k_per_split);
    const int bz1 = blockIdx.y / p.batch2, bz2 = blockIdx.y - bz1 * p.batch2;
    const float* __restrict__ A = p.A + (int64_t)bz1 * p.sA + (int64_t)bz2 * p.sA2;
    const float* __restrict__ B = p.B + (int64_t)bz1 * p.sB + (int64_t)bz2 * p.sB2;
    const int64_t c_off = (int64_t)bz1 * p.sC + (int64_t)bz2 * p.sC2;
    const float* __restrict__ Z = p.zeros;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // Two register sets per operand: tile kt+2 is fetched while tile kt is multiplied and tile kt+1 (fetched one
    // iteration earlier) is split and committed to the other LDS stage -- a global load gets a whole iteration to land.
    StageRegs ra0, rb0, ra1, rb1;
    float rowsum = 0.f;
    auto rsum = [](const StageRegs& r) { return (r.v[0].x + r.v[0].y) + (r.v[0].z + r.v[0].w) + (r.v[1].x + r.v[1].y) + (r.v[1].z + r.v[1].w); };
    const int64_t klen = kend - kbeg;
    int nk;
    // VEC: tiles 0 .. nfull-1 are whole; tile nfull (if any) is the shifted partial tile.  Host guarantees K % 8 == 0, K >= 16.
    const int nfull = (int)(klen / B3_BK);
    const bool has_tail = VEC && (klen - (int64_t)nfull * B3_BK) != 0;
    [[maybe_unused]] B3Offs offa, offb;
    [[maybe_unused]] __amdgpu_buffer_rsrc_t rsa, rsb;
    [[maybe_unused]] unsigned ua = 0, ub = 0, la4 = 0, lb4 = 0;
    if constexpr (VEC) {
        nk = nfull + (has_tail ? 1 : 0);
        b3_offsets<AKC>(offa, p.lda, p.M, m0, tid);
        b3_offsets<BKC>(offb, p.ldb, p.N, n0, tid);
        rsa = operand_rsrc(AKC ? A + m0 * p.lda : A + m0);
        rsb = operand_rsrc(BKC ? B + n0 * p.ldb : B + n0);
        la4 = (unsigned)(4 * p.lda); lb4 = (unsigned)(4 * p.ldb);
        ua = AKC ? 4u : la4; ub = BKC ? 4u : lb4;
    } else {
        nk = (int)((klen + B3_BK - 1) / B3_BK);
    }
    // first k of tile t (VEC): whole tiles in order, then the shifted partial tile; past the end: any valid tile (never used)
    auto tile_k = [&](int t) -> unsigned { return (unsigned)(t < nfull ? kbeg + (int64_t)t * B3_BK : (has_tail ? kend - B3_BK : kbeg)); };
    auto fetch = [&](StageRegs& fa, StageRegs& fb, int t) {
        if constexpr (VEC) {
            const unsigned k = tile_k(t);
            b3_fetch_fast<AKC>(fa, rsa, k * ua, la4, offa);
            b3_fetch_fast<BKC>(fb, rsb, k * ub, lb4, offb);
        } else {
            const int64_t k = kbeg + (int64_t)t * B3_BK;
            b3_fetch<AKC, VEC>(fa, A, p.lda, p.M, kend, m0, k, tid, t < nk, Z);
            b3_fetch<BKC, VEC>(fb, B, p.ldb, p.N, kend, n0, k, tid, t < nk, Z);
        }
    };
    // tile t goes from registers to LDS stage S: zero the repeated half of the shifted partial tile, add the row sums
    auto commit = [&](StageRegs& ca, StageRegs& cb, int t, unsigned char* S) {
        if constexpr (VEC) {
            if (has_tail && t == nfull) { b3_zero_lo<AKC>(ca, tid); b3_zero_lo<BKC>(cb, tid); }
            if constexpr (CS) { if (t < nk) rowsum += rsum(ca); }
        } else {
            if constexpr (CS) rowsum += rsum(ca);          // zeros past the end
        }
        b3_commit<AKC>(ca, S, tid);
        b3_commit<BKC>(cb, S + B3_OPERAND, tid);
    };
    fetch(ra0, rb0, 0);
    fetch(ra1, rb1, 1);
    commit(ra0, rb0, 0, smem_b);
    __syncthreads();

    int cur = 0;
    // one k-tile: fetch tile kt+2 into (fa, fb) [free: its tile was committed an iteration ago], multiply tile kt out of
    // LDS stage `cur`, commit tile kt+1 from (ca, cb) into the other stage
    auto step = [&](int kt, StageRegs& fa, StageRegs& fb, StageRegs& ca, StageRegs& cb) {
        fetch(fa, fb, kt + 2);
        const unsigned char* As = smem_b + cur * B3_STAGE;
        const unsigned char* Bs = As + B3_OPERAND;
        bf16x8 a[2][3], b[2][3];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                a[i][pl] = b3_frag(As + pl * B3_PLANE, wm * 64 + i * 32 + l31, lh);
                b[i][pl] = b3_frag(Bs + pl * B3_PLANE, wn * 64 + i * 32 + l31, lh);
            }
        // six products per accumulator, smallest first; the four accumulators alternate so that dependent MFMAs are 4 apart
#define B3_TERM(PA, PB)                                                       \
        _Pragma("unroll") for (int i = 0; i < 2; ++i)                          \
            _Pragma("unroll") for (int n = 0; n < 2; ++n) acc[i][n] = B3_MFMA(a[i][PA], b[n][PB], acc[i][n]);
        B3_TERM(2, 0)
        B3_TERM(0, 2)
        B3_TERM(1, 1)
        B3_TERM(1, 0)
        B3_TERM(0, 1)
        B3_TERM(0, 0)
#undef B3_TERM
        commit(ca, cb, kt + 1, smem_b + (cur ^ 1) * B3_STAGE);
        __syncthreads();
        cur ^= 1;
    };
    int kt = 0;
    for (; kt + 1 < nk; kt += 2) {
        step(kt, ra0, rb0, ra1, rb1);
        step(kt + 1, ra1, rb1, ra0, rb0);
    }
    if (kt < nk) step(kt, ra0, rb0, ra1, rb1);

    float* smem = reinterpret_cast<float*>(smem_b);
    if constexpr (CS) {
        // thread (row = tid & 127, kh = tid >> 7) holds its half of the row sum
        if (p.asum && tn == 0) {
            smem[tid] = rowsum;
            __syncthreads();
            if (tid < 128 && m0 + tid < p.M) {
                const float s = smem[tid] + smem[tid + 128];
                (p.splitk > 1 ? p.asum_slab + (int64_t)split * p.M : p.asum)[m0 + tid] = s;
            }
            __syncthreads();
        }
    }
    gemm_epilogue<false>(acc, p, smem, make_float4(0.f, 0.f, 0.f, 0.f), tid, wave, lane, wm, wn, l31, lh, m0, n0, tn, split, c_off);
}

template <bool AKC, bool BKC, bool VEC, bool CS>
static int launch_bf3(const GemmParams& p, int64_t batch, hipStream_t st) {
    constexpr size_t lds = 2 * B3_STAGE;
    static_assert(lds >= 4 * 32 * 68 * sizeof(float), "the epilogue's per-wave transposition area must fit");
    auto kern = gemm_bf3_kernel<AKC, BKC, VEC, CS>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return hip_status(e, "hipFuncSetAttribute(gemm_bf3)");
        attr_set = true;
    }
    dim3 grid((unsigned)(p.tiles_m * p.tiles_n * p.splitk), (unsigned)batch);
    hipLaunchKernelGGL(kern, grid, dim3(NT), lds, st, p);
    NNHIP_LAUNCH_CHECK("gemm_bf3_kernel");
    return 0;
}

// called by gemm_f32_ex (gemm.hip) with a fully planned GemmParams when the split-bf16 mode is on
int gemm_bf3_launch(const GemmParams& p, bool a_kmajor, bool b_kmajor, bool vec, bool cs, int64_t batch, hipStream_t st) {
#define B3_CASE(AK, BKM)                                                                        \
    (vec ? launch_bf3<AK, BKM, true, false>(p, batch, st) : launch_bf3<AK, BKM, false, false>(p, batch, st))
    if (cs) {
        if (b_kmajor) return vec ? launch_bf3<false, true, true, true>(p, batch, st) : launch_bf3<false, true, false, true>(p, batch, st);
        return vec ? launch_bf3<false, false, true, true>(p, batch, st) : launch_bf3<false, false, false, true>(p, batch, st);
    }
    if (a_kmajor && b_kmajor) return B3_CASE(true, true);
    if (a_kmajor) return B3_CASE(true, false);
    if (b_kmajor) return B3_CASE(false, true);
    return B3_CASE(false, false);
#undef B3_CASE
}

}  // namespace nnhip
