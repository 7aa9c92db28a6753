// gemm_bf3.hip -- fp32 GEMM on the bf16 matrix cores: every fp32 operand is split EXACTLY into three bf16 pieces
// (x = hi + mid + lo, 8 + 8 + 8 significand bits, by truncation: no rounding anywhere in the split) and the product is
// formed from the six largest of the nine piece products,
//     a*b ~= hi*hi + hi*mid + mid*hi + mid*mid + hi*lo + lo*hi          (dropped: mid*lo, lo*mid, lo*lo: < 2^-21 |a b| worst case, ~2^-23 typical)
// each an exact bf16 x bf16 product accumulated in fp32 by v_mfma_f32_32x32x16_bf16.  The result carries a relative
// error of ~2^-23 per product -- the same order as fp32 rounding itself (an fp32 MFMA / sgemm dot product of length K
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
//     plane[128 rows][16 k], 32-byte rows with the 16-byte halves of every second group of 8 rows swapped (conflict-free
//     ds_read_b128 / ds_write_b128 / ds_write_b64) = 24 KB per stage;
//   * a k-major operand is fetched as float4 (4 k of a row), split, and stored as three 8-byte pieces; an outer-major
//     operand is fetched as 8 coalesced dwords (8 k of one row: lanes <-> rows), split and stored as three 16-byte
//     pieces -- the transposition to k-major happens in the fetch pattern, not in LDS;
//   * the MFMA operand of lane (l31, lh) is 8 consecutive k of row l31: one ds_read_b128 per plane.
#include <stdlib.h>

#include "gemm_common.h"

namespace nnhip {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int B3_BK = 16;
#ifndef B3_DEPTH
#define B3_DEPTH 2
#endif
constexpr int B3_RS = 32;                               // bytes per LDS row: 16 bf16, unpadded; the two 16-byte halves of rows 8-15 (mod 16) are swapped
// (round 5: the 48-byte padded rows kept ds_read_b128 conflict-free but the 8-byte stores of a k-major operand hit 2-way conflicts --
// SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE 0.33 in profiles/r04*_gemm_pmc_bf3.txt.  Swizzled 32-byte rows: 16 lanes of a b128 read or
// store touch 16 different 16-byte bank groups, and the 32 lanes of a b64 store pass cover 256 contiguous bytes.)
__device__ __forceinline__ int b3_off(int row, int byte) { return row * B3_RS + (byte ^ ((row & 8) << 1)); }
constexpr int B3_PLANE = 128 * B3_RS;                   // one plane of one operand
constexpr int B3_OPERAND = 3 * B3_PLANE;
constexpr int B3_STAGE = 2 * B3_OPERAND;                // 24576 bytes

// x = hi + mid + lo exactly; each piece has its low 16 bits clear (a bf16 in the high half)
__device__ __forceinline__ void split3(float x, unsigned& hi, unsigned& mid, unsigned& lo) {
    hi = __float_as_uint(x) & 0xFFFF0000u;
    const float r1 = x - __uint_as_float(hi);
    mid = __float_as_uint(r1) & 0xFFFF0000u;
    lo = __float_as_uint(r1 - __uint_as_float(mid));   // <= 8 significant bits left: already a bf16
}
// {a.high16, b.high16} -> one dword (a in the low half)
__device__ __forceinline__ unsigned pack_hi16(unsigned a, unsigned b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }
// Two elements at once, straight to the packed dwords of the three planes: 4 v_and + 2 v_pk_add_f32 + 3 v_perm = 9 VALU per
// pair (split3 + packing: 11).  The split is issue-bound work next to the MFMAs (see the step schedule), so it counts.
typedef float f32x2_ __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split3_pair(float x0, float x1, unsigned& hi, unsigned& mid, unsigned& lo) {
#ifdef B3_ABL_NOSPLIT      // diagnostic build (wrong values): no split arithmetic -- the ceiling of a kernel fed pre-split planes
    hi = __float_as_uint(x0); mid = __float_as_uint(x1); lo = __float_as_uint(x0) ^ __float_as_uint(x1);
    return;
#endif
    const unsigned h0 = __float_as_uint(x0) & 0xFFFF0000u, h1 = __float_as_uint(x1) & 0xFFFF0000u;
    const f32x2_ r1 = f32x2_{x0, x1} - f32x2_{__uint_as_float(h0), __uint_as_float(h1)};
    const unsigned m0 = __float_as_uint(r1.x) & 0xFFFF0000u, m1 = __float_as_uint(r1.y) & 0xFFFF0000u;
    const f32x2_ r2 = r1 - f32x2_{__uint_as_float(m0), __uint_as_float(m1)};
    hi = pack_hi16(h0, h1);
    mid = pack_hi16(m0, m1);
    lo = pack_hi16(__float_as_uint(r2.x), __float_as_uint(r2.y));
}

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
            unsigned h[2], m[2], l[2];
            split3_pair(r.v[p].x, r.v[p].y, h[0], m[0], l[0]);
            split3_pair(r.v[p].z, r.v[p].w, h[1], m[1], l[1]);
            unsigned char* dst = S + b3_off(rr, k4 * 2);
            *reinterpret_cast<uint2*>(dst) = make_uint2(h[0], h[1]);
            *reinterpret_cast<uint2*>(dst + B3_PLANE) = make_uint2(m[0], m[1]);
            *reinterpret_cast<uint2*>(dst + 2 * B3_PLANE) = make_uint2(l[0], l[1]);
        }
    } else {
        unsigned h[4], m[4], l[4];
        split3_pair(r.v[0].x, r.v[0].y, h[0], m[0], l[0]);
        split3_pair(r.v[0].z, r.v[0].w, h[1], m[1], l[1]);
        split3_pair(r.v[1].x, r.v[1].y, h[2], m[2], l[2]);
        split3_pair(r.v[1].z, r.v[1].w, h[3], m[3], l[3]);
        unsigned char* dst = S + b3_off(tid & 127, (tid >> 7) * 16);
        *reinterpret_cast<uint4*>(dst) = make_uint4(h[0], h[1], h[2], h[3]);
        *reinterpret_cast<uint4*>(dst + B3_PLANE) = make_uint4(m[0], m[1], m[2], m[3]);
        *reinterpret_cast<uint4*>(dst + 2 * B3_PLANE) = make_uint4(l[0], l[1], l[2], l[3]);
    }
}

__device__ __forceinline__ bf16x8 b3_frag(const unsigned char* __restrict__ S, int row, int lh) {
    const uint4 v = *reinterpret_cast<const uint4*>(S + b3_off(row, lh * 16));
    return __builtin_bit_cast(bf16x8, v);
}

#define B3_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

// ---- pieces of the hand-ordered step (b3_step_fast) -----------------------------------------------------------------------
// element e (0..7) of a staged tile, as a reference
__device__ __forceinline__ float& sr_elem(StageRegs& r, int e) {
    float4& v = r.v[e >> 2];
    return (e & 3) == 0 ? v.x : (e & 3) == 1 ? v.y : (e & 3) == 2 ? v.z : v.w;
}
__device__ __forceinline__ float sr_get(const StageRegs& r, int e) {
    const float4& v = r.v[e >> 2];
    return (e & 3) == 0 ? v.x : (e & 3) == 1 ? v.y : (e & 3) == 2 ? v.z : v.w;
}
// the idx-th global load of one operand's tile: k-major 2 x b128 (idx 0, 1), outer-major 8 x b32 (idx 0..7)
template <bool KC>
__device__ __forceinline__ void b3_load_one(StageRegs& r, __amdgpu_buffer_rsrc_t rs, unsigned koff, unsigned ldb4, const B3Offs& t, int idx) {
    if constexpr (KC) {
        const u32x4_ v = __builtin_amdgcn_raw_buffer_load_b128(rs, t.o[idx], koff, 0);
        r.v[idx].x = __uint_as_float(v.x); r.v[idx].y = __uint_as_float(v.y); r.v[idx].z = __uint_as_float(v.z); r.v[idx].w = __uint_as_float(v.w);
    } else {
        sr_elem(r, idx) = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, t.o[0], koff + (unsigned)idx * ldb4, 0));
    }
}
// split of one pair in two halves (issue slots): A = hi / first remainder / mid, B = second remainder and the three packs
struct PairSplit { f32x2_ r1; unsigned h0, h1, m0, m1; };
__device__ __forceinline__ void split_pair_a(PairSplit& s, float x0, float x1) {
#ifdef B3_ABL_NOSPLIT
    s.h0 = __float_as_uint(x0); s.h1 = __float_as_uint(x1); s.m0 = s.h1; s.m1 = s.h0; s.r1 = f32x2_{x0, x1};
    return;
#endif
    s.h0 = __float_as_uint(x0) & 0xFFFF0000u; s.h1 = __float_as_uint(x1) & 0xFFFF0000u;
    s.r1 = f32x2_{x0, x1} - f32x2_{__uint_as_float(s.h0), __uint_as_float(s.h1)};
    s.m0 = __float_as_uint(s.r1.x) & 0xFFFF0000u; s.m1 = __float_as_uint(s.r1.y) & 0xFFFF0000u;
}
__device__ __forceinline__ void split_pair_b(const PairSplit& s, unsigned& hi, unsigned& mid, unsigned& lo) {
#ifdef B3_ABL_NOSPLIT
    hi = s.h0; mid = s.h1; lo = s.m0 ^ s.h0;
    return;
#endif
    const f32x2_ r2 = s.r1 - f32x2_{__uint_as_float(s.m0), __uint_as_float(s.m1)};
    hi = pack_hi16(s.h0, s.h1);
    mid = pack_hi16(s.m0, s.m1);
    lo = pack_hi16(__float_as_uint(r2.x), __float_as_uint(r2.y));
}
// LDS store `w` (0..2 = plane) of group `g` of one operand: k-major groups are (p = g: pairs 2g, 2g+1 -> 8 bytes per plane),
// an outer-major operand is one group (pairs 0..3 -> 16 bytes per plane)
template <bool KC>
__device__ __forceinline__ void b3_store_one(unsigned char* __restrict__ S, int tid, int g, int w, const unsigned (&H)[4], const unsigned (&M)[4],
                                             const unsigned (&L)[4]) {
    const unsigned (&P)[4] = w == 0 ? H : (w == 1 ? M : L);
    if constexpr (KC) {
        const int idx = tid + NT * g;
        unsigned char* dst = S + b3_off(idx >> 2, (idx & 3) * 8) + w * B3_PLANE;
        *reinterpret_cast<uint2*>(dst) = make_uint2(P[2 * g], P[2 * g + 1]);
    } else {
        unsigned char* dst = S + b3_off(tid & 127, (tid >> 7) * 16) + w * B3_PLANE;
        *reinterpret_cast<uint4*>(dst) = make_uint4(P[0], P[1], P[2], P[3]);
    }
}

// One whole k-tile of the vectorised variants, hand-ordered: 24 issue slots, each = one MFMA + a few other instructions,
// separated by scheduling barriers so the compiler keeps them there.
//   slot q:  product q  |  q < 8: fragment read 4+q  |  q < NVM: global load q of tile kt+D
//            |  2 <= q < 18: half a pair of the split of tile kt+1  |  LDS stores of a finished group
// A 32x32x16 bf16 MFMA holds the matrix pipe for 32 cycles (~8 issue slots of which ~5 can carry other work,
// MI355X_MICROARCH.md): the step's ~100 non-MFMA instructions fit under its 24 MFMAs only if they are PLACED between them.
// Left to the compiler (round 2, and a sched_group_barrier pipeline tried first in round 3) the order was: reads, 24 MFMAs
// back to back, THEN the whole split with the matrix pipe idle -> 0.47 of the bf16 MFMA peak.
template <bool AKC, bool BKC, bool CS>
__device__ __forceinline__ void b3_step_fast(f32x16 (&acc)[2][2], StageRegs& fa, StageRegs& fb, const StageRegs& ca, const StageRegs& cb,
                                             unsigned char* __restrict__ smem_b, int cur, __amdgpu_buffer_rsrc_t rsa,
                                             __amdgpu_buffer_rsrc_t rsb, unsigned koa, unsigned kob, unsigned la4, unsigned lb4,
                                             const B3Offs& offa, const B3Offs& offb, int tid, int wm, int wn, int l31, int lh,
                                             float& rowsum) {
    const unsigned char* As = smem_b + cur * B3_STAGE;
    const unsigned char* Bs = As + B3_OPERAND;
    unsigned char* Sa = smem_b + (cur ^ 1) * B3_STAGE;
    unsigned char* Sb = Sa + B3_OPERAND;
    constexpr int NLA = AKC ? 2 : 8, NLB = BKC ? 2 : 8;            // global loads per operand
    // fragments in consumption order: products 0-3 need (a lo, b hi), 4-7 (a hi, b lo), 8-11 (a mid, b mid)
    bf16x8 a[2][3], b[2][3];
    auto read_frag = [&](int f) {
        constexpr int pa[3] = {2, 0, 1}, pb[3] = {0, 2, 1};
        const int q = f >> 2, j = f & 3;
        if (j < 2) a[j][pa[q]] = b3_frag(As + pa[q] * B3_PLANE, wm * 64 + j * 32 + l31, lh);
        else b[j - 2][pb[q]] = b3_frag(Bs + pb[q] * B3_PLANE, wn * 64 + (j - 2) * 32 + l31, lh);
    };
#pragma unroll
    for (int f = 0; f < 4; ++f) read_frag(f);
    PairSplit ps[8];
    unsigned HA[4], MA[4], LA[4], HB[4], MB[4], LB[4];
    constexpr int ta[6] = {2, 0, 1, 1, 0, 0}, tb[6] = {0, 2, 1, 0, 1, 0};   // (piece of a, piece of b) per term, smallest first
#pragma unroll
    for (int q = 0; q < 24; ++q) {
        const int term = q >> 2, i = (q >> 1) & 1, n = q & 1;
        acc[i][n] = B3_MFMA(a[i][ta[term]], b[n][tb[term]], acc[i][n]);
        if (q < 8) read_frag(4 + q);
        if (q < NLA) b3_load_one<AKC>(fa, rsa, koa, la4, offa, q);
        else if (q < NLA + NLB) b3_load_one<BKC>(fb, rsb, kob, lb4, offb, q - NLA);
        if (q >= 2 && q < 18) {                                   // pair j of 8 (0-3: A, 4-7: B): half A at even, half B at odd slots
            const int j = (q - 2) >> 1;
            const StageRegs& src = j < 4 ? ca : cb;
            const int e = (j & 3) * 2;
            if (((q - 2) & 1) == 0) split_pair_a(ps[j], sr_get(src, e), sr_get(src, e + 1));
            else if (j < 4) split_pair_b(ps[j], HA[j], MA[j], LA[j]);
            else split_pair_b(ps[j], HB[j - 4], MB[j - 4], LB[j - 4]);
        }
        // LDS stores, one per slot: a k-major operand's group g (pairs 2g, 2g+1) is complete after slot 5 + 4g (+8 for B),
        // an outer-major operand (one group of four pairs) after slot 9 (17 for B)
        if constexpr (AKC) {
            if (q >= 6 && q < 9) b3_store_one<true>(Sa, tid, 0, q - 6, HA, MA, LA);
            if (q >= 10 && q < 13) b3_store_one<true>(Sa, tid, 1, q - 10, HA, MA, LA);
        } else {
            if (q >= 10 && q < 13) b3_store_one<false>(Sa, tid, 0, q - 10, HA, MA, LA);
        }
        if constexpr (BKC) {
            if (q >= 14 && q < 17) b3_store_one<true>(Sb, tid, 0, q - 14, HB, MB, LB);
            if (q >= 18 && q < 21) b3_store_one<true>(Sb, tid, 1, q - 18, HB, MB, LB);
        } else {
            if (q >= 18 && q < 21) b3_store_one<false>(Sb, tid, 0, q - 18, HB, MB, LB);
        }
        if constexpr (CS) {
            if (q == 21) rowsum += (ca.v[0].x + ca.v[0].y) + (ca.v[0].z + ca.v[0].w);
            if (q == 22) rowsum += (ca.v[1].x + ca.v[1].y) + (ca.v[1].z + ca.v[1].w);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// CS (outer-major A only): every thread also sums the A elements it stages (8 k of ONE row) -> asum[m] = sum_k A[m,k].
// One block's work: logical block L of problem p (the XCD-aware grouped order of gemm.hip over L), batch index `by`.
template <bool AKC, bool BKC, bool VEC, bool CS>
__device__ __forceinline__ void gemm_bf3_block(const GemmParams& p, const int L, const int by, unsigned char* __restrict__ smem_b) {
    static_assert(!CS || !AKC, "row sums of A are only implemented for an outer-major A");
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, lh = lane >> 5;

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
    const int bz1 = by / p.batch2, bz2 = by - bz1 * p.batch2;
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

    // D register sets per operand (tile t lives in set t % D): tile kt+D is fetched while tile kt is multiplied and tile
    // kt+1 (fetched D-1 iterations earlier) is split and committed to the other LDS stage.  An iteration of this kernel is
    // only 24 MFMAs = 768 matrix-pipe cycles (0.32 us), far less than an L2 round trip: with round 2's two sets (a load
    // had ONE iteration to land) both resident blocks of a CU sat in s_waitcnt vmcnt for half of the time -- the kernel
    // ran at 0.47 of the bf16 MFMA peak whatever the tile order.  B3_DEPTH sets give a load D-1 iterations.
    constexpr int D = VEC ? B3_DEPTH : 2;          // (the scalar-load variants have no registers to spare)
    StageRegs ra[D], rb[D];
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
#pragma unroll
    for (int j = 0; j < D; ++j) fetch(ra[j], rb[j], j);
    commit(ra[0], rb[0], 0, smem_b);
    __syncthreads();

    int cur = 0;
    // one k-tile: fetch tile kt+D into (fa, fb) [free: its tile kt was committed an iteration ago], multiply tile kt out of
    // LDS stage `cur`, commit tile kt+1 from (ca, cb) into the other stage.
    // FAST (vectorised variants, whole tiles): the body is ONE basic block whose issue order is pinned.  Round 2's step left
    // the order to the compiler: 12 LDS reads, 24 MFMAs back to back (the last four chained on one accumulator), THEN the
    // ~110 split / pack / LDS-store instructions of the next tile with the matrix pipe idle, then the barrier -- 0.47 of the
    // bf16 MFMA peak, and unmoved by prefetch depth (measured, depth 2..8): the wave was issue-bound, not latency-bound.  A
    // 32x32x16 bf16 MFMA occupies its pipe for 32 cycles = ~8 issue slots, of which ~5 can carry other instructions
    // (MI355X_MICROARCH.md): the ~100 non-MFMA instructions of a step fit between its 24 MFMAs if they are PUT there.
    auto mma = [&](const bf16x8 (&a)[2][3], const bf16x8 (&b)[2][3]) {
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
    };
    auto frags = [&](bf16x8 (&a)[2][3], bf16x8 (&b)[2][3]) {
        const unsigned char* As = smem_b + cur * B3_STAGE;
        const unsigned char* Bs = As + B3_OPERAND;
        // in the order the products above consume them: (a lo, b hi), (a hi, b lo), (a mid, b mid)
        constexpr int pa[3] = {2, 0, 1}, pb[3] = {0, 2, 1};
#pragma unroll
        for (int q = 0; q < 3; ++q) {
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i][pa[q]] = b3_frag(As + pa[q] * B3_PLANE, wm * 64 + i * 32 + l31, lh);
#pragma unroll
            for (int i = 0; i < 2; ++i) b[i][pb[q]] = b3_frag(Bs + pb[q] * B3_PLANE, wn * 64 + i * 32 + l31, lh);
        }
    };
    auto step_fast = [&](int kt, StageRegs& fa, StageRegs& fb, StageRegs& ca, StageRegs& cb) {
        if constexpr (VEC) {
            const int64_t kn = kbeg + (int64_t)(kt + D) * B3_BK;
            const unsigned k = (unsigned)(kn + B3_BK <= kend ? kn : kbeg);     // scalar; past the end: any whole tile (never used)
            b3_step_fast<AKC, BKC, CS>(acc, fa, fb, ca, cb, smem_b, cur, rsa, rsb, k * ua, k * ub, la4, lb4, offa, offb, tid, wm, wn,
                                       l31, lh, rowsum);
            __syncthreads();
            cur ^= 1;
        }
    };
    auto step = [&](int kt, StageRegs& fa, StageRegs& fb, StageRegs& ca, StageRegs& cb) {
        fetch(fa, fb, kt + D);
        bf16x8 a[2][3], b[2][3];
        frags(a, b);
        mma(a, b);
        commit(ca, cb, kt + 1, smem_b + (cur ^ 1) * B3_STAGE);
        __syncthreads();
        cur ^= 1;
    };
    // steps kt < nfast commit a WHOLE tile (kt + 1 < nfull) and take the hand-ordered body -- in a loop of their OWN: with
    // the two kinds of step as alternatives inside one loop body the accumulators got different registers on the two paths
    // and every step ended in 32 v_mov_b64 copies behind an s_nop for the last MFMA's result.
    const int nfast = VEC ? (nfull > 0 ? nfull - 1 : 0) : 0;
    int kt = 0;
    if constexpr (VEC) {
        for (; kt + D <= nfast; kt += D) {
#pragma unroll
            for (int j = 0; j < D; ++j) step_fast(kt + j, ra[j], rb[j], ra[(j + 1) % D], rb[(j + 1) % D]);
        }
    }
    for (; kt < nk; kt += D) {                      // the last steps (the shifted partial tile, nothing left to fetch) and every
#pragma unroll                                      // step of the scalar-load variants
        for (int j = 0; j < D; ++j)
            if (kt + j < nk) step(kt + j, ra[j], rb[j], ra[(j + 1) % D], rb[(j + 1) % D]);
    }

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
__global__ __launch_bounds__(NT, 2) void gemm_bf3_kernel(const GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
    gemm_bf3_block<AKC, BKC, VEC, CS>(p, xcd_order((int)blockIdx.x, (int)gridDim.x), (int)blockIdx.y, smem_b);
}

// the grouped parameter-gradient launch of gemm.hip (gemm_f32_group_kernel) in the split-bf16 mode: same grid layout, same jobs
__global__ __launch_bounds__(NT, 2) void gemm_bf3_group_kernel(const GemmGroup g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
    const int b = (int)blockIdx.x;
    const int L = g.by_job ? b : xcd_order(b, (int)gridDim.x);
    int i = 0;
#pragma unroll
    for (int j = 1; j < GEMM_GROUP_MAX; ++j)
        if (j < g.n && L >= g.start[j]) i = j;
    const int local = L - g.start[i];
    gemm_bf3_block<false, false, true, true>(g.p[i], g.by_job ? xcd_order(local, g.start[i + 1] - g.start[i]) : local, 0, smem_b);
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

int gemm_bf3_group_launch(const GemmGroup& g, int blocks, hipStream_t st) {
    constexpr size_t lds = 2 * B3_STAGE;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf3_group_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return hip_status(e, "hipFuncSetAttribute(gemm_bf3 group)");
        attr_set = true;
    }
    hipLaunchKernelGGL(gemm_bf3_group_kernel, dim3((unsigned)blocks), dim3(NT), lds, st, g);
    NNHIP_LAUNCH_CHECK("gemm_bf3_group_kernel");
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
