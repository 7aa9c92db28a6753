// gemm_pst.hip -- persistent variant of the exact-fp32 MFMA GEMM for grids of several generations of tiles
// (C = alpha A B + bias with A k-major: the Linear FORWARD layout with B = W k-major, neunet/nn/layers/linear.py:48-58, and the
// input-gradient layout dX = dO W with B = W outer-major, linear.py:17-24).
//
// Why: tools/gemm_prof.py shows the k-loop of gemm_f32_kernel at 100 % of the matrix pipe; what a K = 512 tile loses
// (11 %) is fixed cost per generation of 512 tiles -- all blocks store their C tiles together (~5 us for 33.5 MB) and then
// wait together for the first k-tile of their next tile (~2-3 us), with idle matrix pipes both times.  Here a block is
// persistent and its k-steps form ONE stream across its tiles:
//   * the last k-step of tile i fetches the first k-tile of tile i+1 (only the buffer descriptors change -- scalar work;
//     the per-thread offsets are tile-independent because rows past an operand's extent are cut off by the descriptor's
//     num_records instead of being clamped), so there is no prologue between tiles;
//   * the finished tile moves to a second accumulator set and its 64 dword stores per lane go out one per MFMA inside the
//     FIRST k-step of the next tile (buffer_store_dword with scalar row offsets: the only vector instructions added to that
//     step are the 64 alpha/bias FMAs), so there is no store phase either.
// Same arithmetic as gemm_f32_kernel (same k order, same fmaf chain, alpha/bias applied the same way): bit-identical C.
// EPI = 1 is the fused Linear->Swish epilogue (neunet/nn/experimental/linear_swish/linear_swish_cutlass.cu: D = swish(z),
// z = alpha A B^T + bias optionally stored as well): its two outputs double the store traffic, which gemm_f32_kernel pays as
// an exposed burst per generation; here the pending tile goes out one row half under each of the next tile's first TWO k-steps
// (all 128 stores per lane under one step saturate the CU's store path when both resident blocks flush together: measured no
// gain; one per MFMA over two steps: 16384x512->2048 with z, 0.278 -> 0.255 ms).
// EPI = 2 is the Swish-backward epilogue of the input gradient that feeds a fused Linear->Swish (C = (A B) * swish'(z), z read
// from a [M, ldc] tensor that C may alias -- linear_swish's in-place dZ contract): the pending tile goes out in quarters under
// the next tile's first FOUR k-steps, and the z values of a quarter are fetched one k-step before the step that stores it
// (quarter 0 under the tile's own last k-step), so neither their latency nor the stores are exposed.
// EPI = 3 is EPI 1 with the z output replaced by swish'(z) (ACT_SWISH_D: one sigmoid serves both outputs) and EPI = 4 is EPI 2 with
// the saved derivative as a plain multiplier (dact 3) -- round 6: the backward epilogue's sigmoid + polynomial (~64 matrix-pipe
// cycles per element, 6 % of a K = 512 tile) becomes one multiply.
// Conditions (gemm_pst_wanted): A k-major, 16-B aligned rows, K % 32 == 0, K >= 96 (128 with EPI 1 / 3, 160 with EPI 2 / 4),
// M % 128 == 0, no split-K / batch / addend, C and the operand windows addressable with 32-bit byte offsets, more tiles than
// resident slots.  Everything else takes gemm.hip.
#include <stdlib.h>

#include <type_traits>

#include "gemm_common.h"

namespace nnhip {

constexpr int PBK = 32;
using PTile = Tile<PBK, true>;
constexpr int PSTAGE = 2 * PTile::SIZE;                        // A tile + B tile, floats (an outer-major B tile is smaller)

struct PstParams {
    const float* A; const float* B; float* C; const float* bias;
    float* preact;                                             // EPI 1: z output [M, ldc] or null; EPI 2: the z INPUT
    int64_t M, N, K, lda, ldb, ldc;
    float alpha, beta;
    int tiles_m, tiles_n, total;
    int stagger;                                               // 1: the k-step under which a block flushes its pending tile depends on the block
};

// where the pending tile goes
struct PstStore {
    unsigned soff;                                             // byte offset of C[m0][n0]
    float b0, b1;                                              // bias of this lane's two columns (n = 0, 1)
    unsigned vo0, vo1;                                         // this lane's byte offset inside the tile for n = 0, 1; a column
                                                               // >= N gets an offset past num_records: the store is dropped by the
                                                               // bounds check, so the 64 stores need no exec masking / branches
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t pst_rsrc(const float* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, (int)bytes, 0x00020000);
}

// virtual block id -> tile (the XCD-aware grouped order of gemm.hip; total % 8 may be anything, gridDim.x % 8 == 0)
__device__ __forceinline__ void pst_tile(int vb, int total, int tiles_m, int tiles_n, int& tm, int& tn) {
    const int xcd = vb & 7, q = total >> 3, r = total & 7;
    const int L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (vb >> 3);
    constexpr int GM = 8;
    const int in_group = GM * tiles_n;
    const int grp = L / in_group;
    const int first_m = grp * GM;
    const int gsz = min(tiles_m - first_m, GM);
    tm = first_m + (L % in_group) % gsz;
    tn = (L % in_group) / gsz;
}

// The pending tile goes out in QUARTERS: quarter q = row half i = q >> 1 of the wave's 64 rows, accumulator registers
// e = 8 (q & 1) .. + 8 of both column halves (16 dwords per lane).  Row of register e in row half i (+ 4 lh + 64 wm: in vo);
// column n*32 + l31 (+ 64 wn: in vo).
__device__ __forceinline__ constexpr unsigned pst_row(int q, int j) {
    const int i = q >> 1, e = 8 * (q & 1) + j;
    return (unsigned)(i * 32 + (e & 3) + 8 * (e >> 2));
}

// EPI 2: fetch the swish' arguments of quarter q of the tile described by `s`
__device__ __forceinline__ void pst_load_z(float (&z)[2][8], int q, const PstStore& s, __amdgpu_buffer_rsrc_t rz, unsigned ldc4) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const unsigned so = s.soff + pst_row(q, j) * ldc4;
        z[0][j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rz, s.vo0, so, 0));   // (a dropped column reads 0)
        z[1][j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rz, s.vo1, so, 0));
    }
}

// the stores of quarter q of the pending tile
template <int EPI>
__device__ __forceinline__ void pst_flush_q(int q, const f32x16 (&pend)[2][2], const PstStore& ps, const float (&z)[2][8],
                                            __amdgpu_buffer_rsrc_t rc, __amdgpu_buffer_rsrc_t rz, unsigned ldc4, float alpha, float beta) {
    const int i = q >> 1;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int e = 8 * (q & 1) + j;
        const unsigned so = ps.soff + pst_row(q, j) * ldc4;
        float v0 = fmaf(alpha, pend[i][0][e], ps.b0), v1 = fmaf(alpha, pend[i][1][e], ps.b1);
        if constexpr (EPI == 1) {
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v0), rz, ps.vo0, so, 0);   // rz has 0 records without a z output
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v1), rz, ps.vo1, so, 0);
            v0 *= sigmoid_fast_(beta * v0);
            v1 *= sigmoid_fast_(beta * v1);
        }
        if constexpr (EPI == 3) {
            float d0, d1;
            swish_fwd_d_(v0, beta, v0, d0);
            swish_fwd_d_(v1, beta, v1, d1);
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(d0), rz, ps.vo0, so, 0);
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(d1), rz, ps.vo1, so, 0);
        }
        if constexpr (EPI == 2) {
            v0 *= swish_grad_(z[0][j], beta);
            v1 *= swish_grad_(z[1][j], beta);
        }
        if constexpr (EPI == 4) {
            v0 *= z[0][j];
            v1 *= z[1][j];
        }
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v0), rc, ps.vo0, so, 0);
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v1), rc, ps.vo1, so, 0);
    }
}

// one k-step: fetch the next k-tile (whatever tile it belongs to) into (fa, fb), multiply LDS stage `cur`, [flush stores of
// the pending tile, at most one per MFMA,] commit (fa, fb) to the other stage.  FLUSH: bit q = quarter q of the pending tile
// goes out in this step (EPI 2: with the z values in zf).  ZQ (EPI 2): 1 + the quarter whose z values this step fetches into
// zl -- of tile `zs` (quarter 0: the tile being finished) or of the pending tile (quarters 1-3).
// (A two-deep variant -- a second register set, the k-tile after next in flight -- was measured: no change at K = 512, level
// with gemm_f32_kernel at K = 4096; not kept.)
// EDGE (round 6): 1 = the tile's FIRST k-step -- its first MFMA per accumulator takes C = 0 (an inline constant) instead of an
// accumulator that 64 v_mov had to clear; 2 = the tile's LAST k-step -- its last MFMA per accumulator writes the PENDING set (D != C),
// so the finished tile changes registers as part of the MFMA that completes it instead of through 64 more v_mov.  128 vector
// instructions per tile and wave less, on lanes the MFMAs share; same products, same order: bit-identical.
template <int EPI, bool BKM, int FLUSH, int ZQ, int EDGE = 0>
__device__ __forceinline__ void pst_step(f32x16 (&acc)[2][2], float4 (&fa)[4], float4 (&fb)[4], float* __restrict__ smem, int cur,
                                         __amdgpu_buffer_rsrc_t rsa, __amdgpu_buffer_rsrc_t rsb, unsigned koffa, unsigned koffb,
                                         const unsigned (&offa)[4], const unsigned (&offb)[4], int tid, int wm, int wn, int l31,
                                         int lh, f32x16 (&pend)[2][2], const PstStore& ps, const PstStore& zs,
                                         const float (&zf)[2][8], float (&zl)[2][8], __amdgpu_buffer_rsrc_t rc,
                                         __amdgpu_buffer_rsrc_t rz, unsigned ldc4, float alpha, float beta) {
    constexpr bool ZIN = EPI == 2 || EPI == 4;                  // the epilogue reads a [M, ldc] operand (z / the saved derivative)
    constexpr bool TWO = EPI == 1 || EPI == 3;                  // the epilogue writes two outputs
    g2r_fast<PBK>(fa, rsa, koffa, offa);
    g2r_fast<PBK>(fb, rsb, koffb, offb);
    if constexpr (ZIN && ZQ == 1) pst_load_z(zl, 0, zs, rz, ldc4);
    if constexpr (ZIN && ZQ > 1) pst_load_z(zl, ZQ - 1, ps, rz, ldc4);
    const float* As = smem + cur * PSTAGE;
    const float* Bs = As + PTile::SIZE;
#pragma unroll
    for (int g = 0; g < PBK / 8; ++g) {
        float a[2][4], b[2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            frag<PBK, true>(a[i], As, wm * 64 + i * 32 + l31, g, lh);
            frag<PBK, BKM>(b[i], Bs, wn * 64 + i * 32 + l31, g, lh);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    if constexpr (EDGE == 1) {
                        if (g == 0 && j == 0) {
                            const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                            acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][j], b[n][j], zero, 0, 0, 0);
                            continue;
                        }
                    }
                    if constexpr (EDGE == 2) {
                        if (g == PBK / 8 - 1 && j == 3) {
                            pend[i][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][j], b[n][j], acc[i][n], 0, 0, 0);
                            continue;
                        }
                    }
                    acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][j], b[n][j], acc[i][n], 0, 0, 0);
                }
    }
    if constexpr ((FLUSH & 1) != 0) pst_flush_q<EPI>(0, pend, ps, zf, rc, rz, ldc4, alpha, beta);
    if constexpr ((FLUSH & 2) != 0) pst_flush_q<EPI>(1, pend, ps, zf, rc, rz, ldc4, alpha, beta);
    if constexpr ((FLUSH & 4) != 0) pst_flush_q<EPI>(2, pend, ps, zf, rc, rz, ldc4, alpha, beta);
    if constexpr ((FLUSH & 8) != 0) pst_flush_q<EPI>(3, pend, ps, zf, rc, rz, ldc4, alpha, beta);
    {
        float* Sn = smem + (cur ^ 1) * PSTAGE;
        r2s<PBK, true>(fa, Sn, tid);
        r2s<PBK, BKM>(fb, Sn + PTile::SIZE, tid);
    }
    // issue order: the tile's 8 loads (and the 16 z loads) under the first MFMAs, the 8 LDS stores under the last ones, the NW
    // stores one per MFMA under the last NW.  The epilogue arithmetic goes wherever the scheduler likes: fp32 MFMAs and VALU
    // share the lanes, it overlaps with nothing.
    constexpr bool ZLD = ZIN && ZQ != 0;
    constexpr int NQ = (FLUSH & 1) + ((FLUSH >> 1) & 1) + ((FLUSH >> 2) & 1) + ((FLUSH >> 3) & 1);
    constexpr int NW = NQ * (TWO ? 32 : 16);
    static_assert(NW <= 64, "at most one store per MFMA");
#pragma unroll
    for (int i = 0; i < 64; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (i < 8) __builtin_amdgcn_sched_group_barrier(0x020, ZLD ? 3 : 1, 0);
        if (i >= 64 - NW) __builtin_amdgcn_sched_group_barrier(0x040, 1, 0);
        if (i >= 56) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
    }
    __syncthreads();
}

template <int EPI, bool BKM>
__global__ __launch_bounds__(NT, 2) void gemm_pst_kernel(const PstParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, lh = lane >> 5;
    const int nk = (int)(p.K / PBK);
    const unsigned la4 = (unsigned)(4 * p.lda), lb4 = (unsigned)(4 * p.ldb), ldc4 = (unsigned)(4 * p.ldc);
    const unsigned kstep_b = BKM ? 128u : 32u * lb4;              // byte advance of one k-tile in B

    // tile-independent per-thread offsets of the 4 float4 a thread stages per operand tile: (row * ld + k4) * 4 for a k-major
    // operand, (k * ld + n4) * 4 for an outer-major B
    unsigned offa[4], offb[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int idx = tid + NT * q;
        offa[q] = (unsigned)(idx >> 3) * la4 + (unsigned)(idx & 7) * 16u;
        offb[q] = BKM ? (unsigned)(idx >> 3) * lb4 + (unsigned)(idx & 7) * 16u : (unsigned)(idx >> 5) * lb4 + (unsigned)(idx & 31) * 16u;
    }
    // descriptor of an operand's rows [r0, r0 + 128) clipped to R rows: loads of rows past the end return 0
    auto rs_of = [&](const float* P, int64_t ld, int64_t R, int64_t r0) {
        const int64_t rows = min((int64_t)128, R - r0);
        return pst_rsrc(P + r0 * ld, (unsigned)(rows * ld * 4));
    };
    // outer-major B ([K, ldb], the tile's 128 columns start at n0): everything from B[0][n0] to the end of row K-1, cut at
    // column N there; in the rows above, columns >= N of the last tile read the next row's first columns -- finite garbage
    // that only reaches columns >= N of C, which are never stored
    auto rs_of_b = [&](int64_t n0) {
        if constexpr (BKM) return rs_of(p.B, p.ldb, p.N, n0);
        else return pst_rsrc(p.B + n0, (unsigned)(((p.K - 1) * p.ldb + min((int64_t)128, p.N - n0)) * 4));
    };
    const __amdgpu_buffer_rsrc_t rc = pst_rsrc(p.C, (unsigned)(p.M * p.ldc * 4));
    const __amdgpu_buffer_rsrc_t rz = pst_rsrc(p.preact ? p.preact : p.C, p.preact ? (unsigned)(p.M * p.ldc * 4) : 0u);
    const unsigned vo = (unsigned)(wm * 64 + 4 * lh) * ldc4 + (unsigned)(wn * 64 + l31) * 4u;
    auto store_of = [&](int tm, int tn) {
        PstStore s;
        const int64_t col = (int64_t)tn * 128 + wn * 64 + l31;
        s.soff = (unsigned)(((int64_t)tm * 128 * p.ldc + (int64_t)tn * 128) * 4);
        const bool ok0 = col < p.N, ok1 = col + 32 < p.N;
        s.vo0 = ok0 ? vo : 0xFFFFFFF0u;
        s.vo1 = ok1 ? vo + 128u : 0xFFFFFFF0u;
        s.b0 = (p.bias && ok0) ? p.bias[col] : 0.f;
        s.b1 = (p.bias && ok1) ? p.bias[col + 32] : 0.f;
        return s;
    };

    f32x16 acc[2][2], pend[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int e = 0; e < 16; ++e) pend[i][n][e] = 0.f;       // (acc is defined by every tile's first k-step)
    float za[2][8], zb[2][8];                                      // EPI 2 only: z of the quarter being flushed / fetched
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int j = 0; j < 8; ++j) { za[n][j] = 0.f; zb[n][j] = 0.f; }

    int vb = blockIdx.x;
    int tm, tn;
    pst_tile(vb, p.total, p.tiles_m, p.tiles_n, tm, tn);
    __amdgpu_buffer_rsrc_t rsa = rs_of(p.A, p.lda, p.M, (int64_t)tm * 128), rsb = rs_of_b((int64_t)tn * 128);
    float4 ra[4], rb[4];
    g2r_fast<PBK>(ra, rsa, 0u, offa);
    g2r_fast<PBK>(rb, rsb, 0u, offb);
    r2s<PBK, true>(ra, smem, tid);
    r2s<PBK, BKM>(rb, smem + PTile::SIZE, tid);
    __syncthreads();

    int cur = 0;
    PstStore ps = store_of(tm, tn);                               // placeholder until a tile is pending
    // the k-step (1 .. nk - 2, minus one for the two-step flush of EPI 1 / 3) under which this block flushes: eight phases by the
    // block's index inside its XCD (block b runs on XCD b % 8), the two blocks of a CU (index i and i + 32) half a
    // tile apart; stagger == 0: step 1 for everybody
    int flush_at = 1;
    if (EPI != 2 && EPI != 4 && p.stagger) {
        const int idx = (int)blockIdx.x >> 3, phase = (idx + 4 * (idx >> 5)) & 7;
        const int last = nk - 2 - ((EPI == 1 || EPI == 3) ? 1 : 0);      // the last k-step a flush may START under (>= 1: gemm_pst_wanted)
        flush_at = __builtin_amdgcn_readfirstlane(1 + (phase * (last - 1)) / 7);
    }
    // one tile of the block's k-step stream.  PEND: a finished tile is waiting in `pend`; its stores go out under the first
    // k-step (EPI 1, twice the stores, and EPI 2: one row half under each of the first two).  Returns false after the block's
    // last tile.  (The first tile is peeled instead of testing a `have_pend` flag inside one loop: with the flag the kernel took
    // 236 VGPRs -- and spilled with the two-step flush; peeled it takes 152 / 208.)
    auto run_tile = [&](auto pend_tag) -> bool {
        constexpr bool PEND = decltype(pend_tag)::value;
        const int vbn = vb + (int)gridDim.x;
        const bool has_next = vbn < p.total;
        int tmn = tm, tnn = tn;
        if (has_next) pst_tile(vbn, p.total, p.tiles_m, p.tiles_n, tmn, tnn);
        const __amdgpu_buffer_rsrc_t rsan = rs_of(p.A, p.lda, p.M, (int64_t)tmn * 128), rsbn = rs_of_b((int64_t)tnn * 128);
        const PstStore mine = store_of(tm, tn);
#define PST_STEP_E(F, ZQ, ZF, ZL, RA, RB, KT, EDGE) \
    pst_step<EPI, BKM, (PEND ? F : 0), (PEND || ZQ == 1 ? ZQ : 0), EDGE>(acc, ra, rb, smem, cur, RA, RB, (unsigned)(KT) * 128u, (unsigned)(KT) * kstep_b, \
        offa, offb, tid, wm, wn, l31, lh, pend, ps, mine, ZF, ZL, rc, rz, ldc4, p.alpha, p.beta); \
    cur ^= 1
#define PST_STEP(F, ZQ, ZF, ZL, RA, RB, KT) PST_STEP_E(F, ZQ, ZF, ZL, RA, RB, KT, 0)
        // Some k-steps of this tile flush the pending one: all of it under ONE step (EPI 0), a row half under each of TWO (EPI 1),
        // a quarter under each of the first FOUR (EPI 2, which also fetches the NEXT quarter's z values a step ahead, the two z sets
        // swapping roles).  WHICH step (EPI 0 / 1) depends on the block (round 6): the resident blocks start together and do equal
        // work, so with a fixed flush step every block of the chip pushed its 64 KB tile into the fabric inside the same ~3 us --
        // 33.5 MB per generation at 10 TB/s, which the store path answers by stalling the operand loads queued behind it.  With
        // `flush_at` spread over the tile's k-steps by block the same bytes leave at the GEMM's average store rate (< 1 TB/s).
        // (three loops in a row, not one loop with a branch around the flushing step: with the branch inside the loop hipcc
        //  spilled 172-432 bytes per lane)
        // step 0 is the tile's FIRST k-step (EDGE 1: the accumulators start from C = 0), with or without flush work
        // (the FIRST step never flushes -- flush_at >= 1 -- so that it is ONE piece of code: a run-time choice between a flushing
        //  and a plain first step spilled 40-84 B per lane, like the branch inside the loop did)
        int kt = 1;
        if constexpr (EPI == 0) {
            PST_STEP_E(0, 0, za, zb, rsa, rsb, 1, 1);
            if constexpr (PEND) {
                for (; kt < flush_at; ++kt) { PST_STEP(0, 0, za, zb, rsa, rsb, kt + 1); }
                PST_STEP(0xF, 0, za, zb, rsa, rsb, kt + 1);
                ++kt;
            }
            for (; kt + 1 < nk; ++kt) { PST_STEP(0, 0, za, zb, rsa, rsb, kt + 1); }
        } else if constexpr (EPI == 1 || EPI == 3) {
            PST_STEP_E(0, 0, za, zb, rsa, rsb, 1, 1);
            if constexpr (PEND) {
                for (; kt < flush_at; ++kt) { PST_STEP(0, 0, za, zb, rsa, rsb, kt + 1); }
                PST_STEP(0x3, 0, za, zb, rsa, rsb, kt + 1);
                ++kt;
                PST_STEP(0xC, 0, za, zb, rsa, rsb, kt + 1);
                ++kt;
            }
            for (; kt + 1 < nk; ++kt) { PST_STEP(0, 0, za, zb, rsa, rsb, kt + 1); }
        } else {
            PST_STEP_E(0x1, 2, za, zb, rsa, rsb, 1, 1);
            PST_STEP(0x2, 3, zb, za, rsa, rsb, 2);
            PST_STEP(0x4, 4, za, zb, rsa, rsb, 3);
            PST_STEP(0x8, 0, zb, za, rsa, rsb, 4);
            for (kt = 4; kt + 1 < nk; ++kt) { PST_STEP(0, 0, za, zb, rsa, rsb, kt + 1); }
        }
        // last k-step: fetches the first k-tile of the next tile (or, with nothing left, re-reads this one's: never used) and
        // (EPI 2) the z values of this tile's quarter 0
        // (EDGE 2: the step's last MFMAs write `pend` -- the finished tile becomes the pending one without a copy)
        PST_STEP_E(0, 1, zb, za, rsan, rsbn, 0, 2);
#undef PST_STEP
#undef PST_STEP_E
        ps = mine;
        vb = vbn; tm = tmn; tn = tnn; rsa = rsan; rsb = rsbn;
        return has_next;
    };
    if (run_tile(std::false_type{}))
        while (run_tile(std::true_type{})) {}
    // the block's last tile: nothing left to hide its stores under
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if ((EPI == 2 || EPI == 4) && q > 0) pst_load_z(za, q, ps, rz, ldc4);
        pst_flush_q<EPI>(q, pend, ps, za, rc, rz, ldc4, p.alpha, p.beta);
    }
}

// ---- host ---------------------------------------------------------------------------------------------------------
template <int EPI, bool BKM>
static int pst_slots_of() {
    int per_cu = 0, dev = 0;
    hipDeviceProp_t prop;
    const size_t lds = 2 * PSTAGE * sizeof(float);
    const void* k = reinterpret_cast<const void*>(gemm_pst_kernel<EPI, BKM>);
    if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess || hipGetDevice(&dev) != hipSuccess ||
        hipGetDeviceProperties(&prop, dev) != hipSuccess || hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k, NT, lds) != hipSuccess)
        return -1;
    return per_cu > 0 ? prop.multiProcessorCount * per_cu : -1;
}

// resident blocks of the whole chip (LDS admits two per CU for every variant), a multiple of the 8 XCDs
static int pst_slots() {
    static const int slots = []() {
        const int s = min(min(min(pst_slots_of<0, true>(), pst_slots_of<1, true>()), min(pst_slots_of<0, false>(), pst_slots_of<2, false>())),
                          min(pst_slots_of<3, true>(), pst_slots_of<4, false>()));
        return s > 0 ? (s & ~7) : -1;
    }();
    return slots;
}

// which epilogue variant a request maps to, or -1
static int pst_epi(bool b_kmajor, int act, const float* preact, const float* dswish, int dact) {
    if (dswish) return (!b_kmajor && act == ACT_NONE && !preact && (dact == 1 || dact == 3)) ? (dact == 1 ? 2 : 4) : -1;
    if (act == ACT_SWISH) return b_kmajor ? 1 : -1;
    if (act == ACT_SWISH_D) return (b_kmajor && preact) ? 3 : -1;
    return (act == ACT_NONE && !preact) ? 0 : -1;
}

// NNHIP_GEMM_PST: 0 = never, 1 (default) = when the conditions hold, 2 = also for long reductions
bool gemm_pst_wanted(int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc, const float* A, const float* B,
                     const float* C, const float* bias, bool b_kmajor, int act, const float* preact, const float* dswish, int dact) {
    static const int on = []() { const char* e = getenv("NNHIP_GEMM_PST"); return e ? atoi(e) : 1; }();
    if (!on) return false;
    const int epi = pst_epi(b_kmajor, act, preact, dswish, dact);
    if (epi < 0) return false;
    // short reductions only (on = 2 lifts that: developer switch): measured on MI355X, 16384x512->15000 135 -> 141 TFLOP/s and
    // 16384x512->2048 129 -> 133, but K = 4096 shapes -2 % -- there the fixed cost is 1.5 % of a tile and gemm.hip's two-deep
    // prefetch is worth more (with a two-deep prefetch of its own this kernel draws level there, no better: not kept)
    if (on < 2 && K > 1024) return false;
    // (the flush of the Swish epilogues takes two / four k-steps before the tile's last one)
    if ((K % PBK) != 0 || K < ((epi == 2 || epi == 4) ? 5 : (epi == 1 || epi == 3) ? 4 : 3) * PBK || (M % 128) != 0 || N <= 0) return false;
    if (!aligned16(A) || !aligned16(B) || (lda & 3) || (ldb & 3) ||
        ((reinterpret_cast<uintptr_t>(C) | reinterpret_cast<uintptr_t>(preact) | reinterpret_cast<uintptr_t>(dswish) | reinterpret_cast<uintptr_t>(bias)) & 3))
        return false;
    // (C's byte size is the store descriptor's num_records and must stay below the 0xFFFFFFF0 offset that drops a lane)
    if (M * ldc * 4 >= (int64_t)0xFFFF0000 || 128 * lda * 4 + K * 4 >= ((int64_t)1 << 32)) return false;
    if (b_kmajor ? 128 * ldb * 4 + K * 4 >= ((int64_t)1 << 32) : (K + 32) * ldb * 4 >= ((int64_t)1 << 32)) return false;
    const int slots = pst_slots();
    if (slots <= 0) return false;
    const int64_t tiles = (M / 128) * ceil_div(N, 128);
    return tiles > slots;
}

int gemm_pst(const float* A, const float* B, float* C, const float* bias, float* preact, int64_t M, int64_t N, int64_t K,
             int64_t lda, int64_t ldb, int64_t ldc, bool b_kmajor, float alpha, int act, float beta, const float* dswish, int dact,
             hipStream_t st) {
    const int epi = pst_epi(b_kmajor, act, preact, dswish, dact);
    PstParams p;
    p.A = A; p.B = B; p.C = C; p.bias = bias; p.preact = (epi == 2 || epi == 4) ? const_cast<float*>(dswish) : preact;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
    p.alpha = alpha; p.beta = beta;
    p.tiles_m = (int)(M / 128); p.tiles_n = (int)ceil_div(N, 128); p.total = p.tiles_m * p.tiles_n;
    static const int stagger = []() { const char* e = getenv("NNHIP_PST_STAGGER"); return e ? atoi(e) : 1; }();
    p.stagger = stagger;
    const size_t lds = 2 * PSTAGE * sizeof(float);
    const dim3 grid((unsigned)pst_slots()), block(NT);
    if (epi == 4) hipLaunchKernelGGL((gemm_pst_kernel<4, false>), grid, block, lds, st, p);
    else if (epi == 3) hipLaunchKernelGGL((gemm_pst_kernel<3, true>), grid, block, lds, st, p);
    else if (epi == 2) hipLaunchKernelGGL((gemm_pst_kernel<2, false>), grid, block, lds, st, p);
    else if (epi == 1) hipLaunchKernelGGL((gemm_pst_kernel<1, true>), grid, block, lds, st, p);
    else if (b_kmajor) hipLaunchKernelGGL((gemm_pst_kernel<0, true>), grid, block, lds, st, p);
    else hipLaunchKernelGGL((gemm_pst_kernel<0, false>), grid, block, lds, st, p);
    NNHIP_LAUNCH_CHECK("gemm_pst_kernel");
    return 0;
}

}  // namespace nnhip
