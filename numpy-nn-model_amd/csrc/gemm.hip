// gemm.hip -- fp32 MFMA GEMM for gfx950 (MI355X), the engine behind Linear fwd/bwd
// (reference semantics: neunet/nn/layers/linear.py:48-58, 17-24) and fused Linear->Swish.
//
// C[M,N] = op(A) * op(B) (+bias[N]) (act), exact fp32 (v_mfma_f32_32x32x2_f32: a k-ordered fmaf
// chain -- no TF32 on gfx950; the reference's CUDA LinearSwish path is TF32 and only 1e-3 accurate).
//
// Structure (MI355X-first, not a CUTLASS translation):
//   * block = 256 threads = 4 wave64 in a 2x2 grid; block tile 128x128; wave tile 64x64 =
//     2x2 MFMA 32x32 tiles -> 4 x f32x16 accumulators (64 acc registers / lane);
//   * K is walked in BK-deep tiles, global -> registers -> LDS with a 2-stage LDS ring and two register staging sets:
//     tile t+2 is fetched while tile t is multiplied and tile t+1 goes to the other LDS stage -> one __syncthreads per
//     tile, two iterations of 64-cycle MFMAs to hide a load; 2 blocks/CU (2 waves/SIMD) cover each other's barrier/LDS
//     phases.  The fp32 MFMA runs on the vector ALU's lanes, so the loop carries NO vector instruction
//     besides the MFMAs: operands come through buffer descriptors with scalar k offsets (gemm_common.h);
//   * each operand is either "k-major" (reduction dim contiguous in memory) or "outer-major";
//     k-major tiles live in LDS as [128][BK+4] (row stride 36 floats = conflict-free
//     ds_read_b128 of 4 consecutive k per lane), outer-major tiles as [BK][128] (ds_read_b32, the
//     two half-waves read two k rows).  Both feed the same k permutation to A and B:
//     MFMA j of k-group g consumes k = 8g + j (lanes 0-31) and 8g + 4 + j (lanes 32-63);
//   * 1-D grid with an XCD-aware, grouped tile order: block b runs on XCD b%8 (observed, used for
//     speed only); each XCD gets a contiguous run of logical tile ids, walked 8 tile-rows at a time,
//     so the 64 tiles resident on one XCD share A/B panels in that XCD's private 4 MiB L2;
//   * split-K (deterministic: fp32 slabs + a reduce kernel, no atomics) when M*N has too few
//     tiles to fill 256 CUs (e.g. dW = dO^T X with a 16384-long reduction at GPT-tiny).
#include <stdlib.h>

#include <type_traits>

#include "gemm_common.h"

namespace nnhip {

__device__ int g_gemm_progress[8192];      // k-loop progress per (XCD, SE, SH, CU, wave slot parity): see gemm_f32_block


// K loop.  One basic block per iteration; the issue order is pinned with sched_group_barrier so that the
// 2*BK/8 global loads of tile t+1 ride in the shadow of the first MFMAs of tile t (one load per 64-cycle
// MFMA) and the 2*BK/8 LDS stores in the shadow of the last ones -- the ablation (profiles/) showed the
// un-interleaved load + store phases costing 13 % of the loop while ds_reads and the barrier were free.
// SUM: also accumulate the staged A elements into cs (row sums of A).
// (A generic lambda here cost 30 %: the accumulators left their registers.  Keep it a forceinline function.)
template <int BK, bool AKC, bool BKC, bool VEC, bool SUM>
__device__ __forceinline__ void gemm_k_loop(f32x16 (&acc)[2][2], float4 (&ra)[BK / 8], float4 (&rb)[BK / 8], float4& cs,
                                            float* __restrict__ smem, const float* __restrict__ A,
                                            const float* __restrict__ B, const GemmParams& p, int nk, int64_t kbeg,
                                            int64_t kend, int64_t m0, int64_t n0, int tid, int wm, int wn, int l31,
                                            int lh, const float* __restrict__ Z) {
    using TA = Tile<BK, AKC>;
    using TB = Tile<BK, BKC>;
    constexpr int STAGE = TA::SIZE + TB::SIZE;
    constexpr int NLD = 2 * (BK / 8);          // float4 loads (= LDS stores) per thread per tile
    constexpr int NMF = 16 * (BK / 8);         // MFMAs per wave per tile
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = kt + 1 < nk;
        const int64_t k0 = kbeg + (int64_t)(kt + 1) * BK;
        if constexpr (SUM) {
            // ra still holds tile kt (its LDS store at the end of the previous iteration already waited for the
            // loads): summing it HERE costs no extra vmcnt wait; summing it next to the loads stalled the MFMA pipe.
#pragma unroll
            for (int q = 0; q < BK / 8; ++q) { cs.x += ra[q].x; cs.y += ra[q].y; cs.z += ra[q].z; cs.w += ra[q].w; }
        }
        g2r<BK, AKC, VEC>(ra, A, p.lda, p.M, kend, m0, k0, tid, more, Z);
        g2r<BK, BKC, VEC>(rb, B, p.ldb, p.N, kend, n0, k0, tid, more, Z);
        const float* As = smem + cur * STAGE;
        const float* Bs = As + TA::SIZE;
#pragma unroll
        for (int g = 0; g < BK / 8; ++g) {
            float a[2][4], b[2][4];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                frag<BK, AKC>(a[i], As, wm * 64 + i * 32 + l31, g, lh);
                frag<BK, BKC>(b[i], Bs, wn * 64 + i * 32 + l31, g, lh);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int n = 0; n < 2; ++n)
                        acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][j], b[n][j], acc[i][n],
                                                                         0, 0, 0);
        }
        {   // next tile -> other LDS stage (stores of zeros on the last iteration are harmless)
            float* Sn = smem + (cur ^ 1) * STAGE;
            r2s<BK, AKC>(ra, Sn, tid);
            r2s<BK, BKC>(rb, Sn + TA::SIZE, tid);
        }
        // ---- issue-order pipeline for this iteration's scheduling region --------------------------------
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // 1 VMEM read
        }
        __builtin_amdgcn_sched_group_barrier(0x008, NMF - 2 * NLD, 0);
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);   // 1 DS write
        }
        __syncthreads();
        cur ^= 1;
    }
}

// One k-tile of the vectorised variants (operand tiles through g2r_fast, gemm_common.h): fetch a tile into (fa, fb), multiply
// the tile in LDS stage `cur`, commit the tile held in (ca, cb) to the other stage.  The fetched tile is TWO ahead of the one
// being multiplied (the committed one is one ahead): a load has two iterations (~7 us with two blocks per CU) to land, so
// the loop rides through the latency spike of a generation of blocks storing their C tiles together -- with a one-deep
// prefetch that store burst (33.5 MB per 512 tiles = 6.7 us at 5 TB/s) was fully exposed in grids of a few generations
// (a build without stores ran exactly that much faster, EXPERIMENTS.md 5.1d).
// SUM: cs += csf[q] * (the committed A tile) -- csf = 1 for a whole tile, the new-rows mask for the shifted partial tile,
// 0 for the never-used tiles fetched past the end.
template <int BK, bool AKC, bool BKC, bool SUM>
__device__ __forceinline__ void gemm_k_step(f32x16 (&acc)[2][2], float4 (&fa)[BK / 8], float4 (&fb)[BK / 8],
                                            const float4 (&ca)[BK / 8], const float4 (&cb)[BK / 8], float4& cs,
                                            const float (&csf)[BK / 8], float* __restrict__ smem, int cur,
                                            __amdgpu_buffer_rsrc_t rsa, __amdgpu_buffer_rsrc_t rsb, unsigned kfa, unsigned kfb,
                                            const unsigned (&offa)[BK / 8], const unsigned (&offb)[BK / 8], int tid, int wm,
                                            int wn, int l31, int lh) {
    using TA = Tile<BK, AKC>;
    using TB = Tile<BK, BKC>;
    constexpr int STAGE = TA::SIZE + TB::SIZE;
    constexpr int NLD = 2 * (BK / 8);
    constexpr int NMF = 16 * (BK / 8);
    g2r_fast<BK>(fa, rsa, kfa, offa);
    g2r_fast<BK>(fb, rsb, kfb, offb);
    const float* As = smem + cur * STAGE;
    const float* Bs = As + TA::SIZE;
#pragma unroll
    for (int g = 0; g < BK / 8; ++g) {
        float a[2][4], b[2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            frag<BK, AKC>(a[i], As, wm * 64 + i * 32 + l31, g, lh);
            frag<BK, BKC>(b[i], Bs, wn * 64 + i * 32 + l31, g, lh);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int n = 0; n < 2; ++n)
                    acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][j], b[n][j], acc[i][n], 0, 0, 0);
    }
    if constexpr (SUM) {
#pragma unroll
        for (int q = 0; q < BK / 8; ++q) {
            cs.x = fmaf(csf[q], ca[q].x, cs.x); cs.y = fmaf(csf[q], ca[q].y, cs.y);
            cs.z = fmaf(csf[q], ca[q].z, cs.z); cs.w = fmaf(csf[q], ca[q].w, cs.w);
        }
    }
    {
        float* Sn = smem + (cur ^ 1) * STAGE;
        r2s<BK, AKC>(ca, Sn, tid);
        r2s<BK, BKC>(cb, Sn + TA::SIZE, tid);
    }
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // 1 VMEM read
    }
    __builtin_amdgcn_sched_group_barrier(0x008, NMF - 2 * NLD, 0);
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);   // 1 DS write
    }
    __syncthreads();
}

// The partial last k-tile sits in LDS stage `cur` as the LAST BK columns of the k range: k-groups [g0, BK/8) are new.
template <int BK, bool AKC, bool BKC>
__device__ __forceinline__ void tail_mma(f32x16 (&acc)[2][2], const float* __restrict__ smem, int cur, int g0, int wm, int wn,
                                         int l31, int lh) {
    using TA = Tile<BK, AKC>;
    using TB = Tile<BK, BKC>;
    const float* As = smem + cur * (TA::SIZE + TB::SIZE);
    const float* Bs = As + TA::SIZE;
#pragma unroll 1
    for (int g = g0; g < BK / 8; ++g) {
        float a[2][4], b[2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            frag<BK, AKC>(a[i], As, wm * 64 + i * 32 + l31, g, lh);
            frag<BK, BKC>(b[i], Bs, wn * 64 + i * 32 + l31, g, lh);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int n = 0; n < 2; ++n)
                    acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][j], b[n][j], acc[i][n], 0, 0, 0);
    }
}

// CS (outer-major A only): every thread also accumulates the A elements it stages -- with the [BK][128] tile layout a
// thread owns the same 4 A rows (m) in every k-tile -- and the tile_n == 0 blocks reduce them to asum[m] = sum_k A[m,k].
// One block's work: logical block L of problem p (L in [0, tiles * splitk)), batch index `by`.
template <int BK, bool AKC, bool BKC, bool VEC, bool CS = false>
__device__ __forceinline__ void gemm_f32_block(const GemmParams& p, const int L, const int by, float* __restrict__ smem) {
    static_assert(!CS || !AKC, "row sums of A are only implemented for an outer-major A");
    using TA = Tile<BK, AKC>;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, lh = lane >> 5;
#ifdef GEMM_PROF
    long long ts_[5];
    ts_[0] = clock64();
#define GP_STAMP(i) ts_[i] = clock64()
#else
#define GP_STAMP(i) do {} while (0)
#endif
    // ---- block id -> (split, tile_m, tile_n): XCD-aware grouped order --------------------------
    // (A persistent variant -- one block per resident slot walking tiles b, b + grid, ... so that a tile's stores drain under
    //  the same waves' next tile -- was measured in round 2 on top of the buffer-load fetch: no gain on any shape, and the
    //  outer loop cost the (k-major, outer-major) variant 7 % through register allocation.  One block per tile it stays.)
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

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    float4 ra[BK / 8], rb[BK / 8];
    float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (VEC) {
        // host guarantees: K % 8 == 0, K >= BK, 16-B aligned rows, ld < 2^22 (gemm_common.h, "fast operand fetch")
        unsigned offa[BK / 8], offb[BK / 8];
        op_offsets<BK, AKC>(offa, p.lda, p.M, m0, tid);
        op_offsets<BK, BKC>(offb, p.ldb, p.N, n0, tid);
        const int64_t klen = kend - kbeg;
        const int nfull = (int)(klen / BK), rem = (int)(klen - (int64_t)nfull * BK);
        const bool has_tail = rem != 0;
        // descriptor = the operand's rows of this block at k = 0; k-tile at k: byte offset 4k (k-major) or 4k*ld (outer-major)
        const __amdgpu_buffer_rsrc_t rsa = operand_rsrc(AKC ? A + m0 * p.lda : A + m0);
        const __amdgpu_buffer_rsrc_t rsb = operand_rsrc(BKC ? B + n0 * p.ldb : B + n0);
        const unsigned ua = AKC ? 4u : (unsigned)(4 * p.lda), ub = BKC ? 4u : (unsigned)(4 * p.ldb);   // bytes per unit of k
        const unsigned katail = (unsigned)(kend - BK) * ua, kbtail = (unsigned)(kend - BK) * ub;
        const unsigned ka = nfull > 0 ? (unsigned)kbeg * ua : katail;
        const unsigned kb = nfull > 0 ? (unsigned)kbeg * ub : kbtail;
        // tile t of this block's k range: whole tiles in order, then the shifted partial tile; tiles past the end are
        // fetched too (any valid address: the first tile) and never used.  All scalar.
        const unsigned astep = BK * ua, bstep = BK * ub;
        auto ka_of = [&](int t) -> unsigned { return t < nfull ? ka + (unsigned)t * astep : ((t == nfull && has_tail) ? katail : ka); };
        auto kb_of = [&](int t) -> unsigned { return t < nfull ? kb + (unsigned)t * bstep : ((t == nfull && has_tail) ? kbtail : kb); };
        [[maybe_unused]] float tailmask[BK / 8], csf[BK / 8];
        auto set_csf = [&](int t) {                       // weights of tile t in the row sums (CS variants)
            if constexpr (CS) {
#pragma unroll
                for (int q = 0; q < BK / 8; ++q) csf[q] = t < nfull ? 1.f : ((t == nfull && has_tail) ? tailmask[q] : 0.f);
            }
        };
        if constexpr (CS) {
#pragma unroll
            for (int q = 0; q < BK / 8; ++q) tailmask[q] = (tid + NT * q) / 32 >= BK - rem ? 1.f : 0.f;
        }
        float4 ra2[BK / 8], rb2[BK / 8];
        g2r_fast<BK>(ra, rsa, ka_of(0), offa);
        g2r_fast<BK>(rb, rsb, kb_of(0), offb);
        if constexpr (CS) {
            set_csf(0);
#pragma unroll
            for (int q = 0; q < BK / 8; ++q) {
                cs.x = fmaf(csf[q], ra[q].x, cs.x); cs.y = fmaf(csf[q], ra[q].y, cs.y);
                cs.z = fmaf(csf[q], ra[q].z, cs.z); cs.w = fmaf(csf[q], ra[q].w, cs.w);
            }
        }
        r2s<BK, AKC>(ra, smem, tid);
        r2s<BK, BKC>(rb, smem + TA::SIZE, tid);
        g2r_fast<BK>(ra, rsa, ka_of(1), offa);            // tile 1: in flight until the first step commits it
        g2r_fast<BK>(rb, rsb, kb_of(1), offb);
        __syncthreads();
        GP_STAMP(1);
        int cur = 0, kt = 0;
        // Lock-step mode (nnhipSetGemmLockstep / NNHIP_GEMM_LOCKSTEP=1, off by default).  Left alone, the older of
        // the two blocks of a CU wins the MFMA issue arbitration (its k-loop takes 0.80 M ticks, its mate's 1.15 M), so the 64 tiles
        // resident on an XCD run as two cohorts of 32 that drift apart by more than the 16 k-tiles the 4 MiB L2 holds, and every
        // operand panel the cohorts share is fetched from the fabric twice.  Here each wave publishes its k-tile on a board indexed
        // by the physical CU and reads its mate's; a block that got LOCK_LEAD k-tiles ahead runs at the low issue priority until it
        // is as far behind.  Fabric reads of a 4096^3 forward 808 -> 575 MB (dW 655 -> 574); the price is the stagger that hid one
        // block's epilogue behind its mate's MFMAs: +2.7 % time (dW +0.6 %), which is why it is opt-in.
        constexpr int LOCK_LEAD = 4;
        const bool plock = p.lockstep && nfull >= 64 && AKC == BKC;     // (mixed layouts -- dX -- already sit at 541 MB: no gain there)
        int *pmine = nullptr, *ptheirs = nullptr;
        int pmate = 0, pstate = 1;
        if (plock) {
            const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20);   // HW_ID, XCC_ID
            const unsigned cuslot = (xcc << 9) | (((hw >> 13) & 7) << 6) | (((hw >> 12) & 1) << 5) | (((hw >> 8) & 15) << 1);
            pmine = g_gemm_progress + (cuslot | (hw & 1));                 // wave slot parity tells the two blocks of a CU apart
            ptheirs = g_gemm_progress + (cuslot | ((hw & 1) ^ 1));
        }
        for (; kt + 1 < nfull; kt += 2) {
            if (plock) {
                asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(pmate));      // the scalar load issued one pair ago (below)
                const int d = kt - pmate;          // pmate: the mate's k-tile as of one pair ago
                // bang-bang with hysteresis: a block that got LOCK_LEAD k-tiles ahead runs at the low priority until it is as far behind
                // (2 = what the mate has done since the read: both see the same lag)
                if (d >= 48 || d <= -48) { if (pstate != 1) { pstate = 1; __builtin_amdgcn_s_setprio(1); } }
                else if (d > 2 + LOCK_LEAD) { if (pstate != 0) { pstate = 0; __builtin_amdgcn_s_setprio(0); } }
                else if (d < 2 - LOCK_LEAD) { if (pstate != 3) { pstate = 3; __builtin_amdgcn_s_setprio(3); } }
                __hip_atomic_store(pmine, kt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                // scalar, past the scalar cache; consumed at the top of the next pair (a vector load here made the compiler wait for
                // vmcnt(0) at once -- the operand prefetch with it: 0.94 -> 1.32 ms)
                asm volatile("s_load_dword %0, %1, 0x0 glc" : "=&s"(pmate) : "s"(ptheirs));
            }
            set_csf(kt + 1);
            gemm_k_step<BK, AKC, BKC, CS>(acc, ra2, rb2, ra, rb, cs, csf, smem, cur, rsa, rsb, ka_of(kt + 2), kb_of(kt + 2), offa, offb,
                                          tid, wm, wn, l31, lh);
            set_csf(kt + 2);
            gemm_k_step<BK, AKC, BKC, CS>(acc, ra, rb, ra2, rb2, cs, csf, smem, cur ^ 1, rsa, rsb, ka_of(kt + 3), kb_of(kt + 3), offa, offb,
                                          tid, wm, wn, l31, lh);
        }
        if (kt < nfull) {
            set_csf(kt + 1);
            gemm_k_step<BK, AKC, BKC, CS>(acc, ra2, rb2, ra, rb, cs, csf, smem, cur, rsa, rsb, ka_of(kt + 2), kb_of(kt + 2), offa, offb,
                                          tid, wm, wn, l31, lh);
            cur ^= 1;
        }
        if (plock) __builtin_amdgcn_s_setprio(0);
        if (has_tail) {
            tail_mma<BK, AKC, BKC>(acc, smem, cur, (BK - rem) >> 3, wm, wn, l31, lh);
            __syncthreads();                                  // the epilogue re-uses the LDS block
        }
        GP_STAMP(2);
    } else {
        const int nk = (int)((kend - kbeg + BK - 1) / BK);
        const float* __restrict__ Z = p.zeros;
        g2r<BK, AKC, VEC>(ra, A, p.lda, p.M, kend, m0, kbeg, tid, nk > 0, Z);
        g2r<BK, BKC, VEC>(rb, B, p.ldb, p.N, kend, n0, kbeg, tid, nk > 0, Z);
        r2s<BK, AKC>(ra, smem, tid);
        r2s<BK, BKC>(rb, smem + TA::SIZE, tid);
        __syncthreads();
        // (running a sum-free copy of the loop in the tile_n != 0 blocks was tried: two inlined copies made the CS kernel
        // 15 % slower; the row sums cost ~4 % of the loop, so linear.hip only asks for them when a separate column-sum
        // pass over dO would cost more.)
        gemm_k_loop<BK, AKC, BKC, VEC, CS>(acc, ra, rb, cs, smem, A, B, p, nk, kbeg, kend, m0, n0, tid, wm, wn, l31, lh, Z);
    }

#ifdef GEMM_PROF
    long long ets_[5] = {0, 0, 0, 0, 0};
    gemm_epilogue<CS>(acc, p, smem, cs, tid, wave, lane, wm, wn, l31, lh, m0, n0, tn, split, c_off, ets_);
#else
    gemm_epilogue<CS>(acc, p, smem, cs, tid, wave, lane, wm, wn, l31, lh, m0, n0, tn, split, c_off);
#endif
#ifdef GEMM_PROF
    if constexpr (VEC) {
        GP_STAMP(3);                                          // epilogue issued
        __builtin_amdgcn_s_waitcnt(0);                        // ... and its stores acknowledged
        GP_STAMP(4);
        if (p.prof && lane == 0) {
            long long* o = p.prof + ((int64_t)blockIdx.x * 4 + wave) * 12;
            o[0] = ts_[0]; o[1] = ts_[1]; o[2] = ts_[2]; o[3] = ts_[3]; o[4] = ts_[4];
            for (int q_ = 0; q_ < 5; ++q_) o[5 + q_] = ets_[q_];
            o[10] = __builtin_amdgcn_s_getreg((31 << 11) | 4);      // HW_ID: wave/simd/pipe/cu/sh/se
            o[11] = ((long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | (unsigned)(tm * 65536 + tn);   // XCC_ID, tile
        }
    }
#endif
}

template <int BK, bool AKC, bool BKC, bool VEC, bool CS = false>
__global__ __launch_bounds__(NT, (BK <= 16) ? 3 : 2) void gemm_f32_kernel(const GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    gemm_f32_block<BK, AKC, BKC, VEC, CS>(p, xcd_order((int)blockIdx.x, (int)gridDim.x), (int)blockIdx.y, smem);
}

// Several independent parameter-gradient GEMMs (C_i = A_i^T-view B_i, both operands outer-major, + row sums of A_i) as ONE grid:
// logical ids [start[i], start[i+1]) belong to problem i.  Why: a transformer layer's four dW GEMMs (16-64 tiles each, a
// 16384-long reduction) each paid a launch ramp, a split-K epilogue in which every block stores its slab at the same moment,
// and a tail; queued behind each other in one grid the blocks of the next problem start as the previous one's finish
// (measured on the equivalent single GEMM 16384x512->6144: 0.757 ms against 0.868 ms for the four launches).
template <int BK>
__global__ __launch_bounds__(NT, 2) void gemm_f32_group_kernel(const GemmGroup g) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int b = (int)blockIdx.x;
    const int L = g.by_job ? b : xcd_order(b, (int)gridDim.x);
    int i = 0;
#pragma unroll
    for (int j = 1; j < GEMM_GROUP_MAX; ++j)
        if (j < g.n && L >= g.start[j]) i = j;
    const int local = L - g.start[i];
    gemm_f32_block<BK, false, false, true, true>(g.p[i], g.by_job ? xcd_order(local, g.start[i + 1] - g.start[i]) : local, 0, smem);
}

// C[m,n] = sum_s slab[s][m][n] (+bias)(act).  One thread per float4 of a row (N%4 handled).
struct ReduceArgs {
    const float* slab; float* C; float* preact; const float* bias;
    int64_t M, N, ldc;
    int splitk, act; float beta, alpha;
    const float* asum_slab; float* asum; const float* addend; int asum_blocks, vec;
    const float* dswish; int dact;
};
// logical block `blk` of `nblk` (the last r.asum_blocks of them reduce the row-sum partials)
__device__ __forceinline__ void splitk_reduce_body(const ReduceArgs& r, const int blk, const int nblk) {
    const float* __restrict__ slab = r.slab; float* __restrict__ C = r.C; float* __restrict__ preact = r.preact;
    const float* __restrict__ bias = r.bias; const float* __restrict__ asum_slab = r.asum_slab; float* __restrict__ asum = r.asum;
    const float* __restrict__ addend = r.addend; const float* __restrict__ dswish = r.dswish;
    const int64_t M = r.M, N = r.N, ldc = r.ldc;
    const int splitk = r.splitk, act = r.act, vec = r.vec, dact = r.dact;
    const float beta = r.beta, alpha = r.alpha;
    const int64_t total = M * N;
    // the last `asum_blocks` blocks reduce the row-sum partials (their splitk dependent loads must not sit in front of
    // the main loop of the first blocks: that put +5 us on the critical path of every split-K dW)
    const int main_blocks = nblk - r.asum_blocks;
    if (blk >= main_blocks) {
        const int64_t i = (int64_t)(blk - main_blocks) * blockDim.x + threadIdx.x;
        if (i < M) {
            float s = 0.f;
            for (int k = 0; k < splitk; ++k) s += asum_slab[(int64_t)k * M + i];
            asum[i] = s;
        }
        return;
    }
    const int64_t stride = (int64_t)main_blocks * blockDim.x;
    auto finish = [&](float s, int64_t m, int64_t n) {
        s *= alpha;
        if (bias) s += bias[n];
        if (addend) s += addend[m * ldc + n];
        if (dswish) {
            const float x = dswish[m * ldc + n];
            s = dact == 2 ? (x > 0.f ? s : 0.f) : dact == 3 ? s * x : s * swish_grad_(x, beta);
        }
        if (act == ACT_SWISH) {
            if (preact) preact[m * ldc + n] = s;
            s = s * sigmoid_fast_(beta * s);
        } else if (act == ACT_SWISH_D) {
            float d;
            swish_fwd_d_(s, beta, s, d);
            if (preact) preact[m * ldc + n] = d;
        } else if (act == ACT_RELU) {
            s = fmaxf(s, 0.f);
        } else if (act == ACT_SIGMOID) {
            s = sigmoid_fast_(s);
        }
        return s;
    };
    if (vec) {   // N % 4 == 0, ldc % 4 == 0, 16-B aligned slab / C: one float4 per thread, 4 slab loads in flight
        const int64_t total4 = total >> 2;
        const float4* __restrict__ slab4 = reinterpret_cast<const float4*>(slab);
        for (int64_t i = (int64_t)blk * blockDim.x + threadIdx.x; i < total4; i += stride) {
            float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
            int k = 0;
            for (; k + 4 <= splitk; k += 4) {
                const float4 a = slab4[(int64_t)k * total4 + i], b = slab4[(int64_t)(k + 1) * total4 + i];
                const float4 c = slab4[(int64_t)(k + 2) * total4 + i], d = slab4[(int64_t)(k + 3) * total4 + i];
                s.x = (((s.x + a.x) + b.x) + c.x) + d.x; s.y = (((s.y + a.y) + b.y) + c.y) + d.y;
                s.z = (((s.z + a.z) + b.z) + c.z) + d.z; s.w = (((s.w + a.w) + b.w) + c.w) + d.w;
            }
            for (; k < splitk; ++k) {
                const float4 a = slab4[(int64_t)k * total4 + i];
                s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
            }
            const int64_t e = i << 2, m = e / N, n = e - m * N;
            s.x = finish(s.x, m, n); s.y = finish(s.y, m, n + 1); s.z = finish(s.z, m, n + 2); s.w = finish(s.w, m, n + 3);
            *reinterpret_cast<float4*>(C + m * ldc + n) = s;
        }
        return;
    }
    for (int64_t i = (int64_t)blk * blockDim.x + threadIdx.x; i < total; i += stride) {
        float s = 0.f;
        for (int k = 0; k < splitk; ++k) s += slab[(int64_t)k * total + i];
        const int64_t m = i / N, n = i - m * N;
        C[m * ldc + n] = finish(s, m, n);
    }
}


__global__ __launch_bounds__(256) void splitk_reduce_kernel(const ReduceArgs r) { splitk_reduce_body(r, (int)blockIdx.x, (int)gridDim.x); }

struct ReduceGroup {
    ReduceArgs r[GEMM_GROUP_MAX];
    int start[GEMM_GROUP_MAX + 1];
    int n;
};
__global__ __launch_bounds__(256) void splitk_reduce_group_kernel(const ReduceGroup g) {
    const int b = (int)blockIdx.x;
    int i = 0;
#pragma unroll
    for (int j = 1; j < GEMM_GROUP_MAX; ++j)
        if (j < g.n && b >= g.start[j]) i = j;
    splitk_reduce_body(g.r[i], b - g.start[i], g.start[i + 1] - g.start[i]);
}

template <int BK, bool AKC, bool BKC, bool VEC, bool CS = false>
static int launch_variant(const GemmParams& p, int64_t batch, hipStream_t st) {
    constexpr size_t lds = 2 * (Tile<BK, AKC>::SIZE + Tile<BK, BKC>::SIZE) * sizeof(float);
    auto kern = gemm_f32_kernel<BK, AKC, BKC, VEC, CS>;
    static bool attr_set = false;  // per instantiation
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return hip_status(e, "hipFuncSetAttribute(gemm)");
        attr_set = true;
    }
    dim3 grid((unsigned)(p.tiles_m * p.tiles_n * p.splitk), (unsigned)batch);
    hipLaunchKernelGGL(kern, grid, dim3(NT), lds, st, p);
    NNHIP_LAUNCH_CHECK("gemm_f32_kernel");
    return 0;
}

// Host-side GEMM dispatcher (internal API used by linear.hip / api).
int gemm_f32_ex(const float* A, const float* B, float* C, const float* bias, float* preact, int64_t M,
                int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc, bool a_kmajor, bool b_kmajor,
                int64_t batch1, int64_t sA, int64_t sB, int64_t sC, int64_t batch2, int64_t sA2,
                int64_t sB2, int64_t sC2, float alpha, int act, float beta, hipStream_t st, float* asum = nullptr,
                const float* addend = nullptr, const float* dswish = nullptr, int dact = 1);

// gemm_bf3.hip: the same tiles on the bf16 matrix cores (exact 3-way bf16 split of every fp32 operand, 6 products)
int gemm_bf3_launch(const GemmParams& p, bool a_kmajor, bool b_kmajor, bool vec, bool cs, int64_t batch, hipStream_t st);
int gemm_bf3_group_launch(const GemmGroup& g, int blocks, hipStream_t st);
// launches per kernel family since load (nnhipGemmLaunchCount): 0 classic fp32 128x128, 1 persistent fp32, 2 small, 3 split-bf16.
// Host-side counters for tests that must know WHICH kernel produced a result (a "bf16x3" test that only ever reaches the
// small kernel proves nothing about gemm_bf3_kernel).
static int gemm_f32_uneven_split(const float* A, const float* B, float* C, float* asum, int64_t M, int64_t N, int64_t K, int64_t lda,
                                 int64_t ldb, int64_t ldc, int64_t K1, hipStream_t st);
static long long g_gemm_launches[4] = {0, 0, 0, 0};
static int g_gemm_lockstep = []() { const char* e = getenv("NNHIP_GEMM_LOCKSTEP"); return e && atoi(e) == 1 ? 1 : 0; }();
static int g_gemm_mode = -1;    // -1: not initialised (NNHIP_GEMM_MODE decides), 0: exact fp32 MFMA, 1: split-bf16
static int gemm_mode() {
    if (g_gemm_mode < 0) {
        const char* e = getenv("NNHIP_GEMM_MODE");
        g_gemm_mode = (e && (atoi(e) == 1 || e[0] == 'b')) ? 1 : 0;
    }
    return g_gemm_mode;
}

// gemm_pst.hip: the persistent variant for multi-generation forward GEMMs (stores and first-tile loads hidden in the k-step stream)
bool gemm_pst_wanted(int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc, const float* A, const float* B,
                     const float* C, const float* bias, bool b_kmajor, int act, const float* preact, const float* dswish, int dact);
int gemm_pst(const float* A, const float* B, float* C, const float* bias, float* preact, int64_t M, int64_t N, int64_t K,
             int64_t lda, int64_t ldb, int64_t ldc, bool b_kmajor, float alpha, int act, float beta, const float* dswish, int dact,
             hipStream_t st);
int gemm_small_linear_backward(const float* X, const float* W, const float* dO, float* dX, float* dW, float* db, int64_t rows,
                               int64_t in, int64_t out, const float* addend, const float* dact_arg, int dact, float beta,
                               hipStream_t st);
// gemm_small.hip: the latency-optimised kernel for problems of a few 32x32 tiles
bool gemm_small_wanted(int64_t M, int64_t N, int64_t K, int64_t batch, int64_t lda, int64_t ldb, bool a_kmajor, bool b_kmajor);
int gemm_small(const float* A, const float* B, float* C, const float* bias, float* preact, int64_t M, int64_t N, int64_t K,
               int64_t lda, int64_t ldb, int64_t ldc, bool a_kmajor, bool b_kmajor, float alpha, int act, float beta,
               float* asum, const float* addend, const float* dact_arg, int dact, hipStream_t st);

#ifdef GEMM_PROF
static long long* g_gemm_prof = nullptr;
extern "C" void nnhipGemmSetProfile(long long* buf) { g_gemm_prof = buf; }
#endif
int gemm_f32(const float* A, const float* B, float* C, const float* bias, float* preact, int64_t M,
             int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc, bool a_kmajor,
             bool b_kmajor, int64_t batch, int64_t sA, int64_t sB, int64_t sC, int act, float beta,
             hipStream_t st) {
    return gemm_f32_ex(A, B, C, bias, preact, M, N, K, lda, ldb, ldc, a_kmajor, b_kmajor, batch, sA, sB, sC, 1,
                       0, 0, 0, 1.0f, act, beta, st);
}

// C = A^T-view * B with asum[m] = sum_k A[m,k] produced by the same kernel (A outer-major, no batch):
// Linear backward's dW = dO^T X and db = column sums of dO in one pass over dO.
// C = A B + bias + addend (addend laid out like C): forward residual / dX accumulation onto an existing gradient.
int gemm_f32_add(const float* A, const float* B, float* C, const float* bias, const float* addend, int64_t M, int64_t N,
                 int64_t K, int64_t lda, int64_t ldb, int64_t ldc, bool a_kmajor, bool b_kmajor, hipStream_t st) {
    return gemm_f32_ex(A, B, C, bias, nullptr, M, N, K, lda, ldb, ldc, a_kmajor, b_kmajor, 1, 0, 0, 0, 1, 0, 0, 0, 1.0f,
                       ACT_NONE, 1.f, st, nullptr, addend);
}

// C = (A B) * swish'(Z; beta): the input gradient of a Linear whose input is h = swish(z) comes back as dz (C may alias Z).
// Both gradients of a small Linear in one launch (gemm_small.hip: gemm_small_pair_kernel) when each of the two GEMMs would take
// the small kernel anyway.  Returns 1 when it did the work, 0 when the caller should run the two GEMMs, < 0 on error.
int gemm_f32_linear_backward_small(const float* X, const float* W, const float* dO, float* dX, float* dW, float* db, int64_t rows,
                                   int64_t in, int64_t out, const float* addend, const float* dact_arg, int dact, float beta,
                                   hipStream_t st) {
    static const int small_on = []() { const char* e = getenv("NNHIP_GEMM_SMALL"); return e ? atoi(e) : 1; }();
    static const int pair_on = []() { const char* e = getenv("NNHIP_GEMM_PAIR"); return e ? atoi(e) : 1; }();
    if (!small_on || !pair_on || gemm_mode() != 0 || !dX || !dW || rows <= 0 || in <= 0 || out <= 0) return 0;
    if (!gemm_small_wanted(rows, in, out, 1, out, in, true, false) || !gemm_small_wanted(out, in, rows, 1, out, in, false, false)) return 0;
    const int rc = gemm_small_linear_backward(X, W, dO, dX, dW, db, rows, in, out, addend, dact_arg, dact, beta, st);
    return rc ? rc : 1;
}

// dact 1: C = (A B) * swish'(Z; beta); dact 3: C = (A B) * Z (Z holds the derivative the forward pass saved).  C may alias Z.
int gemm_f32_dswish(const float* A, const float* B, float* C, const float* Z, float beta, int64_t M, int64_t N, int64_t K,
                    int64_t lda, int64_t ldb, int64_t ldc, bool a_kmajor, bool b_kmajor, hipStream_t st, int dact) {
    return gemm_f32_ex(A, B, C, nullptr, nullptr, M, N, K, lda, ldb, ldc, a_kmajor, b_kmajor, 1, 0, 0, 0, 1, 0, 0, 0, 1.0f,
                       ACT_NONE, beta, st, nullptr, nullptr, Z, dact);
}

// C = (A B) * [F > 0]: the input gradient of a Linear whose input was h = relu(z), F = h (C must not alias F).
int gemm_f32_drelu(const float* A, const float* B, float* C, const float* F, int64_t M, int64_t N, int64_t K,
                   int64_t lda, int64_t ldb, int64_t ldc, bool a_kmajor, bool b_kmajor, hipStream_t st) {
    return gemm_f32_ex(A, B, C, nullptr, nullptr, M, N, K, lda, ldb, ldc, a_kmajor, b_kmajor, 1, 0, 0, 0, 1, 0, 0, 0, 1.0f,
                       ACT_NONE, 1.f, st, nullptr, nullptr, F, 2);
}

int gemm_f32_asum(const float* A, const float* B, float* C, float* asum, int64_t M, int64_t N, int64_t K, int64_t lda,
                  int64_t ldb, int64_t ldc, bool b_kmajor, hipStream_t st) {
    return gemm_f32_ex(A, B, C, nullptr, nullptr, M, N, K, lda, ldb, ldc, false, b_kmajor, 1, 0, 0, 0, 1, 0, 0, 0, 1.0f,
                       ACT_NONE, 1.f, st, asum);
}

int gemm_f32_ex(const float* A, const float* B, float* C, const float* bias, float* preact, int64_t M,
                int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc, bool a_kmajor, bool b_kmajor,
                int64_t batch1, int64_t sA, int64_t sB, int64_t sC, int64_t batch2, int64_t sA2,
                int64_t sB2, int64_t sC2, float alpha, int act, float beta, hipStream_t st, float* asum,
                const float* addend, const float* dswish, int dact) {
    const int64_t batch = batch1 * batch2;
    if (M <= 0 || N <= 0 || batch <= 0) return 0;
    if (addend && dswish) { set_last_error("gemm: addend and dswish are mutually exclusive"); return NNHIP_EINVAL; }
    if (asum && (a_kmajor || batch != 1 || K <= 0)) { set_last_error("gemm: asum needs an outer-major, unbatched A"); return NNHIP_EINVAL; }
    if (a_kmajor && batch == 1 && !asum && !addend && gemm_mode() == 0 &&
        gemm_pst_wanted(M, N, K, lda, ldb, ldc, A, B, C, bias, b_kmajor, act, preact, dswish, dact)) {
        ++g_gemm_launches[1];
        return gemm_pst(A, B, C, bias, preact, M, N, K, lda, ldb, ldc, b_kmajor, alpha, act, beta, dswish, dact, st);
    }
    static const int small_on = []() { const char* e = getenv("NNHIP_GEMM_SMALL"); return e ? atoi(e) : 1; }();
    if (small_on && gemm_small_wanted(M, N, K, batch, lda, ldb, a_kmajor, b_kmajor)) {
        ++g_gemm_launches[2];
        return gemm_small(A, B, C, bias, preact, M, N, K, lda, ldb, ldc, a_kmajor, b_kmajor, alpha, act, beta, asum, addend,
                          dswish, dact, st);
    }
    {   // one under-filled generation of a dW-type GEMM: uneven two-way split (gemm_f32_uneven_split)
        static const int uneven_on = []() { const char* e = getenv("NNHIP_GEMM_UNEVEN"); return e ? atoi(e) : 1; }();
        const int64_t t = ceil_div(M, BM) * ceil_div(N, BN);
        if (uneven_on && !a_kmajor && !b_kmajor && batch == 1 && !bias && !preact && !addend && !dswish &&
            act == ACT_NONE && alpha == 1.0f && t > 256 && t <= 496 && K >= 4096 && (K & 7) == 0 && (M & 3) == 0 && (N & 3) == 0 &&
            (lda & 3) == 0 && (ldb & 3) == 0 && aligned16(A) && aligned16(B) && K * lda + 128 < ((int64_t)1 << 30) &&
            K * ldb + 128 < ((int64_t)1 << 30)) {
            // (the split that finishes both kinds together is ~3.4 % below K t / 512: a long block runs a little faster while the
            //  other slot of its CU is between short blocks -- swept on the head's 472 tiles: 456 of 512 k-tiles, not 472)
            static const int bias_tiles = []() { const char* e = getenv("NNHIP_UNEVEN_BIAS"); return e ? atoi(e) : 0; }();   // dev knob
            const int64_t K1 = (K * t * 966 / ((int64_t)512 * 1000) + 31) / 32 * 32 + 32 * bias_tiles;
            if (K - K1 >= 256) {
                ++g_gemm_launches[gemm_mode() == 1 ? 3 : 0];
                return gemm_f32_uneven_split(A, B, C, asum, M, N, K, lda, ldb, ldc, K1, st);
            }
        }
    }
    constexpr int BK = 32;     // (a BK = 16 / 3-blocks-per-CU variant was measured in rounds 1 and 2: never ahead; dropped)
    GemmParams p;
    p.A = A; p.B = B; p.C = C; p.bias = bias; p.preact = preact;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
    p.sA = sA; p.sB = sB; p.sC = sC;
    p.sA2 = sA2; p.sB2 = sB2; p.sC2 = sC2; p.batch2 = (int)batch2; p.alpha = alpha;
    p.tiles_m = (int)ceil_div(M, BM);
    p.tiles_n = (int)ceil_div(N, BN);
    p.lockstep = g_gemm_lockstep;
    p.act = act; p.beta = beta;
    p.splitk = 1; p.k_per_split = ceil_div(K > 0 ? K : 1, BK) * BK; p.slab = nullptr;
    p.asum = asum; p.asum_slab = nullptr; p.addend = addend; p.dswish = dswish; p.dact = dact;
#ifdef GEMM_PROF
    p.prof = g_gemm_prof;
#endif
    p.zeros = zero_block();
    if (!p.zeros) { set_last_error("zero block allocation failed"); return NNHIP_ENOMEM; }

    // split-K (deterministic slabs + reduce): few tiles and a long reduction -- or a handful of tiles and ANY reduction of
    // >= 256: a lone block walks its k-tiles at ~2 us each (one 64-MFMA slab, nothing to hide the load latency behind),
    // so the MNIST-MLP's 32x784x128 forward took 57 us in ONE block; 9 blocks of 3 slabs + the reduce take ~12.
    const int64_t tiles = (int64_t)p.tiles_m * p.tiles_n * batch;
    // (few = up to 64 tiles: 48 lone blocks of a 430 x 512 -> 1536 projection -- the notebook's batch-4 GPT -- walk 16 k-tiles at
    //  ~1.7 us each, 27 us for 0.7 GFLOP; split 8 ways + reduce they take ~12: that workload 242 -> 262 it/s.  16 / 48 / 64 / 96 / 128
    //  swept: 242 / 254 / 262 / 263 / 262.)
    static const int few_tiles = []() { const char* e = getenv("NNHIP_SPLITK_FEW"); return e ? atoi(e) : 64; }();   // dev knob
    const bool few = tiles <= few_tiles && K >= 256;
    static const int tile_lim = []() { const char* e = getenv("NNHIP_SPLITK_TILES"); return e ? atoi(e) : 192; }();
    if (batch == 1 && tiles < tile_lim && (K >= 1024 || few)) {
        static const int slots = []() { const char* e = getenv("NNHIP_SPLITK_SLOTS"); return e ? atoi(e) : 512; }();   // dev knobs
        static const int smax = []() { const char* e = getenv("NNHIP_SPLITK_MAX"); return e ? atoi(e) : 32; }();
        int64_t s = slots / tiles;
        const int64_t max_by_k = few ? K / 64 : K / 256;
        if (s > max_by_k) s = max_by_k;
        if (s > smax) s = smax;
        if (few && K < 1024 && s < 4) s = 1;       // not worth the extra launch
        if (s >= 2) {
            p.k_per_split = ceil_div(ceil_div(K, s), BK) * BK;
            p.splitk = (int)ceil_div(K, p.k_per_split);
            if (p.splitk >= 2) {
                p.slab = static_cast<float*>(workspace((size_t)p.splitk * (M * N + (asum ? M : 0)) * sizeof(float)));
                if (!p.slab) { set_last_error("split-K workspace allocation failed"); return NNHIP_ENOMEM; }
                if (asum) p.asum_slab = p.slab + (int64_t)p.splitk * M * N;
            } else {
                p.splitk = 1; p.k_per_split = ceil_div(K, BK) * BK;
            }
        }
    }

    // float4 global loads need 16-B aligned rows along the contiguous dim
    auto vec_ok = [&](const float* P, int64_t ld, int64_t s1, int64_t s2, bool kmajor, int64_t outer) {
        if (!aligned16(P) || (ld & 3) || (batch1 > 1 && (s1 & 3)) || (batch2 > 1 && (s2 & 3))) return false;
        return kmajor ? ((K & 3) == 0) : ((outer & 3) == 0);
    };
    // the vectorised kernels fetch with `tile base + 32-bit offset`, clamp rows and shift the partial last k-tile back
    // (gemm_common.h): that needs whole k-groups, at least one full tile of k and offsets that fit 32 bits
    auto span_ok = [&](int64_t ld, bool kmajor) {       // every byte offset inside a block's operand window fits 32 bits
        return (kmajor ? 128 * ld + K : K * ld + 128) < ((int64_t)1 << 30);
    };
    const bool fast_ok = (K & 7) == 0 && K >= BK && span_ok(lda, a_kmajor) && span_ok(ldb, b_kmajor);
    const bool vec_any = vec_ok(A, lda, sA, sA2, a_kmajor, M) && vec_ok(B, ldb, sB, sB2, b_kmajor, N);
    const bool vec = fast_ok && vec_any;
    {
        const bool slab = p.splitk > 1;
        bool ok = (N & 3) == 0;
        if (slab) ok = ok && aligned16(p.slab);
        else ok = ok && aligned16(C) && (ldc & 3) == 0 && (batch1 <= 1 || (sC & 3) == 0) && (batch2 <= 1 || (sC2 & 3) == 0) &&
                  (!bias || aligned16(bias)) && (!preact || aligned16(preact)) && (!addend || aligned16(addend)) && (!dswish || aligned16(dswish));
        p.cvec = ok ? 1 : 0;
    }

    int rc;
    ++g_gemm_launches[gemm_mode() == 1 ? 3 : 0];
    if (gemm_mode() == 1) {
        rc = gemm_bf3_launch(p, a_kmajor, b_kmajor, vec_any && (K & 7) == 0 && K >= 16 && span_ok(lda, a_kmajor) && span_ok(ldb, b_kmajor),
                             asum != nullptr, batch, st);
    } else {
#define NNHIP_GEMM_CASE(AK, BKM) \
    rc = vec ? launch_variant<32, AK, BKM, true>(p, batch, st) : launch_variant<32, AK, BKM, false>(p, batch, st)
    if (asum) {
        rc = b_kmajor ? (vec ? launch_variant<32, false, true, true, true>(p, batch, st) : launch_variant<32, false, true, false, true>(p, batch, st))
                      : (vec ? launch_variant<32, false, false, true, true>(p, batch, st) : launch_variant<32, false, false, false, true>(p, batch, st));
    } else
    if (a_kmajor && b_kmajor) { NNHIP_GEMM_CASE(true, true); }
    else if (a_kmajor && !b_kmajor) { NNHIP_GEMM_CASE(true, false); }
    else if (!a_kmajor && b_kmajor) { NNHIP_GEMM_CASE(false, true); }
    else { NNHIP_GEMM_CASE(false, false); }
#undef NNHIP_GEMM_CASE
    }
    if (rc) return rc;

    if (p.splitk > 1) {
        const int64_t total = M * N;
        const int rvec = p.cvec && (ldc & 3) == 0 && aligned16(C) && aligned16(p.slab) && (!addend || aligned16(addend)) &&
                         (!preact || aligned16(preact)) && (!dswish || aligned16(dswish));
        const int64_t work = rvec ? total / 4 : total;
        int blocks = (int)(ceil_div(work, 256) < 2048 ? ceil_div(work, 256) : 2048);
        const int asum_blocks = asum ? (int)ceil_div(M, 256) : 0;
        const ReduceArgs r{p.slab, C, preact, bias, M, N, ldc, p.splitk, act, beta, alpha, p.asum_slab, asum, addend, asum_blocks, rvec,
                           dswish, p.dact};
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks + asum_blocks), dim3(256), 0, st, r);
        NNHIP_LAUNCH_CHECK("splitk_reduce_kernel");
    }
    return 0;
}


// launch a planned group with the kernel of the current GEMM mode (exact fp32 MFMA / split-bf16)
static int launch_group(const GemmGroup& g, int blocks, hipStream_t st) {
    if (gemm_mode() == 1) return gemm_bf3_group_launch(g, blocks, st);
    constexpr int BK = 32;
    constexpr size_t lds = 2 * (Tile<BK, false>::SIZE + Tile<BK, false>::SIZE) * sizeof(float);
    auto kern = gemm_f32_group_kernel<BK>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return hip_status(e, "hipFuncSetAttribute(gemm group)");
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(NT), lds, st, g);
    NNHIP_LAUNCH_CHECK("gemm_f32_group_kernel");
    return 0;
}

// ---- one generation that does not fill the chip: uneven two-way split of the reduction -------------------------------------------
// A dW-type GEMM (both operands outer-major) with t tiles, 256 < t < 512, runs ONE generation on the 512 resident slots (2 per CU):
// every CU with two blocks takes the full K while 512 - t slots idle -- the GPT-tiny head's dW[15000, 512] has 472 tiles and ran at
// 0.83 of the peak.  Here every tile's reduction is cut at K1 = K t / 512: t long blocks (k < K1) are dispatched first, the t short
// ones (k >= K1) follow through the spare slots (t (K - K1) = (512 - t) K1: they are done when the long ones are), and the two
// partial results meet in the split-K reduce.  Same kernel as the grouped launch: the two pieces are two "jobs" on one tile set.
static int gemm_f32_uneven_split(const float* A, const float* B, float* C, float* asum, int64_t M, int64_t N, int64_t K, int64_t lda,
                                 int64_t ldb, int64_t ldc, int64_t K1, hipStream_t st) {
    constexpr int BK = 32;
    const size_t floats = 2 * (size_t)(M * N + (asum ? M : 0));
    float* slab = static_cast<float*>(workspace(floats * sizeof(float)));
    if (!slab) { set_last_error("split-K workspace allocation failed"); return NNHIP_ENOMEM; }
    const float* zeros = zero_block();
    if (!zeros) { set_last_error("zero block allocation failed"); return NNHIP_ENOMEM; }
    float* asum_slab = asum ? slab + 2 * (size_t)(M * N) : nullptr;
    GemmGroup g{};
    g.n = 2;
    g.by_job = 1;
    const int tiles_m = (int)ceil_div(M, BM), tiles_n = (int)ceil_div(N, BN), tiles = tiles_m * tiles_n;
    for (int i = 0; i < 2; ++i) {
        GemmParams& p = g.p[i];
        const int64_t k0 = i ? K1 : 0, kn = i ? K - K1 : K1;
        p.A = A + k0 * lda; p.B = B + k0 * ldb; p.C = C; p.M = M; p.N = N; p.K = kn; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
        p.batch2 = 1; p.alpha = 1.f; p.beta = 1.f; p.tiles_m = tiles_m; p.tiles_n = tiles_n;
        p.k_per_split = ceil_div(kn, BK) * BK;
        p.splitk = 2;                                        // "write a slab": each piece is split 0 of its own slab pointer
        p.slab = slab + (size_t)i * M * N; p.zeros = zeros; p.cvec = 1; p.asum = asum;
        p.asum_slab = asum ? asum_slab + (size_t)i * M : nullptr;
        g.start[i] = i * tiles;
    }
    for (int i = 2; i <= GEMM_GROUP_MAX; ++i) g.start[i] = 2 * tiles;
    if (int rc = launch_group(g, 2 * tiles, st)) return rc;
    const int rvec = (ldc & 3) == 0 && aligned16(C);
    const int64_t work = rvec ? M * N / 4 : M * N;
    const int blocks = (int)(ceil_div(work, 256) < 2048 ? ceil_div(work, 256) : 2048);
    const int asum_blocks = asum ? (int)ceil_div(M, 256) : 0;
    const ReduceArgs r{slab, C, nullptr, nullptr, M, N, ldc, 2, ACT_NONE, 1.f, 1.f, asum_slab, asum, nullptr, asum_blocks, rvec, nullptr, 0};
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks + asum_blocks), dim3(256), 0, st, r);
    NNHIP_LAUNCH_CHECK("splitk_reduce_kernel");
    return 0;
}

// ---- grouped parameter-gradient GEMMs (linear.hip: the deferred dW queue) ------------------------------------------------------
// Job i: C_i[M,N] = sum_k A_i[k][m] B_i[k][n] (dW = dO^T X: A = dO [K, M], B = X [K, N], dense), asum_i[m] = sum_k A_i[k][m] (db) or
// null.  A job's reduction is always cut into FOUR chunks (>= 1024 rows each), whatever else is in the group, so its bits do not
// depend on the company it is launched in (data-parallel and single-process steps flush the queue at different points).  Swept on
// the C4 step (chunk rows x jobs per flush): 2048 x 4 jobs 21.66 ms, 4096 x 4: 21.61, 4096 x 8: 21.58, 8192 x 8: 21.55, 8192 x 4 (384
// blocks, an under-filled generation): 22.99 -- few fat chunks (a quarter of the slab traffic) win as long as a flush has >= ~700
// blocks; four chunks keep a small flush (a DP segment boundary) from falling under that.
static int64_t wgrad_chunk(int64_t K) {
    static const int64_t forced = []() { const char* e = getenv("NNHIP_WGRAD_CHUNK"); return (int64_t)(e ? atoi(e) : 0); }();   // dev knob
    if (forced > 0) return ceil_div(forced, (int64_t)32) * 32;
    const int64_t c = ceil_div(ceil_div(K, (int64_t)4), (int64_t)32) * 32;
    return c < 1024 ? 1024 : c;
}
bool gemm_f32_wgrad_group_ok(const WgradJob& j) {
    static const int on = []() { const char* e = getenv("NNHIP_WGRAD_GROUP_KERNEL"); return e ? atoi(e) : 1; }();
    if (!on) return false;                                   // (both GEMM modes: the group kernel exists for each)
    if (j.M <= 0 || j.N <= 0 || j.K < 4096 || (j.K & 7)) return false;
    if ((j.M & 3) || (j.N & 3) || !aligned16(j.A) || !aligned16(j.B) || !aligned16(j.C)) return false;
    if (j.K * j.M + 128 >= ((int64_t)1 << 30) || j.K * j.N + 128 >= ((int64_t)1 << 30)) return false;     // 32-bit operand offsets
    const int64_t tiles = ceil_div(j.M, BM) * ceil_div(j.N, BN);
    return tiles <= 256;                                    // a GEMM that fills the chip on its own gains nothing from company
}

int gemm_f32_wgrad_group(const WgradJob* jobs, int n, hipStream_t st) {
    if (n <= 0) return 0;
    if (n > GEMM_GROUP_MAX) {
        for (int i = 0; i < n; i += GEMM_GROUP_MAX)
            if (int rc = gemm_f32_wgrad_group(jobs + i, n - i < GEMM_GROUP_MAX ? n - i : GEMM_GROUP_MAX, st)) return rc;
        return 0;
    }
    GemmGroup g{};
    ReduceGroup rg{};
    size_t floats = 0;
    for (int i = 0; i < n; ++i) {
        const WgradJob& j = jobs[i];
        const int64_t kps = wgrad_chunk(j.K);
        floats += (size_t)ceil_div(j.K, kps) * (size_t)(j.M * j.N + (j.asum ? j.M : 0));
    }
    float* slab = static_cast<float*>(workspace_arena(1, floats * sizeof(float)));
    if (!slab) { set_last_error("grouped dW workspace allocation failed"); return NNHIP_ENOMEM; }
    const float* zeros = zero_block();
    if (!zeros) { set_last_error("zero block allocation failed"); return NNHIP_ENOMEM; }
    g.n = rg.n = n;
    int blocks = 0, rblocks = 0;
    for (int i = 0; i < n; ++i) {
        const WgradJob& j = jobs[i];
        GemmParams& p = g.p[i];
        p = GemmParams{};
        p.A = j.A; p.B = j.B; p.C = j.C; p.M = j.M; p.N = j.N; p.K = j.K; p.lda = j.M; p.ldb = j.N; p.ldc = j.N;
        p.batch2 = 1; p.alpha = 1.f; p.beta = 1.f;
        p.tiles_m = (int)ceil_div(j.M, BM); p.tiles_n = (int)ceil_div(j.N, BN);
        p.k_per_split = wgrad_chunk(j.K);
        p.splitk = (int)ceil_div(j.K, p.k_per_split);
        p.slab = slab; p.zeros = zeros; p.cvec = 1; p.asum = j.asum;
        slab += (size_t)p.splitk * j.M * j.N;
        if (j.asum) { p.asum_slab = slab; slab += (size_t)p.splitk * j.M; }
        g.start[i] = blocks;
        blocks += p.tiles_m * p.tiles_n * p.splitk;
        const int64_t work = j.M * j.N / 4;
        const int main_blocks = (int)(ceil_div(work, 256) < 2048 ? ceil_div(work, 256) : 2048);
        const int asum_blocks = j.asum ? (int)ceil_div(j.M, 256) : 0;
        rg.r[i] = ReduceArgs{p.slab, j.C, nullptr, nullptr, j.M, j.N, j.N, p.splitk, ACT_NONE, 1.f, 1.f, p.asum_slab, j.asum, nullptr,
                             asum_blocks, 1, nullptr, 0};
        rg.start[i] = rblocks;
        rblocks += main_blocks + asum_blocks;
    }
    for (int i = n; i <= GEMM_GROUP_MAX; ++i) { g.start[i] = blocks; rg.start[i] = rblocks; }
    g_gemm_launches[gemm_mode() == 1 ? 3 : 0] += n;
    if (int rc = launch_group(g, blocks, st)) return rc;
    hipLaunchKernelGGL(splitk_reduce_group_kernel, dim3((unsigned)rblocks), dim3(256), 0, st, rg);
    NNHIP_LAUNCH_CHECK("splitk_reduce_group_kernel");
    return 0;
}

}  // namespace nnhip

extern "C" int nnhipSetGemmMode(int mode) {
    NNHIP_CHECK_ARG(mode == 0 || mode == 1, NNHIP_EINVAL, "nnhipSetGemmMode: 0 = exact fp32 MFMA, 1 = split-bf16 (bf16x3)");
    nnhip::g_gemm_mode = mode;
    return 0;
}
extern "C" int nnhipGetGemmMode(void) { return nnhip::gemm_mode(); }
extern "C" int nnhipSetGemmLockstep(int enable) {
    NNHIP_CHECK_ARG(enable == 0 || enable == 1, NNHIP_EINVAL, "nnhipSetGemmLockstep: 0 = off (default), 1 = the two blocks of a CU keep step");
    nnhip::g_gemm_lockstep = enable;
    return 0;
}
extern "C" int nnhipGetGemmLockstep(void) { return nnhip::g_gemm_lockstep; }
extern "C" int64_t nnhipGemmLaunchCount(int family) {
    return family >= 0 && family < 4 ? (int64_t)nnhip::g_gemm_launches[family] : -1;
}

extern "C" int nnhipGemmF32Ex(const float* A, const float* B, float* C, const float* bias, int64_t M,
                              int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc, int a_kmajor,
                              int b_kmajor, int64_t batch1, int64_t sA1, int64_t sB1, int64_t sC1,
                              int64_t batch2, int64_t sA2, int64_t sB2, int64_t sC2, float alpha,
                              nnhipStream_t stream) {
    NNHIP_CHECK_ARG(A && B && C, NNHIP_EINVAL, "nnhipGemmF32Ex: null operand");
    NNHIP_CHECK_ARG(M >= 0 && N >= 0 && K >= 0 && batch1 >= 0 && batch2 >= 1, NNHIP_EINVAL, "nnhipGemmF32Ex: bad size");
    NNHIP_CHECK_ARG(batch1 * batch2 <= 65535, NNHIP_EINVAL, "nnhipGemmF32Ex: batch1*batch2 must be <= 65535");
    NNHIP_CHECK_ARG(nnhip::aligned4(A) && nnhip::aligned4(B) && nnhip::aligned4(C), NNHIP_EALIGN,
                    "nnhipGemmF32Ex: pointers must be 4-byte aligned");
    return nnhip::gemm_f32_ex(A, B, C, bias, nullptr, M, N, K, lda, ldb, ldc, a_kmajor != 0, b_kmajor != 0,
                              batch1, sA1, sB1, sC1, batch2, sA2, sB2, sC2, alpha, nnhip::ACT_NONE, 1.f,
                              static_cast<hipStream_t>(stream));
}

extern "C" int nnhipGemmF32(const float* A, const float* B, float* C, const float* bias, int64_t M,
                            int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc,
                            int a_kmajor, int b_kmajor, int64_t batch, int64_t strideA,
                            int64_t strideB, int64_t strideC, nnhipStream_t stream) {
    NNHIP_CHECK_ARG(A && B && C, NNHIP_EINVAL, "nnhipGemmF32: null operand");
    NNHIP_CHECK_ARG(M >= 0 && N >= 0 && K >= 0 && batch >= 0, NNHIP_EINVAL, "nnhipGemmF32: negative size");
    NNHIP_CHECK_ARG(batch <= 65535, NNHIP_EINVAL, "nnhipGemmF32: batch must be <= 65535");
    NNHIP_CHECK_ARG(nnhip::aligned4(A) && nnhip::aligned4(B) && nnhip::aligned4(C), NNHIP_EALIGN,
                    "nnhipGemmF32: pointers must be 4-byte aligned");
    return nnhip::gemm_f32(A, B, C, bias, nullptr, M, N, K, lda, ldb, ldc, a_kmajor != 0,
                           b_kmajor != 0, batch, strideA, strideB, strideC, nnhip::ACT_NONE, 1.f,
                           static_cast<hipStream_t>(stream));
}
