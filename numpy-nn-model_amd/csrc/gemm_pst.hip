// gemm_pst.hip -- persistent variant of the exact-fp32 MFMA GEMM for grids of several generations of tiles
// (C = alpha A B^T + bias, both operands k-major: the Linear FORWARD layout, neunet/nn/layers/linear.py:48-58).
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
// Conditions (gemm_pst_wanted): both operands k-major and 16-B aligned rows, K % 32 == 0, K >= 64 (96 with Swish),
// M % 128 == 0, bias-only or Swish epilogue, no split-K / batch, C and the operand windows addressable with 32-bit byte
// offsets, more tiles than resident slots.  Everything else takes gemm.hip.
#include <stdlib.h>

#include <type_traits>

#include "gemm_common.h"

namespace nnhip {

constexpr int PBK = 32;
using PTile = Tile<PBK, true>;
constexpr int PSTAGE = 2 * PTile::SIZE;                        // A tile + B tile, floats

struct PstParams {
    const float* A; const float* B; float* C; const float* bias;
    float* preact;                                             // EPI 1: z output [M, ldc] or null
    int64_t M, N, K, lda, ldb, ldc;
    float alpha, beta;
    int tiles_m, tiles_n, total;
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

// the stores of row half i (32 rows per wave) of the pending tile.  accumulator register e of tile (i, n): row
// i*32 + (e&3) + 8(e>>2) (+ 4 lh + 64 wm: in vo), column n*32 + l31 (+ 64 wn: in vo)
template <int EPI>
__device__ __forceinline__ void pst_flush_half(int i, const f32x16 (&pend)[2][2], const PstStore& ps, __amdgpu_buffer_rsrc_t rc,
                                               __amdgpu_buffer_rsrc_t rz, unsigned ldc4, float alpha, float beta) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const unsigned so = ps.soff + (unsigned)(i * 32 + (e & 3) + 8 * (e >> 2)) * ldc4;
        float v0 = fmaf(alpha, pend[i][0][e], ps.b0), v1 = fmaf(alpha, pend[i][1][e], ps.b1);
        if constexpr (EPI == 1) {
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v0), rz, ps.vo0, so, 0);   // rz has 0 records without a z output
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v1), rz, ps.vo1, so, 0);
            v0 *= sigmoid_fast_(beta * v0);
            v1 *= sigmoid_fast_(beta * v1);
        }
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v0), rc, ps.vo0, so, 0);
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v1), rc, ps.vo1, so, 0);
    }
}

// one k-step: fetch the next k-tile (whatever tile it belongs to) into (fa, fb), multiply LDS stage `cur`, [flush stores of
// the pending tile, one per MFMA,] commit (fa, fb) to the other stage.  FLUSH: bit i = row half i of the pending tile goes out
// in this step.  (A two-deep variant -- a second register set, the k-tile after next in flight -- was measured: no change at
// K = 512, level with gemm_f32_kernel at K = 4096; not kept.)
template <int EPI, int FLUSH>
__device__ __forceinline__ void pst_step(f32x16 (&acc)[2][2], float4 (&fa)[4], float4 (&fb)[4],
                                         float* __restrict__ smem, int cur, __amdgpu_buffer_rsrc_t rsa, __amdgpu_buffer_rsrc_t rsb,
                                         unsigned koff, const unsigned (&offa)[4], const unsigned (&offb)[4], int tid, int wm, int wn,
                                         int l31, int lh, const f32x16 (&pend)[2][2], const PstStore& ps, __amdgpu_buffer_rsrc_t rc,
                                         __amdgpu_buffer_rsrc_t rz, unsigned ldc4, float alpha, float beta) {
    g2r_fast<PBK>(fa, rsa, koff, offa);
    g2r_fast<PBK>(fb, rsb, koff, offb);
    const float* As = smem + cur * PSTAGE;
    const float* Bs = As + PTile::SIZE;
#pragma unroll
    for (int g = 0; g < PBK / 8; ++g) {
        float a[2][4], b[2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            frag<PBK, true>(a[i], As, wm * 64 + i * 32 + l31, g, lh);
            frag<PBK, true>(b[i], Bs, wn * 64 + i * 32 + l31, g, lh);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int n = 0; n < 2; ++n)
                    acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][j], b[n][j], acc[i][n], 0, 0, 0);
    }
    if constexpr ((FLUSH & 1) != 0) pst_flush_half<EPI>(0, pend, ps, rc, rz, ldc4, alpha, beta);
    if constexpr ((FLUSH & 2) != 0) pst_flush_half<EPI>(1, pend, ps, rc, rz, ldc4, alpha, beta);
    {
        float* Sn = smem + (cur ^ 1) * PSTAGE;
        r2s<PBK, true>(fa, Sn, tid);
        r2s<PBK, true>(fb, Sn + PTile::SIZE, tid);
    }
    // issue order: 8 loads under the first MFMAs, 8 LDS stores under the last ones; FLUSH: a store after every MFMA (the Swish
    // epilogue's arithmetic goes wherever the scheduler likes: fp32 MFMAs and VALU share the lanes, it overlaps with nothing)
    if constexpr (FLUSH != 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x040, 1, 0);
        }
#pragma unroll
        for (int i = 0; i < 48; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x040, 1, 0);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x040, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        }
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 48, 0);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        }
    }
    __syncthreads();
}

template <int EPI>
__global__ __launch_bounds__(NT, 2) void gemm_pst_kernel(const PstParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, lh = lane >> 5;
    const int nk = (int)(p.K / PBK);
    const unsigned la4 = (unsigned)(4 * p.lda), lb4 = (unsigned)(4 * p.ldb), ldc4 = (unsigned)(4 * p.ldc);

    // tile-independent per-thread offsets: (row * ld + k4) * 4 for the 4 float4 a thread stages per operand tile
    unsigned offa[4], offb[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int idx = tid + NT * q;
        offa[q] = (unsigned)(idx >> 3) * la4 + (unsigned)(idx & 7) * 16u;
        offb[q] = (unsigned)(idx >> 3) * lb4 + (unsigned)(idx & 7) * 16u;
    }
    // descriptor of an operand's rows [r0, r0 + 128) clipped to R rows: loads of rows past the end return 0
    auto rs_of = [&](const float* P, int64_t ld, int64_t R, int64_t r0) {
        const int64_t rows = min((int64_t)128, R - r0);
        return pst_rsrc(P + r0 * ld, (unsigned)(rows * ld * 4));
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
            for (int e = 0; e < 16; ++e) { acc[i][n][e] = 0.f; pend[i][n][e] = 0.f; }

    int vb = blockIdx.x;
    int tm, tn;
    pst_tile(vb, p.total, p.tiles_m, p.tiles_n, tm, tn);
    __amdgpu_buffer_rsrc_t rsa = rs_of(p.A, p.lda, p.M, (int64_t)tm * 128), rsb = rs_of(p.B, p.ldb, p.N, (int64_t)tn * 128);
    float4 ra[4], rb[4];
    g2r_fast<PBK>(ra, rsa, 0u, offa);
    g2r_fast<PBK>(rb, rsb, 0u, offb);
    r2s<PBK, true>(ra, smem, tid);
    r2s<PBK, true>(rb, smem + PTile::SIZE, tid);
    __syncthreads();

    int cur = 0;
    PstStore ps = store_of(tm, tn);                               // placeholder until a tile is pending
    // one tile of the block's k-step stream.  PEND: a finished tile is waiting in `pend`; its stores go out under the first
    // k-step (EPI 1, twice the stores: one row half under each of the first two).  Returns false after the block's last tile.
    // (The first tile is peeled instead of testing a `have_pend` flag inside one loop: with the flag the kernel took 236 VGPRs --
    // and spilled with the two-step flush; peeled it takes 152 / 208.)
    auto run_tile = [&](auto pend_tag) -> bool {
        constexpr bool PEND = decltype(pend_tag)::value;
        constexpr int F0 = PEND ? (EPI == 1 ? 1 : 3) : 0, F1 = (PEND && EPI == 1) ? 2 : 0;
        const int vbn = vb + (int)gridDim.x;
        const bool has_next = vbn < p.total;
        int tmn = tm, tnn = tn;
        if (has_next) pst_tile(vbn, p.total, p.tiles_m, p.tiles_n, tmn, tnn);
        const __amdgpu_buffer_rsrc_t rsan = rs_of(p.A, p.lda, p.M, (int64_t)tmn * 128), rsbn = rs_of(p.B, p.ldb, p.N, (int64_t)tnn * 128);
        const PstStore mine = store_of(tm, tn);
#define PST_STEP(F, RA, RB, KOFF) \
    pst_step<EPI, F>(acc, ra, rb, smem, cur, RA, RB, KOFF, offa, offb, tid, wm, wn, l31, lh, pend, ps, rc, rz, ldc4, p.alpha, p.beta); \
    cur ^= 1
        int kt = 1;
        PST_STEP(F0, rsa, rsb, 128u);
        if constexpr (EPI == 1) {
            PST_STEP(F1, rsa, rsb, 256u);
            kt = 2;
        }
        for (; kt + 1 < nk; ++kt) { PST_STEP(0, rsa, rsb, (unsigned)(kt + 1) * 128u); }
        // last k-step: fetches the first k-tile of the next tile (or, with nothing left, re-reads this one's: never used)
        PST_STEP(0, rsan, rsbn, 0u);
#undef PST_STEP
        // the finished tile becomes the pending one
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                pend[i][n] = acc[i][n];
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][n][e] = 0.f;
            }
        ps = mine;
        vb = vbn; tm = tmn; tn = tnn; rsa = rsan; rsb = rsbn;
        return has_next;
    };
    if (run_tile(std::false_type{}))
        while (run_tile(std::true_type{})) {}
    // the block's last tile: nothing left to hide its stores under
    pst_flush_half<EPI>(0, pend, ps, rc, rz, ldc4, p.alpha, p.beta);
    pst_flush_half<EPI>(1, pend, ps, rc, rz, ldc4, p.alpha, p.beta);
}

// ---- host ---------------------------------------------------------------------------------------------------------
template <int EPI>
static int pst_slots_of() {
    int per_cu = 0, dev = 0;
    hipDeviceProp_t prop;
    const size_t lds = 2 * PSTAGE * sizeof(float);
    const void* k = reinterpret_cast<const void*>(gemm_pst_kernel<EPI>);
    if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess || hipGetDevice(&dev) != hipSuccess ||
        hipGetDeviceProperties(&prop, dev) != hipSuccess || hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k, NT, lds) != hipSuccess)
        return -1;
    return per_cu > 0 ? prop.multiProcessorCount * per_cu : -1;
}

// resident blocks of the whole chip (LDS admits two per CU for both variants), a multiple of the 8 XCDs
static int pst_slots() {
    static const int slots = []() {
        const int s = min(pst_slots_of<0>(), pst_slots_of<1>());
        return s > 0 ? (s & ~7) : -1;
    }();
    return slots;
}

// NNHIP_GEMM_PST: 0 = never, 1 (default) = when the conditions hold, 2 = also for long reductions
bool gemm_pst_wanted(int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc, const float* A, const float* B,
                     const float* C, const float* bias, int act, const float* preact) {
    static const int on = []() { const char* e = getenv("NNHIP_GEMM_PST"); return e ? atoi(e) : 1; }();
    if (!on) return false;
    if (act != ACT_NONE && act != ACT_SWISH) return false;
    if (act == ACT_NONE && preact) return false;
    // short reductions only (on = 2 lifts that: developer switch): measured on MI355X, 16384x512->15000 135 -> 141 TFLOP/s and
    // 16384x512->2048 129 -> 133, but K = 4096 shapes -2 % -- there the fixed cost is 1.5 % of a tile and gemm.hip's two-deep
    // prefetch is worth more (with a two-deep prefetch of its own this kernel draws level there, no better: not kept)
    if (on < 2 && K > 1024) return false;
    // (the Swish epilogue's second flush step needs a third k-tile)
    if ((K % PBK) != 0 || K < (act == ACT_SWISH ? 3 : 2) * PBK || (M % 128) != 0 || N <= 0) return false;
    if (!aligned16(A) || !aligned16(B) || (lda & 3) || (ldb & 3) ||
        ((reinterpret_cast<uintptr_t>(C) | reinterpret_cast<uintptr_t>(preact) | reinterpret_cast<uintptr_t>(bias)) & 3)) return false;
    // (C's byte size is the store descriptor's num_records and must stay below the 0xFFFFFFF0 offset that drops a lane)
    if (M * ldc * 4 >= (int64_t)0xFFFF0000 || 128 * lda * 4 + K * 4 >= ((int64_t)1 << 32) || 128 * ldb * 4 + K * 4 >= ((int64_t)1 << 32)) return false;
    const int slots = pst_slots();
    if (slots <= 0) return false;
    const int64_t tiles = (M / 128) * ceil_div(N, 128);
    return tiles > slots;
}

int gemm_pst(const float* A, const float* B, float* C, const float* bias, float* preact, int64_t M, int64_t N, int64_t K,
             int64_t lda, int64_t ldb, int64_t ldc, float alpha, int act, float beta, hipStream_t st) {
    PstParams p;
    p.A = A; p.B = B; p.C = C; p.bias = bias; p.preact = preact; p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
    p.alpha = alpha; p.beta = beta;
    p.tiles_m = (int)(M / 128); p.tiles_n = (int)ceil_div(N, 128); p.total = p.tiles_m * p.tiles_n;
    const size_t lds = 2 * PSTAGE * sizeof(float);
    const dim3 grid((unsigned)pst_slots()), block(NT);
    if (act == ACT_SWISH) hipLaunchKernelGGL(gemm_pst_kernel<1>, grid, block, lds, st, p);
    else hipLaunchKernelGGL(gemm_pst_kernel<0>, grid, block, lds, st, p);
    NNHIP_LAUNCH_CHECK("gemm_pst_kernel");
    return 0;
}

}  // namespace nnhip
