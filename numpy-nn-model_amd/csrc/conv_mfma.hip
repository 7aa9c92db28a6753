// conv_mfma.hip -- Conv2d forward / input gradient / weight gradient as implicit GEMM on the fp32 MFMA, built on the Linear
// GEMM's pipeline (gemm.hip): 4 wave64 per block, 64x64 wave tiles of 2x2 32x32x2 MFMA accumulators, a 2-stage LDS ring, tile
// t+2 in flight in registers while tile t is multiplied and tile t+1 goes to LDS, the issue order of a k-step pinned with
// sched_group_barrier, operands through buffer descriptors (an element that does not exist gets an out-of-range offset and
// reads 0 -- the gather has no branch).  No im2col buffer exists anywhere.
// CPU semantics: neunet/nn/layers/conv2d.py:297-355 (forward: zero-pad, dilate W, einsum "bihwkl,oikl->bohw") and :16-115
// (backward: dW / db / dX einsums).
//
// What makes the gather cheap -- the round-3 kernels (conv_igemm_kernel) stepped a (channel, r, s) counter per gathered element,
// ~530 scalar + ~180 vector instructions per 64 MFMAs, and sat at 0.3-0.4 of the fp32 MFMA peak:
//   * forward / dgrad reduce over k = (tap, source channel) with the TAP OUTERMOST.  Inside a k-tile the tap is fixed, so
//     whether a thread's pixel has a source element at all, and where, is ONE computation per tile; the tile's 16 elements per
//     thread differ only by the channel stride, which rides in the buffer load's SCALAR offset.  The weights are repacked once
//     per call into [tap][M][channels] (zero padded to the tile) -- a plain k-major GEMM operand read with 16-byte loads;
//   * wgrad reduces over k = (image, output pixel); a k-tile is 32 (16) consecutive output pixels of one image, the GEMM's
//     columns are (tap, input channel) with the tap fixed per group of channels: again one validity / address computation per
//     tap slot and tile, channel strides as scalar offsets; dO[b, co, :] rows are k-major and read with 16-byte loads; db is the
//     row sum of the dO elements a thread stages anyway; partial tiles of the K chunks meet in a deterministic reduce.
#include <stdlib.h>

#include "conv_common.h"
#include "gemm_common.h"

namespace nnhip {

constexpr unsigned CV_SENT = 0x80000000u;   // vector offset of an element that does not exist: past num_records (< 2 GiB) -> reads 0

__device__ __forceinline__ __amdgpu_buffer_rsrc_t conv_rsrc(const float* base, int64_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float bload1(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff, 0));
}
__device__ __forceinline__ float4 bload4(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff) {
    const u32x4_ t = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0);
    return make_float4(__uint_as_float(t.x), __uint_as_float(t.y), __uint_as_float(t.z), __uint_as_float(t.w));
}

// One k-step's issue order (gemm.hip, gemm_k_step): the tile's NLD loads ride one per MFMA under the first MFMAs, its NDW LDS
// stores under the last; with fewer MFMAs than slots the last slot of each phase takes the rest.
template <int MASK, int CNT>
__device__ __forceinline__ void conv_sgb() { __builtin_amdgcn_sched_group_barrier(MASK, CNT, 0); }   // (the builtin wants literal constants)
template <int I, int N, int MASK, int LASTCNT>
__device__ __forceinline__ void conv_pin_pairs() {                       // N x { 1 MFMA, 1 MASK-instruction }, the last pair takes LASTCNT
    if constexpr (I < N) {
        conv_sgb<0x008, 1>();
        conv_sgb<MASK, (I == N - 1 ? LASTCNT : 1)>();
        conv_pin_pairs<I + 1, N, MASK, LASTCNT>();
    }
}
template <int NMF, int NLD, int NDW>
__device__ __forceinline__ void conv_pin_pipeline() {
    constexpr int N1 = NLD < NMF / 2 ? NLD : NMF / 2;
    constexpr int N2 = NDW < NMF - N1 ? NDW : NMF - N1;
    conv_pin_pairs<0, N1, 0x020, NLD - N1 + 1>();                          // VMEM reads
    if constexpr (NMF - N1 - N2 > 0) conv_sgb<0x008, NMF - N1 - N2>();
    conv_pin_pairs<0, N2, 0x200, NDW - N2 + 1>();                          // DS writes
}

// Each wave's 64x64 result tile -> rows of float4 through its own LDS region (gemm_common.h, gemm_epilogue: a lane owns a strided
// COLUMN of an MFMA accumulator; direct stores are 4x the instructions).  store(row_in_tile, col4_in_tile, float4)
template <class Store>
__device__ __forceinline__ void conv_store_tile(f32x16 (&acc)[2][2], float* __restrict__ smem, int wave, int lane, int l31, int lh,
                                                Store&& store) {
    constexpr int ELD = 68;
    float* E = smem + wave * (32 * ELD);
    const int er = lane >> 4, ec = (lane & 15) * 4;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int e = 0; e < 16; ++e) E[((e & 3) + 8 * (e >> 2) + 4 * lh) * ELD + n * 32 + l31] = acc[i][n][e];
        __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0); E is private to the wave
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int rl = it * 4 + er;
            store(i * 32 + rl, ec, *reinterpret_cast<const float4*>(&E[rl * ELD + ec]));
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
    }
}

// =================================================================================================================================
// forward / dgrad:  Dst[b, m, pixel] = sum_{tap, c} Wr[tap][m][c] * Src[b, c, source pixel of (pixel, tap)]   (+ bias[m])
//   forward : m = output channel, c = input channel, source = X,  source pixel = (y sh - pu + r dh, x sw - pl + s dw)
//   dgrad   : m = input channel,  c = output channel, source = dO, source pixel = ((y + pu - r dh) / sh, (x + pl - s dw) / sw)
//             where both divisions are exact
// Block tile: WM = 2: 128 channels x 128 pixels, BK = 32;  WM = 1: 64 channels x 256 pixels, BK = 16 (layers of <= 64 channels).
// =================================================================================================================================
struct ConvTapArgs {
    const float* Wr;     // repacked weights [taps][Mp][Csp], zero padded
    const float* Src;    // [B][Cs][Hs][Ws]
    const float* bias;   // [M] or null
    float* Dst;          // [B][M][Hd][Wd]
    int B, M, Mp, Cs, Csp, Hs, Ws, Hd, Wd, kh, kw;
    int sh, sw, dh, dw, pu, pl;
    int tiles_m, tiles_n;
    int cvec;            // Hd*Wd % 4 == 0 and Dst 16-B aligned: float4 stores
    // The grid: `main_blocks` blocks that each own a whole tile (pixel tiles [0, tn0) x all channel tiles), then the tail -- pixel
    // tiles [tn0, tiles_n) with the reduction cut into `splits` ranges of `ktps` k-tiles, one block each, whose partial results go to
    // slab[split][m][n - n_start] (n_start = tn0 * BN, row pitch `scnt`); conv_tap_reduce_kernel adds them (+ bias) into Dst
    int main_blocks, tn0, splits, ktps, scnt;
    float* slab;
};

template <int WM>
struct TapCfg {
    static constexpr int WN = 4 / WM, BM = 64 * WM, BN = 64 * WN, BK = WM == 2 ? 32 : 16;
    static constexpr int ALD = BK + 4, NVA = BM * BK / 1024, KPT = 16;      // KPT: source channels per thread per tile
    // both tiles k-major, rows of BK + 4 floats (conflict-free 16-byte LDS accesses with lane = row): a thread's 16 gathered
    // channels of one pixel are 16 consecutive k of one B row -- four ds_write_b128 -- and the MFMA fragments are ds_read_b128
    static constexpr int A_SIZE = BM * ALD, B_SIZE = BN * ALD, STAGE = A_SIZE + B_SIZE;
    static constexpr int NMF = 16 * (BK / 8), NLD = NVA + KPT, NDW = NVA + KPT / 4;
    static constexpr size_t LDS = (size_t)2 * STAGE * sizeof(float);
};

template <int WM>
__device__ __forceinline__ void tap_k_step(f32x16 (&acc)[2][2], float4 (&fa)[TapCfg<WM>::NVA], float (&fb)[16],
                                           const float4 (&ca)[TapCfg<WM>::NVA], const float (&cb)[16], float* __restrict__ smem, int cur,
                                           __amdgpu_buffer_rsrc_t rsA, __amdgpu_buffer_rsrc_t rsB, const unsigned (&voffA)[TapCfg<WM>::NVA],
                                           unsigned soffA, unsigned voffB, int cbase, int Cs, unsigned HWs4,
                                           const int (&sA)[TapCfg<WM>::NVA], int sB, int wm, int wn, int l31, int lh) {
    using C = TapCfg<WM>;
    // ---- fetch (tile t+2) --------------------------------------------------------------------------------------------------
#pragma unroll
    for (int p = 0; p < C::NVA; ++p) fa[p] = bload4(rsA, voffA[p], soffA);
#pragma unroll
    for (int j = 0; j < C::KPT; ++j) {
        const int c = cbase + j;                                         // uniform
        fb[j] = bload1(rsB, c < Cs ? voffB : CV_SENT, (unsigned)c * HWs4);
    }
    // ---- multiply (tile t, LDS stage cur) ----------------------------------------------------------------------------------
    const float* As = smem + cur * C::STAGE;
    const float* Bs = As + C::A_SIZE;
#pragma unroll
    for (int g = 0; g < C::BK / 8; ++g) {
        float a[2][4], b[2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float4 v = *reinterpret_cast<const float4*>(&As[(wm * 64 + i * 32 + l31) * C::ALD + g * 8 + lh * 4]);
            a[i][0] = v.x; a[i][1] = v.y; a[i][2] = v.z; a[i][3] = v.w;
            const float4 w = *reinterpret_cast<const float4*>(&Bs[(wn * 64 + i * 32 + l31) * C::ALD + g * 8 + lh * 4]);
            b[i][0] = w.x; b[i][1] = w.y; b[i][2] = w.z; b[i][3] = w.w;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int n = 0; n < 2; ++n) acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][j], b[n][j], acc[i][n], 0, 0, 0);
    }
    // ---- commit (tile t+1 -> the other stage) ------------------------------------------------------------------------------
    float* An = smem + (cur ^ 1) * C::STAGE;
    float* Bn = An + C::A_SIZE;
#pragma unroll
    for (int p = 0; p < C::NVA; ++p) *reinterpret_cast<float4*>(&An[sA[p]]) = ca[p];
#pragma unroll
    for (int j = 0; j < C::KPT; j += 4) *reinterpret_cast<float4*>(&Bn[sB + j]) = make_float4(cb[j], cb[j + 1], cb[j + 2], cb[j + 3]);
    conv_pin_pipeline<C::NMF, C::NLD, C::NDW>();
    __syncthreads();
}

template <bool DGRAD, int WM>
__global__ __launch_bounds__(256, WM == 2 ? 2 : 3) void conv_tap_kernel(const ConvTapArgs a) {
    using C = TapCfg<WM>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / C::WN, wn = wave % C::WN, l31 = lane & 31, lh = lane >> 5;
    // m fastest: the tiles_m blocks that share a pixel tile are neighbours on one XCD (they read the same source pixels)
    // (each part of the grid has its own XCD-aware order: blocks go to XCDs round-robin by blockIdx, so both parts stay balanced)
    const bool is_tail = (int)blockIdx.x >= a.main_blocks;
    const int L = is_tail ? xcd_order((int)blockIdx.x - a.main_blocks, (int)gridDim.x - a.main_blocks) : xcd_order((int)blockIdx.x, a.main_blocks);
    const int split = is_tail ? L % a.splits : 0, tL = is_tail ? L / a.splits : L;   // the splits of a tile are neighbours too (same pixels)
    const int tm = tL % a.tiles_m, tn = (is_tail ? a.tn0 : 0) + tL / a.tiles_m;
    const int ktps = is_tail ? a.ktps : 1 << 30;
    const int m0 = tm * C::BM;
    const int64_t n0 = (int64_t)tn * C::BN;
    const int HWd = a.Hd * a.Wd, HWs = a.Hs * a.Ws;
    const int64_t N = (int64_t)a.B * HWd;
    const unsigned HWs4 = (unsigned)HWs * 4u;

    // ---- B side: this thread gathers ONE destination pixel, KPT consecutive source channels per tile -----------------------------
    const int px = tid % C::BN;
    const int khalf = __builtin_amdgcn_readfirstlane(tid / C::BN);       // wave-uniform (BN >= 128)
    const int64_t n = n0 + px;
    const bool n_ok = n < N;
    int b = 0, yd = 0, xd = 0;
    if (n_ok) {
        b = (int)(n / HWd);
        const int rem = (int)(n - (int64_t)b * HWd);
        yd = rem / a.Wd;
        xd = rem - yd * a.Wd;
    }
    const int y0 = DGRAD ? yd + a.pu : yd * a.sh - a.pu;
    const int x0 = DGRAD ? xd + a.pl : xd * a.sw - a.pl;
    const unsigned img = (unsigned)b * (unsigned)a.Cs * (unsigned)HWs;   // element offset of image b, channel 0
    const bool unit = a.sh == 1 && a.sw == 1;
    auto tap_voff = [&](int tap) -> unsigned {                           // tap uniform; once per tile
        const int r = tap / a.kw, s = tap - r * a.kw;
        int ys, xs;
        bool ok;
        if constexpr (!DGRAD) {
            ys = y0 + r * a.dh;
            xs = x0 + s * a.dw;
            ok = (unsigned)ys < (unsigned)a.Hs && (unsigned)xs < (unsigned)a.Ws;
        } else {
            const int ty = y0 - r * a.dh, tx = x0 - s * a.dw;
            if (unit) {
                ys = ty; xs = tx;
                ok = (unsigned)ys < (unsigned)a.Hs && (unsigned)xs < (unsigned)a.Ws;
            } else {
                ys = ty / a.sh; xs = tx / a.sw;
                ok = ty >= 0 && tx >= 0 && ys * a.sh == ty && xs * a.sw == tx && ys < a.Hs && xs < a.Ws;
            }
        }
        return (ok && n_ok && r < a.kh) ? (img + (unsigned)(ys * a.Ws + xs)) * 4u : CV_SENT;
    };
    const __amdgpu_buffer_rsrc_t rsB = conv_rsrc(a.Src, (int64_t)a.B * a.Cs * HWs * 4);
    const int sB = px * C::ALD + khalf * C::KPT;                         // LDS slot of this thread's first element of a tile

    // ---- A side: NVA float4 of the repacked weights per tile ----------------------------------------------------------------------
    const int taps = a.kh * a.kw;
    const __amdgpu_buffer_rsrc_t rsA = conv_rsrc(a.Wr, (int64_t)taps * a.Mp * a.Csp * 4);
    unsigned voffA[C::NVA];
    int sA[C::NVA];
#pragma unroll
    for (int p = 0; p < C::NVA; ++p) {
        const int idx = tid + 256 * p, rr = idx / (C::BK / 4), k4 = (idx % (C::BK / 4)) * 4;
        voffA[p] = (unsigned)(rr * a.Csp + k4) * 4u;
        sA[p] = rr * C::ALD + k4;
    }
    const unsigned tapstrideA = (unsigned)a.Mp * (unsigned)a.Csp * 4u;
    const unsigned rowA = (unsigned)m0 * (unsigned)a.Csp * 4u;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // fetch state: the tile to be fetched NEXT is (ftap, fc0); it runs two tiles ahead of the multiply
    const int ctiles = a.Csp / C::BK;
    const int kt0 = split * (is_tail ? a.ktps : 0);                        // this block's k-tiles: [kt0, kt0 + T)
    const int T = (ktps < taps * ctiles - kt0 ? ktps : taps * ctiles - kt0);
    int ftap = kt0 / ctiles, fc0 = (kt0 - (kt0 / ctiles) * ctiles) * C::BK;
    unsigned fvoff = tap_voff(ftap);
    auto advance = [&]() {
        fc0 += C::BK;
        if (fc0 >= a.Csp) {
            fc0 = 0;
            ++ftap;
            fvoff = tap_voff(ftap);                                       // past the last tap: r >= kh -> nothing to read
        }
    };
    float4 ra[C::NVA], ra2[C::NVA];
    float rb[16], rb2[16];
    auto fetch = [&](float4 (&fa)[C::NVA], float (&fb)[16]) {
        const unsigned soffA = (unsigned)ftap * tapstrideA + rowA + (unsigned)fc0 * 4u;
        const int cbase = fc0 + khalf * C::KPT;
#pragma unroll
        for (int p = 0; p < C::NVA; ++p) fa[p] = bload4(rsA, voffA[p], soffA);
#pragma unroll
        for (int j = 0; j < C::KPT; ++j) fb[j] = bload1(rsB, cbase + j < a.Cs ? fvoff : CV_SENT, (unsigned)(cbase + j) * HWs4);
    };
    // prologue: tile 0 -> stage 0, tile 1 in registers
    fetch(ra, rb);
    advance();
#pragma unroll
    for (int p = 0; p < C::NVA; ++p) *reinterpret_cast<float4*>(&smem[sA[p]]) = ra[p];
#pragma unroll
    for (int j = 0; j < C::KPT; j += 4) *reinterpret_cast<float4*>(&smem[C::A_SIZE + sB + j]) = make_float4(rb[j], rb[j + 1], rb[j + 2], rb[j + 3]);
    fetch(ra, rb);
    advance();
    __syncthreads();
    int cur = 0, t = 0;
    for (; t + 1 < T; t += 2) {
        tap_k_step<WM>(acc, ra2, rb2, ra, rb, smem, cur, rsA, rsB, voffA, (unsigned)ftap * tapstrideA + rowA + (unsigned)fc0 * 4u, fvoff,
                       fc0 + khalf * C::KPT, a.Cs, HWs4, sA, sB, wm, wn, l31, lh);
        advance();
        tap_k_step<WM>(acc, ra, rb, ra2, rb2, smem, cur ^ 1, rsA, rsB, voffA, (unsigned)ftap * tapstrideA + rowA + (unsigned)fc0 * 4u, fvoff,
                       fc0 + khalf * C::KPT, a.Cs, HWs4, sA, sB, wm, wn, l31, lh);
        advance();
    }
    if (t < T) {
        tap_k_step<WM>(acc, ra2, rb2, ra, rb, smem, cur, rsA, rsB, voffA, (unsigned)ftap * tapstrideA + rowA + (unsigned)fc0 * 4u, fvoff,
                       fc0 + khalf * C::KPT, a.Cs, HWs4, sA, sB, wm, wn, l31, lh);
    }

    // ---- epilogue: rows = channels m, columns = pixels n -> Dst[b][m][p]  (split launch: slab[split][m][n - n_start]) -----------------
    const bool to_slab = is_tail;
    const int64_t n_start = (int64_t)a.tn0 * C::BN;
    float* sbase = to_slab ? a.slab + (int64_t)split * a.M * a.scnt : nullptr;
    if (a.cvec) {
        const int64_t col = n0 + wn * 64 + (lane & 15) * 4;              // 4 consecutive pixels of one image (HWd % 4 == 0)
        const bool col_ok = col < N;
        const int cb_ = col_ok ? (int)(col / HWd) : 0;
        float* dst = to_slab ? sbase + (col - n_start) : a.Dst + ((int64_t)cb_ * a.M) * HWd + (col - (int64_t)cb_ * HWd);
        const int64_t pitch = to_slab ? a.scnt : HWd;
        const int rbase = m0 + wm * 64;
        const float* bias = to_slab ? nullptr : a.bias;
        conv_store_tile(acc, smem, wave, lane, l31, lh, [&](int row, int, float4 v) {
            const int m = rbase + row;
            if (m < a.M && col_ok) {
                if (bias) { const float bv = bias[m]; v.x += bv; v.y += bv; v.z += bv; v.w += bv; }
                *reinterpret_cast<float4*>(dst + (int64_t)m * pitch) = v;
            }
        });
        return;
    }
#pragma unroll
    for (int nn = 0; nn < 2; ++nn) {
        const int64_t col = n0 + wn * 64 + nn * 32 + l31;
        if (col >= N) continue;
        const int cb_ = (int)(col / HWd);
        float* dst = to_slab ? sbase + (col - n_start) : a.Dst + ((int64_t)cb_ * a.M) * HWd + (col - (int64_t)cb_ * HWd);
        const int64_t pitch = to_slab ? a.scnt : HWd;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                if (m < a.M) dst[(int64_t)m * pitch] = acc[i][nn][e] + ((a.bias && !to_slab) ? a.bias[m] : 0.f);
            }
    }
}

// Dst[b][m][p] = bias[m] + sum_s slab[s][m][n - n_start]  for the pixels n = b HWd + p >= n_start of a split launch (splits added in
// order: deterministic).  One thread per (m, pixel); consecutive threads = consecutive pixels.
__global__ __launch_bounds__(256) void conv_tap_reduce_kernel(const float* __restrict__ slab, const float* __restrict__ bias,
                                                              float* __restrict__ Dst, int splits, int M, int scnt, int64_t n_start,
                                                              int64_t N, int HWd) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t cnt = N - n_start;
    if (idx >= cnt * M) return;
    const int m = (int)(idx / cnt);
    const int64_t j = idx - (int64_t)m * cnt, n = n_start + j;
    const float* src = slab + (int64_t)m * scnt + j;
    float s = 0.f;
    for (int k = 0; k < splits; ++k) s += src[(int64_t)k * M * scnt];
    if (bias) s += bias[m];
    const int64_t b = n / HWd;
    Dst[(b * M + m) * HWd + (n - b * HWd)] = s;
}

// Wr[tap][m][c] (Mp x Csp per tap, zero padded) from W[Cout][Cin][taps]:  forward m = co, c = ci;  dgrad m = ci, c = co
__global__ __launch_bounds__(256) void conv_repack_kernel(const float* __restrict__ W, float* __restrict__ Wr, int Cout, int Cin, int taps,
                                                          int Mp, int Csp, int dgrad) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total = (int64_t)taps * Mp * Csp;
    if (idx >= total) return;
    const int c = (int)(idx % Csp);
    const int m = (int)((idx / Csp) % Mp);
    const int tap = (int)(idx / ((int64_t)Csp * Mp));
    const int M = dgrad ? Cin : Cout, Cs = dgrad ? Cout : Cin;
    float v = 0.f;
    if (m < M && c < Cs) v = W[((int64_t)(dgrad ? c : m) * Cin + (dgrad ? m : c)) * taps + tap];
    Wr[idx] = v;
}

template <bool DGRAD>
static int launch_conv_tap(const float* W, const float* Src, const float* bias, float* Dst, const ConvGeom& g, hipStream_t st) {
    const int M = DGRAD ? g.Cin : g.Cout, Cs = DGRAD ? g.Cout : g.Cin;
    const int Hs = DGRAD ? g.Ho : g.H, Ws = DGRAD ? g.Wo : g.W, Hd = DGRAD ? g.H : g.Ho, Wd = DGRAD ? g.W : g.Wo;
    const int wm = M > 64 ? 2 : 1;
    const int BM = 64 * wm, BN = 256 / wm, BK = wm == 2 ? 32 : 16;
    const int taps = g.kh * g.kw;
    const int Mp = (int)ceil_div(M, BM) * BM, Csp = (int)ceil_div(Cs, BK) * BK;
    const int64_t wr_floats = (int64_t)taps * Mp * Csp;
    const int64_t src_bytes = (int64_t)g.B * Cs * Hs * Ws * 4;
    // (+ one k-tile of channels: the scalar channel offset of a padded channel, added to the out-of-range marker, must not wrap)
    NNHIP_CHECK_ARG(src_bytes + (int64_t)BK * Hs * Ws * 4 < ((int64_t)1 << 31) && wr_floats * 4 < ((int64_t)1 << 31), NNHIP_EINVAL,
                    "conv2d: the implicit-GEMM kernel addresses each operand with 32-bit byte offsets (tensor >= 2 GiB)");
    const int64_t N = (int64_t)g.B * Hd * Wd;
    const int tiles_m = Mp / BM;
    const int64_t tiles_n = ceil_div(N, BN);
    NNHIP_CHECK_ARG(tiles_n * tiles_m < ((int64_t)1 << 27), NNHIP_EINVAL, "conv2d: too many tiles");

    // ---- plan ---------------------------------------------------------------------------------------------------------------------------
    // Every block does the same work, so a grid runs in rounds of the chip's resident blocks (2 per CU at the 128-row tile, 3 at the
    // 64-row one) and a last round with a handful of tiles costs a whole round (64->128 ch 56x56 B64: 1568 tiles = 3.06 rounds
    // of 512 -> 4; measured 0.62 of the MFMA peak where the blocks themselves run at 0.8).  The tiles of the whole rounds get one
    // block each; the pixel tiles of the ragged last round (or of a grid that never fills the chip: 7x7 feature maps) follow in the
    // SAME grid with the reduction cut `splits` ways along the k-tiles -- short blocks that fill the slots the last full round
    // frees -- each split a partial tile in a slab, added in order by a small reduce.
    static const int cus = []() {
        int d = 0, n = 0;
        return (hipGetDevice(&d) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, d) == hipSuccess && n > 0) ? n : 256;
    }();
    static const bool split_on = []() { const char* e = getenv("NNHIP_CONV_TAIL_SPLIT"); return !e || atoi(e) != 0; }();
    const int64_t slots = (int64_t)cus * (wm == 2 ? 2 : 3);
    const int T = taps * (Csp / BK);
    int64_t tn_main = (((tiles_n * tiles_m) / slots) * slots) / tiles_m;   // pixel tiles of the whole rounds
    int64_t tail = tiles_n - tn_main;                                      // pixel tiles left
    int splits = 1, ktps = T;
    if (split_on && tail > 0) {
        // the cut that finishes the tail soonest: s splits of k = ceil(T / s) k-tiles run in ceil(tail blocks * s / slots) rounds of
        // (k + ~1.5) tile times each (a block's pipeline prologue costs about a tile and a half); a split keeps >= 3 k-tiles
        double best = 1e30;
        for (int sp = 1; sp <= 16 && sp <= (T >= 3 ? T / 3 : 1); ++sp) {
            const int k = (int)ceil_div(T, sp), se = (int)ceil_div(T, k);
            const double cost = (double)ceil_div(tail * tiles_m * se, slots) * (k + 1.5);
            if (cost < best - 1e-9) { best = cost; splits = se; ktps = k; }
        }
    }
    if (splits <= 1) { tn_main = tiles_n; tail = 0; splits = 1; ktps = T; }
    const int64_t n_start = tn_main * BN, cnt = N - n_start, scnt = (cnt + 3) / 4 * 4;
    NNHIP_CHECK_ARG(scnt < ((int64_t)1 << 31), NNHIP_EINVAL, "conv2d: too many pixels in the split tail");

    // ---- one workspace reservation: repacked weights, then the tail's slab ----------------------------------------------------------------
    const size_t wr_pad = ((size_t)wr_floats + 63) / 64 * 64;
    const size_t slab_floats = tail > 0 ? (size_t)splits * M * scnt : 0;
    float* Wr = static_cast<float*>(workspace((wr_pad + slab_floats) * sizeof(float)));
    NNHIP_CHECK_ARG(Wr != nullptr, NNHIP_ENOMEM, "conv2d: workspace allocation failed");
    hipLaunchKernelGGL(conv_repack_kernel, dim3((unsigned)ceil_div(wr_floats, 256)), dim3(256), 0, st, W, Wr, g.Cout, g.Cin, taps, Mp, Csp,
                       DGRAD ? 1 : 0);
    NNHIP_LAUNCH_CHECK("conv_repack_kernel");

    ConvTapArgs a;
    a.Wr = Wr; a.Src = Src; a.bias = bias; a.Dst = Dst;
    a.B = g.B; a.M = M; a.Mp = Mp; a.Cs = Cs; a.Csp = Csp; a.Hs = Hs; a.Ws = Ws; a.Hd = Hd; a.Wd = Wd; a.kh = g.kh; a.kw = g.kw;
    a.sh = g.sh; a.sw = g.sw; a.dh = g.dh; a.dw = g.dw; a.pu = g.pu; a.pl = g.pl;
    a.tiles_m = tiles_m; a.tiles_n = (int)tiles_n;
    a.cvec = ((int64_t)Hd * Wd) % 4 == 0 && aligned16(Dst) ? 1 : 0;
    a.slab = nullptr; a.scnt = 0;
    a.main_blocks = (int)(tn_main * tiles_m);
    a.tn0 = (int)tn_main; a.splits = splits; a.ktps = ktps;
    if (tail > 0) { a.slab = Wr + wr_pad; a.scnt = (int)scnt; }
    const dim3 grid((unsigned)(a.main_blocks + tail * tiles_m * splits));
    if (wm == 2) {
        auto kern = conv_tap_kernel<DGRAD, 2>;
        static bool attr = false;
        if (!attr) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)TapCfg<2>::LDS);
            if (e != hipSuccess) return hip_status(e, "hipFuncSetAttribute(conv_tap_kernel)");
            attr = true;
        }
        hipLaunchKernelGGL(kern, grid, dim3(256), TapCfg<2>::LDS, st, a);
    } else {
        hipLaunchKernelGGL((conv_tap_kernel<DGRAD, 1>), grid, dim3(256), TapCfg<1>::LDS, st, a);
    }
    NNHIP_LAUNCH_CHECK(DGRAD ? "conv_tap_kernel<dgrad>" : "conv_tap_kernel<fwd>");
    if (tail > 0) {
        hipLaunchKernelGGL(conv_tap_reduce_kernel, dim3((unsigned)ceil_div(cnt * M, 256)), dim3(256), 0, st, a.slab, bias, Dst, splits, M,
                           (int)scnt, n_start, N, Hd * Wd);
        NNHIP_LAUNCH_CHECK("conv_tap_reduce_kernel");
    }
    return 0;
}

int conv_mfma_forward(const float* X, const float* W, const float* bias, float* O, const ConvGeom& g, hipStream_t st) {
    return launch_conv_tap<false>(W, X, bias, O, g, st);
}
int conv_mfma_dgrad(const float* dO, const float* W, float* dX, const ConvGeom& g, hipStream_t st) {
    return launch_conv_tap<true>(W, dO, nullptr, dX, g, st);
}

// =================================================================================================================================
// wgrad:  dW[co][ci][tap] = sum_{b, p} dO[b, co, p] * X[b, ci, source pixel of (p, tap)],   db[co] = sum_{b, p} dO[b, co, p]
// GEMM rows = co, columns = (tap slot, channel) with CB channels per slot and TPT = BN / CB taps per column tile, reduction =
// (image, output pixel) cut into `chunks` ranges of k-tiles; every block writes its partial tile to a slab, a reduce adds the
// chunks in order and scatters into dW's [co][ci][tap] layout.
// =================================================================================================================================
struct ConvWgArgs {
    const float* X;      // [B][Cin][H][W]
    const float* dO;     // [B][Cout][Ho][Wo]
    float* slab;         // [chunks][Cout][tiles_n * BN]
    float* bslab;        // [chunks][Cout] (row sums of dO) or null
    int B, Cin, H, W, Cout, Ho, Wo, kh, kw, sh, sw, dh, dw, pu, pl;
    int tiles_m, tiles_n, cblks, chunks, tpc, tpi;     // tpc: k-tiles per chunk, tpi: k-tiles per image
    // FLAT mode (small or ragged feature maps): the reduction index is k = b * Ho*Wo + p with no padding per image; a k-tile is
    // BK consecutive k and may span images (2x2 maps: 8 of them).  ktiles = ceil(B Ho Wo / BK); (mP, sP) / (mW, sW): magic
    // numbers of the exact 32-bit divisions by Ho*Wo and by Wo (udiv_magic)
    int ktiles;
    unsigned mP, sP, mW, sW;
};

// n / d for every 32-bit n, d fixed: (m, s) from magic_u32(d) on the host (Granlund-Montgomery; s == 0 <=> d == 1)
__device__ __forceinline__ unsigned udiv_magic(unsigned n, unsigned m, unsigned s) {
    if (s == 0) return n;
    const unsigned t = __umulhi(m, n);
    return (t + ((n - t) >> 1)) >> (s - 1);
}
static void magic_u32(unsigned d, unsigned& m, unsigned& s) {
    if (d <= 1) { m = 0; s = 0; return; }
    s = 0;
    while ((1ull << s) < d) ++s;
    m = (unsigned)((((unsigned long long)1 << 32) * (((unsigned long long)1 << s) - d)) / d + 1);
}

template <int WM>
struct WgCfg {
    static constexpr int WN = 4 / WM, BM = 64 * WM, BN = 64 * WN, BK = WM == 2 ? 32 : 16;
    static constexpr int ALD = BK + 4, NVA = BM * BK / 1024, NCG = 256 / BK, NPB = 16;   // NPB: gathered elements per thread per tile
    static constexpr int A_SIZE = BM * ALD, B_SIZE = BN * ALD, STAGE = A_SIZE + B_SIZE;
    static constexpr int NMF = 16 * (BK / 8);
    static constexpr size_t LDS = (size_t)2 * STAGE * sizeof(float);
};

// AVEC: dO rows can be read with 16-byte loads (Ho*Wo % 4 == 0, 16-B aligned base); else 4 dword loads per float4
template <int WM, int CB, bool AVEC, bool FLAT>
__global__ __launch_bounds__(256, WM == 2 ? 2 : 3) void conv_wgrad_mfma_kernel(const ConvWgArgs a) {
    using C = WgCfg<WM>;
    constexpr int TPT = C::BN / CB;                      // tap slots per column tile
    constexpr int NLD = (AVEC ? C::NVA : 4 * C::NVA) + C::NPB, NDW = C::NVA + C::NPB;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / C::WN, wn = wave % C::WN, l31 = lane & 31, lh = lane >> 5;
    // tiles fastest: the blocks of one K chunk (they read the same dO / X pixels) are neighbours on one XCD
    const int L = xcd_order((int)blockIdx.x, (int)gridDim.x);
    const int tiles = a.tiles_m * a.tiles_n;
    const int chunk = L / tiles, tt = L - chunk * tiles;
    const int tm = tt % a.tiles_m, tn = tt / a.tiles_m;
    const int tg = tn / a.cblks, cblk = tn - tg * a.cblks;   // tap group, channel block
    const int m0 = tm * C::BM, c0 = cblk * CB;
    const int HWo = a.Ho * a.Wo, HW = a.H * a.W;
    const unsigned HW4 = (unsigned)HW * 4u;

    // ---- this block's k-tiles: [t0, t1) of the flattened (image, tile in image) sequence ----------------------------------------
    const int t0 = chunk * a.tpc;
    const int tend = FLAT ? a.ktiles : a.B * a.tpi;
    const int T = (t0 + a.tpc < tend ? t0 + a.tpc : tend) - t0;
    const unsigned Ktot = (unsigned)a.B * (unsigned)HWo;

    // ---- A side (dO): float4 slots (row rr, pixels k4..k4+3) ---------------------------------------------------------------------
    const __amdgpu_buffer_rsrc_t rsA = conv_rsrc(a.dO, (int64_t)a.B * a.Cout * HWo * 4);
    unsigned voffA[C::NVA];
    int sA[C::NVA], k4A[C::NVA];
#pragma unroll
    for (int p = 0; p < C::NVA; ++p) {
        const int idx = tid + 256 * p, rr = idx / (C::BK / 4), k4 = (idx % (C::BK / 4)) * 4;
        voffA[p] = (unsigned)(rr * HWo + k4) * 4u;
        sA[p] = rr * C::ALD + k4;
        k4A[p] = k4;
    }
    // ---- B side (gathered X): pixel kk of the tile, channel group cg; element i is column cg + NCG i ------------------------------
    const __amdgpu_buffer_rsrc_t rsB = conv_rsrc(a.X, (int64_t)a.B * a.Cin * HW * 4);
    const int kk = tid % C::BK, cg = tid / C::BK;
    const int sB = cg * C::ALD + kk;
    const int ho_i = kk / a.Wo, wo_i = kk - (kk / a.Wo) * a.Wo;          // pixel kk of an image's first tile
    const int step_ho = C::BK / a.Wo, step_wo = C::BK - step_ho * a.Wo;
    int dy[TPT], dx[TPT];                                                 // uniform: this block's tap slots
#pragma unroll
    for (int ts = 0; ts < TPT; ++ts) {
        const int tap = tg * TPT + ts;
        const int r = tap / a.kw, s = tap - r * a.kw;
        dy[ts] = r < a.kh ? r * a.dh - a.pu : -(1 << 24);                 // a slot past the last tap: never in range
        dx[ts] = s * a.dw - a.pl;
    }

    // fetch state (two tiles ahead of the multiply): image fb_, first pixel fp0, this thread's pixel (fho, fwo)
    int fb_ = FLAT ? 0 : t0 / a.tpi, fp0 = FLAT ? 0 : (t0 - (t0 / a.tpi) * a.tpi) * C::BK;
    int fho = (fp0 + kk) / a.Wo, fwo = (fp0 + kk) - ((fp0 + kk) / a.Wo) * a.Wo;
    unsigned fk0 = (unsigned)t0 * C::BK;                                  // FLAT: first k of the tile to fetch
    auto advance = [&]() {
        if constexpr (FLAT) { fk0 += C::BK; return; }
        fp0 += C::BK;
        fwo += step_wo;
        fho += step_ho;
        if (fwo >= a.Wo) { fwo -= a.Wo; ++fho; }
        if (fp0 >= HWo) { fp0 = 0; ++fb_; fho = ho_i; fwo = wo_i; }
    };
    float4 ra[C::NVA], ra2[C::NVA];
    float rb[C::NPB], rb2[C::NPB];
    float cs[C::NVA];
#pragma unroll
    for (int p = 0; p < C::NVA; ++p) cs[p] = 0.f;
    auto fetch = [&](float4 (&fa)[C::NVA], float (&fbv)[C::NPB]) {
        if constexpr (FLAT) {
            // ---- A: this thread's float4 slots all sit at the same k4 (256 threads = a multiple of BK/4 slots per row) ----------------
            const unsigned ka = fk0 + (unsigned)k4A[0];
            const unsigned ba = udiv_magic(ka, a.mP, a.sP), pa = ka - ba * (unsigned)HWo;
            const unsigned soffA = (unsigned)m0 * (unsigned)HWo * 4u;
            if constexpr (AVEC) {                                         // Ho*Wo % 4 == 0: the four pixels are in one image
                const unsigned va = ka < Ktot ? (ba * (unsigned)a.Cout * (unsigned)HWo + pa) * 4u : CV_SENT;
#pragma unroll
                for (int p = 0; p < C::NVA; ++p) fa[p] = bload4(rsA, ka < Ktot ? va + voffA[p] - (unsigned)k4A[0] * 4u : CV_SENT, soffA);
            } else {
                unsigned be = ba, pe = pa;
                unsigned ve[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    ve[e] = ka + e < Ktot ? (be * (unsigned)a.Cout * (unsigned)HWo + pe) * 4u : CV_SENT;
                    if (++pe == (unsigned)HWo) { pe = 0; ++be; }
                }
#pragma unroll
                for (int p = 0; p < C::NVA; ++p) {
                    float t4[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) t4[e] = bload1(rsA, ve[e] == CV_SENT ? CV_SENT : ve[e] + voffA[p] - (unsigned)k4A[0] * 4u, soffA);
                    fa[p] = make_float4(t4[0], t4[1], t4[2], t4[3]);
                }
            }
            // ---- B: pixel k = fk0 + kk -> (image, ho, wo) with two exact magic divisions --------------------------------------------
            const unsigned kb = fk0 + (unsigned)kk;
            const unsigned bb = udiv_magic(kb, a.mP, a.sP), pb = kb - bb * (unsigned)HWo;
            const unsigned hb = udiv_magic(pb, a.mW, a.sW), wb = pb - hb * (unsigned)a.Wo;
            unsigned vo[TPT];
            const bool p_ok = kb < Ktot;
            const unsigned imgb = (bb * (unsigned)a.Cin + (unsigned)cg) * (unsigned)HW;
#pragma unroll
            for (int ts = 0; ts < TPT; ++ts) {
                const int ys = (int)hb * a.sh + dy[ts], xs = (int)wb * a.sw + dx[ts];
                const bool ok = p_ok && (unsigned)ys < (unsigned)a.H && (unsigned)xs < (unsigned)a.W;
                vo[ts] = ok ? (imgb + (unsigned)(ys * a.W + xs)) * 4u : CV_SENT;
            }
            const unsigned soffB = (unsigned)c0 * HW4;
#pragma unroll
            for (int i = 0; i < C::NPB; ++i) {
                const int col = C::NCG * i;
                fbv[i] = bload1(rsB, vo[col / CB], soffB + (unsigned)(col % CB) * HW4);
            }
            return;
        }
        const int remk = HWo - fp0;                                       // pixels of this image left from the tile's first one
        const unsigned soffA = (unsigned)((fb_ * a.Cout + m0) * HWo + fp0) * 4u;
#pragma unroll
        for (int p = 0; p < C::NVA; ++p) {
            if constexpr (AVEC) {
                fa[p] = bload4(rsA, k4A[p] < remk ? voffA[p] : CV_SENT, soffA);
            } else {
                float t4[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) t4[e] = bload1(rsA, k4A[p] + e < remk ? voffA[p] + 4u * e : CV_SENT, soffA);
                fa[p] = make_float4(t4[0], t4[1], t4[2], t4[3]);
            }
        }
        unsigned vo[TPT];
        const bool p_ok = kk < remk && fb_ < a.B;
#pragma unroll
        for (int ts = 0; ts < TPT; ++ts) {
            const int ys = fho * a.sh + dy[ts], xs = fwo * a.sw + dx[ts];
            const bool ok = p_ok && (unsigned)ys < (unsigned)a.H && (unsigned)xs < (unsigned)a.W;
            vo[ts] = ok ? (unsigned)((cg * a.H + ys) * a.W + xs) * 4u : CV_SENT;
        }
        const unsigned soffB = (unsigned)(fb_ * a.Cin + c0) * HW4;
#pragma unroll
        for (int i = 0; i < C::NPB; ++i) {
            const int col = C::NCG * i;                                    // compile-time: slot and channel offset of element i
            fbv[i] = bload1(rsB, vo[col / CB], soffB + (unsigned)(col % CB) * HW4);
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // wt = 1 for a tile of this block's range, 0 for the one fetched past its end (the pipeline commits one tile ahead): the row
    // sums of dO count every pixel exactly once across the chunks
    auto commit = [&](const float4 (&ca)[C::NVA], const float (&cbv)[C::NPB], int stage, float wt) {
        float* An = smem + stage * C::STAGE;
        float* Bn = An + C::A_SIZE;
#pragma unroll
        for (int p = 0; p < C::NVA; ++p) {
            *reinterpret_cast<float4*>(&An[sA[p]]) = ca[p];
            cs[p] = fmaf(wt, (ca[p].x + ca[p].y) + (ca[p].z + ca[p].w), cs[p]);
        }
#pragma unroll
        for (int i = 0; i < C::NPB; ++i) Bn[sB + (C::NCG * i) * C::ALD] = cbv[i];
    };
    auto mma = [&](int stage) {
        const float* As = smem + stage * C::STAGE;
        const float* Bs = As + C::A_SIZE;
#pragma unroll
        for (int g = 0; g < C::BK / 8; ++g) {
            float av[2][4], bv[2][4];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float4 v = *reinterpret_cast<const float4*>(&As[(wm * 64 + i * 32 + l31) * C::ALD + g * 8 + lh * 4]);
                av[i][0] = v.x; av[i][1] = v.y; av[i][2] = v.z; av[i][3] = v.w;
                const float4 w = *reinterpret_cast<const float4*>(&Bs[(wn * 64 + i * 32 + l31) * C::ALD + g * 8 + lh * 4]);
                bv[i][0] = w.x; bv[i][1] = w.y; bv[i][2] = w.z; bv[i][3] = w.w;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int n = 0; n < 2; ++n) acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][j], bv[n][j], acc[i][n], 0, 0, 0);
        }
    };

    if (T > 0) {
        fetch(ra, rb);
        advance();
        commit(ra, rb, 0, 1.f);
        fetch(ra, rb);
        advance();
        __syncthreads();
        int cur = 0, t = 0;
        for (; t + 1 < T; t += 2) {
            fetch(ra2, rb2);
            mma(cur);
            commit(ra, rb, cur ^ 1, 1.f);                                  // tile t + 1 < T
            conv_pin_pipeline<C::NMF, NLD, NDW>();
            __syncthreads();
            advance();
            fetch(ra, rb);
            mma(cur ^ 1);
            commit(ra2, rb2, cur, t + 2 < T ? 1.f : 0.f);                  // tile t + 2
            conv_pin_pipeline<C::NMF, NLD, NDW>();
            __syncthreads();
            advance();
        }
        if (t < T) {
            mma(cur);
            __syncthreads();
        }
    }
    // ---- partial tile -> slab[chunk][m][tn * BN + col] ---------------------------------------------------------------------------------
    const int ncols = a.tiles_n * C::BN;
    float* slab = a.slab + ((int64_t)chunk * a.Cout) * ncols + (int64_t)tn * C::BN;
    const int rbase = m0 + wm * 64, cbase = wn * 64;
    conv_store_tile(acc, smem, wave, lane, l31, lh, [&](int row, int col4, float4 v) {
        const int m = rbase + row;
        if (m < a.Cout) *reinterpret_cast<float4*>(slab + (int64_t)m * ncols + cbase + col4) = v;
    });
    if (a.bslab && tn == 0) {
        __syncthreads();
        float* red = smem;                                                 // [BM][BK/4]
#pragma unroll
        for (int p = 0; p < C::NVA; ++p) {
            const int idx = tid + 256 * p;
            red[idx] = cs[p];                                              // idx = rr * (BK/4) + k4/4
        }
        __syncthreads();
        if (tid < C::BM && m0 + tid < a.Cout) {
            float s = 0.f;
#pragma unroll
            for (int q = 0; q < C::BK / 4; ++q) s += red[tid * (C::BK / 4) + q];
            a.bslab[(int64_t)chunk * a.Cout + m0 + tid] = s;
        }
    }
}

// dW[co][ci][tap] = sum_chunks slab[c][co][column of (tap, ci)];  db[co] = sum_chunks bslab[c][co].  A block = 32 float4 column
// groups x 8 chunk lanes: lane l adds chunks l, l + 8, ... (coalesced 512-byte slab rows, 8x the threads of one-thread-per-column:
// the slab is 30 MB and 80 blocks could not pull it at HBM speed -- 24.6 us), the 8 lane sums meet in LDS in lane order.  Fixed
// order: deterministic.  The results are scattered into dW's [co][ci][tap] layout (dW is small).
__global__ __launch_bounds__(256) void conv_wgrad_mfma_reduce_kernel(const float* __restrict__ slab, const float* __restrict__ bslab,
                                                                     float* __restrict__ dW, float* __restrict__ db, int chunks, int Cout,
                                                                     int Cin, int taps, int ncols, int BN, int CB, int cblks, int w_blocks) {
    if ((int)blockIdx.x >= w_blocks) {                                     // the db part: 32 channels x 8 chunk lanes per block
        __shared__ float bred[8][32];
        const int cl = threadIdx.x >> 5, co = ((int)blockIdx.x - w_blocks) * 32 + (threadIdx.x & 31);
        float s = 0.f;
        if (co < Cout && bslab) {
            int c = cl;                                                    // (a one-thread chain of `chunks` dependent L2 round trips took 20 us)
            for (; c + 24 < chunks; c += 32) {
                const float v0 = bslab[(int64_t)c * Cout + co], v1 = bslab[(int64_t)(c + 8) * Cout + co];
                const float v2 = bslab[(int64_t)(c + 16) * Cout + co], v3 = bslab[(int64_t)(c + 24) * Cout + co];
                s += v0; s += v1; s += v2; s += v3;
            }
            for (; c < chunks; c += 8) s += bslab[(int64_t)c * Cout + co];
        }
        bred[cl][threadIdx.x & 31] = s;
        __syncthreads();
        if (cl == 0 && co < Cout && db && bslab) {
#pragma unroll
            for (int l = 1; l < 8; ++l) s += bred[l][threadIdx.x & 31];
            db[co] = s;
        }
        return;
    }
    if (!dW) return;
    __shared__ float4 red[8][32];
    const int n4 = ncols / 4;
    const int cl = threadIdx.x >> 5, qq = threadIdx.x & 31;
    const int64_t idx = (int64_t)blockIdx.x * 32 + qq;
    const bool live = idx < (int64_t)Cout * n4;
    const int co = live ? (int)(idx / n4) : 0, q = live ? (int)(idx - (int64_t)co * n4) : 0;
    const float* src = slab + (int64_t)co * ncols + q * 4;
    const int64_t cstride = (int64_t)Cout * ncols;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live) {
        int c = cl;
        for (; c + 24 < chunks; c += 32) {
            const float4 v0 = *reinterpret_cast<const float4*>(src + (int64_t)c * cstride);
            const float4 v1 = *reinterpret_cast<const float4*>(src + (int64_t)(c + 8) * cstride);
            const float4 v2 = *reinterpret_cast<const float4*>(src + (int64_t)(c + 16) * cstride);
            const float4 v3 = *reinterpret_cast<const float4*>(src + (int64_t)(c + 24) * cstride);
            s.x += v0.x; s.y += v0.y; s.z += v0.z; s.w += v0.w;
            s.x += v1.x; s.y += v1.y; s.z += v1.z; s.w += v1.w;
            s.x += v2.x; s.y += v2.y; s.z += v2.z; s.w += v2.w;
            s.x += v3.x; s.y += v3.y; s.z += v3.z; s.w += v3.w;
        }
        for (; c < chunks; c += 8) {
            const float4 v = *reinterpret_cast<const float4*>(src + (int64_t)c * cstride);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    }
    red[cl][qq] = s;
    __syncthreads();
    if (cl != 0 || !live) return;
#pragma unroll
    for (int l = 1; l < 8; ++l) { const float4 v = red[l][qq]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
    const float out[4] = {s.x, s.y, s.z, s.w};
    const int TPT = BN / CB;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int n = q * 4 + e, tn = n / BN, col = n - tn * BN;
        const int tg = tn / cblks, cblk = tn - tg * cblks;
        const int tap = tg * TPT + col / CB, ci = cblk * CB + col % CB;
        if (tap < taps && ci < Cin) dW[((int64_t)co * Cin + ci) * taps + tap] = out[e];
    }
}

template <int WM, int CB, bool AVEC, bool FLAT>
static int launch_wgrad_one(const ConvWgArgs& a, dim3 grid, hipStream_t st) {
    using C = WgCfg<WM>;
    auto kern = conv_wgrad_mfma_kernel<WM, CB, AVEC, FLAT>;
    static bool attr = false;
    if (!attr && C::LDS > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS);
        if (e != hipSuccess) return hip_status(e, "hipFuncSetAttribute(conv wgrad)");
        attr = true;
    }
    hipLaunchKernelGGL(kern, grid, dim3(256), C::LDS, st, a);
    NNHIP_LAUNCH_CHECK("conv_wgrad_mfma_kernel");
    return 0;
}
template <int WM, int CB>
static int launch_wgrad_cfg(const ConvWgArgs& a, bool avec, bool flat, dim3 grid, hipStream_t st) {
    if (flat) return avec ? launch_wgrad_one<WM, CB, true, true>(a, grid, st) : launch_wgrad_one<WM, CB, false, true>(a, grid, st);
    return avec ? launch_wgrad_one<WM, CB, true, false>(a, grid, st) : launch_wgrad_one<WM, CB, false, false>(a, grid, st);
}

int conv_mfma_wgrad(const float* X, const float* dO, float* dW, float* db, const ConvGeom& g, hipStream_t st) {
    if (!dW && !db) return 0;
    const int wm = g.Cout > 64 ? 2 : 1;
    const int BM = 64 * wm, BN = 256 / wm, BK = wm == 2 ? 32 : 16;
    // channels per tap slot: the smallest power of two >= Cin, at least 32, at most the tile's width
    int CB = 32;
    while (CB < g.Cin && CB < BN) CB *= 2;
    const int TPT = BN / CB, taps = g.kh * g.kw;
    const int cblks = (int)ceil_div(g.Cin, CB), tgroups = (int)ceil_div(taps, TPT);
    const int HWo = g.Ho * g.Wo;
    // (+ one tile of channels / rows: scalar offsets of padded channels, added to the out-of-range marker, must not wrap)
    NNHIP_CHECK_ARG(((int64_t)g.B * g.Cin + BN) * g.H * g.W * 4 < ((int64_t)1 << 31) && ((int64_t)g.B * g.Cout + BM) * HWo * 4 < ((int64_t)1 << 31),
                    NNHIP_EINVAL, "conv2d: the implicit-GEMM kernel addresses each operand with 32-bit byte offsets (tensor >= 2 GiB)");
    ConvWgArgs a;
    a.X = X; a.dO = dO;
    a.B = g.B; a.Cin = g.Cin; a.H = g.H; a.W = g.W; a.Cout = g.Cout; a.Ho = g.Ho; a.Wo = g.Wo; a.kh = g.kh; a.kw = g.kw;
    a.sh = g.sh; a.sw = g.sw; a.dh = g.dh; a.dw = g.dw; a.pu = g.pu; a.pl = g.pl;
    a.tiles_m = (int)ceil_div(g.Cout, BM);
    a.cblks = cblks;
    a.tiles_n = dW ? tgroups * cblks : 1;                                  // db alone: one column tile carries the row sums
    a.tpi = (int)ceil_div(HWo, BK);
    // per-image k-tiles pad every image to a multiple of BK pixels (2x2 maps: 4 of 32 used); when that wastes more than ~3 % the
    // reduction index is flattened over (image, pixel) instead (FLAT: exact divisions by Ho*Wo and Wo per tile, no padding)
    static const int flat_mode = []() { const char* e = getenv("NNHIP_CONV_WGRAD_FLAT"); return e ? atoi(e) : -1; }();   // dev knob: 0 never, 1 always
    const int64_t Ktot = (int64_t)g.B * HWo;
    const bool flat = flat_mode >= 0 ? flat_mode != 0 : (int64_t)a.tpi * BK * 32 > (int64_t)HWo * 33;
    a.ktiles = (int)ceil_div(Ktot, BK);
    magic_u32((unsigned)HWo, a.mP, a.sP);
    magic_u32((unsigned)g.Wo, a.mW, a.sW);
    const int64_t ktiles = flat ? a.ktiles : (int64_t)g.B * a.tpi;
    const int tiles = a.tiles_m * a.tiles_n;
    // one generation of the chip's 512 resident blocks (2 per CU), cut along K; a chunk has at least 4 k-tiles (8: -15 % on the
    // 8x8 ... 16x16 maps of a U-Net body, whose grids do not fill the chip otherwise).  The cut depends on
    // the layer only, not on which outputs were asked for: db alone sums the same chunks in the same order as db next to dW
    int64_t chunks = 512 / (a.tiles_m * tgroups * cblks);
    if (chunks < 1) chunks = 1;
    static const int min_kt = []() { const char* e = getenv("NNHIP_CONV_WGRAD_MINKT"); const int v = e ? atoi(e) : 4; return v < 1 ? 1 : v; }();
    if (chunks > ceil_div(ktiles, min_kt)) chunks = ceil_div(ktiles, min_kt);
    a.tpc = (int)ceil_div(ktiles, chunks);
    a.chunks = (int)ceil_div(ktiles, a.tpc);
    const int ncols = a.tiles_n * BN;
    const size_t slab_floats = (size_t)a.chunks * g.Cout * ncols, b_floats = db ? (size_t)a.chunks * g.Cout : 0;
    float* ws = static_cast<float*>(workspace((slab_floats + b_floats) * sizeof(float)));
    NNHIP_CHECK_ARG(ws != nullptr, NNHIP_ENOMEM, "conv2d wgrad: workspace allocation failed");
    a.slab = ws;
    a.bslab = db ? ws + slab_floats : nullptr;
    const bool avec = HWo % 4 == 0 && aligned16(dO);
    const dim3 grid((unsigned)(tiles * a.chunks));
    int rc;
    if (wm == 2) {
        rc = CB == 32 ? launch_wgrad_cfg<2, 32>(a, avec, flat, grid, st) : CB == 64 ? launch_wgrad_cfg<2, 64>(a, avec, flat, grid, st)
                                                                              : launch_wgrad_cfg<2, 128>(a, avec, flat, grid, st);
    } else {
        rc = CB == 32 ? launch_wgrad_cfg<1, 32>(a, avec, flat, grid, st) : CB == 64 ? launch_wgrad_cfg<1, 64>(a, avec, flat, grid, st)
             : CB == 128 ? launch_wgrad_cfg<1, 128>(a, avec, flat, grid, st) : launch_wgrad_cfg<1, 256>(a, avec, flat, grid, st);
    }
    if (rc) return rc;
    const int w_blocks = dW ? (int)ceil_div((int64_t)g.Cout * (ncols / 4), 32) : 0;
    const int b_blocks = db ? (int)ceil_div(g.Cout, 32) : 0;
    hipLaunchKernelGGL(conv_wgrad_mfma_reduce_kernel, dim3((unsigned)(w_blocks + b_blocks)), dim3(256), 0, st, a.slab, a.bslab, dW, db, a.chunks,
                       g.Cout, g.Cin, taps, ncols, BN, CB, cblks, w_blocks);
    NNHIP_LAUNCH_CHECK("conv_wgrad_mfma_reduce_kernel");
    return 0;
}

}  // namespace nnhip
