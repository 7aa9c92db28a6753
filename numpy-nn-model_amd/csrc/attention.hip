// attention.hip -- fused (flash-style) masked multi-head attention for gfx950, fp32 MFMA, head dim 32 / 64 / 128.
// Semantics = examples/gpt.ipynb cell 2:  scores = QK^T / sqrt(d_model); scores = where(mask == 0, -1e9, scores);
// attn = Softmax(-1)(scores); attn = dropout(attn); ctx = attn V.
// The mask is either (key padding AND causal) -- what the notebook's get_pad_mask & get_sub_mask always produce, carried
// as key_valid[B,Tk] + a flag -- or ANY dense [B,Tq,Tk] mask, handed over as packed bits (nnhipAttentionPackMask).
// The [B,H,T,T] score / attention matrices are never written: the forward keeps a running (max, sum) per query row and
// stores only LSE[B,H,T,2] = (max, log2 sum) in log2 units; the backward recomputes P from it.  Attention dropout
// (neunet/nn/layers/dropout.py:17-37 applied to the attention map) happens inside the kernels: the keep/scale multiplier
// of element (b,h,q,k) is either read from an injected [B,H,Tq,Tk] mask (parity tests against the oracle) or a
// counter-based hash of (seed, b, h, q, k) that the forward and both backward kernels re-evaluate -- never stored.
//
// Register layout trick (all three kernels): every product is arranged so that the softmax axis lands on the
// LANE index of the 32x32 MFMA accumulator (col = lane & 31) and the other axis on the register index:
//   S^T[key, q] = K Q^T     -> lane <-> query q, registers <-> keys: row max / row sum are in-lane reductions
//                              plus ONE cross-half exchange (lanes l and l+32 hold the two key halves);
//   O^T[d, q]  += V^T P^T   -> the B operand of step e is exactly accumulator register e of P^T (lanes 0-31
//                              carry key (e&3)+8(e>>2), lanes 32-63 that key + 4), so P feeds the next MFMA
//                              straight from registers, and the per-query rescale factor is per-lane.
// The MFMA phases read their A operands from LDS one step ahead of the MFMAs that consume them and pin that issue
// order with sched_group_barrier (left alone, hipcc re-uses one register pair and waits lgkmcnt(0) before every
// MFMA pair -- measured 2.3x off the MFMA bound).  K/V (Q/dO) tiles are fetched into registers one tile ahead.
// Q, K, V, ctx live in the [B, T, H*DH] layout of the projections (row stride D = H*DH).
// Kernels are templates on <DH, GEN>: GEN = false is the (key_valid, causal, no dropout) fast path that round 1 tuned at
// DH = 64; GEN = true adds the dense-mask bits and the dropout multiplier to the softmax phase.
#include <math.h>
#include <stdlib.h>

#include "common.h"
#include "attention.h"

namespace nnhip {

// A [rows x DH] tile staged through registers: fetch early (the loads stay in flight during the MFMA phase), commit
// to LDS (row stride DH + 4) at the top of the next iteration.  The per-thread element offsets are computed once
// (tile_offsets); a fetch is then NV loads off a wave-uniform tile pointer -- the per-tile 64-bit address arithmetic and
// the per-row exec-mask branches of round 1 cost ~1500 cycles per tile of a VALU that the fp32 MFMAs share.
// Rows past the end of the tensor re-read its last row (finite data whose scores are masked / whose probabilities are
// zero downstream), so no lane is ever switched off.
template <int NV>
struct TileRegsN { float4 v[NV]; };          // NV*4*NT/DH rows x DH floats over NT threads
template <int NV>
struct TileOffs { unsigned o[NV]; };
template <int DH, int NT, int NV>
__device__ __forceinline__ void tile_offsets(TileOffs<NV>& t, int64_t D, int tid) {
    constexpr int C4 = DH / 4;
#pragma unroll
    for (int p = 0; p < NV; ++p) {
        const int idx = tid + NT * p;
        t.o[p] = (unsigned)(idx / C4) * (unsigned)D + (unsigned)(idx % C4) * 4u;
    }
}
// One buffer descriptor over rows [0, nrows) of this (batch, head) slice, the tile's first row as the SCALAR offset, the thread's
// (row-in-tile, column) offset computed once (TileOffs): a fetch is NV `buffer_load_dwordx4 v, v_off, s[rsrc], s_row0 offen` with no
// vector instruction at all, and rows past the end read 0 through the descriptor's bounds check (finite data whose scores are
// masked / whose probabilities are zero downstream) -- no clamping code, no zero default, no branch.  (Round 3: the per-tile fetch
// was 14 % of a heavy forward block's life -- 64-bit address pairs and register defaults next to the other wave's fp32 MFMAs,
// whose lanes the vector ALU shares.)  The host refuses slices beyond 4 GiB of byte offsets.
typedef unsigned at_u32x4 __attribute__((ext_vector_type(4)));
template <int DH, int NT, int NV>
__device__ __forceinline__ void tile_fetch(TileRegsN<NV>& r, const TileOffs<NV>& t, const float* __restrict__ base, int64_t D, int row0,
                                           int nrows, int tid) {
    const unsigned bytes = nrows > 0 ? (unsigned)(((int64_t)(nrows - 1) * D + DH) * 4) : 0u;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, (int)bytes, 0x00020000);
    const unsigned soff = (unsigned)row0 * (unsigned)D * 4u;            // wave-uniform
#pragma unroll
    for (int p = 0; p < NV; ++p) {
        const at_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, t.o[p] * 4u, soff, 0);
        r.v[p].x = __uint_as_float(v.x); r.v[p].y = __uint_as_float(v.y); r.v[p].z = __uint_as_float(v.z); r.v[p].w = __uint_as_float(v.w);
    }
}
// The pointer form (64-bit address per load, clamped rows): kept for the dK/dV kernel, which is out of scalar registers -- two more
// descriptors there spill (36 B/lane).
template <int DH, int NT, int NV>
__device__ __forceinline__ void tile_fetch_ptr(TileRegsN<NV>& r, const TileOffs<NV>& t, const float* __restrict__ base, int64_t D, int row0,
                                           int nrows, int tid) {
    constexpr int C4 = DH / 4, ROWS = NV * 4 * NT / DH;
    const float* __restrict__ tb = base + (int64_t)row0 * D;          // wave-uniform
    unsigned off[NV];
#pragma unroll
    for (int p = 0; p < NV; ++p) off[p] = t.o[p];
    if (row0 + ROWS > nrows) {                                        // last, partial tile (wave-uniform)
        const int last = nrows - 1 - row0;                            // >= 0: callers only fetch tiles that start inside the tensor
#pragma unroll
        for (int p = 0; p < NV; ++p) {
            const int idx = tid + NT * p;
            off[p] = (unsigned)min(idx / C4, last) * (unsigned)D + (unsigned)(idx % C4) * 4u;
        }
    }
    // (the zero default + wave-uniform guard keep the tile in registers: an unconditional load-into-struct /
    //  store-from-struct pair is turned into memcpys through a stack slot by hipcc -- 144 B/lane of scratch)
#pragma unroll
    for (int p = 0; p < NV; ++p) r.v[p] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row0 < nrows) {
#pragma unroll
        for (int p = 0; p < NV; ++p) r.v[p] = *reinterpret_cast<const float4*>(tb + off[p]);
    }
}
template <int DH, int NT, int NV>
__device__ __forceinline__ void tile_commit(float* __restrict__ S, const TileRegsN<NV>& r, int tid) {
    constexpr int C4 = DH / 4, LD = DH + 4;
#pragma unroll
    for (int p = 0; p < NV; ++p) {
        const int idx = tid + NT * p;
        *reinterpret_cast<float4*>(&S[(idx / C4) * LD + (idx % C4) * 4]) = r.v[p];
    }
}
// padding flag of key kv0 + lane (1 = real token); the wave ballots it into a 64-bit mask per tile
__device__ __forceinline__ int key_flag(const int32_t* __restrict__ kv, int kv0, int Tk, int lane) {
    if (!kv) return 1;
    if (Tk <= 0) return 0;                                   // (no keys at all: nothing to clamp to)
    const int i = kv0 + lane;
    const int v = kv[i < Tk ? i : Tk - 1];                   // unconditional load of a clamped index: no exec-masked region (a
    return i < Tk ? v : 0;                                   // conditional load is its own basic block and may end in vmcnt(0))
}

// first key index of batch b that is not padding (block-wide min; Tk if none) -- decides where causal skipping is legal:
// a query q has a visible unmasked key iff first_valid <= q + shift; a row without one is uniform over ALL keys.
template <int NT>
__device__ __forceinline__ int first_valid_key_scan(const int32_t* __restrict__ kv, int Tk, int tid, int* sh) {
    int best = Tk;
    for (int j = tid; j < Tk; j += NT)
        if (kv[j] != 0) { best = j; break; }
    if (tid == 0) *sh = Tk;
    __syncthreads();
    atomicMin(sh, best);
    __syncthreads();
    return *sh;
}
// `flag0`: key_flag(kv, 0, Tk, lane) of the calling lane -- the same 64 values in every wave.  Almost always the first real token
// is among keys 0..63 (no left padding): then every wave reads it off its own ballot -- no scan loop (a dependent load), no LDS
// atomic, no block barriers in the prologue.  Otherwise all waves take the block-wide scan (the decision is block-uniform).
template <int NT>
__device__ __forceinline__ int first_valid_key(const int32_t* __restrict__ kv, int Tk, int tid, int* sh, int flag0) {
    if (!kv) return 0;
    const unsigned long long m = __ballot(flag0 != 0);
    if (m != 0ull) return (int)__builtin_ctzll(m);
    return first_valid_key_scan<NT>(kv, Tk, tid, sh);
}

// Block -> (batch*head, tile block) map.  Hardware deals consecutive blockIdx round-robin over the 8 XCDs, so a plain
// (bh, blk) = (id / nblk, id % nblk) order puts every light causal block on the even XCDs and every heavy one on the
// odd XCDs.  Here every XCD gets the same mix and walks it heaviest first (all its heaviest tile blocks, then the
// next lighter ones, ...): with 2 resident blocks per CU an alternating heavy/light order left some CUs with three
// heavy blocks out of four (measured 91 us causal vs 97 us non-causal; balanced would be 73).
// Grid = ceil(BH/8)*8*nblk; returns false for padding blocks.
__device__ __forceinline__ bool map_block(int id, int BH, int nblk, bool heavy_is_high, int& bh, int& blk) {
    const int x = id & 7, r = id >> 3;
    const int nbx = (BH + 7) >> 3;
    const int j = r / nbx;
    bh = (r - j * nbx) * 8 + x;
    blk = heavy_is_high ? nblk - 1 - j : j;
    return bh < BH;
}
// (Round 2, measured and dropped: a mirrored wave -> row-group map in every other block, so that a SIMD's two resident waves
//  carry complementary causal work: +-0; starting every slot with one heavy and one light block so that later prologues run
//  under somebody else's MFMA phase: 80 -> 84 us.  See EXPERIMENTS.md 5.8.)
__host__ inline unsigned mapped_grid(int64_t BH, int64_t nblk) { return (unsigned)(ceil_div(BH, 8) * 8 * nblk); }

// ---- MFMA phases -------------------------------------------------------------------------------------------------
// accA += TA_rows . fA^T and accB += TB_rows . fB^T (8*G MFMAs, G = DH/8): the A operands are 32 rows of a k-major LDS
// tile (ta / tb = this lane's row pointer, &T[(r0 + l31) * LD + 4 * lh]), the B operands per-lane register fragments
// f[g][j] = X[lane's row][8g + 4lh + j].  One ds_read_b128 per product per g, issued one g ahead of its 4 MFMAs.
template <int G>
__device__ __forceinline__ void mma2_rows(f32x16& accA, const float* __restrict__ ta, const float (&fA)[G][4],
                                          f32x16& accB, const float* __restrict__ tb, const float (&fB)[G][4]) {
    float4 a[2], c[2];
    a[0] = *reinterpret_cast<const float4*>(ta);
    c[0] = *reinterpret_cast<const float4*>(tb);
#pragma unroll
    for (int g = 0; g < G; ++g) {
        if (g + 1 < G) {
            a[(g + 1) & 1] = *reinterpret_cast<const float4*>(ta + 8 * (g + 1));
            c[(g + 1) & 1] = *reinterpret_cast<const float4*>(tb + 8 * (g + 1));
        }
        const float4 x = a[g & 1], y = c[g & 1];
        accA = AT_MFMA(x.x, fA[g][0], accA);
        accB = AT_MFMA(y.x, fB[g][0], accB);
        accA = AT_MFMA(x.y, fA[g][1], accA);
        accB = AT_MFMA(y.y, fB[g][1], accB);
        accA = AT_MFMA(x.z, fA[g][2], accA);
        accB = AT_MFMA(y.z, fB[g][2], accB);
        accA = AT_MFMA(x.w, fA[g][3], accA);
        accB = AT_MFMA(y.w, fB[g][3], accB);
    }
    // both reads of group g+1 go out in the first half of group g (a read issued right before its first use showed up
    // as an lgkmcnt(0) stall in front of every other MFMA group)
    AT_SCHED_DSRD(2);
#pragma unroll
    for (int g = 0; g < G - 1; ++g) {
        AT_SCHED_MFMA(2);
        AT_SCHED_DSRD(1);
        AT_SCHED_MFMA(2);
        AT_SCHED_DSRD(1);
        AT_SCHED_MFMA(4);
    }
    AT_SCHED_MFMA(8);
}

// one product of the pair above (4*G MFMAs)
template <int G>
__device__ __forceinline__ void mma1_rows(f32x16& acc, const float* __restrict__ ta, const float (&f)[G][4]) {
    float4 a[2];
    a[0] = *reinterpret_cast<const float4*>(ta);
#pragma unroll
    for (int g = 0; g < G; ++g) {
        if (g + 1 < G) a[(g + 1) & 1] = *reinterpret_cast<const float4*>(ta + 8 * (g + 1));
        const float4 x = a[g & 1];
        acc = AT_MFMA(x.x, f[g][0], acc);
        acc = AT_MFMA(x.y, f[g][1], acc);
        acc = AT_MFMA(x.z, f[g][2], acc);
        acc = AT_MFMA(x.w, f[g][3], acc);
    }
    AT_SCHED_DSRD(1);
#pragma unroll
    for (int g = 0; g < G - 1; ++g) {
        AT_SCHED_MFMA(2);
        AT_SCHED_DSRD(1);
        AT_SCHED_MFMA(2);
    }
    AT_SCHED_MFMA(4);
}

// acc[dt] += T^T[d = 32dt + l31][row(e)] * P[e] over the 32 rows carried by accumulator P (16*DT MFMAs, DT = DH/32): the
// A operand of step e is T[(r0 + acc_row(e, lh)) * LD + 32dt + l31] (tl = &T[(r0 + 4lh) * LD + l31]), the B operand
// register e of P.  LDS values are read 4 e-steps ahead.
template <int DT, int LD>
__device__ __forceinline__ void mma_cols(f32x16 (&acc)[DT], const float* __restrict__ tl, const f32x16& P) {
    float v[2][4][DT];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) v[0][i][dt] = tl[((i & 3) + 8 * (i >> 2)) * LD + 32 * dt];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        if (c + 1 < 4) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int e = 4 * (c + 1) + i;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) v[(c + 1) & 1][i][dt] = tl[((e & 3) + 8 * (e >> 2)) * LD + 32 * dt];
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) acc[dt] = AT_MFMA(v[c & 1][i][dt], P[4 * c + i], acc[dt]);
    }
    if constexpr (DT == 2) {
        AT_SCHED_DSRD(4);                 // each (d, d+32) pair is one ds_read2_b32
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                AT_SCHED_MFMA(2);
                AT_SCHED_DSRD(1);
            }
        AT_SCHED_MFMA(8);
    }
}

// two such products sharing the pipeline: accA += TA^T PA, accB += TB^T PB; LDS read 2 e-steps ahead.
template <int DT, int LD>
__device__ __forceinline__ void mma2_cols(f32x16 (&accA)[DT], const float* __restrict__ ta, const f32x16& PA,
                                          f32x16 (&accB)[DT], const float* __restrict__ tb, const f32x16& PB) {
    float va[2][2][DT], vb[2][2][DT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) { va[0][i][dt] = ta[i * LD + 32 * dt]; vb[0][i][dt] = tb[i * LD + 32 * dt]; }
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        if (c + 1 < 8) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int e = 2 * (c + 1) + i;
                const int off = ((e & 3) + 8 * (e >> 2)) * LD;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) { va[(c + 1) & 1][i][dt] = ta[off + 32 * dt]; vb[(c + 1) & 1][i][dt] = tb[off + 32 * dt]; }
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                accA[dt] = AT_MFMA(va[c & 1][i][dt], PA[2 * c + i], accA[dt]);
                accB[dt] = AT_MFMA(vb[c & 1][i][dt], PB[2 * c + i], accB[dt]);
            }
    }
    if constexpr (DT == 2) {
        AT_SCHED_DSRD(4);
#pragma unroll
        for (int c = 0; c < 7; ++c)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                AT_SCHED_MFMA(2);
                AT_SCHED_DSRD(1);
            }
        AT_SCHED_MFMA(8);
    }
}

// per-wave transposed store: acc[dt][e] = X^T[d = 32dt + acc_row(e, lh)][row = l31]  ->  dst rows of DH contiguous floats
template <int DH>
__device__ __forceinline__ void store_transposed(float* __restrict__ E, const f32x16 (&acc)[DH / 32], float mul, float* __restrict__ dst,
                                                 int64_t D, int row0, int nrows, int lane) {
    constexpr int LD = DH + 4, C4 = DH / 4;
    const int l31 = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int dt = 0; dt < DH / 32; ++dt)
#pragma unroll
        for (int e = 0; e < 16; ++e) E[l31 * LD + dt * 32 + acc_row(e, lh)] = acc[dt][e] * mul;
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the staging area is private to the wave
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int it = 0; it < DH / 8; ++it) {
        const int idx = it * 64 + lane;
        const int r = idx / C4, c4 = (idx % C4) * 4;
        if (row0 + r < nrows)
            *reinterpret_cast<float4*>(dst + (int64_t)(row0 + r) * D + c4) = *reinterpret_cast<const float4*>(&E[r * LD + c4]);
    }
}

// visible-key bits of this lane's query for the 64 keys of tile t, already shifted so that bit kl <-> local key kl + 4lh
__device__ __forceinline__ unsigned long long dense_bits(const AttnExtra& x, int b, int q, int Tq, int Tk, int t, int lh) {
    const int nw = (Tk + 63) >> 6;
    const unsigned long long w = q < Tq ? x.mask_bits[((int64_t)b * Tq + q) * nw + t] : 0ull;
    return w >> (4 * lh);
}

// =====================================================================================================
// forward: block = 128 queries (4 waves x 32), loop over 64-key tiles
// =====================================================================================================
template <int DH, bool GEN, int NW>
__global__ __launch_bounds__(64 * NW, DH <= 64 ? 2 : 1) void attn_fwd_kernel(const AttnParams p) {
    constexpr int NT = 64 * NW, BQ = 32 * NW;
    constexpr int LD = DH + 4, G = DH / 8, DT = DH / 32, NV = AT_BK * DH / (4 * NT);
    // one LDS block: [K tile 64 x LD | V tile 64 x LD]; the epilogue re-uses it as [4 waves][32 q][LD]
    constexpr int SM_FLOATS = 2 * AT_BK * LD > NW * 32 * LD ? 2 * AT_BK * LD : NW * 32 * LD;
    __shared__ __attribute__((aligned(16))) float smem[SM_FLOATS];
    __shared__ int sh_fv;
    float* Ks = smem;
    float* Vs = smem + AT_BK * LD;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int qblocks = (p.Tq + BQ - 1) / BQ;
    // pair mode: a block takes query block (qblocks - 1 - k) and then its causal complement k -- every block the same amount of
    // work, the whole grid resident at once, no second round of block launches behind the heavy blocks
    const int nmap = p.pair ? (qblocks + 1) / 2 : qblocks;
    int bh, blk;
    if (!map_block(blockIdx.x, p.B * p.H, nmap, true, bh, blk)) return;
    const int kpair = nmap - 1 - blk;
    const int qb_first = p.pair ? qblocks - 1 - kpair : blk, qb_second = (p.pair && kpair != qblocks - 1 - kpair) ? kpair : -1;
    auto item = [&](const int qb) {
    AT_PROF_DECL;
    const int b = bh / p.H, h = bh - b * p.H;
    const int q0 = qb * BQ + wave * 32;   // this wave's first query
    const int q = q0 + l31;                           // this lane's query
    const float* Qb = p.Q + ((int64_t)b * p.Tq) * p.LQ + (int64_t)h * DH;
    const float* Kb = p.K + ((int64_t)b * p.Tk) * p.LQ + (int64_t)h * DH;
    const float* Vb = p.V + ((int64_t)b * p.Tk) * p.LQ + (int64_t)h * DH;
    const int32_t* kv = p.key_valid ? p.key_valid + (int64_t)b * p.Tk : nullptr;
    const int shift = p.Tk - p.Tq;                    // causal: key j visible to query i iff j <= i + shift
    const bool dense = GEN && p.x.mask_bits != nullptr;
    const bool dropping = GEN && (p.x.drop_mask != nullptr || p.x.drop_threshold != 0u);
    const int64_t drow = ((int64_t)b * p.H + h) * p.Tq + q;             // row of the [B,H,Tq,Tk] dropout mask
    const unsigned seed = GEN ? p.x.drop_seed + at_seed_offset(p.x) : 0u;
    const unsigned rowkey = GEN ? at_rowkey(seed, (unsigned)drow) : 0u;

    // Every global load of the prologue is issued before the first wait: the first K/V tile (every block visits tile 0
    // unless it visits none), the Q fragments, the key-padding scan -- one memory round trip instead of three.
    TileRegsN<NV> kr, vr;
    TileOffs<NV> toff;
    tile_offsets<DH, NT>(toff, p.LQ, tid);
    int kflag = 1;
    tile_fetch<DH, NT>(kr, toff, Kb, p.LQ, 0, p.Tk, tid);
    tile_fetch<DH, NT>(vr, toff, Vb, p.LQ, 0, p.Tk, tid);
    kflag = key_flag(kv, 0, p.Tk, lane);
    // Q fragments, pre-scaled by scale*log2(e) (softmax runs on exp2): lane (q, lh) holds Q[q][8g + 4lh + j]
    float qf[G][4];
    const float qs = p.scale * AT_LOG2E;
#pragma unroll
    for (int g = 0; g < G; ++g) {
        // a query past the end re-reads the last one (finite values; nothing of it is stored): no exec-masked load regions
        const float4 v = *reinterpret_cast<const float4*>(Qb + (int64_t)(q < p.Tq ? q : p.Tq - 1) * p.LQ + 8 * g + 4 * lh);
        qf[g][0] = v.x; qf[g][1] = v.y; qf[g][2] = v.z; qf[g][3] = v.w;
    }

    // How many key tiles does this block visit?  Causal tiles above the block's last query contribute exp(-1e9-m)=0
    // and are skipped -- unless some query of the block has NO visible unmasked key: then the reference's softmax is
    // uniform over ALL keys (every score is the same -1e9) and nothing may be skipped.
    const int fv = first_valid_key<NT>(kv, p.Tk, tid, &sh_fv, kflag);
    int n_tiles = (p.Tk + AT_BK - 1) / AT_BK;
    const bool skip_ok = !dense && p.causal && fv <= qb * BQ + shift;   // every query of the block sees a non-padding key
    if (skip_ok) {
        const int last_key = min(p.Tk - 1, qb * BQ + BQ - 1 + shift);
        n_tiles = last_key < 0 ? 0 : last_key / AT_BK + 1;
    }
    // dense mask: a tile none of this wave's queries can see is skipped when every one of them sees SOME key elsewhere
    const bool row_sees = dense && p.x.row_any && q < p.Tq ? p.x.row_any[(int64_t)b * p.Tq + q] != 0 : (q >= p.Tq);

    f32x16 o[DT];                                       // O^T: o[dt][e] <-> d = 32dt + acc_row(e, lh), query = lane
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[dt][e] = 0.f;
    float m = -INFINITY, l = 0.f;                       // running max (log2 units, shared by the lane pair), partial sum
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int j = 0; j < 4; ++j) qf[g][j] *= qs;

    AT_T(0);
    for (int t = 0; t < n_tiles; ++t) {
        const int kv0 = t * AT_BK;
        __syncthreads();                                // previous tile fully consumed
        AT_T(1);
        tile_commit<DH, NT>(Ks, kr, tid);
        tile_commit<DH, NT>(Vs, vr, tid);
        const unsigned long long vball = __ballot(kflag != 0);
        unsigned long long valid = vball >> (4 * lh);                   // bit j: key kv0 + j + 4lh is a real token
        __syncthreads();
        AT_T(2);
        if (t + 1 < n_tiles) {                          // next tile's loads fly during this tile's MFMAs
            tile_fetch<DH, NT>(kr, toff, Kb, p.LQ, kv0 + AT_BK, p.Tk, tid);
            tile_fetch<DH, NT>(vr, toff, Vb, p.LQ, kv0 + AT_BK, p.Tk, tid);
            kflag = key_flag(kv, kv0 + AT_BK, p.Tk, lane);
        }
        AT_T(3);
        bool skip = skip_ok && kv0 > q0 + 31 + shift;   // wave-uniform: every key of this tile is above the diagonal
        if constexpr (GEN) {
            if (dense) {
                valid = dense_bits(p.x, b, q, p.Tq, p.Tk, t, lh);
                skip = __ballot(valid != 0ull || !row_sees) == 0ull;
            }
        }
        // wave-uniform: the tile's upper 32 keys are above the diagonal for every query of the wave (all of which see a
        // real key): their probabilities are exactly 0, so that half's S, softmax and PV work is dropped
        const bool half = skip_ok && kv0 + 32 > q0 + 31 + shift;
        if (!skip) {
            // ---- S^T = K Q^T (2 key sub-tiles of 32) -------------------------------------------------------
            f32x16 s[2];
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int e = 0; e < 16; ++e) s[kt][e] = 0.f;
            if (half) mma1_rows<G>(s[0], &Ks[l31 * LD + 4 * lh], qf);
            else mma2_rows<G>(s[0], &Ks[l31 * LD + 4 * lh], qf, s[1], &Ks[(32 + l31) * LD + 4 * lh], qf);
            AT_T(4);
            // ---- mask + online softmax (lane <-> query); local key of register (kt, e) = kt*32 + rc(e) + 4lh ------
            const int lim_causal = p.causal ? q + shift - kv0 - 4 * lh : 1 << 30;   // masked iff local index > lim
            const int lim_range = p.Tk - 1 - kv0 - 4 * lh;                            // not a key at all iff local index > lim
            float mx = -INFINITY;
            // wave-uniform: every key of the tile is a real token that every query of the wave may see -- nothing to mask
            // (all tiles left of the diagonal: the common case)
            const bool plain = !(GEN && dense) && vball == ~0ull && kv0 + AT_BK <= p.Tk && (!p.causal || kv0 + AT_BK - 1 <= q0 + shift);
            if (plain) {
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int e = 0; e < 16; ++e) mx = fmaxf(mx, s[kt][e]);
            } else {
#pragma unroll
                for (int kt = 0; kt < 2; ++kt) {
                    if (kt == 1 && half) break;
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int kl = kt * 32 + (e & 3) + 8 * (e >> 2);
                        float v = s[kt][e];
                        if (kl > lim_causal || !((valid >> kl) & 1ull)) v = AT_MASKED2;
                        if (kl > lim_range) v = -INFINITY;
                        s[kt][e] = v;
                        mx = fmaxf(mx, v);
                    }
                }
            }
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m, mx);
            const float alpha = __builtin_amdgcn_exp2f(m - m_new);   // exp2(-inf) = 0 on the first tile
            float ps = 0.f;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                if (kt == 1 && half) break;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    float pv = __builtin_amdgcn_exp2f(s[kt][e] - m_new);
                    ps += pv;                                        // the softmax denominator is over the UN-dropped map
                    if constexpr (GEN) {
                        if (dropping) {
                            const int key = kv0 + kt * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                            pv *= (q < p.Tq && key < p.Tk) ? at_drop_mult(p.x, rowkey, drow, p.Tk, key) : 0.f;
                        }
                    }
                    s[kt][e] = pv;
                }
            }
            l = l * alpha + ps;
            m = m_new;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int e = 0; e < 16; ++e) o[dt][e] *= alpha;
            AT_T(5);
            // ---- O^T += V^T P^T ------------------------------------------------------------------------------
            mma_cols<DT, LD>(o, &Vs[(4 * lh) * LD + l31], s[0]);
            if (!half) mma_cols<DT, LD>(o, &Vs[(32 + 4 * lh) * LD + l31], s[1]);
            AT_T(6);
        }
    }

    // ---- finish: O = O^T / l; (max, log2 sum) in log2 units are kept apart: a fully-masked row has m = -1e9*log2e,
    // where fp32 cannot hold m + log2 l ------------------------------------------------------------------------
    const float lt = l + __shfl_xor(l, 32, 64);
    const float inv = lt > 0.f ? 1.0f / lt : 0.f;
    if (lh == 0 && q < p.Tq)
        *reinterpret_cast<float2*>(p.LSE + 2 * (((int64_t)b * p.H + h) * p.Tq + q)) = make_float2(m, log2f(lt));
    __syncthreads();                                     // K/V tiles are dead: reuse the block as [4 waves][32 q][LD]
    store_transposed<DH>(smem + wave * (32 * LD), o, inv, p.O + ((int64_t)b * p.Tq) * p.D + (int64_t)h * DH, p.D, q0, p.Tq, lane);
    AT_T(7);
    AT_PROF_STORE(p.prof);
    };
    item(qb_first);
    if (qb_second >= 0) {
        __syncthreads();                                     // the epilogue's LDS rows are read before the next item's tiles land
        item(qb_second);
    }
}

// =====================================================================================================
// dK, dV: block = 128 keys (4 waves x 32), loop over 32-query tiles
// S[q,key] = Q K^T (A = Q tile rows from LDS, B = K fragments in registers; lane <-> key, registers <-> queries),
// P = exp2(S - m[q] - l[q]), dP = dO V^T (B = V fragments in registers), dS = P (dP*drop - Dsum[q]) * scale (0 where
// masked), dV^T[d,key] += dO^T (P*drop), dK^T[d,key] += Q^T dS -- P and dS feed those MFMAs as B operands straight from
// their accumulator registers.
// =====================================================================================================
template <int DH, bool GEN, int NW>
__global__ __launch_bounds__(64 * NW, DH <= 64 ? 2 : 1) void attn_bwd_dkdv_kernel(const AttnBwdParams p) {
    constexpr int NT = 64 * NW, BKB = 32 * NW;                 // threads, keys per block
    constexpr int LD = DH + 4, G = DH / 8, DT = DH / 32;
    constexpr int QT = 32;                                     // queries per tile (keeps the DH = 64 kernel at 2 blocks / CU)
    constexpr int NV = QT * DH / (4 * NT);
    constexpr int SM_FLOATS = (NW * 32 * LD > 2 * QT * LD + 4 * QT) ? NW * 32 * LD : 2 * QT * LD + 4 * QT;                     // epilogue staging; the loop uses Q tile, dO tile, max, log2sum, Dsum, any
    static_assert(SM_FLOATS >= 2 * QT * LD + 4 * QT, "tiles must fit");
    __shared__ __attribute__((aligned(16))) float smem[SM_FLOATS];
    __shared__ int sh_fv;
    float* Qs = smem;
    float* dOs = smem + QT * LD;
    float* Ms = smem + 2 * QT * LD;
    float* Ls = Ms + QT;
    float* Ds = Ls + QT;
    float* As = Ds + QT;                                       // GEN: row_any of the tile's queries

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int kblocks = (p.Tk + BKB - 1) / BKB;
    int bh, kb;
    if (!map_block(blockIdx.x, p.B * p.H, kblocks, false, bh, kb)) return;
    const int b = bh / p.H, h = bh - b * p.H;
    const int k0w = kb * BKB + wave * 32;
    const int key = k0w + l31;                                  // this lane's key
    const float* Qb = p.Q + ((int64_t)b * p.Tq) * p.LQ + (int64_t)h * DH;
    const float* dOb = p.dO + ((int64_t)b * p.Tq) * p.D + (int64_t)h * DH;
    const float* Kb = p.K + ((int64_t)b * p.Tk) * p.LQ + (int64_t)h * DH;
    const float* Vb = p.V + ((int64_t)b * p.Tk) * p.LQ + (int64_t)h * DH;
    const float2* LSEb = reinterpret_cast<const float2*>(p.LSE) + ((int64_t)b * p.H + h) * p.Tq;
    const float* Dsb = p.Dsum + ((int64_t)b * p.H + h) * p.Tq;
    const int32_t* kv = p.key_valid ? p.key_valid + (int64_t)b * p.Tk : nullptr;
    const int shift = p.Tk - p.Tq;
    const bool key_in = key < p.Tk;
    const int keyc = key_in ? key : p.Tk - 1;
    const bool dense = GEN && p.x.mask_bitsT != nullptr;
    const bool dropping = GEN && (p.x.drop_mask != nullptr || p.x.drop_threshold != 0u);
    const int nwq = (p.Tq + 63) >> 6;
    const unsigned long long* bitsT = dense && key_in ? p.x.mask_bitsT + ((int64_t)b * p.Tk + key) * nwq : nullptr;
    const int64_t drow0 = ((int64_t)b * p.H + h) * p.Tq;
    const unsigned seed = GEN ? p.x.drop_seed + at_seed_offset(p.x) : 0u;

    // K (pre-scaled by scale*log2e: scores in log2 units, as the forward saved them) and V fragments of this lane's key
    float kf[G][4], vf[G][4];
    const float sl2 = p.scale * AT_LOG2E;
#pragma unroll
    for (int g = 0; g < G; ++g) {
        // a key past the end re-reads the last one (finite values, its dK / dV columns are not stored): no exec-masked loads
        const float4 a = *reinterpret_cast<const float4*>(Kb + (int64_t)keyc * p.LQ + 8 * g + 4 * lh);
        const float4 c = *reinterpret_cast<const float4*>(Vb + (int64_t)keyc * p.LQ + 8 * g + 4 * lh);
        kf[g][0] = a.x; kf[g][1] = a.y; kf[g][2] = a.z; kf[g][3] = a.w;
        vf[g][0] = c.x; vf[g][1] = c.y; vf[g][2] = c.z; vf[g][3] = c.w;
    }
    const bool key_pad = kv && kv[keyc] == 0 && key_in;
    const int fv = first_valid_key<NT>(kv, p.Tk, tid, &sh_fv, key_flag(kv, 0, p.Tk, lane));
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int j = 0; j < 4; ++j) kf[g][j] *= sl2;
    f32x16 dk[DT], dv[DT];                                      // dK^T / dV^T: [dt][e] <-> d = 32dt + acc_row(e,lh), key = lane
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int e = 0; e < 16; ++e) { dk[dt][e] = 0.f; dv[dt][e] = 0.f; }

    // query tiles this block must visit: tile qt may be skipped iff every query of it sees a real key (fv <= qs + shift)
    // AND the whole key block lies above its diagonal; both conditions are monotone in qt, so the visited tiles are
    // [0, qt_lo) (rows that may be fully masked) and [qt_hi, n_qt) (on / below the diagonal).
    const int n_qt = (p.Tq + QT - 1) / QT;
    auto tile_skipped = [&](int qt) { return !dense && p.causal && fv <= qt * QT + shift && kb * BKB > qt * QT + QT - 1 + shift; };
    auto next_tile = [&](int qt) { while (qt < n_qt && tile_skipped(qt)) ++qt; return qt; };

    TileRegsN<NV> qr, gr;
    TileOffs<NV> qoff, goff;
    tile_offsets<DH, NT>(qoff, p.LQ, tid);
    tile_offsets<DH, NT>(goff, p.D, tid);
    float2 ml = make_float2(0.f, 0.f);
    float dsv = 0.f, anyv = 1.f;
    int rows_qs = 0;
    auto fetch_rows = [&](int qs) {                             // unconditional loads of clamped rows (every thread): no exec-masked
        rows_qs = qs;                                           // region; rows past the end are zeroed when they are committed
        const int r = min(qs + (tid & (QT - 1)), p.Tq - 1);
        ml = LSEb[r]; dsv = Dsb[r]; anyv = 1.f;
        if (GEN && dense && p.x.row_any) anyv = p.x.row_any[(int64_t)b * p.Tq + r] ? 1.f : 0.f;
    };
    int qt = next_tile(0);
    if (qt < n_qt) {
        tile_fetch_ptr<DH, NT>(qr, qoff, Qb, p.LQ, qt * QT, p.Tq, tid);
        tile_fetch_ptr<DH, NT>(gr, goff, dOb, p.D, qt * QT, p.Tq, tid);
        fetch_rows(qt * QT);
    }
    while (qt < n_qt) {
        const int qs = qt * QT;
        __syncthreads();
        tile_commit<DH, NT>(Qs, qr, tid);
        tile_commit<DH, NT>(dOs, gr, tid);
        if (tid < QT) {
            const bool ok = rows_qs + tid < p.Tq;
            Ms[tid] = ok ? ml.x : 0.f; Ls[tid] = ok ? ml.y : 0.f; Ds[tid] = ok ? dsv : 0.f; As[tid] = ok ? anyv : 1.f;
        }
        __syncthreads();
        const int qn = next_tile(qt + 1);
        if (qn < n_qt) {
            tile_fetch_ptr<DH, NT>(qr, qoff, Qb, p.LQ, qn * QT, p.Tq, tid);
            tile_fetch_ptr<DH, NT>(gr, goff, dOb, p.D, qn * QT, p.Tq, tid);
            fetch_rows(qn * QT);
        }
        // wave-uniform skip: this wave's 32 keys are above the diagonal for all queries of the tile (which all see a real key)
        bool skip = !dense && p.causal && fv <= qs + shift && k0w > qs + QT - 1 + shift;
        unsigned validq = 0xFFFFFFFFu;                           // dense: bit ql <-> query qs + ql + 4lh visible to this key
        if constexpr (GEN) {
            if (dense) {
                const unsigned long long w = bitsT ? bitsT[qs >> 6] : 0ull;
                const unsigned w32 = (unsigned)(w >> (qs & 63));
                validq = w32 >> (4 * lh);
                skip = __ballot(w32 != 0u) == 0ull && __ballot(As[l31] == 0.f) == 0ull;
            }
        }
        if (!skip) {
            const int lim_causal = p.causal ? key - shift - qs - 4 * lh : -(1 << 30);   // masked iff local query < lim
            const int lim_range = p.Tq - qs - 4 * lh;                                     // a query iff local index < lim
            f32x16 s, dp;
#pragma unroll
            for (int e = 0; e < 16; ++e) { s[e] = 0.f; dp[e] = 0.f; }
            mma2_rows<G>(s, &Qs[l31 * LD + 4 * lh], kf, dp, &dOs[l31 * LD + 4 * lh], vf);
            // wave-uniform: the wave's 32 keys are real tokens visible to all 32 queries of the tile -- nothing to mask
            const bool plain = !(GEN && dense) && __ballot(key_pad || !key_in) == 0ull && qs + QT <= p.Tq &&
                               (!p.causal || k0w + 31 <= qs + shift);
            // ---- P = exp2(S_masked - m[q] - l[q]);  dS = P (dP*drop - Dsum[q]) scale, zero where masked ---------------
#pragma unroll
            for (int e4 = 0; e4 < 4; ++e4) {
                const float4 M4 = *reinterpret_cast<const float4*>(&Ms[8 * e4 + 4 * lh]);
                const float4 L4 = *reinterpret_cast<const float4*>(&Ls[8 * e4 + 4 * lh]);
                const float4 D4 = *reinterpret_cast<const float4*>(&Ds[8 * e4 + 4 * lh]);
                const float Mr[4] = {M4.x, M4.y, M4.z, M4.w}, Lr[4] = {L4.x, L4.y, L4.z, L4.w}, Dr[4] = {D4.x, D4.y, D4.z, D4.w};
                if (plain) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int e = e4 * 4 + r;
                        const float pv = __builtin_amdgcn_exp2f((s[e] - Mr[r]) - Lr[r]);
                        float mult = 1.f;
                        if constexpr (GEN) {
                            if (dropping) {
                                const int64_t row = drow0 + qs + 8 * e4 + r + 4 * lh;
                                mult = at_drop_mult(p.x, at_rowkey(seed, (unsigned)row), row, p.Tk, key);
                            }
                        }
                        s[e] = pv * mult;
                        dp[e] = pv * ((dp[e] * mult - Dr[r]) * p.scale);
                    }
                    continue;
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int e = e4 * 4 + r;
                    const int ql = 8 * e4 + r;                          // local query index minus 4lh
                    const bool masked = ql < lim_causal || key_pad || !((validq >> ql) & 1u);
                    float arg = ((masked ? AT_MASKED2 : s[e]) - Mr[r]) - Lr[r];
                    if (!(key_in && ql < lim_range)) arg = -INFINITY;
                    const float pv = __builtin_amdgcn_exp2f(arg);
                    float mult = 1.f;
                    if constexpr (GEN) {
                        if (dropping && key_in && ql < lim_range) {
                            const int64_t row = drow0 + qs + ql + 4 * lh;
                            mult = at_drop_mult(p.x, at_rowkey(seed, (unsigned)row), row, p.Tk, key);
                        }
                    }
                    s[e] = pv * mult;                                   // dV^T += dO^T (P * drop)
                    dp[e] = (masked ? 0.f : pv) * ((dp[e] * mult - Dr[r]) * p.scale);
                }
            }
            // ---- dV^T += dO^T P ; dK^T += Q^T dS ----------------------------------------------------------------
            mma2_cols<DT, LD>(dv, &dOs[(4 * lh) * LD + l31], s, dk, &Qs[(4 * lh) * LD + l31], dp);
        }
        qt = qn;
    }

    // ---- store dK, dV rows (transpose through LDS) --------------------------------------------------------------
    __syncthreads();
    float* E = smem + wave * (32 * LD);
    store_transposed<DH>(E, dk, 1.0f, p.dK + ((int64_t)b * p.Tk) * p.LQ + (int64_t)h * DH, p.LQ, k0w, p.Tk, lane);
    __builtin_amdgcn_wave_barrier();
    store_transposed<DH>(E, dv, 1.0f, p.dV + ((int64_t)b * p.Tk) * p.LQ + (int64_t)h * DH, p.LQ, k0w, p.Tk, lane);
}

// =====================================================================================================
// dQ: block = 128 queries (4 waves x 32), loop over 64-key tiles (mirror of the forward)
// S^T = K Q^T, P^T = exp2(S^T - m[q] - l[q]) (lane <-> query), dP^T = V dO^T (B = dO fragments in registers),
// dS^T = P^T (dP^T*drop - Dsum[q]) scale, dQ^T[d,q] += K^T dS^T.
// =====================================================================================================
template <int DH, bool GEN, int NW>
__global__ __launch_bounds__(64 * NW, DH <= 64 ? 2 : 1) void attn_bwd_dq_kernel(const AttnBwdParams p) {
    constexpr int NT = 64 * NW, BQ = 32 * NW;
    constexpr int LD = DH + 4, G = DH / 8, DT = DH / 32, NV = AT_BK * DH / (4 * NT);
    constexpr int SM_FLOATS = 2 * AT_BK * LD > NW * 32 * LD ? 2 * AT_BK * LD : NW * 32 * LD;
    __shared__ __attribute__((aligned(16))) float smem[SM_FLOATS];
    __shared__ int sh_fv;
    float* Ks = smem;
    float* Vs = smem + AT_BK * LD;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int qblocks = (p.Tq + BQ - 1) / BQ;
    int bh, qb;
    if (!map_block(blockIdx.x, p.B * p.H, qblocks, true, bh, qb)) return;
    const int b = bh / p.H, h = bh - b * p.H;
    const int q0 = qb * BQ + wave * 32;
    const int q = q0 + l31;
    const float* Qb = p.Q + ((int64_t)b * p.Tq) * p.LQ + (int64_t)h * DH;
    const float* dOb = p.dO + ((int64_t)b * p.Tq) * p.D + (int64_t)h * DH;
    const float* Kb = p.K + ((int64_t)b * p.Tk) * p.LQ + (int64_t)h * DH;
    const float* Vb = p.V + ((int64_t)b * p.Tk) * p.LQ + (int64_t)h * DH;
    const int32_t* kv = p.key_valid ? p.key_valid + (int64_t)b * p.Tk : nullptr;
    const int shift = p.Tk - p.Tq;
    const bool q_in = q < p.Tq;
    // prologue loads all in flight before the first wait: first K/V tile, row statistics, Q / dO / O fragments, padding scan
    TileRegsN<NV> kr, vr;
    TileOffs<NV> toff;
    tile_offsets<DH, NT>(toff, p.LQ, tid);
    int kflag = 1;
    tile_fetch<DH, NT>(kr, toff, Kb, p.LQ, 0, p.Tk, tid);
    tile_fetch<DH, NT>(vr, toff, Vb, p.LQ, 0, p.Tk, tid);
    kflag = key_flag(kv, 0, p.Tk, lane);
    const int qc = q_in ? q : p.Tq - 1;
    const float2 ml = reinterpret_cast<const float2*>(p.LSE)[((int64_t)b * p.H + h) * p.Tq + qc];
    const float* Ob = p.O + ((int64_t)b * p.Tq) * p.D + (int64_t)h * DH;
    const bool dense = GEN && p.x.mask_bits != nullptr;
    const bool dropping = GEN && (p.x.drop_mask != nullptr || p.x.drop_threshold != 0u);
    const int64_t drow = ((int64_t)b * p.H + h) * p.Tq + q;
    const unsigned seed = GEN ? p.x.drop_seed + at_seed_offset(p.x) : 0u;
    const unsigned rowkey = GEN ? at_rowkey(seed, (unsigned)drow) : 0u;
    const bool row_sees = dense && p.x.row_any && q_in ? p.x.row_any[(int64_t)b * p.Tq + q] != 0 : !q_in;

    float qf[G][4], gf[G][4];                                   // Q (pre-scaled, log2 units) and dO fragments of this lane's query
    const float sl2 = p.scale * AT_LOG2E;
    float dpart = 0.f;                                          // this half-wave's part of Dsum[q] = sum_d dO[q,d] O[q,d]
    float4 og[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        // a query past the end re-reads the last one (finite values; nothing of it is stored): no exec-masked load regions
        const float4 a = *reinterpret_cast<const float4*>(Qb + (int64_t)qc * p.LQ + 8 * g + 4 * lh);
        const float4 c = *reinterpret_cast<const float4*>(dOb + (int64_t)qc * p.D + 8 * g + 4 * lh);
        const float4 o4 = *reinterpret_cast<const float4*>(Ob + (int64_t)qc * p.D + 8 * g + 4 * lh);
        qf[g][0] = a.x; qf[g][1] = a.y; qf[g][2] = a.z; qf[g][3] = a.w;
        gf[g][0] = c.x; gf[g][1] = c.y; gf[g][2] = c.z; gf[g][3] = c.w;
        og[g] = o4;
    }
    const int fv = first_valid_key<NT>(kv, p.Tk, tid, &sh_fv, kflag);
#pragma unroll
    for (int g = 0; g < G; ++g) {
        dpart += gf[g][0] * og[g].x + gf[g][1] * og[g].y + gf[g][2] * og[g].z + gf[g][3] * og[g].w;
#pragma unroll
        for (int j = 0; j < 4; ++j) qf[g][j] *= sl2;
    }
    // Dsum (= sum_k dP P, the softmax-backward row term; with dropout O already holds the dropped map, so dO.O is still it)
    // is produced here -- the dO fragments are already in registers -- and saved for the dK/dV kernel, which runs after.
    const float dsum = dpart + __shfl_xor(dpart, 32, 64);
    if (lh == 0 && q_in) p.Dsum[((int64_t)b * p.H + h) * p.Tq + q] = dsum;
    int n_tiles = (p.Tk + AT_BK - 1) / AT_BK;
    const bool skip_ok = !dense && p.causal && fv <= qb * BQ + shift;
    if (skip_ok) {
        const int last_key = min(p.Tk - 1, qb * BQ + BQ - 1 + shift);
        n_tiles = last_key < 0 ? 0 : last_key / AT_BK + 1;
    }
    f32x16 dq[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int e = 0; e < 16; ++e) dq[dt][e] = 0.f;

    for (int t = 0; t < n_tiles; ++t) {
        const int kv0 = t * AT_BK;
        __syncthreads();
        tile_commit<DH, NT>(Ks, kr, tid);
        tile_commit<DH, NT>(Vs, vr, tid);
        const unsigned long long vball = __ballot(kflag != 0);
        unsigned long long valid = vball >> (4 * lh);
        __syncthreads();
        if (t + 1 < n_tiles) {
            tile_fetch<DH, NT>(kr, toff, Kb, p.LQ, kv0 + AT_BK, p.Tk, tid);
            tile_fetch<DH, NT>(vr, toff, Vb, p.LQ, kv0 + AT_BK, p.Tk, tid);
            kflag = key_flag(kv, kv0 + AT_BK, p.Tk, lane);
        }
        bool skip = skip_ok && kv0 > q0 + 31 + shift;
        if constexpr (GEN) {
            if (dense) {
                valid = dense_bits(p.x, b, q, p.Tq, p.Tk, t, lh);
                skip = __ballot(valid != 0ull || !row_sees) == 0ull;
            }
        }
        if (!skip) {
            const int lim_causal = p.causal ? q + shift - kv0 - 4 * lh : 1 << 30;
            const int lim_range = q_in ? p.Tk - 1 - kv0 - 4 * lh : -1;
            const bool half = skip_ok && kv0 + 32 > q0 + 31 + shift;   // upper 32 keys: P = 0 for the whole wave
            const bool plain = !(GEN && dense) && vball == ~0ull && kv0 + AT_BK <= p.Tk && q0 + 32 <= p.Tq &&
                               (!p.causal || kv0 + AT_BK - 1 <= q0 + shift);
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                if (kt == 1 && half) break;
                f32x16 s, dp;
#pragma unroll
                for (int e = 0; e < 16; ++e) { s[e] = 0.f; dp[e] = 0.f; }
                mma2_rows<G>(s, &Ks[(kt * 32 + l31) * LD + 4 * lh], qf, dp, &Vs[(kt * 32 + l31) * LD + 4 * lh], gf);
                if (plain) {                                            // nothing masked in this tile for this wave
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        float mult = 1.f;
                        if constexpr (GEN) {
                            if (dropping) mult = at_drop_mult(p.x, rowkey, drow, p.Tk, kv0 + kt * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh);
                        }
                        dp[e] = __builtin_amdgcn_exp2f((s[e] - ml.x) - ml.y) * ((dp[e] * mult - dsum) * p.scale);
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int kl = kt * 32 + (e & 3) + 8 * (e >> 2);
                        const bool live = kl <= lim_range && kl <= lim_causal && ((valid >> kl) & 1ull);
                        const float arg = live ? (s[e] - ml.x) - ml.y : -INFINITY;
                        float mult = 1.f;
                        if constexpr (GEN) {
                            if (dropping && live) mult = at_drop_mult(p.x, rowkey, drow, p.Tk, kv0 + kl + 4 * lh);
                        }
                        dp[e] = __builtin_amdgcn_exp2f(arg) * ((dp[e] * mult - dsum) * p.scale);
                    }
                }
                mma_cols<DT, LD>(dq, &Ks[(kt * 32 + 4 * lh) * LD + l31], dp);      // dQ^T += K^T dS^T
            }
        }
    }
    __syncthreads();
    store_transposed<DH>(smem + wave * (32 * LD), dq, 1.0f, p.dQ + ((int64_t)b * p.Tq) * p.LQ + (int64_t)h * DH, p.LQ, q0, p.Tq, lane);
}

// ---- dense mask -> packed bits ---------------------------------------------------------------------------------------
// mask [B, Tq, Tk] int32 (non-zero = key visible; the notebook's get_pad_mask & get_sub_mask) ->
//   bits  [B, Tq, ceil(Tk/64)]  one wave per word: lanes <-> 64 consecutive keys (coalesced), ballot
//   bitsT [B, Tk, ceil(Tq/64)]  one wave per word: lanes <-> 64 consecutive queries (stride Tk; the mask is small)
//   row_any [B, Tq]             1 iff the row has a visible key
__global__ __launch_bounds__(256) void attn_pack_mask_kernel(const int32_t* __restrict__ mask, unsigned long long* __restrict__ bits,
                                                             unsigned long long* __restrict__ bitsT, unsigned char* __restrict__ row_any,
                                                             int B, int Tq, int Tk) {
    const int lane = threadIdx.x & 63;
    const int64_t wid = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
    const int nw = (Tk + 63) >> 6, nwq = (Tq + 63) >> 6;
    const int64_t n1 = (int64_t)B * Tq * nw, n2 = (int64_t)B * Tk * nwq, n3 = (int64_t)B * Tq;
    if (wid < n1) {
        const int64_t bq = wid / nw;
        const int w = (int)(wid - bq * nw);
        const int k = w * 64 + lane;
        const int v = k < Tk ? mask[bq * Tk + k] : 0;
        const unsigned long long word = __ballot(v != 0);
        if (lane == 0) bits[wid] = word;
    } else if (wid < n1 + n2) {
        const int64_t j = wid - n1;
        const int64_t bk = j / nwq;
        const int w = (int)(j - bk * nwq);
        const int64_t b = bk / Tk;
        const int k = (int)(bk - b * Tk);
        const int q = w * 64 + lane;
        const int v = q < Tq ? mask[(b * Tq + q) * Tk + k] : 0;
        const unsigned long long word = __ballot(v != 0);
        if (lane == 0) bitsT[j] = word;
    } else if (wid < n1 + n2 + n3) {
        const int64_t bq = wid - n1 - n2;
        int any = 0;
        for (int k = lane; k < Tk; k += 64) any |= mask[bq * Tk + k] != 0;
        const unsigned long long word = __ballot(any != 0);
        if (lane == 0) row_any[bq] = word != 0ull;
    }
}

// materialise the hash dropout multipliers (tests: the unfused path / the oracle run with exactly this mask)
__global__ __launch_bounds__(256) void attn_dropout_mask_kernel(float* __restrict__ out, int64_t rows, int Tk, unsigned seed,
                                                                unsigned threshold, float scale, const unsigned* __restrict__ seed_dev) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * Tk) return;
    if (seed_dev) seed += __hip_atomic_load(seed_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // as the fused kernels do
    const int64_t row = i / Tk;
    const int key = (int)(i - row * Tk);
    out[i] = at_hash(at_rowkey(seed, (unsigned)row), (unsigned)key) >= threshold ? scale : 0.f;
}

}  // namespace nnhip

using namespace nnhip;

static int attn_check(const char* fn, const void* Q, const void* K, const void* V, int64_t B, int64_t H, int64_t Tq,
                      int64_t Tk, int64_t dh, int64_t ld_qkv) {
    NNHIP_CHECK_ARG(ld_qkv == 0 || (ld_qkv >= H * dh && ld_qkv % 4 == 0), NNHIP_EINVAL,
                    "%s: ld_qkv must be 0 or a multiple of 4 that is >= H * head_dim", fn);
    NNHIP_CHECK_ARG(B >= 0 && H > 0 && Tq >= 0 && Tk >= 0, NNHIP_EINVAL, "%s: bad sizes", fn);
    NNHIP_CHECK_ARG(dh == 32 || dh == 64 || dh == 128, NNHIP_EINVAL, "%s: the fused kernels support head_dim 32, 64 and 128", fn);
    NNHIP_CHECK_ARG(Tq < (1 << 24) && Tk < (1 << 24) && B * H < (1 << 24), NNHIP_EINVAL, "%s: sizes too large", fn);
    {   // a (batch, head) slice is addressed through one buffer descriptor with 32-bit byte offsets
        const int64_t ld = ld_qkv > 0 ? ld_qkv : H * dh, tmax = Tq > Tk ? Tq : Tk;
        NNHIP_CHECK_ARG(tmax * (ld > H * dh ? ld : H * dh) * 4 < ((int64_t)1 << 32), NNHIP_EINVAL,
                        "%s: one sequence's rows span more than 4 GiB (T x row stride): split the sequence", fn);
    }
    if (B == 0 || Tq == 0) return 0;
    NNHIP_CHECK_ARG(Q && K && V, NNHIP_EINVAL, "%s: null pointer", fn);
    NNHIP_CHECK_ARG(aligned16(Q) && aligned16(K) && aligned16(V), NNHIP_EALIGN, "%s: Q/K/V must be 16-byte aligned", fn);
    return 0;
}

static int fill_extra(const char* fn, AttnExtra& x, const nnhipAttentionOptions* o, const int32_t*& key_valid, int& causal) {
    x = AttnExtra{};
    if (!o) return 0;
    NNHIP_CHECK_ARG((o->mask_bits == nullptr) == (o->mask_bitsT == nullptr), NNHIP_EINVAL,
                    "%s: mask_bits and mask_bitsT come together (nnhipAttentionPackMask writes both)", fn);
    NNHIP_CHECK_ARG(o->dropout_p >= 0.f && o->dropout_p < 1.f, NNHIP_EINVAL, "%s: dropout_p must be in [0, 1)", fn);
    x.mask_bits = reinterpret_cast<const unsigned long long*>(o->mask_bits);
    x.mask_bitsT = reinterpret_cast<const unsigned long long*>(o->mask_bitsT);
    x.row_any = o->row_any;
    x.drop_mask = o->dropout_mask;
    x.drop_seed = o->dropout_seed;
    x.seed_dev = o->dropout_seed_dev;
    x.drop_scale = o->dropout_p > 0.f ? 1.0f / (1.0f - o->dropout_p) : 1.0f;
    // keep iff hash >= threshold: P(drop) = threshold / 2^32
    const double th = (double)o->dropout_p * 4294967296.0;
    x.drop_threshold = o->dropout_mask ? 0u : (unsigned)(th < 4294967295.0 ? th : 4294967295.0);
    if (x.mask_bits) { key_valid = nullptr; causal = 0; }   // the dense mask is the whole mask
    return 0;
}
#ifdef AT_PROF
static long long* g_prof = nullptr;
extern "C" void nnhipAttentionSetProfile(long long* buf) { g_prof = buf; }
#define AT_SET_PROF(p) (p).prof = g_prof
#else
#define AT_SET_PROF(p) do {} while (0)
#endif
static bool extra_active(const AttnExtra& x) { return x.mask_bits || x.drop_mask || x.drop_threshold != 0u; }

// NW = waves (32-row groups) per block.  4 everywhere; head dim 64 -- the GPT-tiny shape -- can also run 2-wave blocks
// (attn_waves(): 4 resident blocks per CU instead of 2, so more blocks are out of phase with each other).
#define AT_LAUNCH(KERNEL, DHV, NWV, gen, nblk, st, p)                                                                   \
    do {                                                                                                               \
        if (gen) hipLaunchKernelGGL((KERNEL<DHV, true, NWV>), dim3(nblk), dim3(64 * NWV), 0, st, p);                   \
        else hipLaunchKernelGGL((KERNEL<DHV, false, NWV>), dim3(nblk), dim3(64 * NWV), 0, st, p);                      \
    } while (0)
#define AT_DISPATCH(KERNEL, dh, nw, gen, BH, T, st, p)                                                                 \
    do {                                                                                                               \
        if ((dh) == 64 && (nw) == 2) AT_LAUNCH(KERNEL, 64, 2, gen, mapped_grid(BH, ceil_div(T, 64)), st, p);           \
        else if ((dh) == 64) AT_LAUNCH(KERNEL, 64, 4, gen, mapped_grid(BH, ceil_div(T, 128)), st, p);                  \
        else if ((dh) == 32) AT_LAUNCH(KERNEL, 32, 4, gen, mapped_grid(BH, ceil_div(T, 128)), st, p);                  \
        else AT_LAUNCH(KERNEL, 128, 4, gen, mapped_grid(BH, ceil_div(T, 128)), st, p);                                 \
    } while (0)
// Head dim 64 runs 2-wave blocks by default: B64 T256 H8 forward 78.8 -> 75.3 us, backward 242 -> 228 us; T = 1024 +-0 / -3 %.
// NNHIP_ATTN_WAVES = 2 | 4 is the developer A/B switch.
static int attn_waves() {
    static const int w = []() { const char* e = getenv("NNHIP_ATTN_WAVES"); const int v = e ? atoi(e) : 2; return v == 4 ? 4 : 2; }();
    return w;
}

// Pair mode (causal masks, short sequences): a block runs the heaviest remaining row block of its (batch, head) and then that
// block's causal complement -- every block the same work, the whole grid resident at once instead of a second round of block
// launches behind the heavy ones (B64 T256 H8 forward 69.6 -> 64.9 us; the backward kernels sit at 256 VGPRs and spill with the
// second item's loop around them: forward only).  Taken when the halved grid still gives every CU two
// blocks and the sequence has at most 8 row blocks (T = 1024: 16 blocks, -1 %).  *Tgrid: the T the grid macro should see.
static int attn_pair(int64_t BH, int64_t T, int64_t head_dim, int causal, int64_t* Tgrid) {
    static const int mode = []() { const char* e = getenv("NNHIP_ATTN_PAIR"); return e ? atoi(e) : -1; }();   // dev knob: 0 off, 1 always
    const int64_t bq = (head_dim == 64 && attn_waves() == 2) ? 64 : 128;
    const int64_t nblk = ceil_div(T, bq), pairs = (nblk + 1) / 2;
    bool on = causal && nblk >= 2 && nblk <= 8 && BH * pairs >= 512;
    if (mode == 0) on = false;
    if (mode == 1) on = nblk >= 2;
    if (on) *Tgrid = pairs * bq;
    return on ? 1 : 0;
}

extern "C" int nnhipAttentionForwardEx(const float* Q, const float* K, const float* V, const int32_t* key_valid,
                                       float* O, float* LSE, int64_t B, int64_t H, int64_t Tq, int64_t Tk,
                                       int64_t head_dim, int64_t ld_qkv, float scale, int causal,
                                       const nnhipAttentionOptions* opts, nnhipStream_t s) {
    if (int rc = attn_check("nnhipAttentionForward", Q, K, V, B, H, Tq, Tk, head_dim, ld_qkv)) return rc;
    if (B == 0 || Tq == 0) return 0;
    NNHIP_CHECK_ARG(Tk > 0, NNHIP_EINVAL, "nnhipAttentionForward: Tk must be > 0");
    NNHIP_CHECK_ARG(O && LSE && aligned16(O), NNHIP_EINVAL, "nnhipAttentionForward: null / misaligned output");
    AttnParams p;
    if (int rc = fill_extra("nnhipAttentionForward", p.x, opts, key_valid, causal)) return rc;
    p.Q = Q; p.K = K; p.V = V; p.O = O; p.LSE = LSE; p.key_valid = key_valid;
    p.B = (int)B; p.H = (int)H; p.Tq = (int)Tq; p.Tk = (int)Tk; p.D = H * head_dim; p.LQ = ld_qkv ? ld_qkv : p.D; p.scale = scale; p.causal = causal; AT_SET_PROF(p);
    const bool gen = extra_active(p.x);
    if (attn_sb_applicable(B, H, Tq, Tk, head_dim, causal, gen)) { p.pair = 0; return attn_sb_forward(p, (hipStream_t)s); }
    int64_t Tgrid = Tq;
    p.pair = attn_pair(B * H, Tq, head_dim, causal, &Tgrid);
    AT_DISPATCH(attn_fwd_kernel, head_dim, attn_waves(), gen, B * H, Tgrid, (hipStream_t)s, p);
    NNHIP_LAUNCH_CHECK("attn_fwd_kernel");
    return 0;
}

extern "C" int nnhipAttentionForward(const float* Q, const float* K, const float* V, const int32_t* key_valid,
                                     float* O, float* LSE, int64_t B, int64_t H, int64_t Tq, int64_t Tk,
                                     int64_t head_dim, int64_t ld_qkv, float scale, int causal, nnhipStream_t s) {
    return nnhipAttentionForwardEx(Q, K, V, key_valid, O, LSE, B, H, Tq, Tk, head_dim, ld_qkv, scale, causal, nullptr, s);
}

extern "C" int nnhipAttentionBackwardEx(const float* Q, const float* K, const float* V, const int32_t* key_valid,
                                        const float* O, const float* dO, const float* LSE, float* dQ, float* dK,
                                        float* dV, int64_t B, int64_t H, int64_t Tq, int64_t Tk, int64_t head_dim,
                                        int64_t ld_qkv, float scale, int causal, const nnhipAttentionOptions* opts,
                                        nnhipStream_t s) {
    if (int rc = attn_check("nnhipAttentionBackward", Q, K, V, B, H, Tq, Tk, head_dim, ld_qkv)) return rc;
    if (B == 0 || Tq == 0 || Tk == 0) return 0;
    NNHIP_CHECK_ARG(O && dO && LSE && dQ && dK && dV, NNHIP_EINVAL, "nnhipAttentionBackward: null pointer");
    NNHIP_CHECK_ARG(aligned16(dO) && aligned16(O) && aligned16(dQ) && aligned16(dK) && aligned16(dV), NNHIP_EALIGN,
                    "nnhipAttentionBackward: O/dO/dQ/dK/dV must be 16-byte aligned");
    hipStream_t st = (hipStream_t)s;
    float* dsum = static_cast<float*>(workspace((size_t)B * H * Tq * sizeof(float)));
    NNHIP_CHECK_ARG(dsum != nullptr, NNHIP_ENOMEM, "nnhipAttentionBackward: workspace allocation failed");
    AttnBwdParams p;
    if (int rc = fill_extra("nnhipAttentionBackward", p.x, opts, key_valid, causal)) return rc;
    p.Q = Q; p.K = K; p.V = V; p.dO = dO; p.LSE = LSE; p.Dsum = dsum; p.O = O; p.dQ = dQ; p.dK = dK; p.dV = dV;
    p.key_valid = key_valid; p.B = (int)B; p.H = (int)H; p.Tq = (int)Tq; p.Tk = (int)Tk; p.D = H * head_dim;
    p.LQ = ld_qkv ? ld_qkv : p.D; p.scale = scale; p.causal = causal; AT_SET_PROF(p);
    const bool gen = extra_active(p.x);
    if (attn_sb_applicable(B, H, Tq, Tk, head_dim, causal, gen)) return attn_sb_backward(p, st);      // one pass, no Dsum hand-over
    // dQ first: it also produces Dsum, which the dK/dV kernel consumes
    AT_DISPATCH(attn_bwd_dq_kernel, head_dim, attn_waves(), gen, B * H, Tq, st, p);
    NNHIP_LAUNCH_CHECK("attn_bwd_dq_kernel");
    AT_DISPATCH(attn_bwd_dkdv_kernel, head_dim, attn_waves(), gen, B * H, Tk, st, p);
    NNHIP_LAUNCH_CHECK("attn_bwd_dkdv_kernel");
    return 0;
}

extern "C" int nnhipAttentionBackward(const float* Q, const float* K, const float* V, const int32_t* key_valid,
                                      const float* O, const float* dO, const float* LSE, float* dQ, float* dK,
                                      float* dV, int64_t B, int64_t H, int64_t Tq, int64_t Tk, int64_t head_dim,
                                      int64_t ld_qkv, float scale, int causal, nnhipStream_t s) {
    return nnhipAttentionBackwardEx(Q, K, V, key_valid, O, dO, LSE, dQ, dK, dV, B, H, Tq, Tk, head_dim, ld_qkv, scale,
                                    causal, nullptr, s);
}

extern "C" int nnhipAttentionPackMask(const int32_t* mask, uint64_t* mask_bits, uint64_t* mask_bitsT, uint8_t* row_any,
                                      int64_t B, int64_t Tq, int64_t Tk, nnhipStream_t s) {
    NNHIP_CHECK_ARG(B >= 0 && Tq >= 0 && Tk >= 0 && Tq < (1 << 24) && Tk < (1 << 24), NNHIP_EINVAL, "nnhipAttentionPackMask: bad sizes");
    if (B == 0 || Tq == 0 || Tk == 0) return 0;
    NNHIP_CHECK_ARG(mask && mask_bits && mask_bitsT && row_any, NNHIP_EINVAL, "nnhipAttentionPackMask: null pointer");
    const int64_t waves = B * Tq * ceil_div(Tk, 64) + B * Tk * ceil_div(Tq, 64) + B * Tq;
    hipLaunchKernelGGL(attn_pack_mask_kernel, dim3((unsigned)ceil_div(waves, 4)), dim3(256), 0, (hipStream_t)s, mask,
                       reinterpret_cast<unsigned long long*>(mask_bits), reinterpret_cast<unsigned long long*>(mask_bitsT), row_any,
                       (int)B, (int)Tq, (int)Tk);
    NNHIP_LAUNCH_CHECK("attn_pack_mask_kernel");
    return 0;
}

extern "C" int nnhipAttentionDropoutMaskEx(float* out, int64_t B, int64_t H, int64_t Tq, int64_t Tk, float dropout_p,
                                           uint32_t seed, const uint32_t* seed_dev, nnhipStream_t s);
extern "C" int nnhipAttentionDropoutMask(float* out, int64_t B, int64_t H, int64_t Tq, int64_t Tk, float dropout_p,
                                         uint32_t seed, nnhipStream_t s) {
    return nnhipAttentionDropoutMaskEx(out, B, H, Tq, Tk, dropout_p, seed, nullptr, s);
}

extern "C" int nnhipAttentionDropoutMaskEx(float* out, int64_t B, int64_t H, int64_t Tq, int64_t Tk, float dropout_p,
                                           uint32_t seed, const uint32_t* seed_dev, nnhipStream_t s) {
    NNHIP_CHECK_ARG(B >= 0 && H >= 0 && Tq >= 0 && Tk >= 0 && Tk < (1 << 24), NNHIP_EINVAL, "nnhipAttentionDropoutMask: bad sizes");
    NNHIP_CHECK_ARG(dropout_p >= 0.f && dropout_p < 1.f, NNHIP_EINVAL, "nnhipAttentionDropoutMask: dropout_p must be in [0, 1)");
    const int64_t rows = B * H * Tq;
    if (rows == 0 || Tk == 0) return 0;
    NNHIP_CHECK_ARG(out != nullptr, NNHIP_EINVAL, "nnhipAttentionDropoutMask: null pointer");
    const double th = (double)dropout_p * 4294967296.0;
    hipLaunchKernelGGL(attn_dropout_mask_kernel, dim3((unsigned)ceil_div(rows * Tk, 256)), dim3(256), 0, (hipStream_t)s, out, rows,
                       (int)Tk, seed, (unsigned)(th < 4294967295.0 ? th : 4294967295.0), dropout_p > 0.f ? 1.0f / (1.0f - dropout_p) : 1.0f,
                       seed_dev);
    NNHIP_LAUNCH_CHECK("attn_dropout_mask_kernel");
    return 0;
}
