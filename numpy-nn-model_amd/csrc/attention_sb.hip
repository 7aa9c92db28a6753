// attention_sb.hip -- the "super-block" fused attention kernels: causal self-attention at T = 256, head dim 64 (the GPT-tiny
// shape of BASELINE config C4: B64 x H8 x T256), exact fp32 MFMA.  Same semantics and the same LSE format as attention.hip
// (examples/gpt.ipynb cell 2: masked scores are REPLACED by -1e9; a query whose every visible key is padding is uniform over
// ALL keys); the general tiled kernels of attention.hip keep every other shape, dense masks and dropout.
//
// Why a second set of kernels (round 5).  At T = 256 the tiled kernels ran at 0.43 (forward) / 0.32 (backward) of the fp32
// MFMA peak while their tile loops kept the matrix pipe saturated: what was lost were the per-block prologues / epilogues of
// 1-4 tile blocks, two block barriers per tile, the causal imbalance between the waves of a block, and -- in the backward --
// two kernels that each recompute S and dP (7 GEMM-units per tile pair where 5 are needed).  Here:
//   * A block is 8 waves (one per SIMD pair, 1 block per CU) and owns TWO (batch, head) slices A and B.  The 8 x 8 causal
//     triangle of 32 x 32 sub-tiles of A and the triangle of B are dealt so that EVERY wave gets exactly 9 units of work:
//     forward -- wave w runs row group 7-w of A (8-w key groups) and then row group w of B (w+1 key groups);
//     backward -- wave w owns key group w of A (pairs (i, w), i = w..7) and then key group 7-w of B.
//     256 blocks of B64 x H8 are one resident generation: no tail, no second round, no imbalance.
//   * Operands are streamed from L2 straight into the registers the MFMAs read (no LDS ring, no staging registers, no block
//     barrier in the forward): a 32-key K group is 8 float4 per lane (lane <-> key row: the A operand of S^T = K Q^T), a V
//     group 16 float2 per lane (lane <-> two adjacent head-dim columns: coalesced 512-byte wave loads, the A operand of
//     O^T += V^T P^T).  Every operand register is refilled with the NEXT unit's value right behind the MFMAs that read it
//     (a rolling prefetch: a load has a whole unit -- >= 4096 matrix-pipe cycles -- to land), all through one buffer
//     descriptor per tensor slice with the group's row offset in a scalar register.
//   * The softmax keeps a LAZY running maximum (rescale only when a row maximum grows by more than 2^8: the 32 multiplies of
//     O per unit disappear; (m_ref, log2 sum) stays a consistent pair for the backward).
//   * Backward: ONE pass.  The wave that owns key group j computes S, dP, P, dS for the pair (i, j) once (lane <-> key),
//     feeds dV^T += dO^T P and dK^T += Q^T dS from the accumulator registers, transposes dS through 4.6 KB of wave-private
//     LDS and multiplies dQ^T(i) += K_j^T dS^T.  The partial dQ of row group i is summed in an LDS slot in a FIXED order: the
//     schedule is skewed (the s-th pair of wave w is (w + s, w)), so the row groups in flight at any time are distinct; a
//     per-slot contribution count orders the adds (the first contributor stores, the last one adds its part and writes the
//     row group out) -- no atomics, no block barrier in the pair loop, deterministic bits, nine pairs for every wave.
#include <math.h>
#include <stdlib.h>

#include "common.h"
#include "attention.h"

namespace nnhip {

typedef unsigned sb_u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned sb_u32x4 __attribute__((ext_vector_type(4)));
typedef float sb_f32x4 __attribute__((ext_vector_type(4)));

constexpr int SB_T = 256;        // the sequence length these kernels are built for: 8 groups of 32 rows <-> 8 waves
constexpr int SB_NG = 8;
constexpr int SB_DH = 64;
constexpr int SB_LD = SB_DH + 4; // LDS row stride of a staged [32][64] tile (17 chunks of 16 B: conflict-free b128 rows)
constexpr int SB_WAVE_LDS = 32 * SB_LD + 32 * 36;           // backward, per wave: K_j tile + dS patch (floats)
constexpr int SB_TABS = 4 * 32 * SB_LD + 8 * SB_WAVE_LDS;   // backward: offset of the row tables behind the 4 dQ slots and the waves' regions

// Developer instrumentation (tools/attn_sb_prof.py; build.py --variant prof -D SB_PROF): per-wave cycle counts of the kernels' phases,
// accumulated in registers and written to g_sb_prof[(block * 8 + wave) * 16 + phase] at the end.  Compiled out of the library.
#ifdef SB_PROF
__device__ long long* g_sb_prof = nullptr;
#define SB_PROF_DECL long long pacc_[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; long long tlast_ = clock64()
#define SB_T(i) do { const long long t_ = clock64(); pacc_[i] += t_ - tlast_; tlast_ = t_; } while (0)
#define SB_PROF_STORE(wave) do { if (g_sb_prof && (threadIdx.x & 63) == 0) { for (int i_ = 0; i_ < 16; ++i_) g_sb_prof[((int64_t)blockIdx.x * 8 + (wave)) * 16 + i_] = pacc_[i_]; } } while (0)
#define SB_PROF_ARG , long long (&pacc_)[16], long long& tlast_
#define SB_PROF_PASS , pacc_, tlast_
#else
#define SB_PROF_DECL do {} while (0)
#define SB_T(i) do {} while (0)
#define SB_PROF_STORE(wave) do {} while (0)
#define SB_PROF_ARG
#define SB_PROF_PASS
#endif

// ---- operand streaming --------------------------------------------------------------------------------------------------
// The loads of the unit loops are inline asm with a read-write ("+v") destination: the refill of an operand register is
// issued right behind the MFMAs that read it and lands IN PLACE, a whole unit later.  (Left to hipcc, the refill goes to a
// second register set -- 64 more VGPRs -- and the loop ends in 32-64 v_mov behind a vmcnt that drains the prefetch.)  hipcc
// does not count asm loads: every consumer is fenced by an explicit `s_waitcnt vmcnt(N)` that names the register ("+v": no
// consumer can be scheduled above it).  Loads return in order, and in the steady state every operand register has exactly 23
// younger loads in flight when it is needed (8 K + 16 V loads per unit, issued in consumption order), so N is 23 (a few more for the
// K groups whose burst of refills has already gone out); the last unit of an item issues nothing and counts down.
typedef unsigned sb_rsrc __attribute__((ext_vector_type(4)));
__device__ __forceinline__ sb_rsrc sb_make_rsrc(const float* base, unsigned bytes) {
    const unsigned long long a = (unsigned long long)base;
    sb_rsrc r;
    // (readfirstlane: the words are uniform anyway, but only this makes hipcc keep the descriptor in scalar registers everywhere --
    //  under register pressure it otherwise parks it in VGPRs, which a buffer instruction cannot encode)
    r.x = __builtin_amdgcn_readfirstlane((unsigned)a); r.y = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xFFFFu);
    r.z = __builtin_amdgcn_readfirstlane(bytes); r.w = 0x00020000u;
    return r;
}
// (developer ablations, tools/attn_sb_abl.sh: -DSB_ABL_NOLOAD drops the refills of the unit loops -- wrong results, the timing of everything
//  but the operand stream; -DSB_ABL_NOBAR the step barriers of the backward; -DSB_ABL_NOSLOT its dQ slot traffic)
template <int IMM>
__device__ __forceinline__ void sb_issue128(sb_u32x4& r, unsigned voff, sb_rsrc rs, unsigned soff) {
#if defined(SB_ABL_NOLOAD) || defined(SB_ABL_NOROWS)
    return;
#endif
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:%4" : "+v"(r) : "v"(voff), "s"(rs), "s"(soff), "i"(IMM));
}
template <int IMM>
__device__ __forceinline__ void sb_first128(sb_u32x4& r, unsigned voff, sb_rsrc rs, unsigned soff) {
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:%4" : "=v"(r) : "v"(voff), "s"(rs), "s"(soff), "i"(IMM));
}
__device__ __forceinline__ void sb_issue64(sb_u32x2& r, unsigned voff, sb_rsrc rs, unsigned soff) {
#if defined(SB_ABL_NOLOAD) || defined(SB_ABL_NOCOLS)
    return;
#endif
    asm volatile("buffer_load_dwordx2 %0, %1, %2, %3 offen" : "+v"(r) : "v"(voff), "s"(rs), "s"(soff));
}
__device__ __forceinline__ void sb_first64(sb_u32x2& r, unsigned voff, sb_rsrc rs, unsigned soff) {
    asm volatile("buffer_load_dwordx2 %0, %1, %2, %3 offen" : "=v"(r) : "v"(voff), "s"(rs), "s"(soff));
}
// column-type fetch with a RUNNING scalar offset: load, then advance the offset to the next row of the r(e) sequence inside the same
// statement (rows 0 1 2 3 8 9 10 11 ...: + pitch, after every fourth + 5 pitch).  Written as `base + r(e) * pitch` hipcc hoists the
// 3 x 16 products of a pair to its top and runs out of scalar registers (the buffer descriptors then land in VGPRs: no encoding).
__device__ __forceinline__ void sb_issue64_adv(sb_u32x2& r, unsigned voff, sb_rsrc rs, unsigned& soff, unsigned step) {
#if defined(SB_ABL_NOLOAD) || defined(SB_ABL_NOCOLS)
    return;
#endif
    // (readfirstlane: under scalar-register pressure hipcc parks the row step in a VGPR lane, which s_add_u32 cannot read)
    asm volatile("buffer_load_dwordx2 %0, %2, %3, %1 offen\n\ts_add_u32 %1, %1, %4" : "+v"(r), "+s"(soff) : "v"(voff), "s"(rs), "s"(__builtin_amdgcn_readfirstlane(step)) : "scc");   // (s_add writes SCC)
}
// The counts N below are HAND-COUNTED (hipcc does not see the asm loads): one edit to the issue order away from a silent race that only
// a value test at exactly this shape would catch.  -DSB_CHECK (build.py --variant sbcheck -D SB_CHECK) turns every counted wait into
// vmcnt(0) -- always safe, slower -- and tests/test_hip_parity.py::test_attention_sb_counted_waits_match_full_drain compares the two
// builds bit for bit on the C4 shape and the odd-slice / padding cases (round-5 review; tools/collect_profiles.sh runs it).
#ifdef SB_CHECK
#define SB_VM(N) 0
#else
#define SB_VM(N) (N)
#endif
template <int N>
__device__ __forceinline__ void sb_wait(sb_u32x4& r) { asm volatile("s_waitcnt vmcnt(%1)" : "+v"(r) : "i"(SB_VM(N))); }
template <int N>
__device__ __forceinline__ void sb_wait(sb_u32x2& r) { asm volatile("s_waitcnt vmcnt(%1)" : "+v"(r) : "i"(SB_VM(N))); }
template <int N>
__device__ __forceinline__ void sb_wait2(sb_u32x4& r, sb_u32x4& q) { asm volatile("s_waitcnt vmcnt(%2)" : "+v"(r), "+v"(q) : "i"(SB_VM(N))); }
// 32 staged rows of 64 floats (LDS row stride SB_LD) -> 32 tensor rows starting at byte offset `soff` of the slice behind `rs`, as
// whole 256-byte rows: 16 lanes per row, 4 rows per wave-store, the row step in the SCALAR offset (a 64-bit address per store is
// 16 VGPRs that hipcc hoists out of the pair loop)
__device__ __forceinline__ void sb_rows_out(const float* __restrict__ E, sb_rsrc rs, unsigned soff, unsigned pitchB, int lane) {
    const unsigned voff = (unsigned)(lane >> 4) * pitchB + (unsigned)(lane & 15) * 16u;
    const float* __restrict__ src = E + (lane >> 4) * SB_LD + (lane & 15) * 4;
    // all eight LDS reads first, then the eight stores: one LDS round trip, not eight (the stores touch memory nobody in these kernels
    // reads, so they carry no "memory" clobber that would pin each of them behind its own read)
    sb_u32x4 u[8];
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const float4 v = *reinterpret_cast<const float4*>(src + it * 4 * SB_LD);
        u[it].x = __float_as_uint(v.x); u[it].y = __float_as_uint(v.y); u[it].z = __float_as_uint(v.z); u[it].w = __float_as_uint(v.w);
    }
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        // (the s_nop: hipcc does not know that a store is still reading its data registers when the statement ends)
        asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen\n\ts_nop 1" : : "v"(u[it]), "v"(voff), "s"(rs), "s"(soff + (unsigned)it * 4u * pitchB));
    }
}
// acc[dt][e] = X^T[d = 2 (r(e) + 4 lh) + dt][row = l31]  ->  32 rows of 64 floats at byte offset soff of the slice behind rs, through the wave's rows E
__device__ __forceinline__ void sb_store_rows(float* __restrict__ E, const f32x16 (&acc)[2], float mul, sb_rsrc rs, unsigned soff, unsigned pitchB, int lane) {
    const int l31 = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        *reinterpret_cast<float4*>(&E[l31 * SB_LD + 16 * c + 8 * lh]) =
            make_float4(acc[0][4 * c] * mul, acc[1][4 * c] * mul, acc[0][4 * c + 1] * mul, acc[1][4 * c + 1] * mul);
        *reinterpret_cast<float4*>(&E[l31 * SB_LD + 16 * c + 8 * lh + 4]) =
            make_float4(acc[0][4 * c + 2] * mul, acc[1][4 * c + 2] * mul, acc[0][4 * c + 3] * mul, acc[1][4 * c + 3] * mul);
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);       // lgkmcnt(0): the staging rows are private to the wave
    __builtin_amdgcn_wave_barrier();
    sb_rows_out(E, rs, soff, pitchB, lane);
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();          // the staging rows are free again
}

__device__ __forceinline__ float sb_f(unsigned u) { return __uint_as_float(u); }
// a wave-uniform value the compiler must keep in a vector register from here on
__device__ __forceinline__ int sb_vec(int s) { int v; asm volatile("v_mov_b32 %0, %1" : "=v"(v) : "s"(s)); return v; }

// row r(e) of a 32-row group that accumulator register e carries for the lower half-wave (the upper one: + 4)
__device__ __forceinline__ constexpr int sb_row(int e) { return (e & 3) + 8 * (e >> 2); }

// max / sum over the lane pair (l, l + 32) -- the two halves of one query / key column
__device__ __forceinline__ float sb_pair_max(float v) { return fmaxf(v, __shfl_xor(v, 32, 64)); }
__device__ __forceinline__ float sb_pair_sum(float v) { return v + __shfl_xor(v, 32, 64); }

// padding bits of the 256 keys of a sequence as four scalars (an indexed array of them lands in scratch)
struct SbKeyBits { unsigned long long w0, w1, w2, w3; };
__device__ __forceinline__ unsigned sb_valid32(const SbKeyBits& k, int j) {
    const unsigned long long lo = j < 2 ? k.w0 : k.w1, hi = j < 6 ? k.w2 : k.w3;
    const unsigned long long w = j < 4 ? lo : hi;
    return (unsigned)(w >> (32 * (j & 1)));
}

// ballots of the padding flags of one sequence; fv = its first real key (256 if none)
__device__ __forceinline__ void sb_key_bits(const int32_t* __restrict__ kv, int lane, SbKeyBits& k, int& fv) {
    const int f0 = kv[lane], f1 = kv[64 + lane], f2 = kv[128 + lane], f3 = kv[192 + lane];
    k.w0 = __ballot(f0 != 0); k.w1 = __ballot(f1 != 0); k.w2 = __ballot(f2 != 0); k.w3 = __ballot(f3 != 0);
    fv = k.w0 ? (int)__builtin_ctzll(k.w0) : k.w1 ? 64 + (int)__builtin_ctzll(k.w1)
       : k.w2 ? 128 + (int)__builtin_ctzll(k.w2) : k.w3 ? 192 + (int)__builtin_ctzll(k.w3) : SB_T;
}

// =====================================================================================================
// forward
// =====================================================================================================
// One unit = one 32-key group against this wave's 32 queries: S^T = K_j Q^T (32 MFMAs), online softmax (lane <-> query),
// O^T += V_j^T P^T (32 MFMAs).  kreg / vreg hold group j on entry; unless LAST, group j + 1 is on its way into them on exit.
// Row-type refills go out in BURSTS OF FOUR: the four float4 of a lane pair's 128-byte line of each of the 32 rows.  One refill per MFMA
// group (256+ cycles apart, other waves' loads in between) found its line evicted from the 32 KB L1 every time: 32 L2 requests per
// instruction, 4x the useful bytes.  Back to back, the second to fourth load hit the line the first one is fetching.
#define SB_FWD_QK(g)                                                                      \
    do {                                                                                  \
        sb_wait<LAST ? 23 - (g) : ((g) < 4 ? 23 - (g) : 27 - (g))>(kreg[g]);              \
        s = AT_MFMA(sb_f(kreg[g].x), qf[g][0], s);                                        \
        s = AT_MFMA(sb_f(kreg[g].y), qf[g][1], s);                                        \
        s = AT_MFMA(sb_f(kreg[g].z), qf[g][2], s);                                        \
        s = AT_MFMA(sb_f(kreg[g].w), qf[g][3], s);                                        \
        if constexpr (!LAST && ((g) & 3) == 3) {                                          \
            sb_issue128<32 * ((g) - 3)>(kreg[(g) - 3], voffK, rk, soff_next);             \
            sb_issue128<32 * ((g) - 2)>(kreg[(g) - 2], voffK, rk, soff_next);             \
            sb_issue128<32 * ((g) - 1)>(kreg[(g) - 1], voffK, rk, soff_next);             \
            sb_issue128<32 * (g)>(kreg[g], voffK, rk, soff_next);                         \
        }                                                                                 \
    } while (0)
#define SB_FWD_PV(e)                                                                      \
    do {                                                                                  \
        sb_wait<LAST ? 15 - (e) : 23>(vreg[e]);                                           \
        o[0] = AT_MFMA(sb_f(vreg[e].x), s[e], o[0]);                                      \
        o[1] = AT_MFMA(sb_f(vreg[e].y), s[e], o[1]);                                      \
        if constexpr (!LAST) sb_issue64(vreg[e], voffV, rv, soff_next + (unsigned)sb_row(e) * pitchB); \
    } while (0)

template <bool MASKED, bool LAST>
__device__ __forceinline__ void sb_fwd_unit(f32x16 (&o)[2], float& m, float& l, sb_u32x4 (&kreg)[8], sb_u32x2 (&vreg)[16], const float (&qf)[8][4],
                                            const sb_rsrc rk, const sb_rsrc rv, const unsigned voffK, const unsigned voffV,
                                            const unsigned soff_next, const unsigned pitchB, const int lim_causal, const unsigned valid SB_PROF_ARG) {
    f32x16 s;
#pragma unroll
    for (int e = 0; e < 16; ++e) s[e] = 0.f;
    SB_T(1);
    SB_FWD_QK(0); SB_FWD_QK(1); SB_FWD_QK(2); SB_FWD_QK(3); SB_FWD_QK(4); SB_FWD_QK(5); SB_FWD_QK(6); SB_FWD_QK(7);
    SB_T(2);
    float mx = -INFINITY;
    if constexpr (MASKED) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int r = sb_row(e);
            if (r > lim_causal || !((valid >> r) & 1u)) s[e] = AT_MASKED2;
            mx = fmaxf(mx, s[e]);
        }
    } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) mx = fmaxf(mx, s[e]);
    }
    mx = sb_pair_max(mx);
    // lazy running maximum: m only moves when a row maximum outgrows it by more than 2^8 (always on the first unit: m = -inf)
    if (__ballot(mx > m + 8.0f) != 0ull) {
        const float m_new = fmaxf(m, mx);
        const float alpha = m_new == m ? 1.0f : __builtin_amdgcn_exp2f(m - m_new);
        m = m_new;
        l *= alpha;
#pragma unroll
        for (int e = 0; e < 16; ++e) { o[0][e] *= alpha; o[1][e] *= alpha; }
    }
    float ps = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        s[e] = __builtin_amdgcn_exp2f(s[e] - m);
        ps += s[e];
    }
    l += ps;
    SB_T(3);
    SB_FWD_PV(0); SB_FWD_PV(1); SB_FWD_PV(2); SB_FWD_PV(3); SB_FWD_PV(4); SB_FWD_PV(5); SB_FWD_PV(6); SB_FWD_PV(7);
    SB_FWD_PV(8); SB_FWD_PV(9); SB_FWD_PV(10); SB_FWD_PV(11); SB_FWD_PV(12); SB_FWD_PV(13); SB_FWD_PV(14); SB_FWD_PV(15);
    SB_T(4);
}

__global__ __launch_bounds__(512, 2) void attn_sb_fwd_kernel(const AttnParams p) {
    __shared__ __attribute__((aligned(16))) float smem[8 * 32 * SB_LD];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform: everything derived from it stays scalar
    const int l31 = lane & 31, lh = lane >> 5;
    const int BH = p.B * p.H;
    const unsigned pitchB = (unsigned)p.LQ * 4u;                                 // bytes between two rows of Q / K / V
    const unsigned slice_bytes = (unsigned)(((int64_t)(SB_T - 1) * p.LQ + SB_DH) * 4);
    const unsigned voffK = (unsigned)l31 * pitchB + 16u * lh;                    // lane <-> row l31, columns 8 g + 4 lh .. + 3
    const unsigned voffV = 4u * lh * pitchB + 8u * l31;                          // lane <-> columns 2 l31, 2 l31 + 1 of row r(e) + 4 lh
    const float qs = p.scale * AT_LOG2E;
    float* E = smem + wave * (32 * SB_LD);
    SB_PROF_DECL;

#pragma unroll 1
    for (int item = 0; item < 2; ++item) {
        const int bh = 2 * blockIdx.x + item;
        if (bh >= BH) break;
        const int rg = item == 0 ? SB_NG - 1 - wave : wave;                      // this wave's row group: queries 32 rg .. 32 rg + 31
        const int b = bh / p.H, h = bh - b * p.H;
        const float* Qb = p.Q + ((int64_t)b * SB_T) * p.LQ + (int64_t)h * SB_DH;
        const sb_rsrc rk = sb_make_rsrc(p.K + ((int64_t)b * SB_T) * p.LQ + (int64_t)h * SB_DH, slice_bytes);
        const sb_rsrc rv = sb_make_rsrc(p.V + ((int64_t)b * SB_T) * p.LQ + (int64_t)h * SB_DH, slice_bytes);

        // every load of the prologue is in flight before the first wait: key group 0, the Q fragments, the padding flags
        sb_u32x4 kreg[8];
        sb_u32x2 vreg[16];
        sb_first128<0>(kreg[0], voffK, rk, 0u); sb_first128<32>(kreg[1], voffK, rk, 0u); sb_first128<64>(kreg[2], voffK, rk, 0u);
        sb_first128<96>(kreg[3], voffK, rk, 0u); sb_first128<128>(kreg[4], voffK, rk, 0u); sb_first128<160>(kreg[5], voffK, rk, 0u);
        sb_first128<192>(kreg[6], voffK, rk, 0u); sb_first128<224>(kreg[7], voffK, rk, 0u);
        float qf[8][4];
        {
            const float* qrow = Qb + (int64_t)(32 * rg + l31) * p.LQ + 4 * lh;
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const float4 v = *reinterpret_cast<const float4*>(qrow + 8 * g);
                qf[g][0] = v.x; qf[g][1] = v.y; qf[g][2] = v.z; qf[g][3] = v.w;
            }
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) sb_first64(vreg[e], voffV, rv, (unsigned)sb_row(e) * pitchB);
        SbKeyBits kvbits = {~0ull, ~0ull, ~0ull, ~0ull};
        int fv = 0;
        if (p.key_valid) sb_key_bits(p.key_valid + (int64_t)b * SB_T, lane, kvbits, fv);
#pragma unroll
        for (int g = 0; g < 8; ++g)
#pragma unroll
            for (int j = 0; j < 4; ++j) qf[g][j] *= qs;

        // key groups above the diagonal contribute exp(-1e9 - m) = 0 and are skipped -- unless a query of this group has NO visible
        // real key: its softmax is uniform over ALL keys (every score is the same -1e9), so nothing may be skipped
        const int n_units = fv <= 32 * rg ? rg + 1 : SB_NG;
        f32x16 o[2];
#pragma unroll
        for (int e = 0; e < 16; ++e) { o[0][e] = 0.f; o[1][e] = 0.f; }
        float m = -INFINITY, l = 0.f;
        SB_T(0);
        // The order of the key groups is free (online softmax).  ONE loop body per kind of unit -- with the plain and the masked
        // body as alternatives inside one loop, hipcc gives the operand registers different homes per path and copies them around
        // while their loads are in flight: first every group that needs no masking (below the diagonal, all 32 keys real), then the
        // masked ones (padding inside; above the diagonal for a group with fully-masked rows), the diagonal group last.
        unsigned plain = 0u;
#pragma unroll
        for (int j = 0; j < SB_NG; ++j)
            if (j < rg && sb_valid32(kvbits, j) == 0xFFFFFFFFu) plain |= 1u << j;
        unsigned masked = (((1u << n_units) - 1u) & ~plain) & ~(1u << rg);
        // (the prologue fetched group 0: right unless group 0 is not the first one in this order -- then refetch; rare: padding in
        //  the first 32 keys of a sequence)
        const int first = plain ? (int)__builtin_ctz(plain) : masked ? (int)__builtin_ctz(masked) : rg;
        if (first != 0) {
            const unsigned so = (unsigned)(32 * first) * pitchB;
            sb_first128<0>(kreg[0], voffK, rk, so); sb_first128<32>(kreg[1], voffK, rk, so); sb_first128<64>(kreg[2], voffK, rk, so);
            sb_first128<96>(kreg[3], voffK, rk, so); sb_first128<128>(kreg[4], voffK, rk, so); sb_first128<160>(kreg[5], voffK, rk, so);
            sb_first128<192>(kreg[6], voffK, rk, so); sb_first128<224>(kreg[7], voffK, rk, so);
#pragma unroll
            for (int e = 0; e < 16; ++e) sb_first64(vreg[e], voffV, rv, so + (unsigned)sb_row(e) * pitchB);
        }
#pragma unroll 1
        while (plain) {
            plain &= plain - 1u;
            const int jn = plain ? (int)__builtin_ctz(plain) : masked ? (int)__builtin_ctz(masked) : rg;
            sb_fwd_unit<false, false>(o, m, l, kreg, vreg, qf, rk, rv, voffK, voffV, (unsigned)(32 * jn) * pitchB, pitchB, 0, 0u SB_PROF_PASS);
        }
#pragma unroll 1
        while (masked) {
            const int j = (int)__builtin_ctz(masked);
            masked &= masked - 1u;
            const int jn = masked ? (int)__builtin_ctz(masked) : rg;
            sb_fwd_unit<true, false>(o, m, l, kreg, vreg, qf, rk, rv, voffK, voffV, (unsigned)(32 * jn) * pitchB, pitchB, j < rg ? (1 << 20) : -1,
                                     sb_valid32(kvbits, j) >> (4 * lh) SB_PROF_PASS);
        }
        // the diagonal group: key r(e) + 4 lh is visible to query l31 iff it is <= l31
        sb_fwd_unit<true, true>(o, m, l, kreg, vreg, qf, rk, rv, voffK, voffV, 0u, pitchB, l31 - 4 * lh, sb_valid32(kvbits, rg) >> (4 * lh) SB_PROF_PASS);
        SB_T(5);
        // ---- finish: O = O^T / l; (m, log2 sum) stay apart: a fully-masked row has m = -1e9 log2e, where fp32 cannot hold m + log2 l
        const float lt = sb_pair_sum(l);
        const float inv = lt > 0.f ? 1.0f / lt : 0.f;
        const int q0 = 32 * rg;
        if (lh == 0) *reinterpret_cast<float2*>(p.LSE + 2 * ((int64_t)bh * SB_T + q0 + l31)) = make_float2(m, log2f(lt));
        // o[dt][e] = O^T[d = 2 (r(e) + 4 lh) + dt][q = l31] -> rows of the context tensor, through the wave's staging rows
        sb_store_rows(E, o, inv, sb_make_rsrc(p.O + ((int64_t)b * SB_T) * p.D + (int64_t)h * SB_DH, (unsigned)(((int64_t)(SB_T - 1) * p.D + SB_DH) * 4)),
                      (unsigned)q0 * (unsigned)p.D * 4u, (unsigned)p.D * 4u, lane);
        SB_T(6);
    }
    SB_PROF_STORE(wave);
}


// ---- round 6: two forwards that take their K / V operands from LDS tiles instead of streaming them into the MFMA registers ---------
// Why they were tried: the streamed forward's counters show the texture addresser's FIFOs full for two thirds of the kernel
// (SQ_VMEM_TA_ADDR_FIFO_FULL 2.4 M of 3.5 M busy cycles, matrix pipe busy 55 %; profiles/r05d_attention_pmc.md) -- a third of its
// 24 operand loads per unit are "lane <-> key row" (32 rows of 256 B touched by one instruction).  Result (MI355X, B64 H8, same box):
// streamed 55.2-58.9 us; BLOCK-SHARED tiles (all waves of a slice on the same key group per step, four tiles per step fetched once
// per block, a two-deep ring, one block barrier per step) 65 us -- the barrier aligns the two waves of every SIMD and the wave that
// finishes a row group holds the other seven up, nine times per block; WAVE-PRIVATE tiles (below: no barrier, whole-row loads one
// unit ahead) 60-63 us -- the 16 + 24 LDS instructions per unit cost more than the texture addresser's queueing did.  The streamed
// kernel stays the default; the wave-private variant is kept as a tested switch (NNHIP_ATTN_SB_FWD=pw), the block-shared one was
// deleted (EXPERIMENTS.md, round 6).
constexpr int SB_TILE = 32 * SB_LD;                       // floats of one staged tile

// sb_store_rows with COMPILER barriers around the wave-private transposition: E reaches sb_rows_out as a second pointer, and neither
// s_waitcnt nor wave_barrier orders memory accesses for hipcc -- inlined next to other LDS traffic the row reads were scheduled above
// the writes they transpose (right statistics, scrambled output)
__device__ __forceinline__ void sb_store_rows_fenced(float* E, const f32x16 (&acc)[2], float mul, sb_rsrc rs, unsigned soff, unsigned pitchB, int lane) {
    const int l31 = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        *reinterpret_cast<float4*>(&E[l31 * SB_LD + 16 * c + 8 * lh]) =
            make_float4(acc[0][4 * c] * mul, acc[1][4 * c] * mul, acc[0][4 * c + 1] * mul, acc[1][4 * c + 1] * mul);
        *reinterpret_cast<float4*>(&E[l31 * SB_LD + 16 * c + 8 * lh + 4]) =
            make_float4(acc[0][4 * c + 2] * mul, acc[1][4 * c + 2] * mul, acc[0][4 * c + 3] * mul, acc[1][4 * c + 3] * mul);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    sb_rows_out(E, rs, soff, pitchB, lane);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

template <bool MASKED>
__device__ __forceinline__ void sb_fwd_unit_lds(f32x16 (&o)[2], float& m, float& l, const float* __restrict__ Kt, const float* __restrict__ Vt,
                                                const float (&qf)[8][4], const int l31, const int lh, const int lim_causal, const unsigned valid) {
    f32x16 s;
#pragma unroll
    for (int e = 0; e < 16; ++e) s[e] = 0.f;
    const float* __restrict__ krow = Kt + l31 * SB_LD + 4 * lh;          // lane <-> key row l31, columns 8 g + 4 lh .. + 3
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        const float4 k4 = *reinterpret_cast<const float4*>(krow + 8 * g);
        s = AT_MFMA(k4.x, qf[g][0], s);
        s = AT_MFMA(k4.y, qf[g][1], s);
        s = AT_MFMA(k4.z, qf[g][2], s);
        s = AT_MFMA(k4.w, qf[g][3], s);
    }
    float mx = -INFINITY;
    if constexpr (MASKED) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int r = sb_row(e);
            if (r > lim_causal || !((valid >> r) & 1u)) s[e] = AT_MASKED2;
            mx = fmaxf(mx, s[e]);
        }
    } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) mx = fmaxf(mx, s[e]);
    }
    mx = sb_pair_max(mx);
    if (__ballot(mx > m + 8.0f) != 0ull) {                  // lazy running maximum (see sb_fwd_unit)
        const float m_new = fmaxf(m, mx);
        const float alpha = m_new == m ? 1.0f : __builtin_amdgcn_exp2f(m - m_new);
        m = m_new;
        l *= alpha;
#pragma unroll
        for (int e = 0; e < 16; ++e) { o[0][e] *= alpha; o[1][e] *= alpha; }
    }
    float ps = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        s[e] = __builtin_amdgcn_exp2f(s[e] - m);
        ps += s[e];
    }
    l += ps;
    const float* __restrict__ vcol = Vt + (4 * lh) * SB_LD + 2 * l31;    // lane <-> columns 2 l31, 2 l31 + 1 of row r(e) + 4 lh
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const float2 v2 = *reinterpret_cast<const float2*>(vcol + sb_row(e) * SB_LD);
        o[0] = AT_MFMA(v2.x, s[e], o[0]);
        o[1] = AT_MFMA(v2.y, s[e], o[1]);
    }
}


// ---- forward with WAVE-PRIVATE K / V tiles in LDS (round 6; opt-in NNHIP_ATTN_SB_FWD=pw) --------------------------------------------------------
// The streamed kernel's schedule (every wave free-running over its own key groups, no block barrier) with the operand path of the
// LDS kernel: a wave fetches its next (K_j, V_j) pair with whole-row loads (8 + 8 float4 per lane, 4 rows x 256 B per instruction)
// one unit ahead, parks it in its own [2][32][68] LDS tile pair and reads the MFMA fragments from there -- no "lane <-> key row"
// loads (the texture addresser's problem), no barrier (the LDS kernel's problem).  The tile pair is single-buffered: the unit's
// fragments are all in registers before the next pair is written.
__global__ __launch_bounds__(512, 2) void attn_sb_fwd_pw_kernel(const AttnParams p) {
    __shared__ __attribute__((aligned(16))) float smem[8 * 2 * SB_TILE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    const int BH = p.B * p.H;
    const float qs = p.scale * AT_LOG2E;
    float* Kt = smem + wave * 2 * SB_TILE;                  // (also the output staging rows at the end of an item)
    float* Vt = Kt + SB_TILE;
    // this lane's float4 of rows (lane >> 4) + 4 it: a per-lane byte offset computed ONCE; the group and the row step ride in the
    // buffer load's scalar offset (64-bit address arithmetic per load was 0.8 VALU per MFMA on lanes the MFMAs share)
    const unsigned pitchB = (unsigned)p.LQ * 4u;
    const unsigned voffT = (unsigned)(lane >> 4) * pitchB + 16u * (unsigned)(lane & 15);
    const unsigned slice_bytes = (unsigned)(((int64_t)(SB_T - 1) * p.LQ + SB_DH) * 4);
    const int loff = (lane >> 4) * SB_LD + 4 * (lane & 15);
#pragma unroll 1
    for (int item = 0; item < 2; ++item) {
        const int bh = 2 * blockIdx.x + item;
        if (bh >= BH) break;
        const int rg = item == 0 ? SB_NG - 1 - wave : wave;
        const int b = bh / p.H, h = bh - b * p.H;
        const float* __restrict__ Qb = p.Q + ((int64_t)b * SB_T) * p.LQ + (int64_t)h * SB_DH;
        const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.K + ((int64_t)b * SB_T) * p.LQ + (int64_t)h * SB_DH), 0, (int)slice_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.V + ((int64_t)b * SB_T) * p.LQ + (int64_t)h * SB_DH), 0, (int)slice_bytes, 0x00020000);
        float qf[8][4];
        {
            const float* qrow = Qb + (int64_t)(32 * rg + l31) * p.LQ + 4 * lh;
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const float4 v = *reinterpret_cast<const float4*>(qrow + 8 * g);
                qf[g][0] = v.x * qs; qf[g][1] = v.y * qs; qf[g][2] = v.z * qs; qf[g][3] = v.w * qs;
            }
        }
        SbKeyBits kvbits = {~0ull, ~0ull, ~0ull, ~0ull};
        int fv = 0;
        if (p.key_valid) sb_key_bits(p.key_valid + (int64_t)b * SB_T, lane, kvbits, fv);
        // a query with no visible real key is uniform over ALL keys: nothing above the diagonal may be skipped then
        const int n_units = fv <= 32 * rg ? rg + 1 : SB_NG;
        // the streamed kernel's order of the key groups (so that the two kernels agree bit for bit): the groups that need no masking,
        // then the masked ones (padding inside; above the diagonal for a group with fully-masked rows), the diagonal group last
        unsigned plain = 0u;
#pragma unroll
        for (int j = 0; j < SB_NG; ++j)
            if (j < rg && sb_valid32(kvbits, j) == 0xFFFFFFFFu) plain |= 1u << j;
        unsigned masked = (((1u << n_units) - 1u) & ~plain) & ~(1u << rg);
        int j = plain ? (int)__builtin_ctz(plain) : masked ? (int)__builtin_ctz(masked) : rg;
        sb_f32x4 nk[8], nv[8];               // (ext vectors: an array of HIP float4 structs went to scratch, 272 B per lane)
        {
            const unsigned so = (unsigned)(32 * j) * pitchB;
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                nk[it] = __builtin_bit_cast(sb_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rk, voffT, so + (unsigned)(4 * it) * pitchB, 0));
                nv[it] = __builtin_bit_cast(sb_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rv, voffT, so + (unsigned)(4 * it) * pitchB, 0));
            }
        }
        f32x16 o[2];
#pragma unroll
        for (int e = 0; e < 16; ++e) { o[0][e] = 0.f; o[1][e] = 0.f; }
        float m = -INFINITY, l = 0.f;
#pragma unroll 1
        for (int u = 0; u < n_units; ++u) {
            // park the pair that has arrived; the previous unit's fragments were read before its MFMAs (and the staging rows of the
            // previous item's output before that item ended)
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                *reinterpret_cast<sb_f32x4*>(Kt + loff + 4 * it * SB_LD) = nk[it];
                *reinterpret_cast<sb_f32x4*>(Vt + loff + 4 * it * SB_LD) = nv[it];
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            // this unit's class, and the group after it: plain groups first, then the masked ones, the diagonal last
            const bool is_plain = (plain >> j) & 1u;
            if (is_plain) plain &= ~(1u << j); else masked &= ~(1u << j);
            const int jn = plain ? (int)__builtin_ctz(plain) : masked ? (int)__builtin_ctz(masked) : rg;
            if (u + 1 < n_units) {                           // the next pair on its way
                const unsigned so = (unsigned)(32 * jn) * pitchB;
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    nk[it] = __builtin_bit_cast(sb_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rk, voffT, so + (unsigned)(4 * it) * pitchB, 0));
                    nv[it] = __builtin_bit_cast(sb_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rv, voffT, so + (unsigned)(4 * it) * pitchB, 0));
                }
            }
            const unsigned valid = sb_valid32(kvbits, j);
            if (j == rg) sb_fwd_unit_lds<true>(o, m, l, Kt, Vt, qf, l31, lh, l31 - 4 * lh, valid >> (4 * lh));
            else if (is_plain) sb_fwd_unit_lds<false>(o, m, l, Kt, Vt, qf, l31, lh, 0, 0u);
            else sb_fwd_unit_lds<true>(o, m, l, Kt, Vt, qf, l31, lh, j < rg ? (1 << 20) : -1, valid >> (4 * lh));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // the unit's fragment reads are done: the tile pair may be overwritten
            __builtin_amdgcn_wave_barrier();
            j = jn;
        }
        const float lt = sb_pair_sum(l);
        const float inv = lt > 0.f ? 1.0f / lt : 0.f;
        const int q0 = 32 * rg;
        if (lh == 0) *reinterpret_cast<float2*>(p.LSE + 2 * ((int64_t)bh * SB_T + q0 + l31)) = make_float2(m, log2f(lt));
        sb_store_rows_fenced(Kt, o, inv, sb_make_rsrc(p.O + ((int64_t)b * SB_T) * p.D + (int64_t)h * SB_DH, (unsigned)(((int64_t)(SB_T - 1) * p.D + SB_DH) * 4)),
                      (unsigned)q0 * (unsigned)p.D * 4u, (unsigned)p.D * 4u, lane);
    }
}


// =====================================================================================================
// backward: one pass, 5 GEMM-units per (row group i, key group j) pair
// =====================================================================================================
// The wave that owns key group j keeps K_j / V_j fragments (the B operands of S and dP: lane <-> key) and the dK_j^T / dV_j^T
// accumulators in registers and walks the row groups i >= j.  Per pair, in this order:
//   U1  S[q, key]  = Q_i K_j^T           A = Q_i rows   (lane <-> query row, float4)     B = K_j fragments (pre-scaled, log2 units)
//   U2  dP[q, key] = dO_i V_j^T          A = dO_i rows                                   B = V_j fragments
//       P = exp2(S_masked - m[q] - l[q]),  dS = P (dP - Dsum[q]) scale, 0 where masked  (lane <-> key, registers <-> queries)
//   U3  dV_j^T[d, key] += dO_i^T P       A = dO_i columns (lane <-> two head-dim columns, float2)    B = P, from its registers
//   U4  dK_j^T[d, key] += Q_i^T dS       A = Q_i columns                                             B = dS, from its registers
//       dS -> wave-private LDS -> dS^T (lane <-> query)
//   U5  dQ_i^T[d, q]  (+)= K_j^T dS^T    A = K_j columns                                             B = dS^T
// Two rolling operand buffers: R (8 float4: Q_i rows -> dO_i rows -> Q_i' rows ...) and C (16 float2: dO_i columns -> Q_i columns
// -> K_j columns -> dO_i' columns ...).  Every register is refilled in place right behind the MFMAs that read it; the refills with
// a budget of only one unit (dO_i rows, Q_i columns, K_j columns) hit lines that the long-budget loads of the same tile (dO_i
// columns, Q_i rows: issued three units ahead) have already brought in.  Loads return in order and are issued in consumption
// order, 80 per pair, so every wait is a constant.  The units run in the order U1, P, U3, U2, dS, U4, U5: the buffers that U1 empties (R, RB)
// are not needed again before U2 -- a unit and the P pass later -- and C's refill for U4 has U2 to land.
// K_j lives in the wave's LDS tile kt[32 keys][SB_LD] for the whole phase (round 5, third step: its row fragments for U1 and its columns for
// U5 were 24 of the 80 loads of a pair, re-fetched from L2 / fabric for every pair because 256 VGPRs cannot hold them).  RB[g] and C[e]
// are refilled from LDS with plain loads (the compiler waits for those itself); the rolling VMEM stream of a pair is now 56 loads:
//   A, B  (in U1, after g = 3 / 7):  R[g-3..g] <- dO_i rows, RB[g-3..g] <- V_j rows, interleaved          8 + 8
//   C     (in U3):                   C[e] <- Q_i columns                                                   16
//   D, E  (in U2, after g = 3 / 7):  R[g-3..g] <- Q_i' rows                                                4 + 4
//   G     (in U5):                   C[e] <- dO_i' columns                                                 16
// and every wait below counts the loads issued AFTER the one it needs.
#define SB_BWD_BURST_RV(g)                                                                \
    do {                                                                                  \
        sb_issue128<32 * ((g) - 3)>(R[(g) - 3], voffRg, rg, soG_i);                       \
        sb_issue128<32 * ((g) - 3)>(RB[(g) - 3], voffRq, rv, soK_j);                      \
        sb_issue128<32 * ((g) - 2)>(R[(g) - 2], voffRg, rg, soG_i);                       \
        sb_issue128<32 * ((g) - 2)>(RB[(g) - 2], voffRq, rv, soK_j);                      \
        sb_issue128<32 * ((g) - 1)>(R[(g) - 1], voffRg, rg, soG_i);                       \
        sb_issue128<32 * ((g) - 1)>(RB[(g) - 1], voffRq, rv, soK_j);                      \
        sb_issue128<32 * (g)>(R[g], voffRg, rg, soG_i);                                   \
        sb_issue128<32 * (g)>(RB[g], voffRq, rv, soK_j);                                  \
    } while (0)
#define SB_BWD_BURST_R(g)                                                                 \
    do {                                                                                  \
        sb_issue128<32 * ((g) - 3)>(R[(g) - 3], voffRq, rq, soQ_n);                       \
        sb_issue128<32 * ((g) - 2)>(R[(g) - 2], voffRq, rq, soQ_n);                       \
        sb_issue128<32 * ((g) - 1)>(R[(g) - 1], voffRq, rq, soQ_n);                       \
        sb_issue128<32 * (g)>(R[g], voffRq, rq, soQ_n);                                   \
        RB[(g) - 3] = krow[2 * ((g) - 3)]; RB[(g) - 2] = krow[2 * ((g) - 2)]; RB[(g) - 1] = krow[2 * ((g) - 1)]; RB[g] = krow[2 * (g)];   \
    } while (0)
#define SB_BWD_U1(g)                                                                      \
    do {                                                                                  \
        sb_wait<((g) < 4 ? 23 : 27) - ((g) & 3)>(R[g]);                                   \
        s = AT_MFMA(sb_f(R[g].x), sb_f(RB[g].x), s);                                      \
        s = AT_MFMA(sb_f(R[g].y), sb_f(RB[g].y), s);                                      \
        s = AT_MFMA(sb_f(R[g].z), sb_f(RB[g].z), s);                                      \
        s = AT_MFMA(sb_f(R[g].w), sb_f(RB[g].w), s);                                      \
        if constexpr (((g) & 3) == 3) SB_BWD_BURST_RV(g);                                 \
    } while (0)
#define SB_BWD_U2(g)                                                                      \
    do {                                                                                  \
        sb_wait2<((g) < 4 ? 30 : 26) - 2 * ((g) & 3)>(R[g], RB[g]);                       \
        dp = AT_MFMA(sb_f(R[g].x), sb_f(RB[g].x), dp);                                    \
        dp = AT_MFMA(sb_f(R[g].y), sb_f(RB[g].y), dp);                                    \
        dp = AT_MFMA(sb_f(R[g].z), sb_f(RB[g].z), dp);                                    \
        dp = AT_MFMA(sb_f(R[g].w), sb_f(RB[g].w), dp);                                    \
        if constexpr (((g) & 3) == 3) SB_BWD_BURST_R(g);                                  \
    } while (0)
#define SB_BWD_U3(e)                                                                      \
    do {                                                                                  \
        sb_wait<31>(C[e]);                                                                \
        dv[0] = AT_MFMA(sb_f(C[e].x), s[e], dv[0]);                                       \
        dv[1] = AT_MFMA(sb_f(C[e].y), s[e], dv[1]);                                       \
        sb_issue64_adv(C[e], voffCq, rq, so3, ((e) & 3) == 3 ? 5u * pitchQ : pitchQ); \
    } while (0)
#define SB_BWD_U4(e)                                                                      \
    do {                                                                                  \
        sb_wait<23 - (e)>(C[e]);                                                          \
        dk[0] = AT_MFMA(sb_f(C[e].x), dp[e], dk[0]);                                      \
        dk[1] = AT_MFMA(sb_f(C[e].y), dp[e], dk[1]);                                      \
        C[e] = kcol[sb_row(e) * (SB_LD / 2)];                                             \
    } while (0)
#define SB_BWD_U5(e)                                                                      \
    do {                                                                                  \
        dq[0] = AT_MFMA(sb_f(C[e].x), dst[e], dq[0]);                                     \
        dq[1] = AT_MFMA(sb_f(C[e].y), dst[e], dq[1]);                                     \
        sb_issue64_adv(C[e], voffCg, rg, so5, ((e) & 3) == 3 ? 5u * pitchG : pitchG); \
    } while (0)

// `slot`: the LDS rows where the partial dQ of row group i is summed (first: store, otherwise add; last: the sum goes out).
// Everything that does not change while a wave works on one (slice, key group) is passed BY VALUE: gathered in a struct it stays in
// memory across the four inlined copies of this body, and a buffer descriptor reloaded from scratch is a VGPR no buffer
// instruction can encode.
template <bool DIAG>
__device__ __forceinline__ void sb_bwd_pair(f32x16 (&dk)[2], f32x16 (&dv)[2], sb_u32x4 (&R)[8], sb_u32x4 (&RB)[8], sb_u32x2 (&C)[16],
                                            const sb_rsrc rq, const sb_rsrc rk, const sb_rsrc rv, const sb_rsrc rg,      // Q, K, V, dO rows of the slice
                                            const unsigned voffRq, const unsigned voffRg, const unsigned voffCq, const unsigned voffCg,
                                            const unsigned pitchQ, const unsigned pitchG,
                                            const float* __restrict__ Mt,          // LDS: max | log2 sum | Dsum of the slice's 256 queries
                                            float* __restrict__ scratch,           // LDS: this wave's dS patch [32][36]
                                            const float* __restrict__ kt,          // LDS: this wave's K_j tile [32][SB_LD]
                                            const sb_rsrc rdq,                     // global: dQ rows of the slice
                                            const float scale, const float sl2, const bool key_pad, const int j, const int i, const int i_next,
                                            float* __restrict__ slot, int* __restrict__ seq, const int turn, const bool first, const bool last, const int lane SB_PROF_ARG) {
    const int l31 = lane & 31, lh = lane >> 5;
    const float* __restrict__ Lt = Mt + SB_T;
    const float* __restrict__ Dt = Mt + 2 * SB_T;
    const unsigned soQ_i = (unsigned)(32 * i) * pitchQ, soG_i = (unsigned)(32 * i) * pitchG;
    const unsigned soQ_n = (unsigned)(32 * i_next) * pitchQ, soG_n = (unsigned)(32 * i_next) * pitchG;
    const unsigned soK_j = (unsigned)(32 * j) * pitchQ;
    unsigned so3 = soQ_i, so5 = soG_n;                       // running row offsets of the two column-type streams
    // K_j out of the wave's tile: row fragments (lane <-> key l31, 16 bytes at dh 8 g + 4 lh) and columns (key r(e) + 4 lh, dh 2 l31, + 1)
    const sb_u32x4* __restrict__ krow = reinterpret_cast<const sb_u32x4*>(kt + l31 * SB_LD + 4 * lh);        // [g]: + 8 g floats = + 2 g
    const sb_u32x2* __restrict__ kcol = reinterpret_cast<const sb_u32x2*>(kt + 4 * lh * SB_LD + 2 * l31);    // [row * SB_LD / 2]
    f32x16 s, dp;
#pragma unroll
    for (int e = 0; e < 16; ++e) { s[e] = 0.f; dp[e] = 0.f; }
    SB_T(11);
    SB_BWD_U1(0); SB_BWD_U1(1); SB_BWD_U1(2); SB_BWD_U1(3); SB_BWD_U1(4); SB_BWD_U1(5); SB_BWD_U1(6); SB_BWD_U1(7);
    SB_T(1);
    // ---- P (into s); register e <-> query 32 i + r(e) + 4 lh, lane <-> key 32 j + l31 -------------------------------------------------
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float4 M4 = *reinterpret_cast<const float4*>(&Mt[32 * i + 8 * c + 4 * lh]);
        const float4 L4 = *reinterpret_cast<const float4*>(&Lt[32 * i + 8 * c + 4 * lh]);
        const float Mr[4] = {M4.x, M4.y, M4.z, M4.w}, Lr[4] = {L4.x, L4.y, L4.z, L4.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int e = 4 * c + r;
            bool masked = key_pad;
            if constexpr (DIAG) masked = masked || (8 * c + r + 4 * lh < l31);      // the query comes before the key
            s[e] = __builtin_amdgcn_exp2f((masked ? AT_MASKED2 - Mr[r] : fmaf(s[e], sl2, -Mr[r])) - Lr[r]);
        }
    }
    SB_T(3);
    SB_BWD_U3(0); SB_BWD_U3(1); SB_BWD_U3(2); SB_BWD_U3(3); SB_BWD_U3(4); SB_BWD_U3(5); SB_BWD_U3(6); SB_BWD_U3(7);
    SB_BWD_U3(8); SB_BWD_U3(9); SB_BWD_U3(10); SB_BWD_U3(11); SB_BWD_U3(12); SB_BWD_U3(13); SB_BWD_U3(14); SB_BWD_U3(15);
    SB_T(4);
    SB_BWD_U2(0); SB_BWD_U2(1); SB_BWD_U2(2); SB_BWD_U2(3); SB_BWD_U2(4); SB_BWD_U2(5); SB_BWD_U2(6); SB_BWD_U2(7);
    SB_T(2);
    // ---- dS = P (dP - Dsum[q]) scale (into dp), 0 where masked ----------------------------------------------------------------
    const float kscale = key_pad ? 0.f : scale;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float4 D4 = *reinterpret_cast<const float4*>(&Dt[32 * i + 8 * c + 4 * lh]);
        const float Dr[4] = {D4.x, D4.y, D4.z, D4.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int e = 4 * c + r;
            float ds = s[e] * ((dp[e] - Dr[r]) * kscale);
            if constexpr (DIAG) ds = (8 * c + r + 4 * lh < l31) ? 0.f : ds;
            dp[e] = ds;
        }
    }
    SB_T(12);
    // dS -> LDS [query][key] (row stride 36) while the dK MFMAs run; read back with lane <-> query
#pragma unroll
    for (int e = 0; e < 16; ++e) scratch[(sb_row(e) + 4 * lh) * 36 + l31] = dp[e];
    SB_BWD_U4(0); SB_BWD_U4(1); SB_BWD_U4(2); SB_BWD_U4(3); SB_BWD_U4(4); SB_BWD_U4(5); SB_BWD_U4(6); SB_BWD_U4(7);
    SB_BWD_U4(8); SB_BWD_U4(9); SB_BWD_U4(10); SB_BWD_U4(11); SB_BWD_U4(12); SB_BWD_U4(13); SB_BWD_U4(14); SB_BWD_U4(15);
    SB_T(5);
    __builtin_amdgcn_s_waitcnt(0xc07f);       // lgkmcnt(0): the staging rows are private to the wave
    __builtin_amdgcn_wave_barrier();
    float dst[16];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float4 t = *reinterpret_cast<const float4*>(&scratch[l31 * 36 + 8 * c + 4 * lh]);
        dst[4 * c] = t.x; dst[4 * c + 1] = t.y; dst[4 * c + 2] = t.z; dst[4 * c + 3] = t.w;
    }
    f32x16 dq[2];
#pragma unroll
    for (int e = 0; e < 16; ++e) { dq[0][e] = 0.f; dq[1][e] = 0.f; }
    SB_T(6);
    SB_BWD_U5(0); SB_BWD_U5(1); SB_BWD_U5(2); SB_BWD_U5(3); SB_BWD_U5(4); SB_BWD_U5(5); SB_BWD_U5(6); SB_BWD_U5(7);
    SB_BWD_U5(8); SB_BWD_U5(9); SB_BWD_U5(10); SB_BWD_U5(11); SB_BWD_U5(12); SB_BWD_U5(13); SB_BWD_U5(14); SB_BWD_U5(15);
    SB_T(7);
    // ---- this pair's part of dQ_i: dq[dt][e] = dQ^T[d = 2 (r(e) + 4 lh) + dt][q = l31] -> slot row q, columns 16c + 8lh .. + 7 ------
    // The contributions to a slot are added in a FIXED order (deterministic bits): `seq` counts them, this pair's turn is `turn`.
    // The contributor before it ran one pair-time earlier on another wave, so the wait is a formality -- but it is what
    // orders the waves, not a block barrier: the two waves of a SIMD stay out of phase and fill each other's bubbles.
    {
        volatile int* vs = seq;
        while (*vs != turn) __builtin_amdgcn_s_sleep(1);
    }
    // (acquire: no slot access below may be moved above the spin by the compiler -- the hardware keeps a wave's LDS accesses in
    //  order, the compiler knows nothing of the protocol; advisor, round 5)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    SB_T(9);
    float* __restrict__ row = slot + l31 * SB_LD + 8 * lh;
#ifdef SB_ABL_NOSLOT
    if (dq[0][0] == 12345.f && dq[1][5] == 54321.f)     // (keeps the MFMAs alive)
#endif
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        float4 x = make_float4(dq[0][4 * c], dq[1][4 * c], dq[0][4 * c + 1], dq[1][4 * c + 1]);
        float4 y = make_float4(dq[0][4 * c + 2], dq[1][4 * c + 2], dq[0][4 * c + 3], dq[1][4 * c + 3]);
        if (!first) {
            const float4 px = *reinterpret_cast<const float4*>(row + 16 * c), py = *reinterpret_cast<const float4*>(row + 16 * c + 4);
            x.x += px.x; x.y += px.y; x.z += px.z; x.w += px.w;
            y.x += py.x; y.y += py.y; y.z += py.z; y.w += py.w;
        }
        *reinterpret_cast<float4*>(row + 16 * c) = x;
        *reinterpret_cast<float4*>(row + 16 * c + 4) = y;
    }
    if (last) {                               // wave-uniform: the row group is complete -- write it out as whole rows
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        sb_rows_out(slot, rdq, soQ_i, pitchQ, lane);
    }
    // LDS executes a wave's accesses in order: the count moves after this pair's writes (and, for the last one, its reads) of the slot
    // (release: the compiler may not sink a slot store below the count's)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    *reinterpret_cast<volatile int*>(seq) = turn + 1;
    SB_T(8);
}

// step barrier: LDS traffic of this step done, then meet -- and nothing else (a __syncthreads() would also wait for this wave's
// global stores, i.e. drain the operand prefetch with them: vmcnt counts every kind of access)
__device__ __forceinline__ void sb_step_barrier() {
#ifdef SB_ABL_NOBAR
    return;
#endif
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// dK / dV rows of key group j of slice bh, through the wave's staging rows
__device__ __forceinline__ void sb_bwd_store_kv(const AttnBwdParams& p, const int bh, const int j, float* __restrict__ E, const f32x16 (&dk)[2],
                                                const f32x16 (&dv)[2], const int lane) {
    const int b = bh / p.H, h = bh - b * p.H;
    const int64_t base = ((int64_t)b * SB_T) * p.LQ + (int64_t)h * SB_DH;
    const unsigned pitchQ = (unsigned)p.LQ * 4u, bytesQ = (unsigned)(((int64_t)(SB_T - 1) * p.LQ + SB_DH) * 4);
    sb_store_rows(E, dk, 1.0f, sb_make_rsrc(p.dK + base, bytesQ), (unsigned)(32 * j) * pitchQ, pitchQ, lane);
    sb_store_rows(E, dv, 1.0f, sb_make_rsrc(p.dV + base, bytesQ), (unsigned)(32 * j) * pitchQ, pitchQ, lane);
}

// (max, log2 sum, Dsum[q] = sum_d dO[q,d] O[q,d]) of both slices' 256 queries into LDS: wave w takes queries 32 w .. 32 w + 31
__device__ __forceinline__ void sb_bwd_row_tables(const AttnBwdParams& p, float* __restrict__ tabs, const int wave, const int lane) {
    const int l31 = lane & 31, lh = lane >> 5;
#pragma unroll 1
    for (int sl = 0; sl < 2; ++sl) {
        const int bh = 2 * blockIdx.x + sl;
        if (bh >= p.B * p.H) break;
        const int b = bh / p.H, h = bh - b * p.H;
        const int q = 32 * wave + l31;
        const float* go = p.dO + ((int64_t)b * SB_T + q) * p.D + (int64_t)h * SB_DH + 4 * lh;
        const float* oo = p.O + ((int64_t)b * SB_T + q) * p.D + (int64_t)h * SB_DH + 4 * lh;
        float part = 0.f;
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const float4 x = *reinterpret_cast<const float4*>(go + 8 * g), y = *reinterpret_cast<const float4*>(oo + 8 * g);
            part += x.x * y.x + x.y * y.y + x.z * y.z + x.w * y.w;
        }
        const float ds = sb_pair_sum(part);
        const float2 ml = reinterpret_cast<const float2*>(p.LSE)[(int64_t)bh * SB_T + q];
        if (lh == 0) {
            float* t = tabs + sl * 3 * SB_T;
            t[q] = ml.x; t[SB_T + q] = ml.y; t[2 * SB_T + q] = ds;
        }
    }
}

// phase 0: key group `wave` of slice A, pairs (i, j) for i = 7 down to j (steps 0 .. 7 - wave: diagonal last);
// phase 1: key group 7 - wave of slice B, i = j .. 7 (diagonal first).  No block barrier: the dQ slots are ordered by their counts.
template <int PH>
__device__ __forceinline__ void sb_bwd_phase(const AttnBwdParams& p, float* __restrict__ smem, const int wave, const int lane, f32x16 (&dk)[2],
                                             f32x16 (&dv)[2], sb_u32x4 (&R)[8], sb_u32x4 (&RB)[8], sb_u32x2 (&C)[16] SB_PROF_ARG) {
    constexpr int SLOT = 32 * SB_LD;
    const int l31 = lane & 31, lh = lane >> 5;
    const int bh = 2 * blockIdx.x + PH;
    const int j = PH == 0 ? wave : SB_NG - 1 - wave;
    const int npairs = SB_NG - j;
    float* __restrict__ slots = smem;
    float* __restrict__ E = smem + 4 * SLOT + wave * SB_WAVE_LDS;      // the wave's K_j tile [32][SB_LD]; its dK / dV rows are staged here when the phase is over
    float* __restrict__ patch = E + SLOT;                               // the wave's dS transposition patch [32][36]
    if (bh >= p.B * p.H) {                    // block-uniform: an odd number of slices leaves the last block without a B
        if constexpr (PH == 1) sb_bwd_store_kv(p, bh - 1, SB_NG - 1 - j, E, dk, dv, lane);
        return;
    }
    const float* __restrict__ Mt = smem + SB_TABS + PH * 3 * SB_T;
    const unsigned pitchQ = (unsigned)p.LQ * 4u, pitchG = (unsigned)p.D * 4u;
    const unsigned bytesQ = (unsigned)(((int64_t)(SB_T - 1) * p.LQ + SB_DH) * 4), bytesG = (unsigned)(((int64_t)(SB_T - 1) * p.D + SB_DH) * 4);
    const float sl2 = p.scale * AT_LOG2E;
    const int b = bh / p.H, h = bh - b * p.H;
    const int64_t base = ((int64_t)b * SB_T) * p.LQ + (int64_t)h * SB_DH;
    const sb_rsrc rq = sb_make_rsrc(p.Q + base, bytesQ), rk = sb_make_rsrc(p.K + base, bytesQ), rv = sb_make_rsrc(p.V + base, bytesQ);
    const sb_rsrc rg = sb_make_rsrc(p.dO + ((int64_t)b * SB_T) * p.D + (int64_t)h * SB_DH, bytesG);
    const unsigned voffRq = (unsigned)l31 * pitchQ + 16u * lh, voffRg = (unsigned)l31 * pitchG + 16u * lh;   // row-type fetch: lane <-> row l31
    const unsigned voffCq = 4u * lh * pitchQ + 8u * l31, voffCg = 4u * lh * pitchG + 8u * l31;               // column-type: lane <-> columns 2 l31, + 1
    const sb_rsrc rdq = sb_make_rsrc(p.dQ + base, bytesQ);
    int fv = 0;
    SbKeyBits kvbits = {~0ull, ~0ull, ~0ull, ~0ull};
    if (p.key_valid) sb_key_bits(p.key_valid + (int64_t)b * SB_T, lane, kvbits, fv);
    const bool key_pad = !((sb_valid32(kvbits, j) >> l31) & 1u);           // this lane's key is padding
    // the first pair's operands: Q rows / dO columns of its row group, the K_j fragments (lane <-> key row: the B operand of S)
    const int i0 = PH == 0 ? SB_NG - 1 : j;                // the first pair of the phase (see the order of the pairs below)
    const unsigned so0 = (unsigned)(32 * i0) * pitchQ, sok = (unsigned)(32 * j) * pitchQ;
    sb_first128<0>(R[0], voffRq, rq, so0); sb_first128<32>(R[1], voffRq, rq, so0); sb_first128<64>(R[2], voffRq, rq, so0);
    sb_first128<96>(R[3], voffRq, rq, so0); sb_first128<128>(R[4], voffRq, rq, so0); sb_first128<160>(R[5], voffRq, rq, so0);
    sb_first128<192>(R[6], voffRq, rq, so0); sb_first128<224>(R[7], voffRq, rq, so0);
    sb_first128<0>(RB[0], voffRq, rk, sok); sb_first128<32>(RB[1], voffRq, rk, sok); sb_first128<64>(RB[2], voffRq, rk, sok);
    sb_first128<96>(RB[3], voffRq, rk, sok); sb_first128<128>(RB[4], voffRq, rk, sok); sb_first128<160>(RB[5], voffRq, rk, sok);
    sb_first128<192>(RB[6], voffRq, rk, sok); sb_first128<224>(RB[7], voffRq, rk, sok);
#pragma unroll
    for (int e = 0; e < 16; ++e) sb_first64(C[e], voffCg, rg, (unsigned)(32 * i0 + sb_row(e)) * pitchG);
    // ... and while they fly: phase 0 -- the row statistics of both slices go to LDS (every wave 32 queries; one block barrier);
    // phase 1 -- the dK / dV rows of phase 0 go out (their stores are younger than the loads above; the drain below takes both)
    if constexpr (PH == 0) {
        sb_bwd_row_tables(p, smem + SB_TABS, wave, lane);
        __syncthreads();
    } else {
        sb_bwd_store_kv(p, bh - 1, wave, E, dk, dv, lane);
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) { dk[0][e] = 0.f; dk[1][e] = 0.f; dv[0][e] = 0.f; dv[1][e] = 0.f; }
    // the first pair's waits are written for the steady state (62 / 47 younger loads): have its operands landed instead
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(R[0]), "+v"(R[1]), "+v"(R[2]), "+v"(R[3]), "+v"(R[4]), "+v"(R[5]), "+v"(R[6]), "+v"(R[7]));
    asm volatile("" : "+v"(RB[0]), "+v"(RB[1]), "+v"(RB[2]), "+v"(RB[3]), "+v"(RB[4]), "+v"(RB[5]), "+v"(RB[6]), "+v"(RB[7]));
    asm volatile("" : "+v"(C[0]), "+v"(C[1]), "+v"(C[2]), "+v"(C[3]), "+v"(C[4]), "+v"(C[5]), "+v"(C[6]), "+v"(C[7]));
    asm volatile("" : "+v"(C[8]), "+v"(C[9]), "+v"(C[10]), "+v"(C[11]), "+v"(C[12]), "+v"(C[13]), "+v"(C[14]), "+v"(C[15]));
    // K_j into the wave's tile (the staging use of these rows -- phase 0's dK / dV above -- has finished: sb_store_rows ends on lgkmcnt(0))
#pragma unroll
    for (int g = 0; g < 8; ++g) *reinterpret_cast<sb_u32x4*>(E + l31 * SB_LD + 8 * g + 4 * lh) = RB[g];
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    SB_T(0);
#define SB_PAIR_ARGS dk, dv, R, RB, C, rq, rk, rv, rg, voffRq, voffRg, voffCq, voffCg, pitchQ, pitchG, Mt, patch, E, rdq, p.scale, sl2, key_pad, j
    // Order of the pairs (round 5, second schedule): at step t of the block's nine steps EVERY phase-0 wave is on row group 7 - t of
    // slice A (i = 7 down to j, the diagonal pair last) and EVERY phase-1 wave on row group t - 1 of slice B (i = j up to 7, the
    // diagonal pair first; a wave enters phase 1 at step 8 - wave): the up to eight waves that need the same Q_i / dO_i tile ask for it
    // at the same time and find each other's lines in L1 / L2 -- walked one step apart (first schedule: phase 0 upwards from the
    // diagonal) a tile had left the XCD's L2 (128 KB per CU against 450 KB of requests per step) before its next reader came.
    // dQ slots: only one row group per slice is being summed at a time, so two slots per slice, reused every second row group
    // (A: slots 0-1, B: slots 2-3); the contribution COUNT of a slot orders everything -- phase 0: key groups 0, 1, .. i (the
    // diagonal pair writes the row group out), phase 1: key groups i, i - 1, .. 0 (key group 0 writes it out); a slot's second
    // row group starts at the count its first one ends with.  Within a step the waves contribute in ascending wave order in both
    // phases, and a row group's first contribution comes two steps after the previous tenant's last: no wave waits for a later one.
    // (slot index and turn live in VECTOR registers -- sb_vec: as scalars they were the handful of SGPRs too many, and the pair body's buffer
    //  descriptors went to VGPRs, which no buffer instruction encodes)
    int* __restrict__ seqs = reinterpret_cast<int*>(smem + SB_TABS + 2 * 3 * SB_T);
    // turn of the first contribution of row group i in its slot (two slots per slice, a slot's tenants in time order -- A: 7 5 3 1 / 6 4 2 0
    // with i + 1 contributions each, B: 1 3 5 7 / 0 2 4 6): eight 5-bit entries per table
    constexpr unsigned long long TA = 15ull | 18ull << 5 | 12ull << 10 | 14ull << 15 | 7ull << 20 | 8ull << 25 | 0ull << 30 | 0ull << 35;
    constexpr unsigned long long TB = 0ull | 0ull << 5 | 1ull << 10 | 2ull << 15 | 4ull << 20 | 6ull << 25 | 9ull << 30 | 12ull << 35;
    if constexpr (PH == 0) {
#pragma unroll 1
        for (int s = 0; s + 1 < npairs; ++s) {
            const int i = SB_NG - 1 - s;
            const int iv = sb_vec(i), x = iv & 1;
            sb_bwd_pair<false>(SB_PAIR_ARGS, i, i - 1, slots + x * SLOT, seqs + x, (int)((TA >> (5 * iv)) & 31) + j, j == 0, false, lane SB_PROF_PASS);
        }
        const int jv = sb_vec(j), x = jv & 1;
        sb_bwd_pair<true>(SB_PAIR_ARGS, j, j, slots + x * SLOT, seqs + x, (int)((TA >> (5 * jv)) & 31) + jv, j == 0, true, lane SB_PROF_PASS);
    } else {
        {
            const int jv = sb_vec(j), x = 2 + (jv & 1);
            sb_bwd_pair<true>(SB_PAIR_ARGS, j, j + 1 < SB_NG ? j + 1 : j, slots + x * SLOT, seqs + x, (int)((TB >> (5 * jv)) & 31), true, j == 0, lane SB_PROF_PASS);
        }
#pragma unroll 1
        for (int s = 1; s < npairs; ++s) {
            const int i = j + s;
            const int iv = sb_vec(i), x = 2 + (iv & 1);
            sb_bwd_pair<false>(SB_PAIR_ARGS, i, i + 1 < SB_NG ? i + 1 : i, slots + x * SLOT, seqs + x, (int)((TB >> (5 * iv)) & 31) + (iv - j), false, j == 0,
                               lane SB_PROF_PASS);
        }
    }
#undef SB_PAIR_ARGS
    // the last pair's look-ahead loads re-read its own row group: let them land before their registers are anyone else's
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(R[0]), "+v"(R[1]), "+v"(R[2]), "+v"(R[3]), "+v"(R[4]), "+v"(R[5]), "+v"(R[6]), "+v"(R[7]));
    asm volatile("" : "+v"(RB[0]), "+v"(RB[1]), "+v"(RB[2]), "+v"(RB[3]), "+v"(RB[4]), "+v"(RB[5]), "+v"(RB[6]), "+v"(RB[7]));
    asm volatile("" : "+v"(C[0]), "+v"(C[1]), "+v"(C[2]), "+v"(C[3]), "+v"(C[4]), "+v"(C[5]), "+v"(C[6]), "+v"(C[7]));
    asm volatile("" : "+v"(C[8]), "+v"(C[9]), "+v"(C[10]), "+v"(C[11]), "+v"(C[12]), "+v"(C[13]), "+v"(C[14]), "+v"(C[15]));
    // Row groups i < j with a fully-masked query (q < fv: its softmax is uniform over ALL keys) reach this key group too:
    // P = exp2(-1e9 log2e - m - l) is 1 / Tk for such a query and exactly 0 for every other one; dS = 0 (masked), so only dV
    // gets something.  Rare (leading padding): plain loads, no pipeline.
    for (int i = 0; i < j && 32 * i < fv; ++i) {
        f32x16 pm;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float4 M4 = *reinterpret_cast<const float4*>(&Mt[32 * i + 8 * c + 4 * lh]);
            const float4 L4 = *reinterpret_cast<const float4*>(&Mt[SB_T + 32 * i + 8 * c + 4 * lh]);
            pm[4 * c] = __builtin_amdgcn_exp2f((AT_MASKED2 - M4.x) - L4.x); pm[4 * c + 1] = __builtin_amdgcn_exp2f((AT_MASKED2 - M4.y) - L4.y);
            pm[4 * c + 2] = __builtin_amdgcn_exp2f((AT_MASKED2 - M4.z) - L4.z); pm[4 * c + 3] = __builtin_amdgcn_exp2f((AT_MASKED2 - M4.w) - L4.w);
        }
        const float* gcol = p.dO + ((int64_t)b * SB_T + 32 * i + 4 * lh) * p.D + (int64_t)h * SB_DH + 2 * l31;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const float2 g2 = *reinterpret_cast<const float2*>(gcol + (int64_t)sb_row(e) * p.D);
            dv[0] = AT_MFMA(g2.x, pm[e], dv[0]);
            dv[1] = AT_MFMA(g2.y, pm[e], dv[1]);
        }
    }
    if constexpr (PH == 1) sb_bwd_store_kv(p, bh, j, E, dk, dv, lane);      // (phase 0's rows go out under phase 1's first loads)
    SB_T(10);
}

__global__ __launch_bounds__(512, 2) void attn_sb_bwd_kernel(const AttnBwdParams p) {
    // one LDS object: [4 dQ slots][32][68] | [8 waves]{K_j tile [32][68] (dK / dV staging at the phase's end), dS patch [32][36]} |
    // (max, log2 sum, Dsum) of 2 x 256 queries | slot counts
    __shared__ __attribute__((aligned(16))) float smem[SB_TABS + 2 * 3 * SB_T + 8];
    if (threadIdx.x < 8) reinterpret_cast<int*>(smem + SB_TABS + 2 * 3 * SB_T)[threadIdx.x] = 0;      // contribution counts of the dQ slots
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    f32x16 dk[2], dv[2];
    sb_u32x4 R[8], RB[8];
    sb_u32x2 C[16];
    SB_PROF_DECL;
    sb_bwd_phase<0>(p, smem, wave, lane, dk, dv, R, RB, C SB_PROF_PASS);
    sb_bwd_phase<1>(p, smem, wave, lane, dk, dv, R, RB, C SB_PROF_PASS);
    SB_PROF_STORE(wave);
}

}  // namespace nnhip

namespace nnhip {

// NNHIP_ATTN_SB=0 keeps every shape on the tiled kernels of attention.hip (developer A/B switch, read on every call)
static bool sb_enabled() {
    const char* e = getenv("NNHIP_ATTN_SB");
    return !(e && atoi(e) == 0);
}

bool attn_sb_applicable(int64_t B, int64_t H, int64_t Tq, int64_t Tk, int64_t head_dim, int causal, bool gen) {
    return sb_enabled() && !gen && causal && Tq == SB_T && Tk == SB_T && head_dim == SB_DH && B * H >= 1;
}

#ifdef SB_PROF
extern "C" int nnhipAttentionSbSetProfile(long long* buf) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_sb_prof), &buf, sizeof(buf));
}
#endif

int attn_sb_backward(const AttnBwdParams& p, hipStream_t st) {
    const int BH = p.B * p.H;
    hipLaunchKernelGGL(attn_sb_bwd_kernel, dim3((unsigned)((BH + 1) / 2)), dim3(512), 0, st, p);
    NNHIP_LAUNCH_CHECK("attn_sb_bwd_kernel");
    return 0;
}

int attn_sb_forward(const AttnParams& p, hipStream_t st) {
    const int BH = p.B * p.H;
    // NNHIP_ATTN_SB_FWD=pw: wave-private LDS tiles (round 6, measured 5-8 % slower than the streamed default; kept as a tested switch)
    static const bool pw = []() { const char* e = getenv("NNHIP_ATTN_SB_FWD"); return e && e[0] == 'p'; }();
    if (pw) {
        hipLaunchKernelGGL(attn_sb_fwd_pw_kernel, dim3((unsigned)((BH + 1) / 2)), dim3(512), 0, st, p);
        NNHIP_LAUNCH_CHECK("attn_sb_fwd_pw_kernel");
        return 0;
    }
    hipLaunchKernelGGL(attn_sb_fwd_kernel, dim3((unsigned)((BH + 1) / 2)), dim3(512), 0, st, p);
    NNHIP_LAUNCH_CHECK("attn_sb_fwd_kernel");
    return 0;
}

}  // namespace nnhip
