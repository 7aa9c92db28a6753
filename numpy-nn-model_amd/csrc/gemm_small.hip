// gemm_small.hip -- latency-optimised fp32 MFMA GEMM for SMALL problems (the README quick-start MLP: 32x784x128,
// 32x128x10 and their gradients; the conv classifier's 256x784x10 head).
//
// The 128x128-tile kernel of gemm.hip is built for throughput: a problem with a handful of output tiles runs as a few
// blocks that each walk their k-tiles at ~2 us apiece (64 MFMAs of which up to 3/4 multiply padding), plus a split-K
// reduce launch -- 8-15 us per GEMM, and the MNIST-MLP step is five of them.  Here:
//   * one block per 32x32 output tile, 4 or 8 waves per block that SPLIT K between them (wave w takes k-groups
//     w, w+NW, ...; a k-group = 8 consecutive k = 4 MFMAs 32x32x2);
//   * operands go global -> registers -> MFMA directly (no LDS staging, no barriers in the loop): lane (l31, lh)
//     holds A[m0 + l31][8g + 4lh .. +3] -- one float4 for a k-major operand, four coalesced dwords for an outer-major
//     one -- exactly the fragment layout of the big kernel, so the k permutation is the same;
//   * the waves' 32x32 accumulators meet in LDS (conflict-free: lane <-> column) and are summed in wave order
//     (deterministic), then the epilogue of gemm.hip: alpha, bias, addend, activation-gradient mask, activation, preact;
//   * asum (row sums of A = db of a Linear) falls out of the operand registers.
// Exact fp32 (v_mfma_f32_32x32x2_f32), same results as gemm.hip up to summation order.
#include "common.h"

namespace nnhip {

typedef float f32x16 __attribute__((ext_vector_type(16)));
enum { SG_ACT_NONE = 0, SG_ACT_SWISH = 1, SG_ACT_RELU = 2, SG_ACT_SIGMOID = 3 };

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

// one 32x32 output tile (bx, by) of problem p; `red` / `ared`: NW x 1024 and NW x 32 floats of LDS
template <int NW, bool AKM, bool BKM, bool VEC, int U>
__device__ __forceinline__ void sg_tile(const SmallGemmParams& p, int bx, int by, float (*red)[32 * 32], float (*ared)[32]) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int64_t m0 = (int64_t)by * 32, n0 = (int64_t)bx * 32;
    const unsigned groups = (unsigned)((p.K + 7) >> 3), K = (unsigned)p.K;
    // operand windows: everything from the operand's origin to the end of its last row / k-line (gemm_small() checked that this
    // fits 31 bits); the per-lane row offset is loop-invariant
    const unsigned la4 = (unsigned)p.lda * 4u, lb4 = (unsigned)p.ldb * 4u;
    const __amdgpu_buffer_rsrc_t rsa = sg_rsrc(p.A, (unsigned)((AKM ? (p.M - 1) * p.lda + p.K : (p.K - 1) * p.lda + p.M) * 4));
    const __amdgpu_buffer_rsrc_t rsb = sg_rsrc(p.B, (unsigned)((BKM ? (p.N - 1) * p.ldb + p.K : (p.K - 1) * p.ldb + p.N) * 4));
    const bool a_ok = m0 + l31 < p.M, b_ok = n0 + l31 < p.N;
    const unsigned a_row = (unsigned)(m0 + l31) * (AKM ? la4 : 4u), b_row = (unsigned)(n0 + l31) * (BKM ? lb4 : 4u);
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    float asum = 0.f;
    // U k-groups in flight per wave (2 x U float4 of operands).  U = 8: a K = 784 over 8 waves is two round trips to L2;
    // the U = 16 instantiation (k-major operands, 64 < K/8 <= 128 groups per block) makes it one.
    for (unsigned gb = wave; gb < groups; gb += (unsigned)NW * U) {
        float4 a[U], b[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const unsigned k0 = 8u * (gb + (unsigned)u * NW) + 4u * lh;      // a k-group past the last has k0 >= K: reads 0
            a[u] = sg_fetch<AKM, VEC>(rsa, la4, a_row, a_ok, k0, K);
            b[u] = sg_fetch<BKM, VEC>(rsb, lb4, b_row, b_ok, k0, K);
        }
        // all 2U loads are in flight before the first MFMA waits for its operands (left alone, the scheduler sinks each load
        // to just above its use -- `load, s_waitcnt vmcnt(0), mfma` U times: U memory round trips instead of one)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].x, b[u].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].y, b[u].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].z, b[u].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].w, b[u].w, acc, 0, 0, 0);
            asum += (a[u].x + a[u].y) + (a[u].z + a[u].w);
        }
    }
    // accumulator register e holds row (e&3) + 8(e>>2) + 4lh, column l31
#pragma unroll
    for (int e = 0; e < 16; ++e) red[wave][((e & 3) + 8 * (e >> 2) + 4 * lh) * 32 + l31] = acc[e];
    if (p.asum) {
        asum += __shfl_xor(asum, 32, 64);
        if (lh == 0) ared[wave][l31] = asum;
    }
    __syncthreads();
    constexpr int PER = 1024 / (NW * 64);                      // outputs per thread
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int o = tid + i * NW * 64;
        float v = red[0][o];
#pragma unroll
        for (int w = 1; w < NW; ++w) v += red[w][o];
        const int64_t row = m0 + (o >> 5), col = n0 + (o & 31);
        if (row < p.M && col < p.N) {
            v = p.alpha * v + (p.bias ? p.bias[col] : 0.f);
            if (p.addend) v += p.addend[row * p.ldc + col];
            if (p.dact_arg) {
                const float x = p.dact_arg[row * p.ldc + col];
                v = p.dact == 2 ? (x > 0.f ? v : 0.f) : v * swish_grad_(x, p.beta);
            }
            if (p.act == SG_ACT_SWISH) {
                if (p.preact) p.preact[row * p.ldc + col] = v;
                v = v * sigmoid_fast_(p.beta * v);
            } else if (p.act == SG_ACT_RELU) {
                v = fmaxf(v, 0.f);
            } else if (p.act == SG_ACT_SIGMOID) {
                v = sigmoid_fast_(v);
            }
            p.C[row * p.ldc + col] = v;
        }
    }
    if (p.asum && bx == 0 && tid < 32) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) s += ared[w][tid];
        if (m0 + tid < p.M) p.asum[m0 + tid] = s;
    }
}

template <int NW, bool AKM, bool BKM, bool VEC, int U = 8>
__global__ __launch_bounds__(NW * 64) void gemm_small_kernel(const SmallGemmParams p) {
    __shared__ float red[NW][32 * 32];
    __shared__ float ared[NW][32];
    sg_tile<NW, AKM, BKM, VEC, U>(p, (int)blockIdx.x, (int)blockIdx.y, red, ared);
}

// Two independent small problems in ONE launch (blockIdx.z picks): the input gradient dX = dO W (A k-major, B outer-major,
// any epilogue) and the weight gradient dW = dO^T X (+ db; both outer-major) of a small Linear.  At MNIST-MLP scale a launch is
// ~4.7 us of a 45 us step whatever it computes, so two tiles' worth of work should not cost two launches.  The grid is the
// larger of the two tile grids; a block outside its problem's grid exits.
struct SmallLinearBwd {                                    // compact kernel argument of the pair kernel (84 bytes)
    const float* dO; const float* W; const float* X;
    float* dX; float* dW; float* db;
    const float* addend; const float* dact_arg;
    int rows, in, out, dact;
    float beta;
};

template <int NW, bool VEC0>
__global__ __launch_bounds__(NW * 64) void gemm_small_pair_kernel(const SmallLinearBwd q) {
    __shared__ float red[NW][32 * 32];
    __shared__ float ared[NW][32];
    const int bx = (int)blockIdx.x, by = (int)blockIdx.y;
    SmallGemmParams p{};
    p.A = q.dO; p.lda = q.out; p.ldb = q.in; p.ldc = q.in; p.N = q.in; p.alpha = 1.f;
    if (blockIdx.z == 0) {                                 // dX[rows, in] = dO[rows, out] W[out, in]   (+ addend, (.) act')
        p.B = q.W; p.C = q.dX; p.addend = q.addend; p.dact_arg = q.dact_arg; p.dact = q.dact; p.beta = q.beta;
        p.M = q.rows; p.K = q.out; p.a_kmajor = 1;
        if ((int64_t)bx * 32 < p.N && (int64_t)by * 32 < p.M) sg_tile<NW, true, false, VEC0, 8>(p, bx, by, red, ared);
    } else {                                               // dW[out, in] = dO^T[out, rows] X[rows, in],  db = row sums of dO^T
        p.B = q.X; p.C = q.dW; p.asum = q.db; p.beta = 1.f;
        p.M = q.out; p.K = q.rows;
        if ((int64_t)bx * 32 < p.N && (int64_t)by * 32 < p.M) sg_tile<NW, false, false, false, 8>(p, bx, by, red, ared);
    }
}

// Is this problem one for the small kernel?  (gemm.hip asks before planning its own launch.)
bool gemm_small_wanted(int64_t M, int64_t N, int64_t K, int64_t batch, int64_t lda, int64_t ldb, bool a_kmajor, bool b_kmajor) {
    if (batch != 1 || M <= 0 || N <= 0 || K <= 0 || K > 2048) return false;
    // the operand windows are addressed with 32-bit byte offsets below an out-of-range sentinel
    const int64_t ea = a_kmajor ? (M - 1) * lda + K : (K - 1) * lda + M, eb = b_kmajor ? (N - 1) * ldb + K : (K - 1) * ldb + N;
    if (ea * 4 >= ((int64_t)1 << 31) || eb * 4 >= ((int64_t)1 << 31)) return false;
    const int64_t tiles128 = ceil_div(M, 128) * ceil_div(N, 128);
    const int64_t tiles32 = ceil_div(M, 32) * ceil_div(N, 32);
    // a very short reduction (the conv classifier's 256x10 -> 784 input gradient) is all launch + epilogue: 32x32 blocks
    // cover it faster than the 128x128 scalar-load kernel even when there are a few hundred of them
    return (tiles128 <= 8 && tiles32 <= 256) || (K <= 64 && tiles32 <= 512);
}

int gemm_small(const float* A, const float* B, float* C, const float* bias, float* preact, int64_t M, int64_t N, int64_t K,
               int64_t lda, int64_t ldb, int64_t ldc, bool a_kmajor, bool b_kmajor, float alpha, int act, float beta,
               float* asum, const float* addend, const float* dact_arg, int dact, hipStream_t st) {
    SmallGemmParams p;
    p.A = A; p.B = B; p.C = C; p.bias = bias; p.preact = preact; p.addend = addend; p.dact_arg = dact_arg; p.asum = asum;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.alpha = alpha; p.beta = beta; p.act = act; p.dact = dact;
    p.a_kmajor = a_kmajor; p.b_kmajor = b_kmajor;
    // float4 loads for the k-major operands: 16-B aligned rows and K % 4 == 0 (a float4 then never straddles the end of a row)
    const bool vec = (a_kmajor || b_kmajor) && (K & 3) == 0 && (!a_kmajor || (aligned16(A) && (lda & 3) == 0)) &&
                     (!b_kmajor || (aligned16(B) && (ldb & 3) == 0));
    const int64_t groups = (K + 7) >> 3;
    const int nw = groups >= 8 ? 8 : 4;
    dim3 grid((unsigned)ceil_div(N, 32), (unsigned)ceil_div(M, 32));
#define SG_LAUNCH2(NW, V)                                                                                                     \
    do {                                                                                                                      \
        if (a_kmajor && b_kmajor) hipLaunchKernelGGL((gemm_small_kernel<NW, true, true, V>), grid, dim3(NW * 64), 0, st, p);  \
        else if (a_kmajor) hipLaunchKernelGGL((gemm_small_kernel<NW, true, false, V>), grid, dim3(NW * 64), 0, st, p);        \
        else if (b_kmajor) hipLaunchKernelGGL((gemm_small_kernel<NW, false, true, V>), grid, dim3(NW * 64), 0, st, p);        \
        else hipLaunchKernelGGL((gemm_small_kernel<NW, false, false, false>), grid, dim3(NW * 64), 0, st, p);                 \
    } while (0)
#define SG_LAUNCH(NW)                                                                                                         \
    do {                                                                                                                      \
        if (vec) SG_LAUNCH2(NW, true);                                                                                        \
        else SG_LAUNCH2(NW, false);                                                                                           \
    } while (0)
    if (nw == 8 && vec && a_kmajor && b_kmajor && groups > 64 && groups <= 128)
        hipLaunchKernelGGL((gemm_small_kernel<8, true, true, true, 16>), grid, dim3(512), 0, st, p);
    else if (nw == 8) SG_LAUNCH(8);
    else SG_LAUNCH(4);
#undef SG_LAUNCH
#undef SG_LAUNCH2
    NNHIP_LAUNCH_CHECK("gemm_small_kernel");
    return 0;
}

// dX = (dO W) [(.) act'] and dW = dO^T X (+ db) of one Linear in one launch; the caller checked gemm_small_wanted() for both.
int gemm_small_linear_backward(const float* X, const float* W, const float* dO, float* dX, float* dW, float* db, int64_t rows,
                               int64_t in, int64_t out, const float* addend, const float* dact_arg, int dact, float beta,
                               hipStream_t st) {
    SmallLinearBwd q;
    q.dO = dO; q.W = W; q.X = X; q.dX = dX; q.dW = dW; q.db = db; q.addend = addend; q.dact_arg = dact_arg;
    q.rows = (int)rows; q.in = (int)in; q.out = (int)out; q.dact = dact; q.beta = beta;
    const bool vec0 = (out & 3) == 0 && aligned16(dO);
    const int64_t g0 = (out + 7) >> 3, g1 = (rows + 7) >> 3;
    const int nw = (g0 >= 8 || g1 >= 8) ? 8 : 4;
    dim3 grid((unsigned)ceil_div(in, 32), (unsigned)max(ceil_div(rows, 32), ceil_div(out, 32)), 2);
    if (nw == 8) {
        if (vec0) hipLaunchKernelGGL((gemm_small_pair_kernel<8, true>), grid, dim3(512), 0, st, q);
        else hipLaunchKernelGGL((gemm_small_pair_kernel<8, false>), grid, dim3(512), 0, st, q);
    } else {
        if (vec0) hipLaunchKernelGGL((gemm_small_pair_kernel<4, true>), grid, dim3(256), 0, st, q);
        else hipLaunchKernelGGL((gemm_small_pair_kernel<4, false>), grid, dim3(256), 0, st, q);
    }
    NNHIP_LAUNCH_CHECK("gemm_small_pair_kernel");
    return 0;
}

}  // namespace nnhip
