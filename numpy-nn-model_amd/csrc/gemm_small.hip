// gemm_small.hip -- latency-optimised fp32 MFMA GEMM for SMALL problems (the README quick-start MLP: 32x784x128,
// 32x128x10 and their gradients; the conv classifier's 256x784x10 head).
//
// The 128x128-tile kernel of gemm.hip is built for throughput: a problem with a handful of output tiles runs as a few
// blocks that each walk their k-tiles at ~2 us apiece (64 MFMAs of which up to 3/4 multiply padding), plus a split-K
// reduce launch -- 8-15 us per GEMM, and the MNIST-MLP step is five of them.  Here (tile body: gemm_small.h):
//   * one block per 16x16 output tile, 4 or 8 waves per block that SPLIT K between them (wave w takes k-groups
//     w, w+NW, ...; a k-group = 16 consecutive k = 4 MFMAs 16x16x4);
//   * operands go global -> registers -> MFMA directly (no LDS staging, no barriers in the loop): lane (l16, kq) holds
//     A[m0 + l16][16g + 4kq .. +3] -- one float4 for a k-major operand, four coalesced dwords for an outer-major one;
//   * the waves' 16x16 accumulators meet in LDS and are summed in wave order (deterministic), then the epilogue of
//     gemm.hip: alpha, bias, addend, activation-gradient mask, activation, preact;
//   * asum (row sums of A = db of a Linear) falls out of the operand registers.
// Exact fp32 (v_mfma_f32_16x16x4_f32), same results as gemm.hip up to summation order.
#include "gemm_small.h"

namespace nnhip {

template <int NW, bool AKM, bool BKM, bool VEC, int U = 8>
__global__ __launch_bounds__(NW * 64) void gemm_small_kernel(const SmallGemmParams p) {
    __shared__ float red[NW][16 * 16];
    __shared__ float ared[NW][16];
    sg_tile16<NW, AKM, BKM, VEC, U>(p, (int)blockIdx.x, (int)blockIdx.y, red, ared);
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
    __shared__ float red[NW][16 * 16];
    __shared__ float ared[NW][16];
    const int bx = (int)blockIdx.x, by = (int)blockIdx.y;
    SmallGemmParams p{};
    p.A = q.dO; p.lda = q.out; p.ldb = q.in; p.ldc = q.in; p.N = q.in; p.alpha = 1.f;
    if (blockIdx.z == 0) {                                 // dX[rows, in] = dO[rows, out] W[out, in]   (+ addend, (.) act')
        p.B = q.W; p.C = q.dX; p.addend = q.addend; p.dact_arg = q.dact_arg; p.dact = q.dact; p.beta = q.beta;
        p.M = q.rows; p.K = q.out; p.a_kmajor = 1;
        if ((int64_t)bx * 16 < p.N && (int64_t)by * 16 < p.M) sg_tile16<NW, true, false, VEC0, 8>(p, bx, by, red, ared);
    } else {                                               // dW[out, in] = dO^T[out, rows] X[rows, in],  db = row sums of dO^T
        p.B = q.X; p.C = q.dW; p.asum = q.db; p.beta = 1.f;
        p.M = q.out; p.K = q.rows;
        if ((int64_t)bx * 16 < p.N && (int64_t)by * 16 < p.M) sg_tile16<NW, false, false, false, 8>(p, bx, by, red, ared);
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
    const int64_t groups = (K + 15) >> 4;                   // k-groups of 16
    const int nw = groups >= 8 ? 8 : 4;
    dim3 grid((unsigned)ceil_div(N, 16), (unsigned)ceil_div(M, 16));
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
    const int64_t g0 = (out + 15) >> 4, g1 = (rows + 15) >> 4;
    const int nw = (g0 >= 8 || g1 >= 8) ? 8 : 4;
    dim3 grid((unsigned)ceil_div(in, 16), (unsigned)max(ceil_div(rows, 16), ceil_div(out, 16)), 2);
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
