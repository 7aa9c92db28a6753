// Where does a 32x784->128 small-GEMM launch spend its time?  The two tile layouts of gemm_small side by side, s_memtime stamps
// around the tile body: 32x32 tiles (lane <-> row) 17 k cycles / 8.4 us per launch, 16x16 tiles (four lanes per row) 7.4 k / 4.5 us.
//   hipcc --offload-arch=gfx950 -O3 -I numpy-nn-model_amd/csrc tools/probes/small_gemm_phases.hip -o tools/probes/small_gemm_phases
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include "gemm_small.h"
namespace nnhip {
// the 32x32x2-MFMA tile body gemm_small.h had first (lane <-> row, 16 B per lane: 64 cache lines per wave-load)
// one 32x32 output tile (bx, by) of problem p; `red` / `ared`: NW x 1024 and NW x 32 floats of LDS
template <int NW, bool AKM, bool BKM, bool VEC, int U>
__device__ __forceinline__ void sg_tile(const SmallGemmParams& p, int bx, int by, float (*red)[32 * 32], float (*ared)[32],
                                        float* lds_copy = nullptr) {     // lds_copy: also keep C[row][0..32) at lds_copy[row * 32 + col]
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
            if (lds_copy) lds_copy[row * 32 + col] = v;
        }
    }
    if (p.asum && bx == 0 && tid < 32) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) s += ared[w][tid];
        if (m0 + tid < p.M) p.asum[m0 + tid] = s;
    }
}

}  // namespace nnhip
using namespace nnhip;

__global__ __launch_bounds__(512) void probe_kernel(const SmallGemmParams p, long long* stamps) {
    __shared__ float red[8][32 * 32];
    __shared__ float ared[8][32];
    const long long t0 = __builtin_amdgcn_s_memtime();
    sg_tile<8, true, true, true, 16>(p, (int)blockIdx.x, (int)blockIdx.y, red, ared);
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t0; stamps[2 * blockIdx.x + 1] = t1; }
}

__global__ __launch_bounds__(512) void probe16_kernel(const SmallGemmParams p, long long* stamps) {
    __shared__ float red[8][16 * 16];
    __shared__ float ared[8][16];
    const long long t0 = __builtin_amdgcn_s_memtime();
    sg_tile16<8, true, true, true, 8>(p, (int)blockIdx.x, (int)blockIdx.y, red, ared);
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && blockIdx.y == 0) { stamps[2 * (blockIdx.x & 3)] = t0; stamps[2 * (blockIdx.x & 3) + 1] = t1; }
}

int main() {
    const int M = 32, K = 784, N = 128;
    float *X, *W, *b, *O; long long* st;
    hipMalloc(&X, M * K * 4); hipMalloc(&W, N * K * 4); hipMalloc(&b, N * 4); hipMalloc(&O, M * N * 4); hipMalloc(&st, 64 * 8);
    hipMemset(X, 0, M * K * 4); hipMemset(W, 0, N * K * 4); hipMemset(b, 0, N * 4);
    SmallGemmParams p{};
    p.A = X; p.B = W; p.C = O; p.bias = b; p.M = M; p.N = N; p.K = K; p.lda = K; p.ldb = K; p.ldc = N; p.alpha = 1.f; p.beta = 1.f;
    p.act = SG_ACT_RELU; p.a_kmajor = 1; p.b_kmajor = 1;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 5; ++it) {
        hipEventRecord(e0);
        for (int r = 0; r < 100; ++r) hipLaunchKernelGGL(probe_kernel, dim3(N / 32, M / 32), dim3(512), 0, 0, p, st);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long h[8]; hipMemcpy(h, st, 64, hipMemcpyDeviceToHost);
        printf("100 launches back to back: %.2f us each; in-kernel (s_memtime, 100 MHz ticks) block 0: %lld ticks = %.2f us, block 3: %.2f us\n",
               ms * 10.f, h[1] - h[0], (h[1] - h[0]) * 0.01, (h[7] - h[6]) * 0.01);
    }
    for (int it = 0; it < 5; ++it) {
        hipEventRecord(e0);
        for (int r = 0; r < 100; ++r) hipLaunchKernelGGL(probe16_kernel, dim3(N / 16, M / 16), dim3(512), 0, 0, p, st);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long h[8]; hipMemcpy(h, st, 64, hipMemcpyDeviceToHost);
        printf("16x16 tiles, 100 launches back to back: %.2f us each; in-kernel block 0: %lld ticks\n", ms * 10.f, h[1] - h[0]);
    }
    return 0;
}
