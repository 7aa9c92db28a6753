// Developer probe (not part of the library): cost of DEPENDENT v_mfma_f32_32x32x2_f32 chains on gfx950.
//   chains = 1: every MFMA accumulates into the same registers; 2 / 4: that many accumulators round-robin
//   waves per SIMD 1 or 2 (blocks of 256 / 512 threads, one block per CU)
// Build: hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_chain_probe.hip -o tools/probes/mfma_chain_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int CH, int NOPS>
__global__ __launch_bounds__(512) void probe(float* out, int iters, long long* cyc) {
    f32x16 a[4];
    for (int c = 0; c < 4; ++c) for (int e = 0; e < 16; ++e) a[c][e] = 0.f;
    float x = threadIdx.x * 1e-3f, y = 1.0f;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 32; ++u) {
            a[u % CH] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a[u % CH], 0, 0, 0);
            if (NOPS == 1 && (u & 3) == 3) asm volatile("s_waitcnt vmcnt(23)");
            if (NOPS == 2 && (u & 3) == 3) { asm volatile("s_waitcnt vmcnt(23)"); asm volatile("s_nop 0"); }
        }
    }
    const long long t1 = clock64();
    float s = 0.f;
    for (int c = 0; c < 4; ++c) for (int e = 0; e < 16; ++e) s += a[c][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int CH, int NOPS>
void run(const char* name, int threads) {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8);
    const int iters = 200;
    probe<CH, NOPS><<<256, threads>>>(out, iters, cyc);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    probe<CH, NOPS><<<256, threads>>>(out, iters, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[8]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-48s %d waves/SIMD  %.3f ms  cycles per MFMA (wave 0): %.1f   per SIMD-MFMA: %.1f\n", name, threads / 256, ms,
           (double)h[0] / (iters * 32.0), (double)h[0] / (iters * 32.0) / (threads / 256));
    hipFree(out); hipFree(cyc);
}
int main() {
    run<1, 0>("1 chain (fully dependent)", 256);
    run<2, 0>("2 chains alternating", 256);
    run<4, 0>("4 chains", 256);
    run<1, 0>("1 chain (fully dependent)", 512);
    run<2, 0>("2 chains alternating", 512);
    run<4, 0>("4 chains", 512);
    run<1, 1>("1 chain + s_waitcnt every 4", 512);
    run<2, 1>("2 chains + s_waitcnt every 4", 512);
    run<2, 1>("2 chains + s_waitcnt every 4", 256);
    run<1, 2>("1 chain + s_waitcnt + s_nop every 4", 512);
    return 0;
}
