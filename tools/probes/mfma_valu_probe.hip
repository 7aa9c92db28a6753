// Developer probe (not part of the library): how do MFMA and VALU instructions share a gfx950 SIMD?
//   mode 0: every wave issues only MFMAs (two accumulators alternating)                -> MFMA-only time
//   mode 1: every wave issues only VALU fma chains                                      -> VALU-only time
//   mode 2: even waves of a SIMD issue MFMAs, odd waves VALU (two waves per SIMD)       -> cross-wave overlap?
//   mode 3: every wave interleaves K VALU ops after each MFMA (same wave, independent)  -> same-wave overlap?
// Build: hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_valu_probe.hip -o tools/probes/mfma_valu_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE, int K>
__global__ __launch_bounds__(512) void probe(float* out, int iters, long long* cyc) {
    const int wave = threadIdx.x >> 6;
    f32x16 a0, a1;
    for (int e = 0; e < 16; ++e) { a0[e] = 0.f; a1[e] = 0.f; }
    float x = threadIdx.x * 1e-3f, y = 1.0f;
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = x + i;
    const bool do_mfma = MODE == 0 || MODE == 3 || MODE == 4 || (MODE == 2 && (wave >> 2) == 0);   // waves 0-3 -> SIMDs 0-3, waves 4-7 -> SIMDs 0-3 again
    const bool do_valu = MODE == 1 || (MODE == 2 && (wave >> 2) == 1);
    const long long t0 = clock64();
    if (MODE == 3) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
#pragma unroll
                for (int k = 0; k < K; ++k) v[k & 7] = __builtin_fmaf(v[k & 7], 1.0001f, 0.5f);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, K, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
#pragma unroll
                for (int k = 0; k < K; ++k) v[(k + 4) & 7] = __builtin_fmaf(v[(k + 4) & 7], 1.0001f, 0.5f);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, K, 0);
            }
        }
    } else if (MODE == 4) {       // bf16 (XDL) MFMA 32x32x16 + K independent VALU per MFMA, same wave
        bf16x8 ba, bb;
        for (int i = 0; i < 8; ++i) { ba[i] = (__bf16)(x + i); bb[i] = (__bf16)(y - i); }
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ba, bb, a0, 0, 0, 0);
#pragma unroll
                for (int k = 0; k < K; ++k) v[k & 7] = __builtin_fmaf(v[k & 7], 1.0001f, 0.5f);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, K, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bb, ba, a1, 0, 0, 0);
#pragma unroll
                for (int k = 0; k < K; ++k) v[(k + 4) & 7] = __builtin_fmaf(v[(k + 4) & 7], 1.0001f, 0.5f);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, K, 0);
            }
        }
    } else if (do_mfma) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
            }
        }
    } else if (do_valu) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 16; ++u)
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = __builtin_fmaf(v[k], 1.0001f, 0.5f);
        }
    }
    const long long t1 = clock64();
    float s = 0.f;
    for (int e = 0; e < 16; ++e) s += a0[e] + a1[e];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int MODE, int K>
void run(const char* name, int iters) {
    float* out; long long* cyc;
    const int blocks = 256;
    hipMalloc(&out, blocks * 512 * 4);
    hipMalloc(&cyc, blocks * 8 * 8);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((probe<MODE, K>), dim3(blocks), dim3(512), 0, 0, out, 10, cyc);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL((probe<MODE, K>), dim3(blocks), dim3(512), 0, 0, out, iters, cyc);
    hipEventRecord(b); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, a, b);
    long long h[8]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-44s %8.3f ms   cycles/iter: wave0 (mfma side) %8.1f   wave4 (valu side) %8.1f\n", name, ms, (double)h[0] / iters, (double)h[4] / iters);
    hipFree(out); hipFree(cyc);
}

int main() {
    const int iters = 2000;
    // per iteration: MFMA waves issue 16 MFMAs (16 x 64 = 1024 pipe cycles); VALU waves issue 128 fma (128 x 4 = 512 cycles... per wave)
    run<0, 0>("0: all 8 waves MFMA only (16/iter)", iters);
    run<1, 0>("1: all 8 waves VALU only (128 fma/iter)", iters);
    run<2, 0>("2: waves 0-3 MFMA, waves 4-7 VALU", iters);
    run<3, 0>("3: MFMA only via mode-3 path, K=0", iters);
    run<3, 4>("3: each MFMA + 4 VALU (same wave)", iters);
    run<3, 8>("3: each MFMA + 8 VALU (same wave)", iters);
    run<3, 12>("3: each MFMA + 12 VALU (same wave)", iters);
    run<3, 14>("3: each MFMA + 14 VALU (same wave)", iters);
    // bf16 32x32x16 MFMA: 8 passes = 32 cycles each; 16 per iteration per wave, 2 waves per SIMD -> 1024 cycles / iter
    run<4, 0>("4: bf16 MFMA 32x32x16 only (16/iter)", iters);
    run<4, 2>("4: each bf16 MFMA + 2 VALU (same wave)", iters);
    run<4, 4>("4: each bf16 MFMA + 4 VALU (same wave)", iters);
    run<4, 6>("4: each bf16 MFMA + 6 VALU (same wave)", iters);
    run<4, 8>("4: each bf16 MFMA + 8 VALU (same wave)", iters);
    return 0;
}
