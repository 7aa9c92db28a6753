// Developer probe: how fast can 512 workgroups write their 128x128 fp32 C tiles at once (the GEMM epilogue's store pattern:
// per wave 16 x global_store_dwordx4, each covering 4 rows x 256 B of a row-major [M, N] matrix), against a contiguous
// stream of the same bytes?   hipcc --offload-arch=gfx950 -O3 tools/probes/tile_store_probe.hip -o tools/probes/tile_store_probe
#include <hip/hip_runtime.h>
#include <stdio.h>

__global__ __launch_bounds__(256) void tile_store(float* C, int N, int tiles_n, int mode) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    int b = blockIdx.x;
    const int tm = b / tiles_n, tn = b % tiles_n;
    const float4 v = make_float4(tid, b, 1.f, 2.f);
    if (mode == 0) {                       // GEMM pattern
        const int er = lane >> 4, ec = (lane & 15) * 4;
        for (int i = 0; i < 2; ++i)
            for (int it = 0; it < 8; ++it) {
                const long row = (long)tm * 128 + wm * 64 + i * 32 + it * 4 + er, col = (long)tn * 128 + wn * 64 + ec;
                *reinterpret_cast<float4*>(C + row * N + col) = v;
            }
    } else {                               // contiguous: the block's 64 KB as one run
        float4* p = reinterpret_cast<float4*>(C) + (long)b * 4096;
        for (int k = 0; k < 16; ++k) p[k * 256 + tid] = v;
    }
}

int main() {
    const int M = 16384;
    float* C;
    hipMalloc(&C, (size_t)M * 15104 * 4);
    hipEvent_t a, e;
    hipEventCreate(&a); hipEventCreate(&e);
    for (int N : {512, 2048, 15104})
        for (int mode = 0; mode < 2; ++mode) {
            const int tiles_n = N / 128, blocks = (M / 128) * tiles_n;
            for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(tile_store, dim3(blocks), dim3(256), 0, 0, C, N, tiles_n, mode);
            hipDeviceSynchronize();
            hipEventRecord(a);
            for (int r = 0; r < 20; ++r) hipLaunchKernelGGL(tile_store, dim3(blocks), dim3(256), 0, 0, C, N, tiles_n, mode);
            hipEventRecord(e);
            hipDeviceSynchronize();
            float ms;
            hipEventElapsedTime(&ms, a, e);
            const double bytes = (double)M * N * 4;
            printf("N=%5d %-22s %8.1f us per launch  %7.1f GB/s  (%d blocks, %.1f MB)\n", N, mode ? "contiguous 64 KB/block" : "GEMM tile pattern",
                   ms / 20 * 1e3, bytes / (ms / 20 * 1e-3) / 1e9, blocks, bytes / 1e6);
        }
    return 0;
}
