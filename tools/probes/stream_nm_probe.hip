// stream_nm_probe.hip -- what a plain float4 stream kernel with R read streams and W write streams reaches on this part, at the
// footprints of the HBM-bound kernels whose fractions of the 8 TB/s spec the bench line reports: the practical roof to read
// those fractions against (round-5 review: "AdamW: either >= 0.70 or a stream_roof row at the same footprint").
//   hipcc --offload-arch=gfx950 -O3 tools/probes/stream_nm_probe.hip -o /tmp/stream_nm_probe && /tmp/stream_nm_probe
// No arithmetic to speak of (a sum of the inputs, scaled per output), every stream its own buffer, grid-stride float4 loop with
// two float4 per stream in flight per lane, 2048 blocks of 256 threads (8 per CU).  Each case is launched 3 + 20 times back to
// back; the mean of the 20 is printed (the same methodology as bench.py's C3 pass: cold between passes only if the footprint
// exceeds the 256 MiB Infinity Cache).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ntload(const float4* p) {
    const f4 v = __builtin_nontemporal_load(reinterpret_cast<const f4*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}

struct Ptrs { const float4* in[4]; float4* out[3]; };

template <int R, int W, bool NT>
__global__ __launch_bounds__(256) void stream_kernel(Ptrs p, long n4) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += 2 * stride) {
        const long j = i + stride;
        float4 a[R], b[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            a[r] = NT ? ntload(&p.in[r][i]) : p.in[r][i];
            if (j < n4) b[r] = NT ? ntload(&p.in[r][j]) : p.in[r][j];
        }
        float4 s = a[0], t = b[0];
#pragma unroll
        for (int r = 1; r < R; ++r) { s.x += a[r].x; s.y += a[r].y; s.z += a[r].z; s.w += a[r].w; t.x += b[r].x; t.y += b[r].y; t.z += b[r].z; t.w += b[r].w; }
#pragma unroll
        for (int w = 0; w < W; ++w) {
            const float c = 1.0f + 0.5f * w;
            p.out[w][i] = make_float4(s.x * c, s.y * c, s.z * c, s.w * c);
            if (j < n4) p.out[w][j] = make_float4(t.x * c, t.y * c, t.z * c, t.w * c);
        }
    }
}

// `cold`: rotate over enough independent buffer sets that a launch never finds its lines in the 256 MiB Infinity Cache (the state
// bench.py's C3 pass measures its kernels in: every op streams 270-940 MB between two launches of itself); otherwise one set,
// launched back to back (the in-step state of the small C4 tensors, which DO sit in the cache when their consumer runs).
template <int R, int W, bool NT>
static void run(const char* what, long elems, bool cold = false) {
    const long n4 = elems / 4;
    const double set_bytes = (double)(R + W) * elems * 4;
    const int sets = cold ? (int)((1.5e9 + set_bytes - 1) / set_bytes) + 1 : 1;
    Ptrs* ps = new Ptrs[sets];
    float** bufs = new float*[sets * 7];
    for (int s = 0; s < sets; ++s) {
        for (int k = 0; k < R + W; ++k) { CK(hipMalloc(&bufs[s * 7 + k], elems * 4)); CK(hipMemset(bufs[s * 7 + k], 0, elems * 4)); }
        for (int r = 0; r < R; ++r) ps[s].in[r] = (const float4*)bufs[s * 7 + r];
        for (int w = 0; w < W; ++w) ps[s].out[w] = (float4*)bufs[s * 7 + R + w];
    }
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int iters = 24;
    for (int it = 0; it < 4; ++it) hipLaunchKernelGGL((stream_kernel<R, W, NT>), dim3(2048), dim3(256), 0, 0, ps[it % sets], n4);
    CK(hipEventRecord(a));
    for (int it = 0; it < iters; ++it) hipLaunchKernelGGL((stream_kernel<R, W, NT>), dim3(2048), dim3(256), 0, 0, ps[it % sets], n4);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, a, b));
    const double us = ms * 1e3 / iters, bytes = set_bytes;
    printf("%-72s %dR+%dW %s %s %8.1f MB  %8.1f us  %7.1f GB/s  %.3f of 8 TB/s\n", what, R, W, NT ? "nt" : "  ", cold ? "cold" : "warm", bytes / 1e6, us, bytes / us / 1e3,
           bytes / us / 1e3 / 8000.0);
    for (int k = 0; k < sets * 7; ++k) if (k % 7 < R + W) CK(hipFree(bufs[k]));
    delete[] ps; delete[] bufs;
}

int main() {
    run<1, 1, false>("(warm-up)", 64l << 20);
    // ---- C3 (bench.py --workload c3: every op is cold when its turn comes) ----
    run<4, 3, false>("AdamW footprint, C3: one 8192 x 4096 tensor (p g m v -> p m v)", 8192l * 4096, true);
    run<4, 3, true>("AdamW footprint, C3, non-temporal loads", 8192l * 4096, true);
    run<1, 1, false>("CE footprint, C3: 8192 x 4096 logits -> dlogits", 8192l * 4096, true);
    run<1, 1, true>("CE footprint, C3, non-temporal loads", 8192l * 4096, true);
    run<2, 1, false>("RMSNorm / Swish / Softmax backward footprint, C3: 8192 x 4096 (dY X -> dX)", 8192l * 4096, true);
    run<1, 1, false>("RMSNorm / Swish / Softmax forward footprint, C3: 8192 x 4096 (X -> Y)", 8192l * 4096, true);
    // ---- C4 (inside the step: the big tensors stream from HBM, the 33.5 MB activations come out of the Infinity Cache) ----
    run<4, 3, false>("AdamW footprint, C4: 34.3 M parameters", 34283264l, true);
    run<1, 1, false>("CE footprint, C4: 16384 x 15000 logits -> dlogits", 16384l * 15000, true);
    run<1, 1, true>("CE footprint, C4, non-temporal loads", 16384l * 15000, true);
    run<3, 1, false>("RMSNorm backward + residual addend, C4: 16384 x 512 (dY X G -> dX)", 16384l * 512);
    run<3, 1, false>("RMSNorm backward + residual addend, C4, cold", 16384l * 512, true);
    run<1, 1, false>("RMSNorm forward, C4: 16384 x 512 (X -> Y)", 16384l * 512);
    run<1, 1, false>("RMSNorm forward, C4, cold", 16384l * 512, true);
    return 0;
}
