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
#include "adam_device.h"
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

bool gemm_small_wanted(int64_t M, int64_t N, int64_t K, int64_t batch, int64_t lda, int64_t ldb, bool a_kmajor, bool b_kmajor);
// ---- Linear2(relu(Linear1(x))) backward when x needs no gradient: all four parameter gradients from ONE launch ------------
// dW2 = dO^T H, db2 = sum_rows dO  (the first blocks of the 1-D grid: the plain tile body)
// dW1 = dZ^T X1, db1 = sum_rows dZ with dZ = (dO W2) (.) [H > 0] formed ON THE FLY (the other blocks): lane (l16, kq) of a dW1
// tile needs dZ[:, m0 .. m0+15]: the block computes that [rows <= 256] x 16 slice together (dO and W2's 16 columns through LDS,
// one dot product over the `out2 <= 16` classes per element) and the MFMA loop reads its A operand from LDS.  (A first version
// had every lane form its own dZ elements with 17 dependent global loads each: 52 us instead of 29 for the whole C1 step.)
// dZ itself is never written to memory: nobody else needs it when x has no gradient.
// The README quick-start MLP's backward pass was two dependent launches (gemm_small_pair_kernel for layer 2, gemm_small_kernel
// for dW1 / db1, ~4.9 us each at the dependent-launch floor); the second one's only dependence on the first was dZ.
struct SmallMlpBwd {
    const float* X1; const float* H; const float* W2; const float* dO;
    float* dW2; float* db2; float* dW1; float* db1;
    int rows, in1, hid, out2;
};
constexpr int SG_MLP_MAXC = 16;
// Optional: Adam / AdamW applied to each gradient element by the thread that has just produced it ("optimizer in backward":
// the README-MLP step then has no optimizer launch at all).  Tensor order: 0 = W2, 1 = b2, 2 = W1, 3 = b1.
struct SmallMlpAdam {
    float* p[4]; float* m[4]; float* v[4];
    AdamHyper h;
    float* dev_state; const float* grad_div;
    unsigned* ticket;                                      // a zeroed library word: arrival counter of the launch's blocks
    double b1, b2;
    int enabled;
};
struct SgAdamPost {                                        // W1 / b1: nothing in this launch reads them -- update in place, at once
    float* p; float* m; float* v; float* pb; float* mb; float* vb;
    AdamHyper h;                                           // by value: a pointer to it forced the struct into scratch memory
    bool on;
    __device__ __forceinline__ void operator()(int64_t i, float g) const { if (on) adam1(p[i], g, m[i], v[i], h); }
    __device__ __forceinline__ void asum(int64_t i, float g) const { if (on) adam1(pb[i], g, mb[i], vb[i], h); }
};
// W2 / b2 are INPUTS of the dW1 blocks (dZ = dO W2): they may only change once every block has read them.  Their gradients are
// re-stored with agent-scope (write-through) stores and the LAST block of the launch to finish applies their update (ticket).
struct SgPublishPost {
    float* g; float* gb;
    __device__ __forceinline__ void operator()(int64_t i, float v) const { __hip_atomic_store(&g[i], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    __device__ __forceinline__ void asum(int64_t i, float v) const { __hip_atomic_store(&gb[i], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
};

constexpr int SG_MLP_MAXROWS = 256;

template <int NW>
__device__ __forceinline__ void sg_tile16_dz(const SmallMlpBwd& q, int bx, int by, float (*red)[16 * 16], float (*ared)[16],
                                             float* __restrict__ dOs, float* __restrict__ dZs, float* __restrict__ w2s,
                                             const SgAdamPost& post) {
    constexpr int U = 8, BS = NW * 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l16 = lane & 15, kq = lane >> 4;
    const int64_t m0 = (int64_t)by * 16, n0 = (int64_t)bx * 16;               // m = hidden unit, n = input feature, k = batch row
    const unsigned K = (unsigned)q.rows, groups = (K + 15) >> 4;
    // the optimizer state of the element this thread will produce: fetched now, a memory round trip before it is needed
    const int64_t orow = m0 + ((tid & 255) >> 4), ocol = n0 + (tid & 15), oidx = orow * q.in1 + ocol;
    const bool omine = tid < 256 && orow < q.hid && ocol < q.in1;
    float pw = 0.f, mw = 0.f, vw = 0.f;
    if (post.on && omine) { pw = post.p[oidx]; mw = post.m[oidx]; vw = post.v[oidx]; }
    // ---- the block's slice of dZ, computed together: dZ[b][o] = (sum_c dO[b][c] W2[c][o]) [H[b][o] > 0], o = m0 .. m0+15 ------
    for (int i = tid; i < q.rows * 16; i += BS) {
        const int b = i >> 4, c = i & 15;
        dOs[i] = c < q.out2 ? q.dO[(int64_t)b * q.out2 + c] : 0.f;
    }
    __syncthreads();                                                           // (w2s: loaded by the caller)
    for (int i = tid; i < q.rows * 16; i += BS) {
        const int b = i >> 4, oo = i & 15;
        const float h = m0 + oo < q.hid ? q.H[(int64_t)b * q.hid + m0 + oo] : 0.f;
        float sacc = 0.f;
#pragma unroll
        for (int c = 0; c < SG_MLP_MAXC; ++c) sacc = fmaf(dOs[b * 16 + c], w2s[c * 16 + oo], sacc);
        dZs[b * 17 + oo] = h > 0.f ? sacc : 0.f;
    }
    __syncthreads();
    // ---- dW1 tile = dZ^T X1 on the 16x16x4 MFMA, the waves splitting the batch rows; A comes from LDS, B (X1) from global ---
    const unsigned lb4 = (unsigned)q.in1 * 4u;
    const __amdgpu_buffer_rsrc_t rsb = sg_rsrc(q.X1, (unsigned)((int64_t)q.rows * q.in1 * 4));
    const bool b_ok = n0 + l16 < q.in1;
    const unsigned b_row = (unsigned)(n0 + l16) * 4u;
    sg_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    float asum = 0.f;
    auto dz = [&](unsigned b) -> float { return b < K ? dZs[b * 17 + l16] : 0.f; };
    for (unsigned gb = wave; gb < groups; gb += (unsigned)NW * U) {
        float4 a[U], b[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const unsigned k0 = 16u * (gb + (unsigned)u * NW) + 4u * kq;
            b[u] = sg_fetch<false, false>(rsb, lb4, b_row, b_ok, k0, K);
            a[u] = make_float4(dz(k0), dz(k0 + 1), dz(k0 + 2), dz(k0 + 3));
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].x, b[u].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].y, b[u].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].z, b[u].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].w, b[u].w, acc, 0, 0, 0);
            asum += (a[u].x + a[u].y) + (a[u].z + a[u].w);
        }
    }
#pragma unroll
    for (int v = 0; v < 4; ++v) red[wave][(4 * kq + v) * 16 + l16] = acc[v];
    asum += __shfl_xor(asum, 16, 64);
    asum += __shfl_xor(asum, 32, 64);
    if (kq == 0) ared[wave][l16] = asum;
    __syncthreads();
    if (tid < 256) {
        float v = red[0][tid];
#pragma unroll
        for (int w = 1; w < NW; ++w) v += red[w][tid];
        if (omine) {
            q.dW1[oidx] = v;
            if (post.on) {
                adam1(pw, v, mw, vw, post.h);
                post.p[oidx] = pw; post.m[oidx] = mw; post.v[oidx] = vw;
            }
        }
    }
    if (q.db1 && bx == 0 && tid < 16) {
        float sacc = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) sacc += ared[w][tid];
        if (m0 + tid < q.hid) {
            q.db1[m0 + tid] = sacc;
            post.asum(m0 + tid, sacc);
        }
    }
}

template <int NW>
__global__ __launch_bounds__(NW * 64) void gemm_small_mlp_bwd_kernel(const SmallMlpBwd q, const SmallMlpAdam ad) {
    __shared__ float red[NW][16 * 16];
    __shared__ float ared[NW][16];
    __shared__ float ash[5];
    // 1-D grid: the dW2 tiles first, then the dW1 tiles -- every block has work (and every block takes the arrival ticket below)
    const int tx0 = (q.hid + 15) >> 4, ty0 = (q.out2 + 15) >> 4, tx1 = (q.in1 + 15) >> 4;
    const int nb0 = tx0 * ty0, id = (int)blockIdx.x;
    const bool first = id < nb0;
    const int bx = first ? id % tx0 : (id - nb0) % tx1, by = first ? id / tx0 : (id - nb0) / tx1;
    AdamHyper h = ad.h;                                    // (a local copy: modifying the by-value argument put it in scratch)
    __shared__ int last;
    if (threadIdx.x == 0) last = 0;
    if (ad.enabled) adam_dev_begin(h, ad.dev_state, ad.b1, ad.b2, ad.grad_div, ash);
    // Arrival ticket (optimizer inside): W2 / b2 are INPUTS of the dW1 blocks, so they may only change once every dW1 block has
    // read them and every dW2 block has published its gradients.  A dW1 block checks in as soon as its W2 slice is in LDS (the
    // atomics then run under its GEMM), a dW2 block when its stores have left; the last arrival updates W2 / b2 and advances the
    // optimizer's device state.  Two levels: block b checks in at word 1 + b % 28 and the last arrival of each word at word 0
    // (one word for all ~400 blocks made this kernel 33 us instead of 7: read-modify-writes on ONE address retire at ~25 M/s).
    // The thread that checks in is lane 0 of the LAST wave: with 32 batch rows that wave has no k-group of the GEMM to wait for.
    auto arrive = [&]() -> int {
        const unsigned n = gridDim.x, b = blockIdx.x;
        const unsigned ways = n < 28u ? n : 28u, s1 = b % ways;
        const unsigned n1 = n / ways + (s1 < n % ways ? 1u : 0u);
        if (atomicAdd(ad.ticket + 1 + s1, 1u) != n1 - 1) return 0;
        __hip_atomic_store(ad.ticket + 1 + s1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return atomicAdd(ad.ticket, 1u) == ways - 1 ? 1 : 0;
    };
    const bool usher = threadIdx.x == (NW - 1) * 64;
    if (first) {                                           // dW2[out2, hid] = dO^T[out2, rows] H[rows, hid],  db2 = row sums of dO^T
        SmallGemmParams p{};
        p.A = q.dO; p.lda = q.out2; p.B = q.H; p.ldb = q.hid; p.C = q.dW2; p.ldc = q.hid; p.asum = q.db2;
        p.M = q.out2; p.N = q.hid; p.K = q.rows; p.alpha = 1.f; p.beta = 1.f;
        if (ad.enabled) {
            const SgPublishPost post{q.dW2, q.db2};
            sg_tile16<NW, false, false, false, 8, SgPublishPost>(p, bx, by, red, ared, nullptr, post);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the agent-scope gradient stores above are in L2 after this
            __syncthreads();
            if (usher) last = arrive();
        } else {
            sg_tile16<NW, false, false, false, 8>(p, bx, by, red, ared);
        }
    } else {
        __shared__ float dOs[SG_MLP_MAXROWS * 16], dZs[SG_MLP_MAXROWS * 17], w2s[256];
        if (threadIdx.x < 256) {
            const int c = threadIdx.x >> 4, oo = threadIdx.x & 15;
            const int64_t m0 = (int64_t)by * 16;
            w2s[threadIdx.x] = (c < q.out2 && m0 + oo < q.hid) ? q.W2[(int64_t)c * q.hid + m0 + oo] : 0.f;
        }
        if (ad.enabled) {
            __syncthreads();                               // every thread's W2 element has arrived (the LDS write waited for it)
            if (usher) last = arrive();
        }
        const SgAdamPost post{ad.p[2], ad.m[2], ad.v[2], ad.p[3], ad.m[3], ad.v[3], h, ad.enabled != 0};
        sg_tile16_dz<NW>(q, bx, by, red, ared, dOs, dZs, w2s, post);
    }
    if (ad.enabled) {
        __syncthreads();
        if (last) {
            constexpr int BS = NW * 64, E = 4;
            const int n2 = q.out2 * q.hid, n = n2 + q.out2;
            for (int base = 0; base < n; base += BS * E) {     // all loads of E elements per thread in flight together
                float g[E], pp[E], mm[E], vv[E];
                float *ptr_p[E], *ptr_m[E], *ptr_v[E];
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const int i = base + e * BS + (int)threadIdx.x;
                    const bool ok = i < n, w = i < n2;
                    const int j = ok ? (w ? i : i - n2) : 0;
                    ptr_p[e] = (w ? ad.p[0] : ad.p[1]) + j; ptr_m[e] = (w ? ad.m[0] : ad.m[1]) + j; ptr_v[e] = (w ? ad.v[0] : ad.v[1]) + j;
                    g[e] = ok ? __hip_atomic_load((w ? q.dW2 : q.db2) + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
                    pp[e] = ok ? *ptr_p[e] : 0.f; mm[e] = ok ? *ptr_m[e] : 0.f; vv[e] = ok ? *ptr_v[e] : 0.f;
                }
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    if (base + e * BS + (int)threadIdx.x < n) {
                        adam1(pp[e], g[e], mm[e], vv[e], h);
                        *ptr_p[e] = pp[e]; *ptr_m[e] = mm[e]; *ptr_v[e] = vv[e];
                    }
                }
            }
            if (threadIdx.x == 0) {
                __hip_atomic_store(ad.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (ad.dev_state) adam_dev_advance(ad.dev_state, ad.b1, ad.b2);
            }
        }
    }
}

// rows <= 256, out2 <= 16, operands below 2 GiB; returns 1 when it did the work, 0 when the caller should take the general path
int gemm_small_mlp_backward(const float* X1, const float* H, const float* W2, const float* dO, float* dW2, float* db2, float* dW1,
                            float* db1, int64_t rows, int64_t in1, int64_t hid, int64_t out2, hipStream_t st,
                            const SmallMlpAdam* adam) {
    if (rows <= 0 || rows > SG_MLP_MAXROWS || out2 <= 0 || out2 > SG_MLP_MAXC || in1 <= 0 || hid <= 0) return 0;
    if (!gemm_small_wanted(out2, hid, rows, 1, out2, hid, false, false) || !gemm_small_wanted(hid, in1, rows, 1, hid, in1, false, false)) return 0;
    if (rows * in1 * 4 >= ((int64_t)1 << 31) || rows * hid * 4 >= ((int64_t)1 << 31)) return 0;
    SmallMlpBwd q;
    q.X1 = X1; q.H = H; q.W2 = W2; q.dO = dO; q.dW2 = dW2; q.db2 = db2; q.dW1 = dW1; q.db1 = db1;
    q.rows = (int)rows; q.in1 = (int)in1; q.hid = (int)hid; q.out2 = (int)out2;
    const int nw = ((rows + 15) >> 4) >= 8 ? 8 : 4;
    dim3 grid((unsigned)(ceil_div(hid, 16) * ceil_div(out2, 16) + ceil_div(in1, 16) * ceil_div(hid, 16)));
    SmallMlpAdam ad{};
    if (adam) ad = *adam;
    if (nw == 8) hipLaunchKernelGGL((gemm_small_mlp_bwd_kernel<8>), grid, dim3(512), 0, st, q, ad);
    else hipLaunchKernelGGL((gemm_small_mlp_bwd_kernel<4>), grid, dim3(256), 0, st, q, ad);
    NNHIP_LAUNCH_CHECK("gemm_small_mlp_bwd_kernel");
    return 1;
}

// The same with Adam / AdamW applied in the epilogues.  pmv: 12 device pointers {p, m, v} x {W2, b2, W1, b1}; step >= 1: host
// stepping (bias corrections of `step`); step == 0: the optimizer handle's device state (graph replay).
int fused_optimizer_state(void* opt, int step, float** dev_state, const float** grad_div);   // optim.hip
int gemm_small_mlp_backward_adam(const float* X1, const float* H, const float* W2, const float* dO, float* dW2, float* db2, float* dW1,
                                 float* db1, int64_t rows, int64_t in1, int64_t hid, int64_t out2, void* opt, float* const* pmv,
                                 double lr, double b1, double b2, double eps, double wd, int step, int decay_mode, float grad_scale,
                                 hipStream_t st) {
    SmallMlpAdam ad{};
    for (int t = 0; t < 4; ++t) { ad.p[t] = pmv[3 * t]; ad.m[t] = pmv[3 * t + 1]; ad.v[t] = pmv[3 * t + 2]; }
    ad.h = make_hyper(lr, b1, b2, eps, wd, step > 0 ? step : 1, decay_mode, grad_scale);
    ad.b1 = b1; ad.b2 = b2; ad.enabled = 1;
    unsigned* sync = sync_words();
    if (!sync) { set_last_error("mlp backward: sync words allocation failed"); return NNHIP_ENOMEM; }
    ad.ticket = sync + SYNC_MLP;
    if (int rc = fused_optimizer_state(opt, step, &ad.dev_state, &ad.grad_div)) return rc;
    return gemm_small_mlp_backward(X1, H, W2, dO, dW2, db2, dW1, db1, rows, in1, hid, out2, st, &ad);
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
