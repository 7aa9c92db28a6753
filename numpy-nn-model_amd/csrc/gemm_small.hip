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
        // (k-groups in flight per wave sized to the problem: a body that fetches 8 multiplies the groups past the end as zeros)
        if ((int64_t)bx * 16 < p.N && (int64_t)by * 16 < p.M) {
            if (((p.K + 15) >> 4) <= NW) sg_tile16<NW, true, false, VEC0, 1>(p, bx, by, red, ared);
            else sg_tile16<NW, true, false, VEC0, 8>(p, bx, by, red, ared);
        }
    } else {                                               // dW[out, in] = dO^T[out, rows] X[rows, in],  db = row sums of dO^T
        p.B = q.X; p.C = q.dW; p.asum = q.db; p.beta = 1.f;
        p.M = q.out; p.K = q.rows;
        if ((int64_t)bx * 16 < p.N && (int64_t)by * 16 < p.M) {
            const int64_t g = (p.K + 15) >> 4;
            if (g <= NW) sg_tile16<NW, false, false, false, 1>(p, bx, by, red, ared);
            else if (g <= 2 * NW) sg_tile16<NW, false, false, false, 2>(p, bx, by, red, ared);
            else sg_tile16<NW, false, false, false, 8>(p, bx, by, red, ared);
        }
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
    unsigned* err;                                         // the library's device error word (host-mapped; runtime.hip)
    double b1, b2;
    int enabled;
};
constexpr int SG_MLP_MAXROWS = 256;
// dW2 / db2 blocks: sg_tile16 hands the element (and row sum) a thread has just produced back to the kernel
struct SgCapturePost {
    float* v; float* s;
    __device__ __forceinline__ void operator()(int64_t, float x) const { *v = x; }
    __device__ __forceinline__ void asum(int64_t, float x) const { *s = x; }
};

// One dW1 tile (bx, by) = dZ^T X1 with the block's [rows] x 16 slice of dZ built in LDS first.  Every global load the tile needs
// -- the optimizer state of the element a thread will produce, the W2 slice, dO, H, the X1 fragments -- is ISSUED before the first
// wait (`after_issue()` runs there: the optimizer's device state is resolved behind the same round trip), `w2_read()` is called
// once the block's W2 values have arrived (the arrival ticket of the optimizer-inside variant).
template <int NW, int U, class F1, class F2>
__device__ __forceinline__ void sg_tile16_dz(const SmallMlpBwd& q, int bx, int by, float (*red)[16 * 16], float (*ared)[16],
                                             float* __restrict__ dOs, float* __restrict__ dZs, float* __restrict__ w2s,
                                             float* const (&pmv)[6], bool adam_on, const AdamHyper& h, F1 after_issue, F2 w2_read) {
    constexpr int BS = NW * 64, PRE = 4;                    // U: k-groups in flight per wave (1 when every wave has at most one)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l16 = lane & 15, kq = lane >> 4;
    const int64_t m0 = (int64_t)by * 16, n0 = (int64_t)bx * 16;               // m = hidden unit, n = input feature, k = batch row
    const unsigned K = (unsigned)q.rows, groups = (K + 15) >> 4;
    const int total = q.rows * 16;
    // (1) optimizer state of this thread's element, (2) W2 slice, (3) dO, (4) H, (5) X1 fragments: all in flight together
    const int64_t orow = m0 + ((tid & 255) >> 4), ocol = n0 + (tid & 15), oidx = orow * q.in1 + ocol;
    const bool omine = tid < 256 && orow < q.hid && ocol < q.in1;
    const bool bmine = bx == 0 && tid < 16 && m0 + tid < q.hid;
    float pw = 0.f, mw = 0.f, vw = 0.f, pbv = 0.f, mbv = 0.f, vbv = 0.f;
    if (adam_on) {
        const int64_t oi = omine ? oidx : 0, bi = bmine ? m0 + tid : 0;       // clamped: unconditional loads, no exec-masked regions
        pw = pmv[0][oi]; mw = pmv[1][oi]; vw = pmv[2][oi];
        pbv = pmv[3][bi]; mbv = pmv[4][bi]; vbv = pmv[5][bi];
    }
    float w2r = 0.f;
    {
        const int c = (tid & 255) >> 4, oo = tid & 15;
        const bool ok = tid < 256 && c < q.out2 && m0 + oo < q.hid;
        w2r = q.W2[ok ? (int64_t)c * q.hid + m0 + oo : 0];
        w2r = ok ? w2r : 0.f;
    }
    float dOr[PRE], Hr[PRE];
#pragma unroll
    for (int e = 0; e < PRE; ++e) {
        const int i = tid + e * BS, b = i >> 4, c = i & 15;
        const bool okd = i < total && c < q.out2, okh = i < total && m0 + c < q.hid;
        dOr[e] = q.dO[okd ? (int64_t)b * q.out2 + c : 0];
        Hr[e] = q.H[okh ? (int64_t)b * q.hid + m0 + c : 0];
    }
    const unsigned lb4 = (unsigned)q.in1 * 4u;
    const __amdgpu_buffer_rsrc_t rsb = sg_rsrc(q.X1, (unsigned)((int64_t)q.rows * q.in1 * 4));
    const bool b_ok = n0 + l16 < q.in1;
    const unsigned b_row = (unsigned)(n0 + l16) * 4u;
    float4 bf[U];                                                              // first (for rows <= 512: only) pass of the k loop
#pragma unroll
    for (int u = 0; u < U; ++u) bf[u] = sg_fetch<false, false>(rsb, lb4, b_row, b_ok, 16u * ((unsigned)wave + (unsigned)u * NW) + 4u * kq, K);
    __builtin_amdgcn_sched_barrier(0);
    after_issue();
    // ---- the block's slice of dZ, computed together: dZ[b][o] = (sum_c dO[b][c] W2[c][o]) [H[b][o] > 0], o = m0 .. m0+15 ------
    if (tid < 256) w2s[tid] = w2r;
#pragma unroll
    for (int e = 0; e < PRE; ++e) {
        const int i = tid + e * BS, c = i & 15;
        if (i < total) dOs[i] = c < q.out2 ? dOr[e] : 0.f;
    }
    for (int i = tid + PRE * BS; i < total; i += BS) {
        const int b = i >> 4, c = i & 15;
        dOs[i] = c < q.out2 ? q.dO[(int64_t)b * q.out2 + c] : 0.f;
    }
    __syncthreads();
    w2_read();
    auto dz_of = [&](int i, float hval) {
        const int b = i >> 4, oo = i & 15;
        float sacc = 0.f;
#pragma unroll
        for (int c = 0; c < SG_MLP_MAXC; ++c) sacc = fmaf(dOs[b * 16 + c], w2s[c * 16 + oo], sacc);
        dZs[b * 17 + oo] = hval > 0.f ? sacc : 0.f;
    };
#pragma unroll
    for (int e = 0; e < PRE; ++e) {
        const int i = tid + e * BS;
        if (i < total) dz_of(i, m0 + (i & 15) < q.hid ? Hr[e] : 0.f);
    }
    for (int i = tid + PRE * BS; i < total; i += BS)
        dz_of(i, m0 + (i & 15) < q.hid ? q.H[(int64_t)(i >> 4) * q.hid + m0 + (i & 15)] : 0.f);
    __syncthreads();
    // ---- dW1 tile = dZ^T X1 on the 16x16x4 MFMA, the waves splitting the batch rows; A comes from LDS, B (X1) from global ---
    sg_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    float asum = 0.f;
    auto dz = [&](unsigned b) -> float { return b < K ? dZs[b * 17 + l16] : 0.f; };
    for (unsigned gb = wave; gb < groups; gb += (unsigned)NW * U) {
        float4 a[U], b[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const unsigned k0 = 16u * (gb + (unsigned)u * NW) + 4u * kq;
            if (gb == (unsigned)wave) b[u] = bf[u];
            else b[u] = sg_fetch<false, false>(rsb, lb4, b_row, b_ok, k0, K);
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
#ifndef MLP_NO_W1ADAM
            if (adam_on) {
                adam1(pw, v, mw, vw, h);
                pmv[0][oidx] = pw; pmv[1][oidx] = mw; pmv[2][oidx] = vw;
            }
#endif
        }
    }
    if (q.db1 && bx == 0 && tid < 16) {
        float sacc = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) sacc += ared[w][tid];
        if (bmine) {
            q.db1[m0 + tid] = sacc;
            if (adam_on) {
                adam1(pbv, sacc, mbv, vbv, h);
                pmv[3][m0 + tid] = pbv; pmv[4][m0 + tid] = mbv; pmv[5][m0 + tid] = vbv;
            }
        }
    }
}

// Optimizer inside (ad.enabled): every gradient element is handed to Adam by the thread that produced it, its state fetched
// before the GEMM.  W1 / b1 are read by nobody in this launch.  W2 / b2 are inputs of the dW1 blocks (dZ = dO W2): a dW2 block
// applies its update only once EVERY block of the launch has checked in -- a dW1 block when its W2 slice has arrived, a dW2 block
// at once -- which it learns by polling (it only ever waits for blocks that wait for nobody).  A stepper block computes the next
// step's bias corrections (two double pow) while everybody works and stores them once everybody has read the state.
// The arrival board has NO read-modify-write whose result anyone waits for: block b adds 1 to word b % 28 of the launch's bank
// (fire and forget), a poller loads the 28 words with one wave-load and compares each with the number of blocks that map to it.
// Two banks alternate by a launch epoch (a device word, so graph replays advance it): the closer -- dW2 block 0, once it has
// seen everybody -- clears the OTHER bank (last used by the previous launch, which is over) and advances the epoch; nothing is
// ever reset under a poller's eyes.  In-kernel stamps of the first version (one returning atomic per block into 28 counters, their
// last arrivals into one more; a "last poller clears" counter): a dW1 block spent 3.8 us between its W2 slice and its stores,
// most of it the usher's atomic round trips in front of the block's next barrier; "everybody is here" became visible 3 us after
// the last block was, and the stepper's returning atomic + clears added 1.4 us behind that: 9.8 us for 6.1 us of gradients.
constexpr int SG_MLP_WAYS = 28, SG_MLP_BANK0 = 32, SG_MLP_BANK1 = 64;    // word offsets from ad.ticket (word 0 = the epoch)
template <int NW>
__global__ __launch_bounds__(NW * 64) void gemm_small_mlp_bwd_kernel(const SmallMlpBwd q, const SmallMlpAdam ad) {
    __shared__ float red[NW][16 * 16];
    __shared__ float ared[NW][16];
    __shared__ float ash[5];
    __shared__ int flag;
    // 1-D grid: the dW2 tiles first, then the dW1 tiles, then (device stepping) the stepper
    const int tx0 = (q.hid + 15) >> 4, ty0 = (q.out2 + 15) >> 4, tx1 = (q.in1 + 15) >> 4;
    const int nb0 = tx0 * ty0, id = (int)blockIdx.x, tid = (int)threadIdx.x, lane = tid & 63;
    const bool first = id < nb0;
    const int bx = first ? id % tx0 : (id - nb0) % tx1, by = first ? id / tx0 : (id - nb0) / tx1;
    AdamHyper h = ad.h;                                    // (a local copy: modifying the by-value argument put it in scratch)
    const bool on = ad.enabled != 0;
    const bool stepper = on && ad.dev_state != nullptr;    // one extra block (the last) whose only job is the optimizer's step
    const unsigned n = gridDim.x - (stepper ? 1u : 0u);    // blocks that check in
    // the launch epoch: requested now by every thread that will need it (same address: one request per wave), used much later
    // (only by the threads that will: every wave of 400 blocks asking for one address is a queue at one L2 channel)
    unsigned epoch = 0;
    const bool stepper_blk = stepper && (unsigned)id == n;
    if (on && (tid == 0 || (stepper_blk && tid < 64) || (first && (tid >= (NW - 1) * 64 || (id == 0 && tid < 64)))))
        epoch = __hip_atomic_load(ad.ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    auto bank = [&](unsigned e) { return ad.ticket + ((e & 1u) ? SG_MLP_BANK1 : SG_MLP_BANK0); };
    auto arrive = [&]() {                                  // one thread; nobody waits for the add
        __hip_atomic_fetch_add(bank(epoch) + (unsigned)id % SG_MLP_WAYS, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    auto everybody_here = [&]() -> bool {                  // a whole wave: lane w looks at word w
        bool ok = true;
        // bounded by WALL time (s_memrealtime: 100 MHz), not by a spin count: a legitimately slow arrival -- a debugger, a profiler
        // replaying the kernel, a time-sliced device -- must not abort training (advisor, round 4); 20 s without it is a broken device
        const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
        for (;;) {
            unsigned seen = 0, want = 0;
            if (lane < SG_MLP_WAYS) {
                seen = __hip_atomic_load(bank(epoch) + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                want = n / SG_MLP_WAYS + ((unsigned)lane < n % SG_MLP_WAYS ? 1u : 0u);
            }
            ok = __ballot(seen < want) == 0ull;
            if (ok || __builtin_amdgcn_s_memrealtime() - t_start > 2000000000ull) break;
            __builtin_amdgcn_s_sleep(2);
        }
        // The host only launches this variant when every block that polls fits on the chip next to blocks that wait for nobody
        // (mlp_adam_grid_fits below), so "everybody is here" must come true; several seconds without it means the device state
        // is broken (a stale ticket bank, a foreign writer).  The block then SKIPS its update and raises the library's device
        // error word (pinned host memory, runtime.hip): the launch ends normally, the context survives, and every later library
        // entry that checks the word -- this entry's next call, the optimizer's step, nnhipDeviceError() -- answers
        // NNHIP_EDEVICE until nnhipClearDeviceError() (round-5 review: a status code, not __builtin_trap()).  The reference's
        // error convention for the same class of failure is printf + exit(1) (linear_cublaslt_no_manual_mem.cu:91-94).
        if (!ok && lane == 0 && ad.err)
            __hip_atomic_store(ad.err, (unsigned)NNHIP_DEVERR_MLP_BARRIER, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return ok;
    };
    if (stepper_blk) {
        if (tid < 64) {
            AdamDevNext nx{};
            if (tid == 0) nx = adam_dev_next(ad.dev_state, ad.b1, ad.b2);      // while everybody else works
            if (everybody_here() && tid == 0) adam_dev_commit(ad.dev_state, nx, ad.b1, ad.b2);
        }
        return;
    }
    AdamDevRaw raw;
    if (on) adam_dev_issue(raw, ad.dev_state, ad.grad_div);
    if (first) {                                           // dW2[out2, hid] = dO^T[out2, rows] H[rows, hid],  db2 = row sums of dO^T
        SmallGemmParams p{};
        p.A = q.dO; p.lda = q.out2; p.B = q.H; p.ldb = q.hid; p.C = q.dW2; p.ldc = q.hid; p.asum = q.db2;
        p.M = q.out2; p.N = q.hid; p.K = q.rows; p.alpha = 1.f; p.beta = 1.f;
        if (!on) {
            if (((q.rows + 15) >> 4) <= NW) sg_tile16<NW, false, false, false, 1>(p, bx, by, red, ared);
            else sg_tile16<NW, false, false, false, 8>(p, bx, by, red, ared);
            return;
        }
        const int64_t row = (int64_t)by * 16 + ((tid & 255) >> 4), col = (int64_t)bx * 16 + (tid & 15), idx = row * q.hid + col;
        const bool mine = tid < 256 && row < q.out2 && col < q.hid;
        const bool bmine = bx == 0 && tid < 16 && (int64_t)by * 16 + tid < q.out2;
        const int64_t wi = mine ? idx : 0, bi = bmine ? (int64_t)by * 16 + tid : 0;
        float pw = ad.p[0][wi], mw = ad.m[0][wi], vw = ad.v[0][wi];
        float pb = ad.p[1][bi], mb = ad.m[1][bi], vb = ad.v[1][bi];
        adam_dev_resolve(h, raw, ad.dev_state, ad.b1, ad.b2, ad.grad_div, ash);
        if (tid == 0) arrive();                            // this block has read the optimizer's state (and reads no W2)
        float g = 0.f, gb = 0.f;
        const SgCapturePost post{&g, &gb};
        if (((q.rows + 15) >> 4) <= NW) sg_tile16<NW, false, false, false, 1, SgCapturePost>(p, bx, by, red, ared, nullptr, post);
        else sg_tile16<NW, false, false, false, 8, SgCapturePost>(p, bx, by, red, ared, nullptr, post);
        if (tid >= (NW - 1) * 64) {                        // the last wave polls (normally everybody checked in long ago)
            const bool ok = everybody_here();
            if (lane == 0) flag = ok ? 1 : 0;
        }
        __syncthreads();
        if (flag) {
            if (mine) { adam1(pw, g, mw, vw, h); ad.p[0][idx] = pw; ad.m[0][idx] = mw; ad.v[0][idx] = vw; }
            if (bmine) { adam1(pb, gb, mb, vb, h); ad.p[1][bi] = pb; ad.m[1][bi] = mb; ad.v[1][bi] = vb; }
            if (id == 0 && tid < 64) {                     // the closer: the other bank for the next launch, then the epoch
                if (lane < SG_MLP_WAYS) __hip_atomic_store(bank(epoch + 1u) + lane, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane == 0) __hip_atomic_store(ad.ticket, epoch + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    } else {
        __shared__ float dOs[SG_MLP_MAXROWS * 16], dZs[SG_MLP_MAXROWS * 17], w2s[256];
        float* const pmv[6] = {ad.p[2], ad.m[2], ad.v[2], ad.p[3], ad.m[3], ad.v[3]};
        auto resolve = [&]() { if (on) adam_dev_resolve(h, raw, ad.dev_state, ad.b1, ad.b2, ad.grad_div, ash); };
        auto checkin = [&]() { if (on && tid == 0) arrive(); };
        // a short batch (rows <= 16 NW: the README MLP's 32 rows on 4 waves) leaves a wave at most ONE k-group: the 1-group body --
        // the 8-group body multiplies the seven groups past the end as zeros, 28 dependent MFMAs (1.5 of the tile's 2.4 us)
        if (((q.rows + 15) >> 4) <= NW) sg_tile16_dz<NW, 1>(q, bx, by, red, ared, dOs, dZs, w2s, pmv, on, h, resolve, checkin);
        else sg_tile16_dz<NW, 8>(q, bx, by, red, ared, dOs, dZs, w2s, pmv, on, h, resolve, checkin);
    }
}

// The optimizer-in-backward launch holds an in-kernel arrival barrier: the dW2 blocks (and the stepper) poll until every block has
// checked in.  That is only safe while the pollers cannot occupy every resident slot -- the blocks they wait for (dW1 tiles, which
// wait for nobody) must always find room.  Pollers are dispatched first (lowest block ids), so the rule is: pollers <= half of the
// blocks the chip holds at once (occupancy query, cached per block size).
static bool mlp_adam_grid_fits(int nw, int64_t pollers) {
    static int cap[16][2] = {};                            // per device (the occupancy of one part says nothing about another)
    int dev = 0, cus = 0, per = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return false;
    int& c = cap[dev][nw == 8];
    if (c == 0) {
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return false;
        const hipError_t e = nw == 8 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, gemm_small_mlp_bwd_kernel<8>, 512, 0)
                                     : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, gemm_small_mlp_bwd_kernel<4>, 256, 0);
        if (e != hipSuccess || per <= 0 || cus <= 0) return false;
        c = per * cus;
    }
    return pollers * 2 <= c;
}

// rows <= 256, out2 <= 16, operands below 2 GiB; returns 1 when it did the work, 0 when the caller should take the general path
// The envelope of the launch below (with_adam: including the optimizer-in-backward variant's barrier rule); 1 = it will do the work
int gemm_small_mlp_fits(int64_t rows, int64_t in1, int64_t hid, int64_t out2, int with_adam) {
    if (rows <= 0 || rows > SG_MLP_MAXROWS || out2 <= 0 || out2 > SG_MLP_MAXC || in1 <= 0 || hid <= 0) return 0;
    if (!gemm_small_wanted(out2, hid, rows, 1, out2, hid, false, false) || !gemm_small_wanted(hid, in1, rows, 1, hid, in1, false, false)) return 0;
    if (rows * in1 * 4 >= ((int64_t)1 << 31) || rows * hid * 4 >= ((int64_t)1 << 31)) return 0;
    const int nw = ((rows + 15) >> 4) >= 8 ? 8 : 4;
    if (with_adam && !mlp_adam_grid_fits(nw, ceil_div(hid, 16) * ceil_div(out2, 16) + 1)) return 0;
    return 1;
}

int gemm_small_mlp_backward(const float* X1, const float* H, const float* W2, const float* dO, float* dW2, float* db2, float* dW1,
                            float* db1, int64_t rows, int64_t in1, int64_t hid, int64_t out2, hipStream_t st,
                            const SmallMlpAdam* adam) {
    if (!gemm_small_mlp_fits(rows, in1, hid, out2, 0)) return 0;
    SmallMlpBwd q;
    q.X1 = X1; q.H = H; q.W2 = W2; q.dO = dO; q.dW2 = dW2; q.db2 = db2; q.dW1 = dW1; q.db1 = db1;
    q.rows = (int)rows; q.in1 = (int)in1; q.hid = (int)hid; q.out2 = (int)out2;
    const int nw = ((rows + 15) >> 4) >= 8 ? 8 : 4;
    dim3 grid((unsigned)(ceil_div(hid, 16) * ceil_div(out2, 16) + ceil_div(in1, 16) * ceil_div(hid, 16)));
    SmallMlpAdam ad{};
    if (adam) ad = *adam;
    if (ad.enabled && ad.dev_state) grid.x += 1;           // the stepper block
    // a hidden layer so wide that its dW2 tiles (the polling blocks) could fill the chip: no in-kernel barrier -- the caller
    // runs the plain backward launch and the separate optimizer step (same values)
    if (ad.enabled && !mlp_adam_grid_fits(nw, ceil_div(hid, 16) * ceil_div(out2, 16) + 1)) return 0;
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
    ad.err = device_error_word();
    if (int rc = device_error_status("nnhipLinearReLULinearBackwardAdam")) return rc;   // an earlier launch timed out: say so, do not pile on
    if (int rc = serialize_shared_state(st)) return rc;     // one arrival board per process (runtime.hip)
    const SharedStateUse in_use(st);
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
