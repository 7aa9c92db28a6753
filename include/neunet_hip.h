/* neunet_hip.h -- C ABI of libneunet_hip.so: the MI355X (gfx950) drop-in for the CUDA .so files
 * that neunet's experimental layer binds through ctypes (reference:
 * neunet/nn/experimental/utils.py:64-92 -- ctypes.CDLL + getattr + argtypes).
 *
 * Conventions (differences from the reference's exports are deliberate and listed here):
 *   - every entry point returns int: 0 = ok, >0 = hipError_t, <0 = NNHIP_E* argument error
 *     (the reference returns void and printf+exit()s on failure, e.g.
 *     linear_cublaslt_no_manual_mem.cu:91-94,117-120; cross_entropy.cu:287-291);
 *   - every launch takes an explicit stream (hipStream_t passed as void*) as its LAST argument
 *     (the reference's Linear and CrossEntropy exports implicitly use stream 0);
 *   - sizes are int64_t (the reference mixes int / size_t: linear.py:46-48 vs .cu:114);
 *   - all tensor pointers are DEVICE pointers to C-contiguous fp32 (labels: int32); the caller owns
 *     and pre-allocates every buffer (reference: utils.py:72-82, `xp.empty` before each call);
 *     optional pointers may be NULL where noted;
 *   - argument ORDER and MEANING follow the reference export each function replaces (cited).
 * The library never synchronises the device inside a compute entry point.
 */
#ifndef NEUNET_HIP_H
#define NEUNET_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* nnhipStream_t; /* hipStream_t */

#define NNHIP_OK 0
#define NNHIP_EINVAL (-1)   /* bad argument (null pointer, negative size, bad enum) */
#define NNHIP_EALIGN (-2)   /* pointer not 4-byte aligned */
#define NNHIP_ENOMEM (-3)   /* workspace allocation failed */
#define NNHIP_EDEVICE (-5)  /* a kernel of an EARLIER launch found the device state broken and raised the library's device error
                             * word; sticky until nnhipClearDeviceError() (-4 is NNHIP_ECOMM, below) */

/* ---- library ------------------------------------------------------------------------------ */
int nnhipVersion(void);
/* Human-readable text for the last non-zero status returned on the calling thread. */
const char* nnhipGetLastErrorString(void);
/* Free the grow-only device workspace (split-K slabs, column-sum partials).
 * Replaces cleanupCudaMemory() (linear_cublaslt_no_manual_mem.cu:186, linear_cutlass.cu:107,
 * linear_swish_cutlass_evt_full.cu:820).  Synchronises the device. */
int nnhipCleanup(void);
/* Device-side errors (ABI 210).  A kernel cannot return a status, and the reference's convention for a failure inside its CUDA
 * path is printf + exit(1) (linear_cublaslt_no_manual_mem.cu:91-94).  Here a kernel that finds the device state broken -- today
 * only the optimizer-in-backward launch whose arrival barrier saw no progress for 20 s -- skips its side effect, stores a code in
 * a library-owned word of pinned host memory and ends normally.  nnhipDeviceError() reads that word without synchronising:
 * 0, or NNHIP_EDEVICE with the story in nnhipGetLastErrorString().  nnhipLinearReLULinearBackwardAdam and the optimizer step
 * entries check it on entry, so a training loop stops at its next step instead of losing the context to a trap.
 * nnhipClearDeviceError() resets the word.  nnhipRaiseDeviceErrorForTest() stores `code` from a one-block kernel on `stream`
 * (the tests' way to exercise the path without a broken device). */
int nnhipDeviceError(void);
int nnhipClearDeviceError(void);
int nnhipRaiseDeviceErrorForTest(int32_t code, nnhipStream_t stream);
/* Grow the workspace to at least `bytes` now (e.g. before capturing a hipGraph). */
int nnhipWorkspaceReserve(int64_t bytes);
/* locked != 0: the workspace may no longer move -- a launch that needs more than it holds returns NNHIP_ENOMEM instead
 * of freeing the block a captured hipGraph still points into.  nnhipCleanup() unlocks. */
int nnhipWorkspaceLock(int locked);

/* GEMM arithmetic of every Linear / attention GEMM that runs on the 128x128-tile kernel:
 *   0 (default)  exact fp32: v_mfma_f32_32x32x2_f32, a k-ordered fmaf chain;
 *   1            split-bf16 ("bf16x3"): every fp32 operand is split EXACTLY into three bf16 pieces and six of the nine
 *                piece products are accumulated in fp32 on the bf16 matrix cores -- relative error <= ~2^-23 per
 *                product (the order of fp32 rounding itself), up to 2.67x the fp32 MFMA rate.  Opt-in; also
 *                NNHIP_GEMM_MODE=1 in the environment.  Small problems (gemm_small) stay exact either way. */
int nnhipSetGemmMode(int mode);
int nnhipGetGemmMode(void);
/* Lock-step mode of the exact-fp32 128x128-tile kernel for reductions of >= 2048 (off by default; also NNHIP_GEMM_LOCKSTEP=1):
 * the two blocks resident on a CU hold each other to within a few k-tiles through issue priorities, so the tiles that share an
 * operand panel in an XCD's L2 fetch it from the fabric once instead of twice (4096^3 forward: 808 -> 575 MB of L2->fabric
 * reads, dW 655 -> 574; dX, already at 541, is left alone) at +2.7 % kernel time (dW +0.6 %).  Results are bit-identical either way (the arithmetic and its order do not change).  For
 * deployments where the fabric is shared with collectives.  ABI 205 */
int nnhipSetGemmLockstep(int enable);
int nnhipGetGemmLockstep(void);
/* Launches since the library was loaded, per GEMM kernel family: 0 = classic fp32 128x128 tiles (gemm_f32_kernel),
 * 1 = persistent fp32 (gemm_pst_kernel), 2 = small-problem kernel (gemm_small*), 3 = split-bf16 (gemm_bf3_kernel);
 * -1 for any other argument.  Host-side bookkeeping for tests that must know which kernel produced a result.  ABI 203 */
int64_t nnhipGemmLaunchCount(int family);

/* ---- a1/a2 Linear  (replaces cudaLinearModuleForward/Backward,
 *      linear_cublaslt_no_manual_mem.cu:114,142 and linear_cutlass.cu:40,67) ------------------ */
/* O[rows,out] = X[rows,in] * W[out,in]^T + b[out]      (b may be NULL) */
int nnhipLinearModuleForward(const float* X, const float* W, const float* b, float* O,
                             int64_t rows, int64_t in_features, int64_t out_features,
                             nnhipStream_t stream);
/* dX[rows,in] = dO*W ; dW[out,in] = dO^T*X ; db[out] = sum_rows dO.  Any of dX/dW/db may be NULL
 * (skipped). */
int nnhipLinearModuleBackward(const float* X, const float* W, const float* dO, float* dX,
                              float* dW, float* db, int64_t rows, int64_t in_features,
                              int64_t out_features, nnhipStream_t stream);
/* Extensions with an addend folded into the GEMM epilogue (both NULL-able; NULL == the plain entry points):
 *   forward : O  = X*W^T + b + addend        (addend [rows,out]: the residual of `x + linear(h)`, neunet/autograd.py add)
 *   backward: dX = dO*W + dX_addend          (dX_addend [rows,in]: a gradient X has already received -- replaces the
 *                                             separate accumulation of Tensor.apply_grad, neunet/autograd.py:85-93)
 * The addend is read once per output float4 (may alias nothing that is written). db is produced by the dW GEMM itself. */
int nnhipLinearModuleForwardEx(const float* X, const float* W, const float* b, const float* addend, float* O,
                               int64_t rows, int64_t in_features, int64_t out_features, nnhipStream_t stream);
int nnhipLinearModuleBackwardEx(const float* X, const float* W, const float* dO, const float* dX_addend, float* dX,
                                float* dW, float* db, int64_t rows, int64_t in_features, int64_t out_features,
                                nnhipStream_t stream);

/* Deferred parameter gradients (extension; the reference computes each layer's dW where its backward runs, linear.py:17-24).
 * enable != 0: from now on the Linear backward entry points (nnhipLinearModuleBackward[Ex|Act], nnhipLinearSwishBackward) QUEUE
 * a dW/db GEMM that is too small to fill the chip alone (<= 256 output tiles of 128x128, >= 4096 rows, 16-B aligned operands;
 * either GEMM mode); dX is still computed by the call itself.  nnhipWeightGradFlush launches everything queued as ONE grid and
 * ONE reduce -- a transformer layer's four dW GEMMs stop paying four launch ramps, four simultaneous slab epilogues and four
 * tails.  A job's reduction is always cut into four chunks, so a gradient's bits do not depend on what else was in the queue.
 * Contract while a job is queued: its X and dO stay alive and unmodified, its dW/db are not read.  enable == 0: flush on
 * `stream`, then every later dW is launched where it is asked for (the default).  nnhipWeightGradPending: queued jobs.
 * The switch and the queue are ONE per process, for ONE host thread and ONE stream at a time (like the workspace): while it is
 * on, a backward entry called from another thread or on another stream is queued all the same -- a caller that defers must be
 * the only caller until it has flushed (Tensor.backward() switches it on for the duration of its own tape walk only).  ABI 204 */
int nnhipWeightGradDefer(int32_t enable, nnhipStream_t stream);
int nnhipWeightGradFlush(nnhipStream_t stream);
/* Only the queued dW/db GEMMs (one grouped launch + its reduce) on `stream`, which may be a SIDE stream: the caller orders it behind
 * the producers of the queued calls' X / dO, keeps those alive until it has joined `stream` again, and reads dW / db only after the
 * join.  The launch keeps its split-K slabs in a block of its own, so kernels of the main stream may run next to it.  Everything
 * else a flush does (conv reduces, RMSNorm column sums) stays with nnhipWeightGradFlush.  ABI 210 */
int nnhipWeightGradFlushGemms(nnhipStream_t stream);
int nnhipWeightGradPending(void);

/* ---- a6 fused Linear -> Swish  (replaces cudaLinearSwishForward/Backward,
 *      linear_swish_cutlass_evt_full.cu:558-570, 680-695) ------------------------------------- */
/* z = X*W^T + b ; O = z*sigmoid(beta*z).  If save_preactivation != 0, z is also written to
 * `preact` (must be non-NULL then).  save_preactivation == 2 (ABI 210, no reference counterpart): `preact` receives
 * swish'(z) = s + beta*O*(1 - s), s = sigmoid(beta*z), instead of z -- the same sigmoid serves both outputs, and the backward
 * pass (nnhipLinearSwishBackward with recompute_preactivation = 2, nnhipLinearInputGradScaled, nnhipLinearModuleBackwardAct with
 * act_grad = 3) only multiplies: in an fp32 MFMA kernel every vector instruction of an epilogue is paid in matrix-pipe time. */
int nnhipLinearSwishForward(const float* X, const float* W, const float* b, float* O, float* preact,
                            int64_t M, int64_t K, int64_t N, float swish_beta,
                            int save_preactivation, nnhipStream_t stream);
/* tmp[M,N]: if recompute_preactivation == 0 it holds z on entry (saved by the forward); 2: it holds swish'(z) (forward with
 * save_preactivation = 2); otherwise (1) it is scratch and z is recomputed into it.  On exit tmp holds dZ (reference in-place contract,
 * ...evt_full.cu:707-711).  dX/dW/db may be NULL. */
int nnhipLinearSwishBackward(const float* X, const float* W, const float* b, const float* dO,
                             float* tmp, float* dX, float* dW, float* db, int64_t M, int64_t K,
                             int64_t N, float swish_beta, int recompute_preactivation,
                             nnhipStream_t stream);

/* ---- general fp32 MFMA GEMM (net-new; used by the entries above and by the batched matmul of
 *      SURVEY 8f-1, neunet/autograd.py:192-230) -------------------------------------------------
 * C[b] (M x N, row-major, ldc) = op(A[b]) * op(B[b]) (+ bias[N]) ;  b = 0..batch-1.
 *   a_kmajor != 0: A element (m,k) at A[m*lda + k]   (reduction dim contiguous)
 *   a_kmajor == 0: A element (m,k) at A[k*lda + m]
 *   b_kmajor != 0: B element (k,n) at B[n*ldb + k]
 *   b_kmajor == 0: B element (k,n) at B[k*ldb + n]
 * strideA/B/C: element offsets between consecutive batches (0 = shared operand). */
int nnhipGemmF32(const float* A, const float* B, float* C, const float* bias, int64_t M, int64_t N,
                 int64_t K, int64_t lda, int64_t ldb, int64_t ldc, int a_kmajor, int b_kmajor,
                 int64_t batch, int64_t strideA, int64_t strideB, int64_t strideC,
                 nnhipStream_t stream);

/* Same, with an output scale and a two-level batch: batch index z = z1*batch2 + z2 addresses
 * A + z1*sA1 + z2*sA2 (likewise B, C):  C = alpha * op(A) op(B) (+ bias).  Lets attention read Q/K/V
 * straight out of their [B,T,H*dh] projection buffers and write the context back in that layout -- no
 * transpose copies (examples/gpt.ipynb cell 2 transposes/reshapes on the host side instead). */
int nnhipGemmF32Ex(const float* A, const float* B, float* C, const float* bias, int64_t M, int64_t N,
                   int64_t K, int64_t lda, int64_t ldb, int64_t ldc, int a_kmajor, int b_kmajor,
                   int64_t batch1, int64_t sA1, int64_t sB1, int64_t sC1, int64_t batch2, int64_t sA2,
                   int64_t sB2, int64_t sC2, float alpha, nnhipStream_t stream);

/* ---- a12 ReLU (net-new export; reference CPU: neunet/nn/activations.py:40-59) --------------- */
int nnhipReLUForward(float* out, const float* in, int64_t size, nnhipStream_t stream);
/* dIn = dOut * (out > 0), `out` = forward output */
int nnhipReLUBackward(float* dIn, const float* dOut, const float* out, int64_t size,
                      nnhipStream_t stream);

/* Input gradient of a Linear whose input was h = swish(z) (the FFN's fc_2 after the fused Linear->Swish fc_1,
 * examples/gpt.ipynb cell 2): dZ[rows,in] = (dO[rows,out] * W[out,in]) (.) swish'(Z; beta), i.e. the dX GEMM of
 * nnhipLinearModuleBackward with the element-wise Swish backward (neunet/nn/activations.py:223-232) applied in its
 * epilogue.  dZ may alias Z (each element is read once, then written).  Feed dZ to nnhipLinearModuleBackward of fc_1. */
int nnhipLinearInputGradSwish(const float* dO, const float* W, const float* Z, float* dZ, int64_t rows,
                              int64_t in_features, int64_t out_features, float swish_beta, nnhipStream_t stream);

/* The same with the derivative in hand: dZ = (dO * W) (.) D, D = what nnhipLinearSwishForward(save_preactivation = 2) left in
 * `preact`.  dZ may alias D.  ABI 210 */
int nnhipLinearInputGradScaled(const float* dO, const float* W, const float* D, float* dZ, int64_t rows,
                               int64_t in_features, int64_t out_features, nnhipStream_t stream);

/* The same for h = relu(z): dZ = (dO * W) (.) [F > 0] with F = the ReLU's forward output (activations.py:44-45); dZ must
 * not alias F (F is still the Linear's input for its dW). */
int nnhipLinearInputGradReLU(const float* dO, const float* W, const float* F, float* dZ, int64_t rows,
                             int64_t in_features, int64_t out_features, nnhipStream_t stream);
/* Both of the above plus the parameter gradients in one call: dZ = (dO * W) (.) act'(act_arg) -- act_grad 1: swish'(act_arg = Z;
 * beta), dZ may alias Z; 2: [act_arg = F > 0], dZ must not alias F; 3 (ABI 210): act_arg is the saved derivative swish'(z) itself, a plain
 * multiplier, dZ may alias it -- and dW = dO^T X, db = column sums of dO (either may be NULL),
 * i.e. _LinearTensor.grad_fn (linear.py:17-24) followed by the activation's backward.  A small layer (the README MLP's
 * 128 -> 10 head) gets dZ, dW and db from ONE launch. */
int nnhipLinearModuleBackwardAct(const float* X, const float* W, const float* dO, const float* act_arg, int32_t act_grad,
                                 float beta, float* dZ, float* dW, float* db, int64_t rows, int64_t in_features,
                                 int64_t out_features, nnhipStream_t stream);
/* Backward of out = Linear2(relu(Linear1(X1))) when X1 needs no gradient (README quick-start MLP, README.md:57-71): dW2 =
 * dO^T H, db2, dW1 = dZ^T X1, db1 with dZ = (dO W2) (.) [H > 0] formed inside the dW1 tiles (never written) -- ONE launch where
 * nnhipLinearModuleBackwardAct + nnhipLinearModuleBackward take two dependent ones.  H = the ReLU output [rows, hidden].
 * rows <= 256, out2 <= 16, small layers only: NNHIP_EINVAL otherwise (use the two calls).  ABI 203 */
int nnhipLinearReLULinearBackward(const float* X1, const float* H, const float* W2, const float* dO, float* dW2, float* db2,
                                  float* dW1, float* db1, int64_t rows, int64_t in1, int64_t hidden, int64_t out2,
                                  nnhipStream_t stream);
/* The same with the optimizer inside: every gradient element is handed to Adam / AdamW (neunet/optim.py:17-33, 52-69) by the
 * thread that produced it -- the README-MLP step then has no optimizer launch.  Gradients are still written.  pmv: 12 device
 * pointers {param, m, v} x {W2, b2, W1, b1}.  step >= 1: host stepping; step == 0: device stepping through `opt`
 * (nnhipCreateFusedOptimizer + SetStep / SetHyper).  Hyper-parameters as nnhipFusedAdamWMultiTensorStep.  ABI 203 */
int nnhipLinearReLULinearBackwardAdam(const float* X1, const float* H, const float* W2, const float* dO, float* dW2, float* db2,
                                      float* dW1, float* db1, int64_t rows, int64_t in1, int64_t hidden, int64_t out2,
                                      void* opt, float* const* pmv, double lr, double beta1, double beta2, double eps,
                                      double weight_decay, int32_t step, int32_t decay_mode, float grad_scale,
                                      nnhipStream_t stream);
/* 1 when the two entries above (with_adam 0 / 1) take these sizes on the current device, 0 when they would answer NNHIP_EINVAL
 * -- for a caller that decides BEFORE it defers the launch (neunet_hip defers it to optimizer.step() so that the README MLP's
 * default training step gets the backward + Adam launch without opting in).  No device work.  ABI 209 */
int nnhipLinearReLULinearBackwardFits(int64_t rows, int64_t in1, int64_t hidden, int64_t out2, int32_t with_adam);
/* O = act(X*W^T + b), activation in the GEMM epilogue: 1 = swish(beta) without saving z, 2 = relu, 3 = sigmoid.
 * One launch for what `act(Linear(x))` is on the reference's tape (linear.py:48-58 + activations.py); the host side
 * uses it when an activation module is applied to a Linear output nobody else has looked at yet. */
int nnhipLinearActivationForward(const float* X, const float* W, const float* b, float* O, int64_t rows,
                                 int64_t in_features, int64_t out_features, int32_t activation, float beta,
                                 nnhipStream_t stream);

/* ---- a5 Swish  (replaces cudaSwishForward/Backward, swish.cu:50,65) ------------------------- */
int nnhipSwishForward(float* out, const float* in, float beta, int64_t size, nnhipStream_t stream);
int nnhipSwishBackward(float* dIn, const float* dOut, const float* in, float beta, int64_t size,
                       nnhipStream_t stream);

/* ---- a7 fused Swish-and-mul / SwiGLU gate  (replaces cudaFusedSwishAndMul[Backward],
 *      fused_swish_and_mul.cu:60,76).  in rows = [gate(hidden), up(hidden)], size = #OUTPUT elems */
int nnhipFusedSwishAndMul(float* out, const float* in, float beta, int64_t hidden, int64_t size,
                          nnhipStream_t stream);
int nnhipFusedSwishAndMulBackward(float* dIn, const float* dOut, const float* in, float beta,
                                  int64_t hidden, int64_t size, nnhipStream_t stream);

/* ---- a8 Softmax  (replaces cudaSoftmaxForward/Backward, softmax.cu:144,229).
 *      Slice s = outer*stride + inner starts at element outer*slice_size*stride + inner; its
 *      slice_size elements are `stride` apart (stride == 1: softmax over contiguous rows). */
int nnhipSoftmaxForward(float* out, const float* in, int64_t num_slices, int64_t slice_size,
                        int64_t stride, nnhipStream_t stream);
int nnhipSoftmaxBackward(float* dX, const float* dY, const float* Y, int64_t num_slices,
                         int64_t slice_size, int64_t stride, nnhipStream_t stream);

/* Attention-score softmax with scale and mask fused (examples/gpt.ipynb cell 2):
 *   y = softmax_j( masked(b,i,j) ? -1e9 : scale * x[b,h,i,j] ),  x,y: [B,H,Tq,Tk]
 *   masked = (key_valid && key_valid[b*Tk + j] == 0) || (causal && j > i + Tk - Tq).   key_valid may be NULL.
 * Backward: dX = masked ? 0 : scale * (dY - sum_j dY*Y) * Y. */
int nnhipMaskedSoftmaxForward(float* out, const float* in, const int32_t* key_valid, int64_t B, int64_t H,
                              int64_t Tq, int64_t Tk, float scale, int causal, nnhipStream_t stream);
int nnhipMaskedSoftmaxBackward(float* dX, const float* dY, const float* Y, const int32_t* key_valid,
                               int64_t B, int64_t H, int64_t Tq, int64_t Tk, float scale, int causal,
                               nnhipStream_t stream);

/* The same with an additional dense mask [B,Tq,Tk] int32 (0 = masked; broadcast over heads; NULL = none): what the notebook's
 * MultiHeadAttention.forward receives (get_pad_mask & get_sub_mask, cell 7), for callers that want the attention map. */
int nnhipMaskedSoftmaxForwardEx(float* out, const float* in, const int32_t* key_valid, const int32_t* dense_mask, int64_t B,
                                int64_t H, int64_t Tq, int64_t Tk, float scale, int causal, nnhipStream_t stream);
int nnhipMaskedSoftmaxBackwardEx(float* dX, const float* dY, const float* Y, const int32_t* key_valid,
                                 const int32_t* dense_mask, int64_t B, int64_t H, int64_t Tq, int64_t Tk, float scale,
                                 int causal, nnhipStream_t stream);

/* Fused (flash-style) attention, head_dim 32 / 64 / 128: same math as  QK^T*scale -> mask(-1e9) -> softmax -> dropout -> *V
 * above, but the [B,H,Tq,Tk] score matrix is never written.  O/dO are [B,T,H*head_dim] (the projection layout); Q/K/V/dQ/dK/dV
 * are [B,T,*] with row stride ld_qkv floats (0 = H*head_dim; 3*H*head_dim when they are the three column blocks of one fused
 * q|k|v projection buffer).  LSE [B,H,Tq,2] = (row max, log2 row sum) of each masked score row in log2 units, saved by the
 * forward and consumed by the backward; the pair is kept apart because a fully-masked row has max = -1e9*log2(e),
 * where fp32 cannot hold max + log(sum). */
int nnhipAttentionForward(const float* Q, const float* K, const float* V, const int32_t* key_valid, float* O,
                          float* LSE, int64_t B, int64_t H, int64_t Tq, int64_t Tk, int64_t head_dim, int64_t ld_qkv,
                          float scale, int causal, nnhipStream_t stream);
/* dQ, dK, dV from (Q, K, V, O, dO, LSE): P is recomputed tile by tile; deterministic (no atomics): one kernel owns
 * 128-key blocks (dK, dV), one owns 128-query blocks (dQ). */
int nnhipAttentionBackward(const float* Q, const float* K, const float* V, const int32_t* key_valid, const float* O,
                           const float* dO, const float* LSE, float* dQ, float* dK, float* dV, int64_t B, int64_t H,
                           int64_t Tq, int64_t Tk, int64_t head_dim, int64_t ld_qkv, float scale, int causal,
                           nnhipStream_t stream);
/* The same with the rest of examples/gpt.ipynb cell 2 inside the kernels (all fields optional; a NULL struct == the calls
 * above):
 *   mask_bits / mask_bitsT / row_any: an arbitrary dense [B,Tq,Tk] mask (what the notebook builds with get_pad_mask &
 *     get_sub_mask and passes down), packed by nnhipAttentionPackMask; when given it REPLACES key_valid and causal;
 *   attention dropout (the notebook's self.dropout(softmax(scores)), dropout.py:17-37): the multiplier of element
 *     (b,h,q,k) is dropout_mask[b,h,q,k] (0 or 1/(1-p): an injected mask, for parity tests against the oracle) or, with
 *     dropout_mask == NULL and dropout_p > 0, (hash(dropout_seed, b, h, q, k) >= p*2^32) / (1-p) -- a counter-based hash
 *     that the forward and both backward kernels re-evaluate (nothing is stored); nnhipAttentionDropoutMask writes the
 *     same multipliers out.  Pass the SAME options to the forward and the backward of one step. */
typedef struct nnhipAttentionOptions {
    const uint64_t* mask_bits;    /* [B, Tq, ceil(Tk/64)] */
    const uint64_t* mask_bitsT;   /* [B, Tk, ceil(Tq/64)] */
    const uint8_t* row_any;       /* [B, Tq] (NULL: no tile skipping under a dense mask) */
    const float* dropout_mask;    /* [B, H, Tq, Tk] */
    float dropout_p;
    uint32_t dropout_seed;
    const uint32_t* dropout_seed_dev; /* NULL, or a device word added to dropout_seed when the kernel runs: point it at a
                                       * per-step counter so that a captured hipGraph draws a fresh mask on every replay */
} nnhipAttentionOptions;
int nnhipAttentionForwardEx(const float* Q, const float* K, const float* V, const int32_t* key_valid, float* O,
                            float* LSE, int64_t B, int64_t H, int64_t Tq, int64_t Tk, int64_t head_dim, int64_t ld_qkv,
                            float scale, int causal, const nnhipAttentionOptions* opts, nnhipStream_t stream);
int nnhipAttentionBackwardEx(const float* Q, const float* K, const float* V, const int32_t* key_valid, const float* O,
                             const float* dO, const float* LSE, float* dQ, float* dK, float* dV, int64_t B, int64_t H,
                             int64_t Tq, int64_t Tk, int64_t head_dim, int64_t ld_qkv, float scale, int causal,
                             const nnhipAttentionOptions* opts, nnhipStream_t stream);
/* mask [B,Tq,Tk] int32 (non-zero = visible) -> mask_bits, mask_bitsT, row_any (one launch). */
int nnhipAttentionPackMask(const int32_t* mask, uint64_t* mask_bits, uint64_t* mask_bitsT, uint8_t* row_any, int64_t B,
                           int64_t Tq, int64_t Tk, nnhipStream_t stream);
/* out [B,H,Tq,Tk] = the hash dropout multipliers the fused kernels use for (dropout_p, seed). */
int nnhipAttentionDropoutMask(float* out, int64_t B, int64_t H, int64_t Tq, int64_t Tk, float dropout_p, uint32_t seed,
                              nnhipStream_t stream);
/* The same with an optional device word added to the seed (the fused kernels' dropout_seed_dev: a per-step counter that lets a
 * captured hipGraph draw a fresh mask on every replay) -- what the UNFUSED attention path (need_weights = True) multiplies its
 * attention map by, so that both paths draw from the library's one counter hash.  ABI 208 */
int nnhipAttentionDropoutMaskEx(float* out, int64_t B, int64_t H, int64_t Tq, int64_t Tk, float dropout_p, uint32_t seed,
                                const uint32_t* seed_dev, nnhipStream_t stream);

/* ---- a9 fused CrossEntropy forward+backward  (replaces cudaCrossEntropyForwardBackward,
 *      cross_entropy.cu:249-260).
 *   logits [n_rows, logits_stride]; loss[n_rows], lse[n_rows] outputs; labels int32.
 *   reduction 'n' | 'm' | 's'.  Gradient scale for 'm' = 1/n_non_ignore, taken from the int
 *   argument, or -- if n_non_ignore_dev != NULL -- read from that device int (no host sync).
 *   dlogits: where to write (softmax - onehot)*scale.  NULL = in place over `logits`
 *   (the reference's behaviour, cross_entropy.cu:211); rows with label == ignore_index get zero
 *   gradient and zero loss.  dlogits has the same row stride as logits. */
int nnhipCrossEntropyForwardBackward(float* logits, float* loss, float* lse, const int32_t* labels,
                                     int64_t logits_stride, int32_t ignore_index, int64_t n_rows,
                                     int64_t n_cols, char reduction, int64_t n_non_ignore,
                                     const int32_t* n_non_ignore_dev, float* dlogits,
                                     nnhipStream_t stream);
/* out_count[0] = #{i : labels[i] != ignore_index}   (device scalar; replaces the host-side
 * cp.sum(labels != ignore_index).item() of cross_entropy.py:72) */
int nnhipCountNotEqual(const int32_t* labels, int64_t n, int32_t ignore_index, int32_t* out_count,
                       nnhipStream_t stream);
/* The 'mean' denominator on its own (labels int16/32/64): count_out[0] = #{labels != ignore_index} (int) and/or
 * denom_out[0] = that count -- or, with class weights, sum of class_weight[label] over those rows -- as a FLOAT.
 * Data-parallel use: every rank back-propagates reduction 's', writes its local denominator into the extra slot of the
 * gradient bucket with this call, and the all-reduced slot becomes the optimizer's gradient divisor
 * (nnhipFusedOptimizerSetGradDivisor): the global mean without a host read. */
int nnhipCrossEntropyDenominator(const void* labels, int32_t label_bytes, int64_t n, int64_t ignore_index,
                                 const float* class_weight_or_null, int64_t n_cols, int32_t* count_out_or_null,
                                 float* denom_out_or_null, nnhipStream_t stream);
/* out[0] = sum(loss_rows) ('s'), or sum/count ('m', count from count_dev) -- the device-side
 * reduction cross_entropy.py:98-101 does with cupy. */
int nnhipReduceLoss(const float* loss_rows, int64_t n_rows, char reduction,
                    const int32_t* count_dev, float* out, nnhipStream_t stream);
/* The three calls above as one launch (reduction 'm' or 's'): d(logits) (NULL = in place), per-row loss, lse, the reduced
 * loss (device scalar) and, for 'm', the non-ignored count (device int, also the 'mean' denominator).
 * = nnhipCrossEntropyLossEx with int32 labels and no class weights. */
int nnhipCrossEntropyLoss(float* logits, float* dlogits_or_null, float* loss_rows, float* lse, const int32_t* labels,
                          int64_t logits_stride, int32_t ignore_index, int64_t n_rows, int64_t n_cols, char reduction,
                          float* loss_out, int32_t* count_out, nnhipStream_t stream);

/* The whole CrossEntropyLoss (neunet/nn/losses.py:59-126) in ONE launch, any reduction:
 *   labels: int16 / int32 / int64 (label_bytes = 2 / 4 / 8; losses.py:100 accepts the three);
 *   class_weight [n_cols] or NULL: loss_i = -logp[y_i] * w[y_i]; 'm' divides by sum_i w[y_i] over non-ignored rows
 *     (losses.py:115-118); gradient rows are scaled by w[y_i];
 *   reduction 'n': loss_rows only (loss_out may be NULL); 'm' / 's': loss_out[0] as well;
 *   count_out (NULL-able): #{labels != ignore_index} for 'm'.
 * The 'mean' denominator is derived from the labels inside the launch (no count launch, no host sync); the per-block
 * loss sums meet in the last block to finish (arrival ticket).  A label outside [0, n_cols) that is not ignore_index
 * gives zero loss and zero gradient.  Rows of any width (looped above 16384 columns). */
int nnhipCrossEntropyLossEx(float* logits, float* dlogits_or_null, float* loss_rows, float* lse, const void* labels,
                            int32_t label_bytes, const float* class_weight_or_null, int64_t logits_stride,
                            int64_t ignore_index, int64_t n_rows, int64_t n_cols, char reduction,
                            float* loss_out_or_null, int32_t* count_out_or_null, nnhipStream_t stream);

/* A small classifier head and its loss in one launch: logits[rows, classes] = X * W^T + b (Linear.forward, linear.py:48-58), then
 * exactly nnhipCrossEntropyLossEx on them (out of place: dlogits != logits).  1 <= rows <= 256, 1 <= classes <= 32,
 * 1 <= in_features <= 2048, else NNHIP_EINVAL -- the two entries called separately give the same results.  At README-MLP scale
 * (32 x 128 -> 10) a launch is ~1/8 of the training step whatever it computes. */
int nnhipLinearCrossEntropyLoss(const float* X, const float* W, const float* b, float* logits, float* dlogits,
                                float* loss_rows, float* lse, const void* labels, int32_t label_bytes,
                                const float* class_weight_or_null, int64_t ignore_index, int64_t rows, int64_t in_features,
                                int64_t classes, char reduction, float* loss_out_or_null, int32_t* count_out_or_null,
                                nnhipStream_t stream);

/* ---- a10 RMSNorm  (replaces RMSNormForward/Backward, rmsnorm.cu:116-140, 282-308) ---------- */
/* X_std[rows] = sqrt(mean(x^2)+eps) always written.  X_norm[rows,cols] may be NULL (not stored;
 * the backward recomputes it) -- the reference always stores it. */
int nnhipRMSNormForward(const float* X, const float* weight, const float* bias_or_null, float* Y,
                        float* X_std, float* X_norm_or_null, int64_t rows, int64_t cols, float eps,
                        nnhipStream_t stream);
/* X_norm is accepted for signature parity and ignored (recomputed from X and X_std).  dW / db are finished inside the
 * same launch (per-block column partials + arrival ticket; the last blocks to arrive column-sum them).  Rows of any
 * width (looped above 16384 columns, like rmsnorm.cu:17-113). */
int nnhipRMSNormBackward(const float* dY, const float* X, const float* weight, const float* X_std,
                         const float* X_norm_unused, float* dX, float* dW, float* db_or_null,
                         int64_t rows, int64_t cols, nnhipStream_t stream);
/* dX = RMSNorm gradient + dX_addend (NULL-able): folds the accumulation onto a gradient X already holds. */
int nnhipRMSNormBackwardEx(const float* dY, const float* X, const float* weight, const float* X_std,
                           const float* X_norm_unused, const float* dX_addend, float* dX, float* dW,
                           float* db_or_null, int64_t rows, int64_t cols, nnhipStream_t stream);

/* ---- a11 fused AdamW  (replaces FusedAdamWStep, fused_adamw.cu:58-71 -- one tensor) --------- */
/* decay_mode 0: decoupled weight decay (AdamW, neunet/optim.py:52-69);
 * decay_mode 1: L2 decay folded into the gradient (Adam, neunet/optim.py:17-33).
 * grad_scale multiplies g on load (1.0 = reference behaviour; 1/world after a DP sum-all-reduce).
 * Hyper-parameters are DOUBLE (the reference passes float): the CPU path forms (1-beta) and
 * 1-beta^step from Python doubles (optim.py:27-31); (float)0.999 would put a 1.3e-5 relative error
 * into (1-beta2) and hence into v.  They are rounded to fp32 exactly where NumPy rounds them. */
int nnhipFusedAdamWStep(float* p, const float* g, float* m, float* v, double lr, double beta1,
                        double beta2, double eps, double weight_decay, int32_t step, int64_t n,
                        int32_t decay_mode, float grad_scale, nnhipStream_t stream);

/* ---- a11 multi-tensor AdamW  (replaces CreateFusedOptimizer / DestroyFusedOptimizer /
 *      FusedAdamWStep(void*,...), fused_adamw_multitensor.cu:308-333) -------------------------- */
void* nnhipCreateFusedOptimizer(void);
int nnhipDestroyFusedOptimizer(void* opt);
/* p/g/m/v: HOST arrays of n_tensors DEVICE pointers; sizes: HOST array of element counts.
 * One kernel launch for all tensors.  Tables are re-uploaded only when they changed. */
int nnhipFusedAdamWMultiTensorStep(void* opt, int32_t n_tensors, float* const* p,
                                   const float* const* g, float* const* m, float* const* v,
                                   const int64_t* sizes, double lr, double beta1, double beta2,
                                   double eps, double weight_decay, int32_t step, int32_t decay_mode,
                                   float grad_scale, nnhipStream_t stream);

/* Device-driven stepping for hipGraph replay (a captured launch freezes its by-value arguments): after
 * nnhipFusedOptimizerSetStep(opt, t) AND nnhipFusedOptimizerSetHyper(...), calls of nnhipFusedAdamWMultiTensorStep with
 * step == 0 take the step count (advanced by the kernel itself), lr, weight_decay and grad_scale from device memory;
 * the lr / weight_decay / grad_scale ARGUMENTS of such a call are ignored.  Both setters are stream-ordered launches
 * (no synchronisation): call SetHyper between replays to run an LR schedule or change the DP gradient scale. */
int nnhipFusedOptimizerSetStep(void* opt, int32_t step, nnhipStream_t stream);
int nnhipFusedOptimizerSetHyper(void* opt, double lr, double weight_decay, float grad_scale, nnhipStream_t stream);
/* Gradients are additionally divided by divisor_dev[0] (a device float; NULL switches it off): with
 * CrossEntropy(ignore_index) under data parallelism every rank back-propagates the SUM loss and the all-reduced count
 * of non-ignored targets (one extra float in the gradient bucket) is the divisor -- no host read, no scale pass. */
int nnhipFusedOptimizerSetGradDivisor(void* opt, const float* divisor_dev_or_null);

/* ---- a3/a4 Conv2d  (net-new exports; reference CPU: neunet/nn/layers/conv2d.py:297-355, 16-115)
 *   X [B,Cin,H,W], W [Cout,Cin,kh,kw], bias [Cout] or NULL, O [B,Cout,Ho,Wo]; NCHW fp32.
 *   pad = (up, down, left, right) as Conv2d.build resolves it (conv2d.py:237-243);
 *   Ho = (H+pu+pd-dh*(kh-1)-1)/sh+1 (conv2d.py:245-258), likewise Wo. */
typedef struct nnhipConv2dDesc {
    int64_t B, Cin, H, W, Cout, kh, kw;
    int64_t sh, sw, dh, dw;
    int64_t pu, pd, pl, pr;
} nnhipConv2dDesc;
int nnhipConv2dForward(const float* X, const float* W, const float* bias, float* O,
                       const nnhipConv2dDesc* d, nnhipStream_t stream);
/* dX/dW/db may be NULL (skipped). */
int nnhipConv2dBackward(const float* X, const float* W, const float* dO, float* dX, float* dW,
                        float* db, const nnhipConv2dDesc* d, nnhipStream_t stream);

/* ---- Embedding (reference CPU: neunet/nn/layers/embedding.py:61-75 via
 *      Tensor.__getitem__, neunet/autograd.py:895-916) ------------------------------------------------
 * out[p,:] = weight[ids[p],:] * scale + (pe ? pe[p % seq_len,:] : 0);  ids int32 (negative = from the end).
 * Backward reproduces the reference's ASSIGNMENT semantics (autograd.py:909-910): for repeated ids only
 * the last occurrence contributes:  dW[v,:] = scale * grad_out[last_pos(v),:], 0 for unused rows. */
int nnhipEmbeddingForward(float* out, const float* weight, const int32_t* ids, const float* pe,
                          int64_t n_ids, int64_t dim, int64_t seq_len, int64_t vocab, float scale,
                          nnhipStream_t stream);
int nnhipEmbeddingBackward(float* dW, const float* grad_out, const int32_t* ids, int64_t n_ids,
                           int64_t dim, int64_t vocab, float scale, nnhipStream_t stream);
/* out[i] = (ids[i] != value) as int32: the key-padding mask of examples/gpt.ipynb cell 7 (get_pad_mask:
 * (x != pad_idx).astype(int)) without leaving the library (torch would run a compare and a cast kernel).  ABI 203 */
int nnhipNotEqualInt32(int32_t* out, const int32_t* ids, int64_t n, int32_t value, nnhipStream_t stream);
/* neunet.argmax (neunet/__init__.py:132-139 = np.argmax cast to int32): out[o, j] = index of the FIRST maximum of
 * x[o, :, j] over the middle axis of the C-contiguous view [outer, n, inner] (axis = -1: inner = 1; axis = None: outer = inner
 * = 1, n = numel); a NaN counts as the maximum, the first NaN wins (NumPy's rule).  out int32 [outer, inner].  Bit-exact: a
 * comparison network on (value, index) pairs, no arithmetic.  n == 0 with outer * inner > 0 is NNHIP_EINVAL (NumPy raises
 * ValueError).  ABI 208 */
int nnhipArgmaxF32(int32_t* out, const float* x, int64_t outer, int64_t n, int64_t inner, nnhipStream_t stream);
/* Dropout (neunet/nn/layers/dropout.py:17-37): out[i] = in[i] * m(i), m(i) = 1/(1-p) with probability 1-p, else 0, from a
 * counter-based hash of (seed + *seed_dev, i) -- never stored: the backward pass calls the same entry with the upstream
 * gradient as `in` (same seed) and gets dX = dY * m.  seed_dev (optional device uint32, e.g. a step counter) makes a
 * captured hipGraph draw a fresh mask on every replay.  out may alias in.  ABI 203 */
int nnhipDropout(float* out, const float* in, int64_t n, float p, uint32_t seed, const uint32_t* seed_dev,
                 nnhipStream_t stream);
/* *word += by (one thread): the per-step device counter that nnhipDropout / the attention kernels add to their dropout
 * seed, advanced on the stream in front of a graph replay.  ABI 203 */
int nnhipIncrementU32(uint32_t* word, uint32_t by, nnhipStream_t stream);

/* ---- the rest of the conv-classifier step (examples/convolutional_digits_classifier.ipynb) ----
 * LeakyReLU (neunet/nn/activations.py:60-84): f = x <= 0 ? alpha*x : x ; dx = dy * (f <= 0 ? alpha : 1). */
int nnhipLeakyReLUForward(float* out, const float* in, float alpha, int64_t size, nnhipStream_t stream);
int nnhipLeakyReLUBackward(float* dIn, const float* dOut, const float* out, float alpha, int64_t size,
                           nnhipStream_t stream);
/* Sigmoid (activations.py:9-28): f = 1/(1+exp(-x)) ; dx = dy * f * (1 - f), `out` = forward output. */
int nnhipSigmoidForward(float* out, const float* in, int64_t size, nnhipStream_t stream);
int nnhipSigmoidBackward(float* dIn, const float* dOut, const float* out, int64_t size, nnhipStream_t stream);
/* MaxPool2d (neunet/nn/layers/maxpool2d.py:85-249), NCHW, dilation 1.  pad = (up, down, left, right), padded
 * with -inf; argmax[b,c,ho,wo] = r*kw + s of the FIRST maximum (np.nanargmax).  Backward gathers, so
 * overlapping windows accumulate deterministically. */
typedef struct nnhipPool2dDesc {
    int64_t B, C, H, W, kh, kw, sh, sw, pu, pd, pl, pr;
    int64_t dh, dw;   /* dilation (maxpool2d.py:170-186: taps at r*dh, s*dw; 0 is read as 1).  ABI 203 */
} nnhipPool2dDesc;
int nnhipMaxPool2dForward(float* out, int32_t* argmax, const float* X, const nnhipPool2dDesc* d,
                          nnhipStream_t stream);
int nnhipMaxPool2dBackward(float* dX, const float* dY, const int32_t* argmax, const nnhipPool2dDesc* d,
                           nnhipStream_t stream);
/* MaxPool2d(LeakyReLU(X; alpha)) forward / backward as ONE launch each (alpha > 0): the composition
 * neunet/nn/activations.py:72-84 -> neunet/nn/layers/maxpool2d.py:85-249 of the conv classifier.  `out`/`argmax` as above;
 * the backward takes the pooled forward output and returns the gradient of the LeakyReLU's INPUT. */
int nnhipMaxPool2dLeakyForward(float* out, int32_t* argmax, const float* X, float alpha, const nnhipPool2dDesc* d,
                               nnhipStream_t stream);
int nnhipMaxPool2dLeakyBackward(float* dX, const float* dY, const int32_t* argmax, const float* pooled, float alpha,
                                const nnhipPool2dDesc* d, nnhipStream_t stream);
/* Conv2d weight / bias gradient straight from the gradient of a MaxPool2d that consumed the conv's output, through an optional
 * LeakyReLU (the backward of conv2d.py:16-115 composed with maxpool2d.py:11-82 and activations.py:72-84, restricted to dW, db):
 *   dW, db of   P = MaxPool2d([LeakyReLU(] Conv2d(X) [; alpha)])   given dP, the pool's window-local arg-max and -- with the
 *   activation -- the pooled output P (its sign is the activation's slope at the arg-max); pooled = NULL: no activation.
 * For a conv whose INPUT needs no gradient and whose output nobody else reads (the conv classifier's first layer): the conv-output
 * gradient [B,Cout,Ho,Wo] is never materialised and the pool's backward launch is not needed.  Pool windows must tile the conv
 * output exactly (kernel == stride, no padding, no dilation, Ho % kh == Wo % kw == 0, kh*kw <= 16) and the conv must be in the
 * small-channel 3x3 domain of the LDS-resident weight-gradient kernel: nnhipConv2dWeightGradPooledOk(conv, pool) = 1 says so
 * (0 otherwise; the pool descriptor's B, C, H, W are the conv output's).  Same values as nnhipMaxPool2d[Leaky]Backward followed by
 * nnhipConv2dBackward(dX = NULL).  ABI 206 */
int nnhipConv2dWeightGradPooledOk(const nnhipConv2dDesc* conv, const nnhipPool2dDesc* pool);
int nnhipConv2dWeightGradPooled(const float* X, const float* dP, const int32_t* argmax, const float* pooled, float alpha,
                                float* dW, float* db, const nnhipConv2dDesc* conv, const nnhipPool2dDesc* pool,
                                nnhipStream_t stream);
/* P = MaxPool2d(2, 2)(LeakyReLU(Conv2d(X); alpha)) and the pool's window-local arg-max in ONE launch (alpha = 1: no activation)
 * -- conv2d.py:297-355 -> activations.py:79-81 -> maxpool2d.py:85-249 for a 3x3, unit-stride, unit-dilation conv with few input
 * channels (Cin <= 4 < Cout <= 16) whose output the windows tile exactly and a batch large enough that one thread per window
 * fills the chip (...Ok = 1; else 0: call the three entries).  The conv output is not written: the chain's backward needs only
 * the arg-max and P (nnhipMaxPool2dLeakyBackward, nnhipConv2dWeightGradPooled).  The same products in the same order as the separate
 * entries (values agree to an ulp or two: the compiler's choice of fused multiply-adds differs between the kernels).  ABI 207 */
int nnhipConv2dLeakyMaxPoolForwardOk(const nnhipConv2dDesc* conv, const nnhipPool2dDesc* pool);
int nnhipConv2dLeakyMaxPoolForward(const float* X, const float* W, const float* bias, float alpha, float* P, int32_t* argmax,
                                   const nnhipConv2dDesc* conv, const nnhipPool2dDesc* pool, nnhipStream_t stream);
/* BatchNorm2d (neunet/nn/layers/batchnorm2d.py:57-115, 11-54), X [B,C,HW].  training != 0: batch mean / biased
 * variance per channel, running = momentum*running + (1-momentum)*stat (the reference's convention; running_*
 * may be NULL); else the running statistics are used.  save_mean / save_inv [C] feed the backward.
 * weight / bias [C] may both be NULL (affine = False). */
int nnhipBatchNorm2dForward(const float* X, const float* weight, const float* bias, float* Y, float* save_mean,
                            float* save_inv, float* running_mean, float* running_var, int64_t B, int64_t C,
                            int64_t HW, float eps, float momentum, int training, nnhipStream_t stream);
int nnhipBatchNorm2dBackward(const float* dY, const float* X, const float* weight, const float* save_mean,
                             const float* save_inv, float* dX, float* dW, float* db, int64_t B, int64_t C,
                             int64_t HW, nnhipStream_t stream);
/* The tail of the reference's conv classifier (examples/convolutional_digits_classifier.ipynb cell 2) as ONE launch, without any block
 * waiting for another.  Producer side: nnhipConv2dLeakyMaxPoolForwardStats = nnhipConv2dLeakyMaxPoolForward that also leaves, per block
 * of 64 pooling windows and per output channel, (mean, M2 = sum of squared deviations) of the pooled values in stats
 * [blocks][Cout][2]; nnhipConv2dLeakyMaxPoolStatsBlocks = that block count (0: no statistics variant for the geometry).
 * Consumer side: nnhipBatchNorm2dLinearSigmoidMSE = BatchNorm2d(training) on X [B,C,HW] (statistics combined from the pairs, Chan
 * et al., in a fixed order) -> reshape [B, C*HW] -> Linear(W [N, C*HW], b [N] or NULL) -> Sigmoid -> MSELoss against target [B,N].
 * Y / save_mean / save_inv / running statistics as nnhipBatchNorm2dForward (equal to rounding: combined instead of two-pass
 * statistics), pred [B,N] = the Sigmoid output, dz [B,N] = d(loss)/d(Linear output), loss[0] = mean squared error.
 * C <= 16, N <= 16, C*HW % 4 == 0 and <= 2048, B <= 4096 (...Fits answers 1 / 0); nstat * count == B * HW.  ABI 209 */
int nnhipConv2dLeakyMaxPoolStatsBlocks(const nnhipConv2dDesc* conv, const nnhipPool2dDesc* pool);
int nnhipConv2dLeakyMaxPoolForwardStats(const float* X, const float* W, const float* bias, float alpha, float* P, int32_t* argmax,
                                        const nnhipConv2dDesc* conv, const nnhipPool2dDesc* pool, float* stats, nnhipStream_t stream);
int nnhipBatchNorm2dLinearSigmoidMSEFits(int64_t B, int64_t C, int64_t HW, int64_t N);
int nnhipBatchNorm2dLinearSigmoidMSE(const float* X, const float* stats, int64_t nstat, int64_t count, const float* bn_weight,
                                     const float* bn_bias, float* Y, float* save_mean, float* save_inv, float* running_mean,
                                     float* running_var, int64_t B, int64_t C, int64_t HW, float eps, float momentum, const float* W,
                                     const float* b, int64_t N, const float* target, float* pred, float* dz, float* loss,
                                     nnhipStream_t stream);
/* MSELoss (neunet/nn/losses.py:9-22): loss[0] = sum((pred-target)^2)/n ; dpred = 2 (pred-target)/n (may be NULL). */
int nnhipMSELossForwardBackward(const float* pred, const float* target, float* loss, float* dpred, int64_t n,
                                nnhipStream_t stream);
/* MSELoss(Sigmoid(z), target) with the Sigmoid backward folded in: `pred` = the sigmoid output, dz_out = d(loss)/dz
 * (neunet/nn/losses.py:9-22 composed with neunet/nn/activations.py:12-13; the conv classifier's last two modules). */
int nnhipMSELossSigmoidForwardBackward(const float* pred, const float* target, float* loss, float* dz_out, int64_t n,
                                       nnhipStream_t stream);

/* ---- gradient-bucket helpers for data-parallel training (net-new) ---------------- */
/* x[i] *= alpha */
int nnhipScale(float* x, float alpha, int64_t n, nnhipStream_t stream);
/* out[r,c] = in[r,c] * scale[r * scale_stride], scale_stride 0 (one device scalar) or 1 (a factor per row): a loss node's product
 * with its upstream gradient (cross_entropy.py:111-114 `y_pred.apply_grad(grad_y_pred * grad)`).  ABI 208 */
int nnhipScaleRows(float* out, const float* in, const float* scale, int64_t rows, int64_t cols, int64_t scale_stride,
                   nnhipStream_t stream);
/* out[i] = a[i] + b[i]   (Tensor.apply_grad accumulation, neunet/autograd.py:85-93) */
int nnhipAdd(float* out, const float* a, const float* b, int64_t n, nnhipStream_t stream);
/* out[i] = a[i] * b[i]   (Dropout mask application, neunet/nn/layers/dropout.py:17-37) */
int nnhipMul(float* out, const float* a, const float* b, int64_t n, nnhipStream_t stream);

/* ---- (e) the data-parallel exchange: RCCL behind this ABI (net-new; SURVEY 8b "add AllReduce*", 8e, 7 step 7) ------------
 * The reference has no collective (neunet/autograd.py:8-14: device is "cpu" | "cuda"); these are what a binder that holds
 * plain device pointers (CuPy arrays through neunet/nn/experimental/utils.py:64-92) needs for the ONE exchange of the path:
 * a SUM all-reduce of the flat fp32 gradient bucket (Module.parameters() order, neunet/nn/modules.py:23-39) after
 * backward() and before optimizer.step() -- the optimizer's grad_scale = 1/world (or nnhipFusedOptimizerSetGradDivisor)
 * turns the sum into the mean.  One process per GPU; hipSetDevice(LOCAL_RANK) before nnhipCommInitRank.
 * librccl is bound with dlopen at the first call here (a copy already mapped into the process -- e.g. torch's -- is
 * shared; NNHIP_RCCL_LIB overrides the path): libneunet_hip.so itself loads on a box without RCCL.  Failures return
 * NNHIP_ECOMM with RCCL's text in nnhipGetLastErrorString().  Collectives are enqueued on the caller's stream, in
 * order with the kernels that produced the buffer; they may be captured into a hipGraph like any other launch.  ABI 208 */
#define NNHIP_ECOMM (-4)             /* RCCL missing, or an RCCL call failed */
#define NNHIP_UNIQUE_ID_BYTES 128    /* sizeof(ncclUniqueId) */
typedef struct nnhipComm* nnhipComm_t;
/* Rank 0: fill `id` (NNHIP_UNIQUE_ID_BYTES HOST bytes) -- ncclGetUniqueId; ship it to every rank out of band
 * (a file, a socket, MPI_Bcast, torch's TCPStore). */
int nnhipCommUniqueId(void* id);
/* Every rank, collectively: join the communicator `id` names as `rank` of `world` on the current device. */
int nnhipCommInitRank(nnhipComm_t* comm, const void* id, int rank, int world);
int nnhipCommDestroy(nnhipComm_t comm);                    /* NULL is a no-op */
int nnhipCommRank(nnhipComm_t comm, int* rank, int* world); /* either out pointer may be NULL */
/* Which librccl was bound (path as passed to dlopen; `version` = ncclGetVersion or 0); both out arguments may be NULL. */
int nnhipCommLibrary(char* path, int64_t path_bytes, int* version);
/* buf[i] = sum over ranks of buf[i], in place, n floats (n == 0: no call).  Deterministic for a fixed communicator
 * (same algorithm, same rank order every step), so replicas that start identical stay bit-identical. */
int nnhipAllReduceSumF32(nnhipComm_t comm, float* buf, int64_t n, nnhipStream_t stream);
/* Same with RCCL's pre-scaled average (sum / world). */
int nnhipAllReduceAvgF32(nnhipComm_t comm, float* buf, int64_t n, nnhipStream_t stream);
/* buf on every rank = buf of `root` (identical initial parameters: SURVEY 8e "broadcast from rank 0 or same seed"). */
int nnhipBroadcastF32(nnhipComm_t comm, float* buf, int64_t n, int root, nnhipStream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* NEUNET_HIP_H */
