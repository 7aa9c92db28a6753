// linear.hip -- Linear and fused Linear->Swish entry points on top of the MFMA GEMM.
// CPU semantics: neunet/nn/layers/linear.py:48-58 (fwd), :17-24 (bwd); fused path = Swish(Linear(x)),
// neunet/nn/activations.py:208-233.
#include <mutex>
#include <vector>

#include "common.h"

namespace nnhip {
bool gemm_f32_wgrad_group_ok(const WgradJob& j);
int gemm_f32_wgrad_group(const WgradJob* jobs, int n, hipStream_t st);
int gemm_f32(const float* A, const float* B, float* C, const float* bias, float* preact, int64_t M,
             int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc, bool a_kmajor,
             bool b_kmajor, int64_t batch, int64_t sA, int64_t sB, int64_t sC, int act, float beta,
             hipStream_t st);
int gemm_f32_add(const float* A, const float* B, float* C, const float* bias, const float* addend, int64_t M, int64_t N,
                 int64_t K, int64_t lda, int64_t ldb, int64_t ldc, bool a_kmajor, bool b_kmajor, hipStream_t st);
int gemm_f32_dswish(const float* A, const float* B, float* C, const float* Z, float beta, int64_t M, int64_t N, int64_t K,
                    int64_t lda, int64_t ldb, int64_t ldc, bool a_kmajor, bool b_kmajor, hipStream_t st, int dact = 1);
int mul_f32(float* out, const float* a, const float* b, int64_t n, hipStream_t st);       // elementwise.hip
int gemm_f32_drelu(const float* A, const float* B, float* C, const float* F, int64_t M, int64_t N, int64_t K,
                   int64_t lda, int64_t ldb, int64_t ldc, bool a_kmajor, bool b_kmajor, hipStream_t st);
int gemm_f32_asum(const float* A, const float* B, float* C, float* asum, int64_t M, int64_t N, int64_t K, int64_t lda,
                  int64_t ldb, int64_t ldc, bool b_kmajor, hipStream_t st);
int gemm_f32_linear_backward_small(const float* X, const float* W, const float* dO, float* dX, float* dW, float* db, int64_t rows,
                                   int64_t in, int64_t out, const float* addend, const float* dact_arg, int dact, float beta,
                                   hipStream_t st);
struct SmallMlpAdam;
int gemm_small_mlp_backward(const float* X1, const float* H, const float* W2, const float* dO, float* dW2, float* db2, float* dW1,
                            float* db1, int64_t rows, int64_t in1, int64_t hid, int64_t out2, hipStream_t st,
                            const SmallMlpAdam* adam);
int gemm_small_mlp_fits(int64_t rows, int64_t in1, int64_t hid, int64_t out2, int with_adam);
int gemm_small_mlp_backward_adam(const float* X1, const float* H, const float* W2, const float* dO, float* dW2, float* db2, float* dW1,
                                 float* db1, int64_t rows, int64_t in1, int64_t hid, int64_t out2, void* opt, float* const* pmv,
                                 double lr, double b1, double b2, double eps, double wd, int step, int decay_mode, float grad_scale,
                                 hipStream_t st);
int colsum(const float* X, int64_t rows, int64_t cols, int64_t ld, float* out, hipStream_t st);
int swish_backward_inplace(float* z_inout, const float* dY, float beta, int64_t n, hipStream_t st);
int fill_f32(float* p, float v, int64_t n, hipStream_t st);
enum { ACT_NONE = 0, ACT_SWISH = 1, ACT_RELU = 2, ACT_SIGMOID = 3, ACT_SWISH_D = 4 };

// ---- the deferred parameter-gradient queue (nnhipWeightGradDefer) ------------------------------------------------------------
// While deferral is on, a dW (+db) GEMM that would not fill the chip on its own is not launched by the backward call that asks
// for it: the job is queued and nnhipWeightGradFlush launches everything queued as ONE grid + ONE reduce (gemm.hip:
// gemm_f32_wgrad_group).  The caller keeps X and dO alive and unchanged until the flush.
static std::mutex g_wq_mu;
static std::vector<WgradJob> g_wq;
static int g_wq_on = 0;
bool wgrad_defer_on() { return g_wq_on != 0; }
static hipStream_t g_wq_stream = nullptr;
static int wq_flush_locked() {
    if (g_wq.empty()) return 0;
    const int rc = gemm_f32_wgrad_group(g_wq.data(), (int)g_wq.size(), g_wq_stream);
    g_wq.clear();
    return rc;
}
// true: queued (*rc: status of a flush this forced); false: not a candidate -- launch it yourself
static bool wq_offer(const WgradJob& j, hipStream_t st, int* rc) {
    std::lock_guard<std::mutex> lk(g_wq_mu);
    *rc = 0;
    if (!g_wq_on || !gemm_f32_wgrad_group_ok(j)) return false;
    if (!g_wq.empty() && g_wq_stream != st) *rc = wq_flush_locked();   // one queue, one stream
    g_wq_stream = st;
    g_wq.push_back(j);
    return true;
}

// dact_arg / dact (optional): dX = (dO W + addend) (.) act'(dact_arg) -- 1: swish'(z; beta), 2: relu mask [f > 0] (gemm.hip)
static int linear_backward(const float* X, const float* W, const float* dO, float* dX, float* dW,
                           float* db, int64_t rows, int64_t in, int64_t out, hipStream_t st, const float* dX_addend = nullptr,
                           const float* dact_arg = nullptr, int dact = 0, float beta = 1.f) {
    int rc = 0;
    if (rows == 0) {  // empty batch: sums over nothing are zero (X / dO / dX may be null pointers of empty arrays)
        if (dW) rc = fill_f32(dW, 0.f, out * in, st);
        if (!rc && db) rc = fill_f32(db, 0.f, out, st);
        return rc;
    }
    // a small layer: both gradients from ONE launch (a launch is ~4.7 us of the 45 us MNIST-MLP step whatever it computes)
    if (dX && dW && !(dX_addend && dact_arg)) {
        rc = gemm_f32_linear_backward_small(X, W, dO, dX, dW, db, rows, in, out, dX_addend, dact_arg, dact, beta, st);
        if (rc) return rc < 0 ? rc : 0;
    }
    // (Round 3, measured and dropped: running the parameter-gradient side -- dW GEMM, split-K reduce, db -- on a side stream
    //  next to the dX GEMM, fork/join through events.  Two concurrently dispatched full-grid GEMMs do not fill each other's
    //  tails on this part, they slow each other down: 16384x512->512 backward 148 -> 166 us, C4 step 22.24 -> 22.33 ms.)
    // dX[rows,in] = dO[rows,out] * W[out,in]         A k-major (k = out), B outer-major
    if (dX) {
        if (dact_arg && (dact == 1 || dact == 3)) rc = gemm_f32_dswish(dO, W, dX, dact_arg, beta, rows, in, out, out, in, in, true, false, st, dact);
        else if (dact_arg) rc = gemm_f32_drelu(dO, W, dX, dact_arg, rows, in, out, out, in, in, true, false, st);
        else rc = gemm_f32_add(dO, W, dX, nullptr, dX_addend, rows, in, out, out, in, in, true, false, st);
    }
    if (rc) return rc;
    // dW[out,in] = dO^T[out,rows] * X[rows,in]        both outer-major (k = rows)
    // db[out] = sum_rows dO.  For in_features <= 2048 it rides in the dW kernel (every thread sums the dO elements it
    // stages: ~4 % of that GEMM, cheaper than the two launches of a column-sum pass -- measured 9 vs 20 us at
    // 16384x512->512, 14 vs 33 us at 16384x512->2048); wider layers keep the separate HBM-bound pass (4096^2: 19 us vs +46).
    const bool fuse_db = dW && db && in <= 2048;
    if (dW) {
        if (!wq_offer(WgradJob{dO, X, dW, fuse_db ? db : nullptr, out, in, rows}, st, &rc))
            rc = gemm_f32_asum(dO, X, dW, fuse_db ? db : nullptr, out, in, rows, out, in, in, false, st);
    }
    if (rc) return rc;
    if (db && !fuse_db) rc = colsum(dO, rows, out, out, db, st);
    return rc;
}
}  // namespace nnhip

using namespace nnhip;

static int check_linear(const char* fn, const void* X, const void* W, int64_t rows, int64_t in, int64_t out) {
    NNHIP_CHECK_ARG(rows >= 0 && in >= 0 && out >= 0, NNHIP_EINVAL, "%s: negative size", fn);
    if (rows == 0) return 0;  // empty batch: pointers of empty arrays may be null
    NNHIP_CHECK_ARG(X && W, NNHIP_EINVAL, "%s: null X/W", fn);
    NNHIP_CHECK_ARG(aligned4(X) && aligned4(W), NNHIP_EALIGN, "%s: misaligned pointer", fn);
    return 0;
}

extern "C" int nnhipLinearModuleForward(const float* X, const float* W, const float* b, float* O,
                                        int64_t rows, int64_t in_features, int64_t out_features,
                                        nnhipStream_t stream) {
    return nnhipLinearModuleForwardEx(X, W, b, nullptr, O, rows, in_features, out_features, stream);
}

extern "C" int nnhipLinearModuleForwardEx(const float* X, const float* W, const float* b, const float* addend,
                                          float* O, int64_t rows, int64_t in_features, int64_t out_features,
                                          nnhipStream_t stream) {
    if (int rc = check_linear("nnhipLinearModuleForward", X, W, rows, in_features, out_features)) return rc;
    if (rows == 0) return 0;
    NNHIP_CHECK_ARG(O != nullptr, NNHIP_EINVAL, "nnhipLinearModuleForward: null output");
    NNHIP_CHECK_ARG(aligned4(addend), NNHIP_EALIGN, "nnhipLinearModuleForward: misaligned addend");
    return gemm_f32_add(X, W, O, b, addend, rows, out_features, in_features, in_features, in_features,
                        out_features, true, true, (hipStream_t)stream);
}

// Deferred parameter gradients (extension; the reference launches each layer's dW where its backward runs, linear.py:17-24).
// enable != 0: from now on the Linear backward entry points QUEUE their dW/db GEMMs when those are too small to fill the chip
// alone (everything else about the calls is unchanged: dX is computed at once) and nnhipWeightGradFlush(stream) launches the queue
// as one grid.  Until the flush the caller must keep the X and dO buffers of the queued calls alive and unmodified, and must not
// read dW/db.  enable == 0: flush what is queued on `stream`, then launch every later dW where it is asked for.  ABI 204
extern "C" int nnhipWeightGradDefer(int32_t enable, nnhipStream_t stream) {
    std::lock_guard<std::mutex> lk(g_wq_mu);
    int rc = 0;
    if (!enable) {
        if (!g_wq.empty()) g_wq_stream = (hipStream_t)stream;
        rc = wq_flush_locked();
        if (int rc2 = nnhip::conv_reduce_flush(stream)) rc = rc ? rc : rc2;
        if (int rc3 = nnhip::colsum_flush(stream)) rc = rc ? rc : rc3;
    }
    g_wq_on = enable ? 1 : 0;
    return rc;
}
extern "C" int nnhipWeightGradFlush(nnhipStream_t stream) {
    std::lock_guard<std::mutex> lk(g_wq_mu);
    if (!g_wq.empty() && g_wq_stream != (hipStream_t)stream) g_wq_stream = (hipStream_t)stream;
    int rc = wq_flush_locked();
    if (int rc2 = nnhip::conv_reduce_flush(stream)) rc = rc ? rc : rc2;
    if (int rc3 = nnhip::colsum_flush(stream)) rc = rc ? rc : rc3;
    return rc;
}
// Only the queued GEMMs (the grouped launch + its reduce), on `stream` -- which may be a side stream: the launch reads the queued
// calls' X / dO (the caller orders `stream` behind their producers and keeps them alive until it has joined `stream` again), writes
// its slabs to a block of its own and dW / db where the calls asked.  The conv reduces and RMSNorm column sums queued next to them
// stay for nnhipWeightGradFlush on the stream their producers run on.  ABI 210
extern "C" int nnhipWeightGradFlushGemms(nnhipStream_t stream) {
    std::lock_guard<std::mutex> lk(g_wq_mu);
    if (!g_wq.empty()) g_wq_stream = (hipStream_t)stream;
    return wq_flush_locked();
}
extern "C" int nnhipWeightGradPending(void) {
    std::lock_guard<std::mutex> lk(g_wq_mu);
    return (int)g_wq.size();
}

extern "C" int nnhipLinearModuleBackward(const float* X, const float* W, const float* dO, float* dX,
                                         float* dW, float* db, int64_t rows, int64_t in_features,
                                         int64_t out_features, nnhipStream_t stream) {
    return nnhipLinearModuleBackwardEx(X, W, dO, nullptr, dX, dW, db, rows, in_features, out_features, stream);
}

extern "C" int nnhipLinearModuleBackwardEx(const float* X, const float* W, const float* dO, const float* dX_addend,
                                           float* dX, float* dW, float* db, int64_t rows, int64_t in_features,
                                           int64_t out_features, nnhipStream_t stream) {
    if (int rc = check_linear("nnhipLinearModuleBackward", X, W, rows, in_features, out_features)) return rc;
    NNHIP_CHECK_ARG(rows == 0 || dO != nullptr, NNHIP_EINVAL, "nnhipLinearModuleBackward: null dO");
    NNHIP_CHECK_ARG(aligned4(dX_addend), NNHIP_EALIGN, "nnhipLinearModuleBackward: misaligned dX_addend");
    return linear_backward(X, W, dO, dX, dW, db, rows, in_features, out_features, (hipStream_t)stream, dX_addend);
}

extern "C" int nnhipLinearSwishForward(const float* X, const float* W, const float* b, float* O,
                                       float* preact, int64_t M, int64_t K, int64_t N, float swish_beta,
                                       int save_preactivation, nnhipStream_t stream) {
    if (int rc = check_linear("nnhipLinearSwishForward", X, W, M, K, N)) return rc;
    if (M == 0) return 0;
    NNHIP_CHECK_ARG(O != nullptr, NNHIP_EINVAL, "nnhipLinearSwishForward: null output");
    NNHIP_CHECK_ARG(!save_preactivation || preact, NNHIP_EINVAL,
                    "nnhipLinearSwishForward: save_preactivation set but preact is null");
    NNHIP_CHECK_ARG(save_preactivation >= 0 && save_preactivation <= 2, NNHIP_EINVAL,
                    "nnhipLinearSwishForward: save_preactivation must be 0, 1 (z) or 2 (swish'(z))");
    return gemm_f32(X, W, O, b, save_preactivation ? preact : nullptr, M, N, K, K, K, N, true, true, 1, 0,
                    0, 0, save_preactivation == 2 ? ACT_SWISH_D : ACT_SWISH, swish_beta, (hipStream_t)stream);
}

extern "C" int nnhipLinearSwishBackward(const float* X, const float* W, const float* b, const float* dO,
                                        float* tmp, float* dX, float* dW, float* db, int64_t M, int64_t K,
                                        int64_t N, float swish_beta, int recompute_preactivation,
                                        nnhipStream_t stream) {
    if (int rc = check_linear("nnhipLinearSwishBackward", X, W, M, K, N)) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (M == 0) return linear_backward(X, W, dO, dX, dW, db, M, K, N, st);
    NNHIP_CHECK_ARG(dO && tmp, NNHIP_EINVAL, "nnhipLinearSwishBackward: null dO/tmp");
    NNHIP_CHECK_ARG(recompute_preactivation >= 0 && recompute_preactivation <= 2, NNHIP_EINVAL,
                    "nnhipLinearSwishBackward: recompute_preactivation must be 0 (tmp = z), 1 (recompute z) or 2 (tmp = swish'(z))");
    int rc = 0;
    if (recompute_preactivation == 1)  // z = X W^T + b into tmp
        rc = gemm_f32(X, W, tmp, b, nullptr, M, N, K, K, K, N, true, true, 1, 0, 0, 0, ACT_NONE, 1.f, st);
    if (rc) return rc;
    if (recompute_preactivation == 2) rc = mul_f32(tmp, dO, tmp, M * N, st);   // tmp holds swish'(z) (forward mode 2): dZ = dO * tmp
    else rc = swish_backward_inplace(tmp, dO, swish_beta, M * N, st);  // tmp <- dZ = dO * swish'(z)
    if (rc) return rc;
    return linear_backward(X, W, tmp, dX, dW, db, M, K, N, st);
}

// dZ[rows,in] = (dO[rows,out] * W[out,in]) (.) swish'(Z[rows,in]; beta): the input gradient of a Linear fed by
// h = swish(z), with the Swish backward folded into the dX GEMM's epilogue.  dZ may alias Z.
extern "C" int nnhipLinearInputGradSwish(const float* dO, const float* W, const float* Z, float* dZ, int64_t rows,
                                         int64_t in_features, int64_t out_features, float swish_beta,
                                         nnhipStream_t stream) {
    NNHIP_CHECK_ARG(rows >= 0 && in_features >= 0 && out_features >= 0, NNHIP_EINVAL, "nnhipLinearInputGradSwish: negative size");
    if (rows == 0 || in_features == 0) return 0;
    NNHIP_CHECK_ARG(dO && W && Z && dZ, NNHIP_EINVAL, "nnhipLinearInputGradSwish: null pointer");
    NNHIP_CHECK_ARG(aligned4(dO) && aligned4(W) && aligned4(Z) && aligned4(dZ), NNHIP_EALIGN,
                    "nnhipLinearInputGradSwish: misaligned pointer");
    return gemm_f32_dswish(dO, W, dZ, Z, swish_beta, rows, in_features, out_features, out_features, in_features,
                           in_features, true, false, (hipStream_t)stream);
}

// The same with the derivative already in hand: dZ = (dO W) (.) D, D = the swish'(z) that nnhipLinearSwishForward(save_preactivation
// = 2) left in `preact`.  dZ may alias D.
extern "C" int nnhipLinearInputGradScaled(const float* dO, const float* W, const float* D, float* dZ, int64_t rows,
                                          int64_t in_features, int64_t out_features, nnhipStream_t stream) {
    NNHIP_CHECK_ARG(rows >= 0 && in_features >= 0 && out_features >= 0, NNHIP_EINVAL, "nnhipLinearInputGradScaled: negative size");
    if (rows == 0 || in_features == 0) return 0;
    NNHIP_CHECK_ARG(dO && W && D && dZ, NNHIP_EINVAL, "nnhipLinearInputGradScaled: null pointer");
    NNHIP_CHECK_ARG(aligned4(dO) && aligned4(W) && aligned4(D) && aligned4(dZ), NNHIP_EALIGN,
                    "nnhipLinearInputGradScaled: misaligned pointer");
    return gemm_f32_dswish(dO, W, dZ, D, 1.f, rows, in_features, out_features, out_features, in_features,
                           in_features, true, false, (hipStream_t)stream, 3);
}

// Backward of a Linear whose input was h = act(z) of the previous layer, in one call: dZ = (dO W) (.) act'(arg) -- act_grad 1:
// swish'(arg = z; beta), dZ may alias arg; 2: relu mask [arg = h > 0], dZ must not alias arg -- plus dW = dO^T X and db.  What
// nnhipLinearInputGradSwish / ReLU followed by nnhipLinearModuleBackward(dX = NULL) compute; small layers (the MNIST-MLP's
// 128 -> 10 head) get all three results from ONE launch.
extern "C" int nnhipLinearModuleBackwardAct(const float* X, const float* W, const float* dO, const float* act_arg,
                                            int32_t act_grad, float beta, float* dZ, float* dW, float* db, int64_t rows,
                                            int64_t in_features, int64_t out_features, nnhipStream_t stream) {
    if (int rc = check_linear("nnhipLinearModuleBackwardAct", X, W, rows, in_features, out_features)) return rc;
    NNHIP_CHECK_ARG(act_grad >= 1 && act_grad <= 3, NNHIP_EINVAL, "nnhipLinearModuleBackwardAct: act_grad must be 1 (swish), 2 (relu) or 3 (act_arg is the saved derivative)");
    NNHIP_CHECK_ARG(rows == 0 || (dO && act_arg && dZ), NNHIP_EINVAL, "nnhipLinearModuleBackwardAct: null pointer");
    NNHIP_CHECK_ARG(rows == 0 || act_grad != 2 || act_arg != dZ, NNHIP_EINVAL, "nnhipLinearModuleBackwardAct: dZ aliases the ReLU output");
    NNHIP_CHECK_ARG(aligned4(dO) && aligned4(act_arg) && aligned4(dZ) && aligned4(dW) && aligned4(db), NNHIP_EALIGN,
                    "nnhipLinearModuleBackwardAct: misaligned pointer");
    return linear_backward(X, W, dO, dZ, dW, db, rows, in_features, out_features, (hipStream_t)stream, nullptr, act_arg, act_grad, beta);
}

// O = act(X W^T + b) with the activation in the GEMM epilogue: activation 1 = swish(beta) (no pre-activation saved; use
// nnhipLinearSwishForward to keep z), 2 = relu, 3 = sigmoid.  What `act(Linear(x))` computes on the reference's tape
// (neunet/nn/layers/linear.py:48-58 followed by activations.py:54-56 / 221-233 / 20-25) in one launch.
extern "C" int nnhipLinearActivationForward(const float* X, const float* W, const float* b, float* O, int64_t rows,
                                            int64_t in_features, int64_t out_features, int32_t activation, float beta,
                                            nnhipStream_t stream) {
    if (int rc = check_linear("nnhipLinearActivationForward", X, W, rows, in_features, out_features)) return rc;
    NNHIP_CHECK_ARG(activation >= 1 && activation <= 3, NNHIP_EINVAL, "nnhipLinearActivationForward: activation must be 1 (swish), 2 (relu) or 3 (sigmoid)");
    if (rows == 0) return 0;
    NNHIP_CHECK_ARG(O != nullptr, NNHIP_EINVAL, "nnhipLinearActivationForward: null output");
    return gemm_f32(X, W, O, b, nullptr, rows, out_features, in_features, in_features, in_features, out_features, true, true,
                    1, 0, 0, 0, activation, beta, (hipStream_t)stream);
}

// dZ[rows,in] = (dO[rows,out] * W[out,in]) (.) [F[rows,in] > 0]: the input gradient of a Linear fed by h = relu(z), F = h,
// with the ReLU backward (neunet/nn/activations.py:44-45) folded into the dX GEMM's epilogue.  dZ must not alias F.
extern "C" int nnhipLinearInputGradReLU(const float* dO, const float* W, const float* F, float* dZ, int64_t rows,
                                        int64_t in_features, int64_t out_features, nnhipStream_t stream) {
    NNHIP_CHECK_ARG(rows >= 0 && in_features >= 0 && out_features >= 0, NNHIP_EINVAL, "nnhipLinearInputGradReLU: negative size");
    if (rows == 0 || in_features == 0) return 0;
    NNHIP_CHECK_ARG(dO && W && F && dZ && F != dZ, NNHIP_EINVAL, "nnhipLinearInputGradReLU: null or aliased pointer");
    NNHIP_CHECK_ARG(aligned4(dO) && aligned4(W) && aligned4(F) && aligned4(dZ), NNHIP_EALIGN,
                    "nnhipLinearInputGradReLU: misaligned pointer");
    return gemm_f32_drelu(dO, W, dZ, F, rows, in_features, out_features, out_features, in_features, in_features, true, false,
                          (hipStream_t)stream);
}

// Backward of out = Linear2(relu(Linear1(X1))) for an input X1 that needs no gradient (the first layer of a model): dW2, db2,
// dW1, db1 from dO, with the hidden gradient dZ = (dO W2) (.) [H > 0] formed inside the dW1 tiles and never written.  H = the
// ReLU output.  Small problems only (rows <= 256, out2 <= 16, both weight gradients within gemm_small's range): returns
// NNHIP_EINVAL otherwise -- callers fall back to nnhipLinearModuleBackwardAct + nnhipLinearModuleBackward, which compute the
// same four tensors (dW1 / db1 up to the summation order of dZ's dot products).
extern "C" int nnhipLinearReLULinearBackward(const float* X1, const float* H, const float* W2, const float* dO, float* dW2,
                                             float* db2, float* dW1, float* db1, int64_t rows, int64_t in1, int64_t hidden,
                                             int64_t out2, nnhipStream_t stream) {
    NNHIP_CHECK_ARG(rows >= 0 && in1 > 0 && hidden > 0 && out2 > 0, NNHIP_EINVAL, "nnhipLinearReLULinearBackward: bad sizes");
    NNHIP_CHECK_ARG(dW2 && db2 && dW1 && db1, NNHIP_EINVAL, "nnhipLinearReLULinearBackward: null output");
    NNHIP_CHECK_ARG(rows == 0 || (X1 && H && W2 && dO), NNHIP_EINVAL, "nnhipLinearReLULinearBackward: null input");
    NNHIP_CHECK_ARG(aligned4(X1) && aligned4(H) && aligned4(W2) && aligned4(dO) && aligned4(dW2) && aligned4(db2) && aligned4(dW1) &&
                        aligned4(db1), NNHIP_EALIGN, "nnhipLinearReLULinearBackward: misaligned pointer");
    hipStream_t st = (hipStream_t)stream;
    if (rows == 0) {
        int rc = fill_f32(dW2, 0.f, out2 * hidden, st);
        if (!rc) rc = fill_f32(db2, 0.f, out2, st);
        if (!rc) rc = fill_f32(dW1, 0.f, hidden * in1, st);
        if (!rc) rc = fill_f32(db1, 0.f, hidden, st);
        return rc;
    }
    const int rc = gemm_small_mlp_backward(X1, H, W2, dO, dW2, db2, dW1, db1, rows, in1, hidden, out2, st, nullptr);
    if (rc < 0) return rc;
    NNHIP_CHECK_ARG(rc == 1, NNHIP_EINVAL, "nnhipLinearReLULinearBackward: outside the small-problem range (rows <= 256, out2 <= 16)");
    return 0;
}

// 1 when nnhipLinearReLULinearBackward (with_adam 0) / ...BackwardAdam (with_adam 1) will take these sizes on the current device,
// 0 when they would return NNHIP_EINVAL: lets a caller that defers the launch decide before it gives up the per-layer path.
extern "C" int nnhipLinearReLULinearBackwardFits(int64_t rows, int64_t in1, int64_t hidden, int64_t out2, int32_t with_adam) {
    if (rows <= 0 || in1 <= 0 || hidden <= 0 || out2 <= 0) return 0;
    return gemm_small_mlp_fits(rows, in1, hidden, out2, with_adam != 0);
}

// The same backward with the optimizer inside ("optimizer in backward"): every gradient element is handed to Adam / AdamW
// (neunet/optim.py:17-33, 52-69) by the thread that produced it, so the README-MLP step needs no optimizer launch.  The
// gradients are still written (param.grad semantics).  pmv: 12 device pointers, {param, m, v} for W2, b2, W1, b1 in that order.
// step >= 1: host-side stepping (the caller's optimizer.t + 1); step == 0: device-side stepping through `opt` (a handle from
// nnhipCreateFusedOptimizer with SetStep / SetHyper done), which is what a captured hipGraph needs.  Hyper-parameters as in
// nnhipFusedAdamWMultiTensorStep.  rows >= 1 (an empty batch has nothing to fuse: call the optimizer).
extern "C" int nnhipLinearReLULinearBackwardAdam(const float* X1, const float* H, const float* W2, const float* dO, float* dW2,
                                                 float* db2, float* dW1, float* db1, int64_t rows, int64_t in1, int64_t hidden,
                                                 int64_t out2, void* opt, float* const* pmv, double lr, double beta1, double beta2,
                                                 double eps, double weight_decay, int32_t step, int32_t decay_mode,
                                                 float grad_scale, nnhipStream_t stream) {
    NNHIP_CHECK_ARG(rows >= 1 && in1 > 0 && hidden > 0 && out2 > 0 && step >= 0, NNHIP_EINVAL, "nnhipLinearReLULinearBackwardAdam: bad sizes");
    NNHIP_CHECK_ARG(decay_mode == 0 || decay_mode == 1, NNHIP_EINVAL, "nnhipLinearReLULinearBackwardAdam: decay_mode must be 0 or 1");
    NNHIP_CHECK_ARG(X1 && H && W2 && dO && dW2 && db2 && dW1 && db1 && pmv, NNHIP_EINVAL, "nnhipLinearReLULinearBackwardAdam: null pointer");
    for (int i = 0; i < 12; ++i) NNHIP_CHECK_ARG(pmv[i] != nullptr && aligned4(pmv[i]), NNHIP_EINVAL, "nnhipLinearReLULinearBackwardAdam: null / misaligned optimizer tensor");
    NNHIP_CHECK_ARG(pmv[0] == W2, NNHIP_EINVAL, "nnhipLinearReLULinearBackwardAdam: pmv[0] must be W2 (order: W2, b2, W1, b1)");
    const int rc = gemm_small_mlp_backward_adam(X1, H, W2, dO, dW2, db2, dW1, db1, rows, in1, hidden, out2, opt, pmv, lr, beta1, beta2, eps,
                                                weight_decay, step, decay_mode, grad_scale, (hipStream_t)stream);
    if (rc < 0) return rc;
    NNHIP_CHECK_ARG(rc == 1, NNHIP_EINVAL, "nnhipLinearReLULinearBackwardAdam: outside the small-problem range (rows <= 256, out2 <= 16), or a hidden "
                    "layer so wide that its polling blocks would fill the chip: run nnhipLinearReLULinearBackward + the optimizer step instead");
    return 0;
}
