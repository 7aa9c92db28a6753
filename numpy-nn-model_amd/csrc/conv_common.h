// conv_common.h -- what conv2d.hip (entry points, direct small-channel kernels) and conv_mfma.hip (MFMA implicit GEMM) share.
#pragma once
#include "common.h"

namespace nnhip {

struct ConvGeom {
    int B, Cin, H, W, Cout, kh, kw, sh, sw, dh, dw, pu, pl, Ho, Wo;
};

// conv_mfma.hip: implicit-GEMM Conv2d on the fp32 MFMA with the Linear GEMM's pipeline.
//   forward : O  = conv(X, W) + bias          (bias may be null)
//   dgrad   : dX = conv^T(dO, W)
//   wgrad   : dW (and/or db) from X, dO       (either may be null)
// Every tensor below 2 GiB (32-bit byte offsets); otherwise NNHIP_EINVAL.  All three take any stride / dilation / padding.
int conv_mfma_forward(const float* X, const float* W, const float* bias, float* O, const ConvGeom& g, hipStream_t st);
int conv_mfma_dgrad(const float* dO, const float* W, float* dX, const ConvGeom& g, hipStream_t st);
int conv_mfma_wgrad(const float* X, const float* dO, float* dW, float* db, const ConvGeom& g, hipStream_t st);

}  // namespace nnhip
