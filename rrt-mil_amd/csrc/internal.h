// internal.h -- host-side launch functions shared between the kernel files and api.cpp
#pragma once
#include <hip/hip_runtime.h>
#include "common.h"
#include "../../include/rrt_hip.h"

GridDev to_dev(const rrt_grid& g);

hipError_t launch_ln_partition(const float* x, const float* gamma, const float* beta, float* u,
                               int dim, const GridDev& g, hipStream_t st);

struct LinearEpilogue {
  int prec;              // MFMA operand precision: 0 f32 (exact), 1 bf16, 2 f16 (fp32 accumulate)
  const float* bias;     // [N] or null
  int q_cols;            // columns [0,q_cols) scaled by q_scale after bias
  float q_scale;
  // un-partition + residual (used when resid != null): C row = token, A row = slot
  const float* resid;
  GridDev g;
};
hipError_t launch_linear(const float* A, const float* B, float* C, int M, int N, int K,
                         const LinearEpilogue& ep, hipStream_t st);

hipError_t launch_region_attention(const float* qkv, const float* pe_w, float* o, int n_regions,
                                   int P, int dim, int heads, int epeg_k, hipStream_t st);

// fused qkv projection + EPEG + attention per (region, head) (rmsa_fused.hip); P in (112,144], head dim 64
bool rmsa_fused_supported(int P, int D, int heads, int epeg_k);
bool rmsa_fused_supported_rows(long n_rows, int D);
hipError_t launch_rmsa_fused(const float* U, const float* Wqkv, const float* bqkv, const float* pe_w,
                             float* O, int n_regions, int P, int D, int heads, int epeg_k, int prec,
                             hipStream_t st);

hipError_t launch_crmsa_logits(const float* x1, const float* gamma, const float* beta,
                               const float* phi, float* mean_rstd, float* logits, int dim, int k,
                               const GridDev& g8, hipStream_t st);
hipError_t launch_crmsa_combine(const float* x1, const float* gamma, const float* beta,
                                const float* mean_rstd, const float* logits, float* wdisp,
                                float* rep, int dim, int k, const GridDev& g8, hipStream_t st);
// mean_rstd == nullptr: x1 is LN(x1) already, region-major [Np8, dim] (crmsa_mlp path)
hipError_t launch_crmsa_mlp_logits(const float* hid, const float* w2, float* logits, int rows, int hdim,
                                   int k, hipStream_t st);
hipError_t launch_crmsa_dispatch_ln(const float* x1, const float* x0, const float* wdisp,
                                    const float* rep2, const float* gamma,
                                    const float* beta, float* y, int dim, int k, const GridDev& g8,
                                    hipStream_t st);
hipError_t launch_layernorm(const float* x1, const float* x0, const float* gamma,
                            const float* beta, float* y, int L, int dim, hipStream_t st);
