// Internal launch helpers shared by nets.hip / prior.hip (not part of the C ABI).
#pragma once
#include "egx_common.h"

struct EgxSeg {
  const float* p;
  int w, ld;
};

// out = act(cat(segs) W^T + b) + res      (W: [N,K] torch layout)
int egx_launch_linear(hipStream_t st, int M, int N, const EgxSeg* segs, int nseg, const float* W, const float* b,
                      int act, float slope, const float* res, int ldr, float* out, int ldo);
// one GEMM description for the paired launcher; ldw = 0 means "K" (weights may also be a column block of a wider matrix)
struct EgxLin {
  int M, N;
  EgxSeg segs[4];
  int nseg;
  const float* W;
  int ldw;
  const float* b;
  int act;
  float slope;
  const float* res;
  int ldr;
  float* out;
  int ldo;
  int bf16 = 0;  // 1: operands rounded to bf16, products on v_mfma_f32_32x32x16_bf16, fp32 accumulate (config-5 policy inference)
};
int egx_launch_linear_pair(hipStream_t st, const EgxLin& A, const EgxLin& B);
int egx_launch_gru_pointwise(hipStream_t st, const float* gi, const float* gh, const float* hprev, int ldh, float* hout,
                             int ldo, int M, int H);
int egx_launch_gru_pointwise_first(hipStream_t st, const float* gi, const float* b_hh, float* hout, int ldo, int M, int H);
int egx_launch_cont6d_to_aa(hipStream_t st, const float* xb6, int n, float* out, int ldo);
int egx_launch_posenc(hipStream_t st, const float* dist, const float* time, int A, float* out);

struct RegWeights {
  const float* in_w; const float* in_b;
  const float* blk_w[20]; const float* blk_b[20];
  const float* out_w; const float* out_b;
  // optional copies of the weights in MFMA operand lane order (egx_prior_weights.reg_packed_*), all three or none
  const f32x4* pk_in;    // [4 column groups][47 chunks][64 lanes] float4   (K 370 zero-padded to 376)
  const f32x4* pk_blk;   // [20 layers][4][16][64]
  const f32x4* pk_out;   // [5][16][64]                                    (N 159 padded to 160 by repeating the last row)
};
int egx_launch_regressor_fused(hipStream_t st, const RegWeights& w, const float* Y, const float* betas, int A, int M,
                               float* out_Yb);
int egx_launch_linear_one(hipStream_t st, const EgxLin& A);  // honours EgxLin::bf16 (always the 32x32 split-K kernel)
// y[t][a][c] += y[t-1][a][c] for t = 0..T-1 with y[-1] = x_last[a][c] (row stride x_ld): residual chain of the decoder
void egx_launch_frame_scan(hipStream_t st, float* y, const float* x_last, int x_ld, int A, int width, int T);
