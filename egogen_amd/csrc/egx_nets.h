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
int egx_launch_gru_pointwise(hipStream_t st, const float* gi, const float* gh, const float* hprev, int ldh, float* hout,
                             int ldo, int M, int H);
int egx_launch_cont6d_to_aa(hipStream_t st, const float* xb6, int n, float* out, int ldo);
int egx_launch_posenc(hipStream_t st, const float* dist, const float* time, int A, float* out);

// y[t][a][c] += y[t-1][a][c] for t = 0..T-1 with y[-1] = x_last[a][c] (row stride x_ld): residual chain of the decoder
void egx_launch_frame_scan(hipStream_t st, float* y, const float* x_last, int x_ld, int A, int width, int T);

// ---- dense3.hip: dense layers on the bf16 matrix pipe, operands as three bf16 planes in MFMA fragment order -------------
// Packed image of a row-major fp32 matrix [R, K]: [2 ceil(R/32) row tiles of 16][K/32 k-steps][3 planes][64 lanes] x 16 bytes.
struct D3Pack {
  const float* src;
  int R, K, ld, col0;   // rows, columns taken, leading dimension, first column
  void* dst;
  int S_total, s0;      // k-steps per row tile of the destination buffer, k-step this image starts at
  int transpose = 0;    // 1: src is [K rows (reduction index), R columns] - the image is that of the transposed block
};
void egx_launch_pack3(hipStream_t st, const D3Pack* jobs, int njobs);  // njobs <= 4, one launch

// out = act(A B^T + bias) + res.  A: packed activations (k-steps sa0 .. sa0 + S of a buffer with SA k-steps per row tile);
// B: packed weights [N, K = 32 S].  Outputs: fp32 row-major `out` and / or packed `out3` (the consumer's A operand: this
// layer's 32-column tile nt is k-step s30 + nt of a buffer with S3 k-steps per row tile; needs N % 32 == 0).  `batches`
// > 1: the same layer for several row blocks (A advances by batch_strideA fragments, out3 by batch_stride3, fp32 rows by
// batch_rows_out).
struct D3Plain {
  const bf16x8* A = nullptr;
  int SA = 0, sa0 = 0;
  const bf16x8* B = nullptr;
  int S = 0;
  const float* bias = nullptr;
  const float* res = nullptr;
  int ldr = 0;
  float* out = nullptr;
  int ldo = 0;
  bf16x8* out3 = nullptr;
  int S3 = 0, s30 = 0;
  int M = 0, N = 0, act = 0;
  float slope = 0.f;
  int batches = 1;
  size_t batch_strideA = 0, batch_stride3 = 0;
  int batch_rows_out = 0;
  int prec = 0;   // 0: three planes, six partial products (fp32-equivalent); 2: two planes, three products (16 operand bits);
                  // 1: leading plane only (operands rounded to bf16).  See dense3.hip::d3_planes.
  // ---- training (update3.hip)
  float* out_act = nullptr;       // fp32 [M, N] (ld ldact): the activation BEFORE the residual is added (saved for backward)
  int ldact = 0;
  const float* dact = nullptr;    // gradient launches: the packed outputs are v x act'(dact[m][n]) (fp32 `out` stays v)
  int lddact = 0, dact_code = 0;
  float dact_slope = 0.f;
  bf16x8* out3T = nullptr;        // transposed packed output: image rows = this layer's columns, reduction index = its rows
  int S3T = 0, s3T0 = 0;          //   (k-step s3T0 + row tile / 2 of a buffer with S3T k-steps per row tile)
  float* bias_out = nullptr;      // n_split > 0: columns >= n_split are not stored to `out`; column n_split goes to bias_out[m]
  int n_split = 0;
};
void egx_launch_dense3_n(hipStream_t st, const D3Plain* ps, int n);   // up to four independent layers, one launch
void egx_launch_dense3(hipStream_t st, const D3Plain& p);
void egx_launch_dense3_pair(hipStream_t st, const D3Plain& p, const D3Plain& q);  // two independent layers, one launch
void egx_launch_dense3_triple(hipStream_t st, const D3Plain& p, const D3Plain& q, const D3Plain& r);
// egx_ppo_loss_packed without its clearing launch (ppo.hip): the caller has cleared out_terms earlier on the same stream
int egx_ppo_loss_packed_precleared(const float* zp, const float* value, const float* act, const float* adv, const float* ret,
                                   const float* logp_old, const float* adv_stats, const float* scale, float adv_eps, float min_logvar,
                                   float max_logvar, float eps_clip, float vf_coef, float ent_coef, int num_rows, float* g_zp,
                                   float* g_value, float* out_terms, void* stream_);
void egx_launch_posenc3(hipStream_t st, const float* dist, const float* time, int n, float* out, int ld, void* out3, int S3, int s0,
                        void* out3T = nullptr, int S3T = 0, int col0T = 0, float* zero6 = nullptr /* six floats cleared by the launch */);

// One GRU cell step (gate order r, z, n; weights [3H, K] packed): see egx_gru3_kernel.
struct D3Gru {
  const bf16x8* Ai = nullptr;   // x side
  int SAi = 0, sai0 = 0;
  const bf16x8* Bi = nullptr;
  int Si = 0;
  const float* bias_i = nullptr;
  const float* gi_in = nullptr;   // [M, 3H] added to the x-side product (null = 0)
  float* gi_out = nullptr;        // [M, 3H] or null (may alias gi_in)
  const bf16x8* Ah = nullptr;   // h side (null: zero previous state)
  int SAh = 0, sah0 = 0;
  const bf16x8* Bh = nullptr;
  int Sh = 0;
  const float* bias_h = nullptr;
  const float* h_prev = nullptr;  // fp32 [M, H] (ld ldh) or null
  int ldh = 0;
  float* h_out = nullptr;         // fp32 [M, H] (ld ldo) or null
  int ldo = 0;
  bf16x8* h_out3 = nullptr;       // packed, k-steps s30 .. s30 + H / 32 of a buffer with S3 per row tile, or null
  int S3 = 0, s30 = 0;
  int M = 0, H = 0;
  int prec = 0;   // as D3Plain::prec
  // ---- training
  float* gh_out = nullptr;        // [M, 3H]: the h-side pre-activations (saved for backward)
  bf16x8* h_out3T = nullptr;      // transposed packed h: image rows col0T .. col0T + H, reduction index = batch rows
  int S3T = 0, s3T0 = 0, col0T = 0;
};
int egx_launch_gru3(hipStream_t st, const D3Gru& g);
int egx_launch_gru3_pair(hipStream_t st, const D3Gru& g0, const D3Gru& g1);   // two cells (same launch)

// Fused regressor on packed weights (dense3.hip): in_fc split into its marker / xb / betas column blocks
struct RegWeights3 {
  const bf16x8 *in_m, *in_xb, *in_b3;   // packed in_fc[:, 0:201] (7 k-steps), [:, 201:360] (5), [:, 360:370] (1)
  const float* in_b;
  const bf16x8* blk;                    // [20 layers] packed [128,128]
  const float* blk_b;                   // [20][128]
  const bf16x8* out;                    // packed [159 -> 160,128]
  const float* out_b;
};
int egx_launch_regressor3(hipStream_t st, const RegWeights3& w, const float* Y, const float* betas, int A, int M, float* out_Yb);

// Fused VPoser encoder mean on packed weights (dense3.hip): fc1 [512,63] (2 k-steps), fc2 [512,512] (16), mu [32,512] (16)
struct VpWeights3 {
  const bf16x8 *fc1, *fc2, *mu;
  const float *b1, *b2, *bmu;
};
int egx_launch_vposer3(hipStream_t st, const VpWeights3& w, const float* x, int x_ld, int n, float* out);

// MoshRegressor tail (models_GAMMA_primitive.py:208-219 + baseops.py:119-162): xb6[159] -> xb[93] =
// transl3 | 22 x (6D -> Gram-Schmidt rotmat -> axis-angle) | hands 24.  One call per (row, item j in 0..22).
__device__ __forceinline__ void egx_cont6d_item(const float* src, float* dst, int j) {
  if (j == 22) {  // transl + hands copied through
    dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2];
    for (int e = 0; e < 24; ++e) dst[69 + e] = src[135 + e];
    return;
  }
  const float* a = src + 3 + 6 * j;  // viewed as (3,2): a1 = (a[0],a[2],a[4]), a2 = (a[1],a[3],a[5])
  float b1[3] = {a[0], a[2], a[4]}, a2[3] = {a[1], a[3], a[5]};
  const float n1 = fmaxf(sqrtf(b1[0] * b1[0] + b1[1] * b1[1] + b1[2] * b1[2]), 1e-12f);
  b1[0] /= n1; b1[1] /= n1; b1[2] /= n1;
  const float d = b1[0] * a2[0] + b1[1] * a2[1] + b1[2] * a2[2];
  float b2[3] = {a2[0] - d * b1[0], a2[1] - d * b1[1], a2[2] - d * b1[2]};
  const float n2 = fmaxf(sqrtf(b2[0] * b2[0] + b2[1] * b2[1] + b2[2] * b2[2]), 1e-12f);
  b2[0] /= n2; b2[1] /= n2; b2[2] /= n2;
  const float b3[3] = {b1[1] * b2[2] - b1[2] * b2[1], b1[2] * b2[0] - b1[0] * b2[2], b1[0] * b2[1] - b1[1] * b2[0]};
  const float R[9] = {b1[0], b2[0], b3[0], b1[1], b2[1], b3[1], b1[2], b2[2], b3[2]};  // columns b1,b2,b3
  float aa[3];
  egx_tgm_rotmat_to_aa(R, aa);
  dst[3 + 3 * j + 0] = aa[0]; dst[3 + 3 * j + 1] = aa[1]; dst[3 + 3 * j + 2] = aa[2];
}

