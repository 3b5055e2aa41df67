// Dense-layer kernels for the rollout networks (C-VAE marker predictor, body regressor, VPoser encoder,
// policy) on gfx950.  The layers are tiny (M = agents or agents*18 rows, K,N <= 1536), so the design goal is
// latency: one 32x32 output tile per workgroup, K split across the 4 waves of the workgroup (each wave runs
// an independent v_mfma_f32_32x32x2_f32 chain over a quarter of K straight from global memory, 16-byte
// loads along K for both operands), partial tiles reduced through LDS, and bias / activation / residual
// fused into the epilogue.  Inputs may be the concatenation of up to 4 row-major segments (the reference
// builds them with torch.cat: models_GAMMA_primitive.py:93,257; models_policy_ppo.py:305) - no copy is made.
// Weights are read in torch's own [N,K] layout so the same storage serves the autograd update path.
#include "egx_nets.h"

namespace {

struct Seg {
  const float* p;
  int w;   // width (columns)
  int ld;  // leading dimension (floats)
};

struct LinArgs {
  Seg x[4];
  int nseg;
  const float* W;   // [N,K], ld = ldw
  int ldw;
  const float* bias;  // [N] or null
  const float* res;   // [M,N] residual added after the activation, or null
  int ldr;
  float* y;
  int ldy;
  int M, N, K;
  int act;      // 0 none, 1 tanh, 2 relu, 3 leaky relu
  float slope;
};

typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));

__device__ __forceinline__ float seg_load1(const LinArgs& a, int row, int k) {
  int kk = k;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    if (s < a.nseg) {
      if (kk < a.x[s].w) return a.x[s].p[(size_t)row * a.x[s].ld + kk];
      kk -= a.x[s].w;
    }
  }
  return 0.f;
}

__device__ __forceinline__ f32x4 load_x4(const LinArgs& a, int row, int k) {
  int kk = k;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    if (s < a.nseg) {
      if (kk + 3 < a.x[s].w) return *reinterpret_cast<const f32x4u*>(a.x[s].p + (size_t)row * a.x[s].ld + kk);
      if (kk < a.x[s].w) break;  // straddles the end of this segment
      kk -= a.x[s].w;
    }
  }
  f32x4 v;
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = (k + e < a.K) ? seg_load1(a, row, k + e) : 0.f;
  return v;
}

__device__ __forceinline__ f32x4 load_w4(const LinArgs& a, int col, int k) {
  const float* p = a.W + (size_t)col * a.ldw + k;
  if (k + 3 < a.K) return *reinterpret_cast<const f32x4u*>(p);
  f32x4 v;
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = (k + e < a.K) ? p[e] : 0.f;
  return v;
}

__device__ __forceinline__ float apply_act(float v, int act, float slope) {
  switch (act) {
    case 1: return tanhf(v);
    case 2: return fmaxf(v, 0.f);
    case 3: return v > 0.f ? v : v * slope;
    default: return v;
  }
}

}  // namespace

__global__ __launch_bounds__(256) void egx_linear_kernel(LinArgs a) {
  __shared__ float red[3][16 * 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 31, h = lane >> 5;
  const int m0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
  const int row = min(m0 + i, a.M - 1), col = min(n0 + i, a.N - 1);
  const int nchunk = (a.K + 7) >> 3;
  const int per = (nchunk + 3) >> 2;
  const int c0 = wave * per, c1 = min(nchunk, c0 + per);
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  if (c0 < c1) {
    f32x4 xa = load_x4(a, row, c0 * 8 + 4 * h), wb = load_w4(a, col, c0 * 8 + 4 * h);
    for (int c = c0; c < c1; ++c) {
      f32x4 xn = xa, wn = wb;
      if (c + 1 < c1) {
        xn = load_x4(a, row, (c + 1) * 8 + 4 * h);
        wn = load_w4(a, col, (c + 1) * 8 + 4 * h);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[e], wb[e], acc, 0, 0, 0);
      xa = xn;
      wb = wn;
    }
  }
  if (wave > 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave - 1][r * 64 + lane] = acc[r];
  }
  __syncthreads();
  if (wave == 0) {
    const int n = n0 + i;
    const float b = (a.bias && n < a.N) ? a.bias[n] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + (r & 3) + 8 * (r >> 2) + 4 * h;
      float v = acc[r] + red[0][r * 64 + lane] + red[1][r * 64 + lane] + red[2][r * 64 + lane] + b;
      v = apply_act(v, a.act, a.slope);
      if (m < a.M && n < a.N) {
        if (a.res) v += a.res[(size_t)m * a.ldr + n];
        a.y[(size_t)m * a.ldy + n] = v;
      }
    }
  }
}

// GRU gate math (torch.nn.GRU / GRUCell, gate order r,z,n): gi = x W_ih^T + b_ih, gh = h W_hh^T + b_hh
__global__ void egx_gru_pointwise_kernel(const float* __restrict__ gi, const float* __restrict__ gh,
                                         const float* __restrict__ hprev, int ldh, float* __restrict__ hout, int ldo,
                                         int M, int H) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= M * H) return;
  const int m = idx / H, c = idx % H;
  const float* gim = gi + (size_t)m * 3 * H;
  const float* ghm = gh + (size_t)m * 3 * H;
  const float r = 1.f / (1.f + expf(-(gim[c] + ghm[c])));
  const float z = 1.f / (1.f + expf(-(gim[H + c] + ghm[H + c])));
  const float nn = tanhf(gim[2 * H + c] + r * ghm[2 * H + c]);
  const float hp = hprev ? hprev[(size_t)m * ldh + c] : 0.f;
  hout[(size_t)m * ldo + c] = (1.f - z) * nn + z * hp;
}

// MoshRegressor tail (models_GAMMA_primitive.py:208-219 + baseops.py:119-162): xb6[n,159] ->
// xb[n,93] = transl3 | 22 x (6D -> Gram-Schmidt rotmat -> axis-angle) | hands 24.  One thread per (row, joint).
__global__ void egx_cont6d_to_aa_kernel(const float* __restrict__ xb6, int n, float* __restrict__ out, int ldo) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n * 23) return;
  const int row = idx / 23, j = idx % 23;
  const float* src = xb6 + (size_t)row * 159;
  float* dst = out + (size_t)row * ldo;
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

// positional_encoding (models_policy_ppo.py:276-285) of dist and time: out[b, 0:64] / out[b, 64:128]
__global__ void egx_posenc_kernel(const float* __restrict__ dist, const float* __restrict__ time, int A,
                                  float* __restrict__ out) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= A * 128) return;
  const int b = idx >> 7, c = idx & 127;
  const float x = (c < 64) ? dist[b] : time[b];
  const int k = (c & 63) >> 1;
  const float f = x * exp2f((float)k);
  out[idx] = (c & 1) ? cosf(f) : sinf(f);
}

// GAMMAPPOPolicy.forward tail (crowd_ppo/ppo_policy.py:169-178): clamp logvar, sigma = exp(logvar)^0.5,
// act = mu + sigma * eps (Normal.sample with injected standard-normal noise), log-prob of Independent(Normal,1).
__global__ void egx_sample_action_kernel(const float* __restrict__ mu, float* __restrict__ logvar,
                                         const float* __restrict__ eps, float min_lv, float max_lv, int deterministic,
                                         int n, float* __restrict__ act, float* __restrict__ logp) {
  const int row = blockIdx.x;  // one 128-thread block per row
  const int c = threadIdx.x;
  __shared__ float sh[128];
  const size_t i = (size_t)row * 128 + c;
  const float lv = fminf(fmaxf(logvar[i], min_lv), max_lv);
  logvar[i] = lv;
  const float sigma = sqrtf(expf(lv));
  const float e = deterministic ? 0.f : eps[i];
  const float a = mu[i] + sigma * e;
  act[i] = a;
  const float d = a - mu[i];
  sh[c] = -(d * d) / (2.f * sigma * sigma) - logf(sigma) - 0.91893853320467274178f;
  __syncthreads();
  for (int s = 64; s > 0; s >>= 1) {
    if (c < s) sh[c] += sh[c + s];
    __syncthreads();
  }
  if (c == 0 && logp) logp[row] = sh[0];
}

// GAE (ppo_policy.py:105-140 -> tianshou _gae_return [upstream]): values v[n+1,A] of obs_0..obs_n, rew / term [n,A]
// (time-major).  v_s_ = v[t+1] * (1 - terminated); end_flag = terminated, forced at the last stored step of every
// env (the sub-buffer's unfinished tail); float64 scan like tianshou's numpy path.
__global__ void egx_gae_kernel(const float* __restrict__ v, const float* __restrict__ rew, const int* __restrict__ term,
                               int n, int A, double gamma, double lam, float* __restrict__ returns, float* __restrict__ adv) {
  const int a = blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= A) return;
  double gae = 0.0;
  for (int t = n - 1; t >= 0; --t) {
    const bool tm = term[(size_t)t * A + a] != 0;
    const double vs = v[(size_t)t * A + a];
    const double vn = tm ? 0.0 : (double)v[(size_t)(t + 1) * A + a];
    const double delta = (double)rew[(size_t)t * A + a] + vn * gamma - vs;
    const bool end = tm || (t == n - 1);
    gae = delta + (end ? 0.0 : gamma * lam) * gae;
    adv[(size_t)t * A + a] = (float)gae;
    returns[(size_t)t * A + a] = (float)(gae + vs);
  }
}

// ---- internal launchers ---------------------------------------------------------------------------
int egx_launch_linear(hipStream_t st, int M, int N, const EgxSeg* segs, int nseg, const float* W, const float* b,
                      int act, float slope, const float* res, int ldr, float* out, int ldo) {
  LinArgs a;
  a.nseg = nseg;
  int K = 0;
  for (int s = 0; s < 4; ++s) {
    a.x[s].p = s < nseg ? segs[s].p : nullptr;
    a.x[s].w = s < nseg ? segs[s].w : 0;
    a.x[s].ld = s < nseg ? segs[s].ld : 0;
    if (s < nseg) K += segs[s].w;
  }
  a.W = W; a.ldw = K; a.bias = b; a.res = res; a.ldr = ldr; a.y = out; a.ldy = ldo;
  a.M = M; a.N = N; a.K = K; a.act = act; a.slope = slope;
  dim3 grid(egx_ceil_div(M, 32), egx_ceil_div(N, 32));
  hipLaunchKernelGGL(egx_linear_kernel, grid, dim3(256), 0, st, a);
  return EGX_OK;
}

int egx_launch_gru_pointwise(hipStream_t st, const float* gi, const float* gh, const float* hprev, int ldh, float* hout,
                             int ldo, int M, int H) {
  hipLaunchKernelGGL(egx_gru_pointwise_kernel, dim3(egx_ceil_div(M * H, 256)), dim3(256), 0, st, gi, gh, hprev, ldh, hout,
                     ldo, M, H);
  return EGX_OK;
}

int egx_launch_cont6d_to_aa(hipStream_t st, const float* xb6, int n, float* out, int ldo) {
  hipLaunchKernelGGL(egx_cont6d_to_aa_kernel, dim3(egx_ceil_div(n * 23, 256)), dim3(256), 0, st, xb6, n, out, ldo);
  return EGX_OK;
}

int egx_launch_posenc(hipStream_t st, const float* dist, const float* time, int A, float* out) {
  hipLaunchKernelGGL(egx_posenc_kernel, dim3(egx_ceil_div(A * 128, 256)), dim3(256), 0, st, dist, time, A, out);
  return EGX_OK;
}

// ---- C ABI -------------------------------------------------------------------------------------------
extern "C" int egx_linear(const egx_linear_desc* d, void* stream_) {
  EGX_REQUIRE(d, "null descriptor");
  EGX_REQUIRE(d->num_rows > 0 && d->out_features > 0 && d->num_segments >= 1 && d->num_segments <= 4, "bad sizes");
  EGX_REQUIRE(d->weight && d->out, "null weight/out");
  EgxSeg segs[4];
  int K = 0;
  for (int s = 0; s < d->num_segments; ++s) {
    segs[s] = {d->seg_ptr[s], d->seg_width[s], d->seg_ld[s]};
    EGX_REQUIRE(segs[s].p && segs[s].w > 0 && segs[s].ld >= segs[s].w, "bad input segment");
    K += segs[s].w;
  }
  EGX_REQUIRE(d->weight_ld == 0 || d->weight_ld == K, "weight_ld other than K is not supported");
  EGX_REQUIRE(d->activation >= 0 && d->activation <= 3, "unknown activation");
  egx_launch_linear(static_cast<hipStream_t>(stream_), d->num_rows, d->out_features, segs, d->num_segments, d->weight,
                    d->bias, d->activation, d->leaky_slope, d->residual, d->residual_ld, d->out,
                    d->out_ld > 0 ? d->out_ld : d->out_features);
  EGX_HIP_CHECK(hipGetLastError());
  return EGX_OK;
}

extern "C" int egx_gru_pointwise(const float* gi, const float* gh, const float* h_prev, int h_prev_ld, float* h_out,
                                 int h_out_ld, int num_rows, int hidden, void* stream_) {
  EGX_REQUIRE(gi && gh && h_out && num_rows > 0 && hidden > 0, "bad arguments");
  egx_launch_gru_pointwise(static_cast<hipStream_t>(stream_), gi, gh, h_prev, h_prev_ld, h_out, h_out_ld, num_rows, hidden);
  EGX_HIP_CHECK(hipGetLastError());
  return EGX_OK;
}

extern "C" int egx_cont6d_to_aa(const float* xb6, int num_rows, float* out, int out_ld, void* stream_) {
  EGX_REQUIRE(xb6 && out && num_rows > 0 && out_ld >= 93, "bad arguments");
  egx_launch_cont6d_to_aa(static_cast<hipStream_t>(stream_), xb6, num_rows, out, out_ld);
  EGX_HIP_CHECK(hipGetLastError());
  return EGX_OK;
}

extern "C" int egx_posenc(const float* dist, const float* time, int num_agents, float* out, void* stream_) {
  EGX_REQUIRE(dist && time && out && num_agents > 0, "bad arguments");
  egx_launch_posenc(static_cast<hipStream_t>(stream_), dist, time, num_agents, out);
  EGX_HIP_CHECK(hipGetLastError());
  return EGX_OK;
}

extern "C" int egx_sample_action(const float* mu, float* logvar, const float* eps, float min_logvar, float max_logvar,
                                 int deterministic, int num_rows, float* act, float* logp, void* stream_) {
  EGX_REQUIRE(mu && logvar && act && num_rows > 0 && (deterministic || eps), "bad arguments");
  hipLaunchKernelGGL(egx_sample_action_kernel, dim3(num_rows), dim3(128), 0, static_cast<hipStream_t>(stream_), mu, logvar, eps,
                     min_logvar, max_logvar, deterministic, num_rows, act, logp);
  EGX_HIP_CHECK(hipGetLastError());
  return EGX_OK;
}

extern "C" int egx_gae(const float* values, const float* rew, const int32_t* terminated, int num_steps, int num_agents,
                       double gamma, double gae_lambda, float* returns, float* adv, void* stream_) {
  EGX_REQUIRE(values && rew && terminated && returns && adv && num_steps > 0 && num_agents > 0, "bad arguments");
  hipLaunchKernelGGL(egx_gae_kernel, dim3(egx_ceil_div(num_agents, 128)), dim3(128), 0, static_cast<hipStream_t>(stream_),
                     values, rew, terminated, num_steps, num_agents, gamma, gae_lambda, returns, adv);
  EGX_HIP_CHECK(hipGetLastError());
  return EGX_OK;
}
