// Fused forward+gradient kernels for the PPO update (crowd_ppo/ppo_policy.py:189-241) and the GRU gate math of the
// policy's two encoders, so that the autograd graph of one minibatch is ~40 kernels instead of ~300 tiny elementwise ones.
#include <algorithm>
#include <cstring>
#include "egx_common.h"

namespace {
constexpr float LOG_SQRT_2PI = 0.91893853320467274178f;

}  // namespace

__device__ __forceinline__ float egx_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// One wave per transition (two of the 128 action dimensions per lane), rows dealt to the waves of the grid.  Writes d loss /
// d mu, d logvar (raw, pre-clamp), d value and accumulates the loss terms: out_terms[0..5] = loss, clip, vf, ent, kld,
// approx_kl (all already multiplied by `scale`) - per-wave partial sums, combined per block, six atomics per block.
__global__ __launch_bounds__(256) void egx_ppo_loss_kernel(const float* __restrict__ mu, const float* __restrict__ logvar,
                                                          const float* __restrict__ value, const float* __restrict__ act,
                                                          const float* __restrict__ adv, const float* __restrict__ ret,
                                                          const float* __restrict__ logp_old, const float* __restrict__ adv_stats,
                                                          const float* __restrict__ scale_ptr, float adv_eps, float min_lv,
                                                          float max_lv, float eps_clip, float vf_coef, float ent_coef, int n,
                                                          int stride /* row pitch of mu, logvar, g_mu, g_logvar */,
                                                          float* __restrict__ g_mu, float* __restrict__ g_logvar,
                                                          float* __restrict__ g_value, float* __restrict__ out_terms) {
  __shared__ float sh[4][6];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float scale = scale_ptr[0];
  float acc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int row = blockIdx.x * 4 + wave; row < n; row += gridDim.x * 4) {
    float lv[2], inv_var[2], diff[2], lp_d = 0.f, ent_d = 0.f, musq_d = 0.f;
    bool pass[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int d = lane + 64 * h;
      const size_t i = (size_t)row * stride + d;
      const float lv_raw = logvar[i], m = mu[i];
      lv[h] = fminf(fmaxf(lv_raw, min_lv), max_lv);
      pass[h] = (lv_raw >= min_lv) && (lv_raw <= max_lv);  // clamp backward
      inv_var[h] = expf(-lv[h]);
      diff[h] = act[(size_t)row * 128 + d] - m;
      lp_d += -0.5f * diff[h] * diff[h] * inv_var[h] - 0.5f * lv[h] - LOG_SQRT_2PI;
      ent_d += 0.5f + LOG_SQRT_2PI + 0.5f * lv[h];
      musq_d += m * m;
    }
    const float lp = egx_wave_sum(lp_d), ent = egx_wave_sum(ent_d), musq = egx_wave_sum(musq_d);
    float A = adv[row];
    if (adv_stats) A = (A - adv_stats[0]) / (adv_stats[1] + adv_eps);
    const float ratio = expf(lp - logp_old[row]);
    const float s1 = ratio * A;
    const float rc = fminf(fmaxf(ratio, 1.f - eps_clip), 1.f + eps_clip);
    const float s2 = rc * A;
    // d(-min(s1,s2))/d lp: the clamp passes gradient inside [1-eps,1+eps]; torch splits ties of min() evenly, and inside
    // the range s1 == s2 with identical derivatives, so the total is A*ratio there and when s1 < s2, else 0
    const bool in_range = (ratio >= 1.f - eps_clip) && (ratio <= 1.f + eps_clip);
    float dclip_dlp;
    if (s1 < s2) dclip_dlp = -A * ratio;
    else if (s1 > s2) dclip_dlp = in_range ? -A * ratio : 0.f;
    else dclip_dlp = in_range ? -A * ratio : -0.5f * A * ratio;  // tie outside the range: only the s1 half carries gradient
    const float v = value[row];
    const float dv = ret[row] - v;
    // gradients (loss = scale * sum_rows [clip + vf_coef*vf - ent_coef*ent])
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const size_t i = (size_t)row * stride + lane + 64 * h;
      g_mu[i] = scale * dclip_dlp * (diff[h] * inv_var[h]);
      const float dlp_dlv = 0.5f * (diff[h] * diff[h] * inv_var[h] - 1.f);
      g_logvar[i] = pass[h] ? scale * (dclip_dlp * dlp_dlv - ent_coef * 0.5f) : 0.f;
    }
    if (lane == 0) g_value[row] = scale * vf_coef * (-2.f) * dv;
    const float clip_l = -fminf(s1, s2), vf_l = dv * dv;
    acc[0] += scale * (clip_l + vf_coef * vf_l - ent_coef * ent);
    acc[1] += scale * clip_l;
    acc[2] += scale * vf_l;
    acc[3] += scale * ent;
    acc[4] += scale * 0.5f * musq / 128.f;
    acc[5] += scale * (logp_old[row] - lp);
  }
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 6; ++k) sh[wave][k] = acc[k];
  }
  __syncthreads();
  if (threadIdx.x < 6) atomicAdd(out_terms + threadIdx.x, (sh[0][threadIdx.x] + sh[1][threadIdx.x]) + (sh[2][threadIdx.x] + sh[3][threadIdx.x]));
}

// GRU gate math backward: recomputes r, z, n from (gi, gh, hprev) and turns dh into d gi, d gh, d hprev.
__global__ void egx_gru_pointwise_bwd_kernel(const float* __restrict__ gi, const float* __restrict__ gh,
                                             const float* __restrict__ hprev, const float* __restrict__ dh, int M, int H,
                                             float* __restrict__ dgi, float* __restrict__ dgh, float* __restrict__ dhprev) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= M * H) return;
  const int m = idx / H, c = idx % H;
  const size_t b3 = (size_t)m * 3 * H;
  const float r = 1.f / (1.f + expf(-(gi[b3 + c] + gh[b3 + c])));
  const float z = 1.f / (1.f + expf(-(gi[b3 + H + c] + gh[b3 + H + c])));
  const float ghn = gh[b3 + 2 * H + c];
  const float nn = tanhf(gi[b3 + 2 * H + c] + r * ghn);
  const float hp = hprev ? hprev[idx] : 0.f;
  const float g = dh[idx];
  const float dn = g * (1.f - z), dz = g * (hp - nn);
  const float dan = dn * (1.f - nn * nn);
  const float dar = dan * ghn * r * (1.f - r);
  const float daz = dz * z * (1.f - z);
  dgi[b3 + c] = dar; dgh[b3 + c] = dar;
  dgi[b3 + H + c] = daz; dgh[b3 + H + c] = daz;
  dgi[b3 + 2 * H + c] = dan; dgh[b3 + 2 * H + c] = dan * r;
  if (dhprev) dhprev[idx] = g * z;
}

// ---- dense-layer glue of the PPO update (the GEMMs themselves are plain library calls) -----------------------
// forward: a = act(z) in place; out = a + res when a residual is given (a is kept for the backward pass)
__global__ void egx_act_fwd_kernel(float* __restrict__ z, const float* __restrict__ res, float* __restrict__ out, size_t n4,
                                   int act, float slope) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  float4 v = reinterpret_cast<float4*>(z)[i];
  float* e = reinterpret_cast<float*>(&v);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float x = e[k];
    e[k] = (act == 1) ? tanhf(x) : (act == 2) ? fmaxf(x, 0.f) : (act == 3) ? (x > 0.f ? x : x * slope) : x;
  }
  reinterpret_cast<float4*>(z)[i] = v;
  if (res) {
    const float4 r = reinterpret_cast<const float4*>(res)[i];
    reinterpret_cast<float4*>(out)[i] = make_float4(v.x + r.x, v.y + r.y, v.z + r.z, v.w + r.w);
  }
}

// backward: g = dy * act'(a) (derivative expressed through the saved output a) and db[n] += sum_m g[m][n].
// Block = 64 columns x 4 row phases; blockIdx.y splits the rows, partial column sums meet in db through atomics.
__global__ __launch_bounds__(256) void egx_act_bwd_colsum_kernel(const float* __restrict__ dy, const float* __restrict__ a,
                                                                 float* __restrict__ g, float* __restrict__ db, int M, int N,
                                                                 int rows_per_block, int act, float slope) {
  __shared__ float part[4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int n = blockIdx.x * 64 + tx;
  const int m0 = blockIdx.y * rows_per_block, m1 = min(M, m0 + rows_per_block);
  float s = 0.f;
  if (n < N) {
    for (int m = m0 + ty; m < m1; m += 4) {
      const size_t idx = (size_t)m * N + n;
      float v = dy[idx];
      if (act != 0) {
        const float y = a[idx];
        v *= (act == 1) ? (1.f - y * y) : (act == 2) ? (y > 0.f ? 1.f : 0.f) : (y > 0.f ? 1.f : slope);
      }
      if (g) g[idx] = v;
      s += v;
    }
  }
  part[ty][tx] = s;
  __syncthreads();
  if (ty == 0 && n < N && db) atomicAdd(db + n, part[0][tx] + part[1][tx] + part[2][tx] + part[3][tx]);
}

// ---- minibatch assembly: one launch gathers the rows idx[0..n) of up to 8 row-major fp32 tensors --------------
struct GatherArgs {  // scalar fields only (a runtime-indexed array inside kernel arguments is copied to scratch)
  const float *s0, *s1, *s2, *s3, *s4, *s5, *s6, *s7;
  float *d0, *d1, *d2, *d3, *d4, *d5, *d6, *d7;
  int w0, w1, w2, w3, w4, w5, w6, w7;
  const long long* idx;
  int n;
};
__device__ __forceinline__ void gather_seg(const float* __restrict__ s, float* __restrict__ d, int w, long long src_row, int dst_row) {
  if (!s || w <= 0) return;
  const float* sr = s + (size_t)src_row * w;
  float* dr = d + (size_t)dst_row * w;
  for (int c = threadIdx.x; c < w; c += blockDim.x) dr[c] = sr[c];
}
__global__ __launch_bounds__(256) void egx_gather_rows_kernel(GatherArgs a) {
  const int r = blockIdx.x;
  const long long src = a.idx[r];
  gather_seg(a.s0, a.d0, a.w0, src, r); gather_seg(a.s1, a.d1, a.w1, src, r);
  gather_seg(a.s2, a.d2, a.w2, src, r); gather_seg(a.s3, a.d3, a.w3, src, r);
  gather_seg(a.s4, a.d4, a.w4, src, r); gather_seg(a.s5, a.d5, a.w5, src, r);
  gather_seg(a.s6, a.d6, a.w6, src, r); gather_seg(a.s7, a.d7, a.w7, src, r);
}

// mean and UNBIASED standard deviation of the minibatch advantages (ppo_policy.py:195-197: adv.mean(), adv.std())
__global__ __launch_bounds__(256) void egx_adv_stats_kernel(const float* __restrict__ adv, int n, float* __restrict__ out) {
  __shared__ double red[256];
  __shared__ double s_mean;
  double a = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) a += adv[i];
  red[threadIdx.x] = a;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if (threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x == 0) s_mean = red[0] / n;
  __syncthreads();
  const double mean = s_mean;
  double b = 0.0;  // second pass around the mean (two-pass variance)
  for (int i = threadIdx.x; i < n; i += 256) { const double d = adv[i] - mean; b += d * d; }
  red[threadIdx.x] = b;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if (threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out[0] = (float)mean;
    out[1] = (n > 1) ? (float)sqrt(red[0] / (n - 1)) : nanf("");
  }
}

// episode bookkeeping of the collector (tianshou Collector.collect [upstream]: running return / length per env, sums over
// the episodes that finished in this step); single workgroup, A is a few hundred
__global__ __launch_bounds__(256) void egx_track_episode_kernel(const float* __restrict__ rew, const int* __restrict__ term, int A,
                                                                float* __restrict__ ep_ret, float* __restrict__ ep_len,
                                                                float* __restrict__ done_sums /* ret, len, count */) {
  __shared__ float s[3][256];
  float a = 0.f, b = 0.f, c = 0.f;
  for (int i = threadIdx.x; i < A; i += 256) {
    const float r = ep_ret[i] + rew[i], l = ep_len[i] + 1.f;
    const bool d = term[i] != 0;
    if (d) { a += r; b += l; c += 1.f; }
    ep_ret[i] = d ? 0.f : r;
    ep_len[i] = d ? 0.f : l;
  }
  s[0][threadIdx.x] = a; s[1][threadIdx.x] = b; s[2][threadIdx.x] = c;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if (threadIdx.x < st)
      for (int k = 0; k < 3; ++k) s[k][threadIdx.x] += s[k][threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x < 3) done_sums[threadIdx.x] += s[threadIdx.x][0];
}

// One launch per vector step of the collector instead of seven copies + the bookkeeping kernel: the observation the
// environment just produced goes into rollout slot t+1, reward / termination flag into slot t, and the extra workgroup
// (blockIdx == A) runs the episode bookkeeping above.
__global__ __launch_bounds__(256) void egx_rollout_store_kernel(const float* __restrict__ state, const float* __restrict__ ego,
                                                                const float* __restrict__ dist, const float* __restrict__ time,
                                                                const float* __restrict__ rew, const int* __restrict__ term, int A,
                                                                float* __restrict__ state_dst, float* __restrict__ ego_dst,
                                                                float* __restrict__ dist_dst, float* __restrict__ time_dst,
                                                                float* __restrict__ rew_dst, int* __restrict__ term_dst,
                                                                float* __restrict__ ep_ret, float* __restrict__ ep_len,
                                                                float* __restrict__ done_sums) {
  const int a = blockIdx.x;
  if (a < A) {
    const f32x4* s4 = reinterpret_cast<const f32x4*>(state + (size_t)a * 804);   // 804 = 201 float4
    f32x4* d4 = reinterpret_cast<f32x4*>(state_dst + (size_t)a * 804);
    if (threadIdx.x < 201) d4[threadIdx.x] = s4[threadIdx.x];
    else if (threadIdx.x < 217) {
      const int e = threadIdx.x - 201;
      reinterpret_cast<f32x4*>(ego_dst + (size_t)a * 64)[e] = reinterpret_cast<const f32x4*>(ego + (size_t)a * 64)[e];
    } else if (threadIdx.x == 217) {
      dist_dst[a] = dist[a];
      time_dst[a] = time[a];
      if (rew_dst) { rew_dst[a] = rew[a]; term_dst[a] = term[a]; }
    }
    return;
  }
  if (!ep_ret) return;
  __shared__ float s[3][256];
  float x = 0.f, y = 0.f, c = 0.f;
  for (int i = threadIdx.x; i < A; i += 256) {
    const float r = ep_ret[i] + rew[i], l = ep_len[i] + 1.f;
    const bool d = term[i] != 0;
    if (d) { x += r; y += l; c += 1.f; }
    ep_ret[i] = d ? 0.f : r;
    ep_len[i] = d ? 0.f : l;
  }
  s[0][threadIdx.x] = x; s[1][threadIdx.x] = y; s[2][threadIdx.x] = c;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if (threadIdx.x < st)
      for (int k = 0; k < 3; ++k) s[k][threadIdx.x] += s[k][threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x < 3) done_sums[threadIdx.x] += s[threadIdx.x][0];
}

// ---- optimiser step over flat buffers: gradient-norm clip of a prefix + AdamW (torch.optim.AdamW semantics) ----------
// pass 1: per-block partial sums of squares of g[0..n_clip) (double accumulation, fixed order -> deterministic); block 0
// also advances the step counter, so that pass 2 (a later kernel on the same stream) reads the new value everywhere.
__global__ __launch_bounds__(256) void egx_sumsq_partial_kernel(const float* __restrict__ g, size_t n, float* __restrict__ partials,
                                                                float* __restrict__ step) {
  __shared__ double red[256];
  double a = 0.0;
  const size_t stride = (size_t)gridDim.x * 256 * 4;
  for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4; i + 3 < n; i += stride) {
    const float4 v = *reinterpret_cast<const float4*>(g + i);
    a += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    for (size_t i = n & ~(size_t)3; i < n; ++i) a += (double)g[i] * g[i];
    *step += 1.f;
  }
  red[threadIdx.x] = a;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if (threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x == 0) partials[blockIdx.x] = (float)red[0];
}

// pass 1b (one workgroup): total norm -> clip coefficient; bias corrections of the new step count
__global__ __launch_bounds__(256) void egx_adamw_consts_kernel(const float* __restrict__ partials, int n_partials, int do_clip,
                                                               float max_norm, double lr, double b1, double b2,
                                                               const float* __restrict__ step, float* __restrict__ consts) {
  __shared__ double red[256];
  double a = 0.0;
  for (int i = threadIdx.x; i < n_partials; i += 256) a += partials[i];
  red[threadIdx.x] = a;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if (threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const double t = (double)*step;
    const float bc1 = (float)(1.0 - pow(b1, t)), bc2 = (float)(1.0 - pow(b2, t));
    float coef = 1.f;
    if (do_clip) coef = fminf(max_norm / ((float)sqrt(red[0]) + 1e-6f), 1.f);  // torch.nn.utils.clip_grad_norm_
    consts[0] = coef;
    consts[1] = (float)(lr / bc1);
    consts[2] = sqrtf(bc2);
  }
}

// pass 2: AdamW on 4 elements per thread: decoupled weight decay, exp_avg lerp, exp_avg_sq, bias corrections - the arithmetic of torch's fused
// multi-tensor AdamW functor (aten/src/ATen/native/cuda/fused_adam_utils.cuh [upstream torch]).
__global__ __launch_bounds__(256) void egx_adamw_flat_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                             float* __restrict__ v, size_t n, size_t n_clip,
                                                             const float* __restrict__ consts /* clip coef, step size, sqrt(bc2) */,
                                                             double lr, double b1, double b2, double eps, double wd) {
  const float coef = consts[0], step_size = consts[1], bc2s = consts[2];
  const size_t i0 = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i0 >= n) return;
  const int cnt = (int)((n - i0 < 4) ? (n - i0) : 4);
  float gr[4], pv[4], mv[4], vv[4];
  if (cnt == 4) {
    *reinterpret_cast<float4*>(gr) = *reinterpret_cast<const float4*>(g + i0);
    *reinterpret_cast<float4*>(pv) = *reinterpret_cast<const float4*>(p + i0);
    *reinterpret_cast<float4*>(mv) = *reinterpret_cast<const float4*>(m + i0);
    *reinterpret_cast<float4*>(vv) = *reinterpret_cast<const float4*>(v + i0);
  } else {
    for (int e = 0; e < cnt; ++e) { gr[e] = g[i0 + e]; pv[e] = p[i0 + e]; mv[e] = m[i0 + e]; vv[e] = v[i0 + e]; }
  }
  const double omb1 = 1.0 - b1, omb2 = 1.0 - b2, lrwd = lr * wd;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    if (e < cnt) {
      // operand types as in torch's functor: fp32 state, double hyper-parameters (the mixed expressions run in double)
      float gg = gr[e];
      if (i0 + e < n_clip) gg *= coef;
      pv[e] = (float)((double)pv[e] - lrwd * (double)pv[e]);
      mv[e] = (float)((double)mv[e] + omb1 * ((double)gg - (double)mv[e]));
      vv[e] = (float)(b2 * (double)vv[e] + omb2 * (double)gg * (double)gg);
      const float denom = (float)((double)(sqrtf(vv[e]) / bc2s) + eps);
      pv[e] -= step_size * mv[e] / denom;
    }
  }
  if (cnt == 4) {
    *reinterpret_cast<float4*>(p + i0) = *reinterpret_cast<const float4*>(pv);
    *reinterpret_cast<float4*>(m + i0) = *reinterpret_cast<const float4*>(mv);
    *reinterpret_cast<float4*>(v + i0) = *reinterpret_cast<const float4*>(vv);
  } else {
    for (int e = 0; e < cnt; ++e) { p[i0 + e] = pv[e]; m[i0 + e] = mv[e]; v[i0 + e] = vv[e]; }
  }
}

// clears the six sums (hipMemsetAsync of 24 bytes is TWO fill launches: a 16-byte part and a tail)
__global__ void egx_zero_terms_kernel(float* __restrict__ t) {
  if (threadIdx.x < 6) t[threadIdx.x] = 0.f;
}

static int ppo_loss_blocks(int num_rows) { return std::max(1, std::min(256, egx_ceil_div(num_rows, 8))); }   // two rows per wave

extern "C" int egx_ppo_loss(const float* mu, const float* logvar, const float* value, const float* act, const float* adv,
                            const float* ret, const float* logp_old, const float* adv_stats, const float* scale,
                            float adv_eps, float min_logvar, float max_logvar, float eps_clip, float vf_coef, float ent_coef,
                            int num_rows, float* g_mu, float* g_logvar, float* g_value, float* out_terms, void* stream_) {
  EGX_REQUIRE(mu && logvar && value && act && adv && ret && logp_old && scale && g_mu && g_logvar && g_value && out_terms &&
                  num_rows > 0, "bad arguments");
  hipStream_t st = static_cast<hipStream_t>(stream_);
  hipLaunchKernelGGL(egx_zero_terms_kernel, dim3(1), dim3(64), 0, st, out_terms);
  hipLaunchKernelGGL(egx_ppo_loss_kernel, dim3(ppo_loss_blocks(num_rows)), dim3(256), 0, st, mu, logvar, value, act, adv, ret, logp_old,
                     adv_stats, scale, adv_eps, min_logvar, max_logvar, eps_clip, vf_coef, ent_coef, num_rows, 128, g_mu, g_logvar,
                     g_value, out_terms);
  EGX_HIP_CHECK(hipGetLastError());
  return EGX_OK;
}

// the update chain's form (csrc/update3.hip): out_terms was cleared by an earlier launch of the chain (egx_launch_posenc3)
int egx_ppo_loss_packed_precleared(const float* zp, const float* value, const float* act, const float* adv, const float* ret,
                                   const float* logp_old, const float* adv_stats, const float* scale, float adv_eps, float min_logvar,
                                   float max_logvar, float eps_clip, float vf_coef, float ent_coef, int num_rows, float* g_zp,
                                   float* g_value, float* out_terms, void* stream_) {
  EGX_REQUIRE(zp && value && act && adv && ret && logp_old && scale && g_zp && g_value && out_terms && num_rows > 0, "bad arguments");
  hipStream_t st = static_cast<hipStream_t>(stream_);
  hipLaunchKernelGGL(egx_ppo_loss_kernel, dim3(ppo_loss_blocks(num_rows)), dim3(256), 0, st, zp, zp + 128, value, act, adv, ret, logp_old, adv_stats,
                     scale, adv_eps, min_logvar, max_logvar, eps_clip, vf_coef, ent_coef, num_rows, 256, g_zp, g_zp + 128, g_value,
                     out_terms);
  EGX_HIP_CHECK(hipGetLastError());
  return EGX_OK;
}

extern "C" int egx_ppo_loss_packed(const float* zp, const float* value, const float* act, const float* adv, const float* ret,
                                   const float* logp_old, const float* adv_stats, const float* scale, float adv_eps, float min_logvar,
                                   float max_logvar, float eps_clip, float vf_coef, float ent_coef, int num_rows, float* g_zp,
                                   float* g_value, float* out_terms, void* stream_) {
  EGX_REQUIRE(zp && value && act && adv && ret && logp_old && scale && g_zp && g_value && out_terms && num_rows > 0, "bad arguments");
  hipStream_t st = static_cast<hipStream_t>(stream_);
  hipLaunchKernelGGL(egx_zero_terms_kernel, dim3(1), dim3(64), 0, st, out_terms);
  hipLaunchKernelGGL(egx_ppo_loss_kernel, dim3(ppo_loss_blocks(num_rows)), dim3(256), 0, st, zp, zp + 128, value, act, adv, ret, logp_old, adv_stats,
                     scale, adv_eps, min_logvar, max_logvar, eps_clip, vf_coef, ent_coef, num_rows, 256, g_zp, g_zp + 128, g_value,
                     out_terms);
  EGX_HIP_CHECK(hipGetLastError());
  return EGX_OK;
}

extern "C" int egx_gru_pointwise_bwd(const float* gi, const float* gh, const float* h_prev, const float* dh, int num_rows,
                                     int hidden, float* dgi, float* dgh, float* dh_prev, void* stream_) {
  EGX_REQUIRE(gi && gh && dh && dgi && dgh && num_rows > 0 && hidden > 0, "bad arguments");
  const int n = num_rows * hidden;
  hipLaunchKernelGGL(egx_gru_pointwise_bwd_kernel, dim3(egx_ceil_div(n, 256)), dim3(256), 0, static_cast<hipStream_t>(stream_),
                     gi, gh, h_prev, dh, num_rows, hidden, dgi, dgh, dh_prev);
  EGX_HIP_CHECK(hipGetLastError());
  return EGX_OK;
}

extern "C" int egx_act_fwd(float* z, const float* res, float* out, int num_rows, int width, int act, float slope, void* stream_) {
  EGX_REQUIRE(z && num_rows > 0 && width > 0 && (!res || out), "bad arguments");
  EGX_REQUIRE(((size_t)num_rows * width) % 4 == 0, "rows x width must be a multiple of 4");
  EGX_REQUIRE(act >= 0 && act <= 3, "unknown activation");
  const size_t n4 = (size_t)num_rows * width / 4;
  hipLaunchKernelGGL(egx_act_fwd_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream_), z,
                     res, out, n4, act, slope);
  EGX_HIP_CHECK(hipGetLastError());
  return EGX_OK;
}

extern "C" int egx_act_bwd_colsum(const float* dy, const float* a, float* g, float* db_accum, int num_rows, int width, int act,
                                  float slope, void* stream_) {
  EGX_REQUIRE(dy && num_rows > 0 && width > 0 && (act == 0 || a) && (g || db_accum), "bad arguments");
  EGX_REQUIRE(act >= 0 && act <= 3, "unknown activation");
  const int strips = egx_ceil_div(width, 64);
  int splits = std::max(1, std::min(egx_ceil_div(num_rows, 32), 512 / strips));  // enough blocks to fill the chip
  const int rpb = egx_ceil_div(num_rows, splits);
  splits = egx_ceil_div(num_rows, rpb);
  hipLaunchKernelGGL(egx_act_bwd_colsum_kernel, dim3(strips, splits), dim3(256), 0, static_cast<hipStream_t>(stream_), dy, a, g,
                     db_accum, num_rows, width, rpb, act, slope);
  EGX_HIP_CHECK(hipGetLastError());
  return EGX_OK;
}

extern "C" int egx_gather_rows(const int64_t* idx, int num_rows, int num_tensors, const float* const* src, const int* width,
                               float* const* dst, void* stream_) {
  EGX_REQUIRE(idx && src && width && dst && num_rows > 0 && num_tensors > 0 && num_tensors <= 8, "bad arguments");
  GatherArgs a;
  std::memset(&a, 0, sizeof(a));
  const float** sp[8] = {&a.s0, &a.s1, &a.s2, &a.s3, &a.s4, &a.s5, &a.s6, &a.s7};
  float** dp[8] = {&a.d0, &a.d1, &a.d2, &a.d3, &a.d4, &a.d5, &a.d6, &a.d7};
  int* wp[8] = {&a.w0, &a.w1, &a.w2, &a.w3, &a.w4, &a.w5, &a.w6, &a.w7};
  for (int t = 0; t < num_tensors; ++t) {
    EGX_REQUIRE(src[t] && dst[t] && width[t] > 0, "null tensor or non-positive width");
    *sp[t] = src[t]; *dp[t] = dst[t]; *wp[t] = width[t];
  }
  a.idx = reinterpret_cast<const long long*>(idx);
  a.n = num_rows;
  hipLaunchKernelGGL(egx_gather_rows_kernel, dim3(num_rows), dim3(256), 0, static_cast<hipStream_t>(stream_), a);
  EGX_HIP_CHECK(hipGetLastError());
  return EGX_OK;
}

extern "C" int egx_adv_stats(const float* adv, int n, float* out_mean_std, void* stream_) {
  EGX_REQUIRE(adv && out_mean_std && n > 0, "bad arguments");
  hipLaunchKernelGGL(egx_adv_stats_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream_), adv, n, out_mean_std);
  EGX_HIP_CHECK(hipGetLastError());
  return EGX_OK;
}

extern "C" int egx_track_episode(const float* rew, const int32_t* term, int num_agents, float* ep_ret, float* ep_len,
                                 float* done_sums, void* stream_) {
  EGX_REQUIRE(rew && term && ep_ret && ep_len && done_sums && num_agents > 0, "bad arguments");
  hipLaunchKernelGGL(egx_track_episode_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream_), rew, term, num_agents,
                     ep_ret, ep_len, done_sums);
  EGX_HIP_CHECK(hipGetLastError());
  return EGX_OK;
}

extern "C" int egx_rollout_store(const float* state, const float* egosensing, const float* dist, const float* time, const float* rew,
                                 const int32_t* term, int num_agents, float* state_dst, float* ego_dst, float* dist_dst, float* time_dst,
                                 float* rew_dst, int32_t* term_dst, float* ep_ret, float* ep_len, float* done_sums, void* stream_) {
  EGX_REQUIRE(state && egosensing && dist && time && state_dst && ego_dst && dist_dst && time_dst && num_agents > 0, "bad arguments");
  EGX_REQUIRE((!rew_dst && !term_dst && !ep_ret) || (rew && term && rew_dst && term_dst), "reward / termination source or slot missing");
  EGX_REQUIRE(!ep_ret || (ep_len && done_sums && rew && term), "episode bookkeeping needs ep_len, done_sums, rew, term");
  hipLaunchKernelGGL(egx_rollout_store_kernel, dim3(num_agents + (ep_ret ? 1 : 0)), dim3(256), 0, static_cast<hipStream_t>(stream_), state,
                     egosensing, dist, time, rew, term, num_agents, state_dst, ego_dst, dist_dst, time_dst, rew_dst, term_dst, ep_ret, ep_len,
                     done_sums);
  EGX_HIP_CHECK(hipGetLastError());
  return EGX_OK;
}

extern "C" size_t egx_adamw_workspace_floats(void) { return 1024 + 8; }

extern "C" int egx_adamw_clip_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n, size_t n_clip,
                                   float max_norm, double lr, double beta1, double beta2, double eps, double weight_decay,
                                   float* step, float* workspace, void* stream_) {
  EGX_REQUIRE(param && grad && exp_avg && exp_avg_sq && step && workspace && n > 0 && n_clip <= n, "bad arguments");
  hipStream_t st = static_cast<hipStream_t>(stream_);
  const int nb = 1024;
  const bool do_clip = max_norm > 0.f && n_clip > 0;
  hipLaunchKernelGGL(egx_sumsq_partial_kernel, dim3(nb), dim3(256), 0, st, grad, do_clip ? n_clip : (size_t)0, workspace, step);
  hipLaunchKernelGGL(egx_adamw_consts_kernel, dim3(1), dim3(256), 0, st, workspace, nb, do_clip ? 1 : 0, max_norm, lr, beta1, beta2,
                     step, workspace + nb);
  const size_t blocks = (n + 1023) / 1024;
  hipLaunchKernelGGL(egx_adamw_flat_kernel, dim3((unsigned)blocks), dim3(256), 0, st, param, grad, exp_avg, exp_avg_sq, n, n_clip,
                     workspace + nb, lr, beta1, beta2, eps, weight_decay);
  EGX_HIP_CHECK(hipGetLastError());
  return EGX_OK;
}
