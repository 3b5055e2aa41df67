// Fused forward+gradient kernels for the PPO update (crowd_ppo/ppo_policy.py:189-241) and the GRU gate math of the
// policy's two encoders, so that the autograd graph of one minibatch is ~40 kernels instead of ~300 tiny elementwise ones.
#include "egx_common.h"

namespace {
constexpr float LOG_SQRT_2PI = 0.91893853320467274178f;

__device__ __forceinline__ float block_sum128(float v, float* sh) {
  sh[threadIdx.x] = v;
  __syncthreads();
  for (int s = 64; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
    __syncthreads();
  }
  const float r = sh[0];
  __syncthreads();
  return r;
}
}  // namespace

// One 128-thread block per transition.  Writes d loss / d mu, d logvar (raw, pre-clamp), d value and accumulates the
// loss terms: out_terms[0..5] = loss, clip, vf, ent, kld, approx_kl (all already multiplied by `scale`).
__global__ __launch_bounds__(128) void egx_ppo_loss_kernel(const float* __restrict__ mu, const float* __restrict__ logvar,
                                                          const float* __restrict__ value, const float* __restrict__ act,
                                                          const float* __restrict__ adv, const float* __restrict__ ret,
                                                          const float* __restrict__ logp_old, const float* __restrict__ adv_stats,
                                                          const float* __restrict__ scale_ptr, float adv_eps, float min_lv,
                                                          float max_lv, float eps_clip, float vf_coef, float ent_coef, int n,
                                                          float* __restrict__ g_mu, float* __restrict__ g_logvar,
                                                          float* __restrict__ g_value, float* __restrict__ out_terms) {
  __shared__ float sh[128];
  const int row = blockIdx.x, d = threadIdx.x;
  const size_t i = (size_t)row * 128 + d;
  const float scale = scale_ptr[0];
  const float lv_raw = logvar[i];
  const float lv = fminf(fmaxf(lv_raw, min_lv), max_lv);
  const bool pass = (lv_raw >= min_lv) && (lv_raw <= max_lv);  // clamp backward
  const float inv_var = expf(-lv);
  const float diff = act[i] - mu[i];
  const float lp_d = -0.5f * diff * diff * inv_var - 0.5f * lv - LOG_SQRT_2PI;
  const float ent_d = 0.5f + LOG_SQRT_2PI + 0.5f * lv;
  const float lp = block_sum128(lp_d, sh);
  const float ent = block_sum128(ent_d, sh);
  const float musq = block_sum128(mu[i] * mu[i], sh);
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
  g_mu[i] = scale * dclip_dlp * (diff * inv_var);
  const float dlp_dlv = 0.5f * (diff * diff * inv_var - 1.f);
  g_logvar[i] = pass ? scale * (dclip_dlp * dlp_dlv - ent_coef * 0.5f) : 0.f;
  if (d == 0) {
    g_value[row] = scale * vf_coef * (-2.f) * dv;
    const float clip_l = -fminf(s1, s2), vf_l = dv * dv;
    atomicAdd(out_terms + 0, scale * (clip_l + vf_coef * vf_l - ent_coef * ent));
    atomicAdd(out_terms + 1, scale * clip_l);
    atomicAdd(out_terms + 2, scale * vf_l);
    atomicAdd(out_terms + 3, scale * ent);
    atomicAdd(out_terms + 4, scale * 0.5f * musq / 128.f);
    atomicAdd(out_terms + 5, scale * (logp_old[row] - lp));
  }
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

extern "C" int egx_ppo_loss(const float* mu, const float* logvar, const float* value, const float* act, const float* adv,
                            const float* ret, const float* logp_old, const float* adv_stats, const float* scale,
                            float adv_eps, float min_logvar, float max_logvar, float eps_clip, float vf_coef, float ent_coef,
                            int num_rows, float* g_mu, float* g_logvar, float* g_value, float* out_terms, void* stream_) {
  EGX_REQUIRE(mu && logvar && value && act && adv && ret && logp_old && scale && g_mu && g_logvar && g_value && out_terms &&
                  num_rows > 0, "bad arguments");
  hipStream_t st = static_cast<hipStream_t>(stream_);
  EGX_HIP_CHECK(hipMemsetAsync(out_terms, 0, 6 * sizeof(float), st));
  hipLaunchKernelGGL(egx_ppo_loss_kernel, dim3(num_rows), dim3(128), 0, st, mu, logvar, value, act, adv, ret, logp_old,
                     adv_stats, scale, adv_eps, min_logvar, max_logvar, eps_clip, vf_coef, ent_coef, num_rows, g_mu, g_logvar,
                     g_value, out_terms);
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
