// Dense-layer kernels for the rollout networks (C-VAE marker predictor, body regressor, VPoser encoder,
// policy) on gfx950.  The layers are tiny (M = agents or agents*18 rows, K,N <= 1536), so the design goal is
// latency: one 32x32 output tile per workgroup, K split across the 4 waves of the workgroup (each wave runs
// an independent v_mfma_f32_32x32x2_f32 chain over a quarter of K straight from global memory, 16-byte
// loads along K for both operands), partial tiles reduced through LDS, and bias / activation / residual
// fused into the epilogue.  Inputs may be the concatenation of up to 4 row-major segments (the reference
// builds them with torch.cat: models_GAMMA_primitive.py:93,257; models_policy_ppo.py:305) - no copy is made.
// Weights are read in torch's own [N,K] layout so the same storage serves the autograd update path.
#include <mutex>
#include "egx_nets.h"

namespace {

// NOTE: no arrays inside the kernel-argument struct - a runtime-indexed member array makes hipcc copy the whole
// struct to scratch and turns every operand load into scratch + flat traffic (measured: 5x slower).
struct LinArgs {
  const float *xp0, *xp1, *xp2, *xp3;  // up to 4 concatenated input segments
  int xw0, xw1, xw2, xw3;              // widths (0 = unused)
  int xl0, xl1, xl2, xl3;              // leading dimensions
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

// 4 consecutive k of the concatenated input starting at local offset kk of segment (p,w,l); elements past the end of
// the segment come from the next segment (pn,wn,ln) or are zero.  Segment widths are >= 4, so two segments suffice.
__device__ __forceinline__ f32x4 load_straddle(const float* p, int w, int l, const float* pn, int wn, int ln, int row, int kk) {
  f32x4 v;
  const float* r0 = p + (size_t)row * l;
  const float* r1 = pn ? pn + (size_t)row * ln : nullptr;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int idx = kk + e;
    float x = 0.f;
    if (idx < w) x = r0[idx];
    else if (r1 && idx - w < wn) x = r1[idx - w];
    v[e] = x;
  }
  return v;
}

__device__ __forceinline__ f32x4 load_x4(const LinArgs& a, int row, int k) {
  int kk = k;
  if (kk + 3 < a.xw0) return *reinterpret_cast<const f32x4u*>(a.xp0 + (size_t)row * a.xl0 + kk);
  if (kk < a.xw0) return load_straddle(a.xp0, a.xw0, a.xl0, a.xp1, a.xw1, a.xl1, row, kk);
  kk -= a.xw0;
  if (kk + 3 < a.xw1) return *reinterpret_cast<const f32x4u*>(a.xp1 + (size_t)row * a.xl1 + kk);
  if (kk < a.xw1) return load_straddle(a.xp1, a.xw1, a.xl1, a.xp2, a.xw2, a.xl2, row, kk);
  kk -= a.xw1;
  if (kk + 3 < a.xw2) return *reinterpret_cast<const f32x4u*>(a.xp2 + (size_t)row * a.xl2 + kk);
  if (kk < a.xw2) return load_straddle(a.xp2, a.xw2, a.xl2, a.xp3, a.xw3, a.xl3, row, kk);
  kk -= a.xw2;
  if (kk + 3 < a.xw3) return *reinterpret_cast<const f32x4u*>(a.xp3 + (size_t)row * a.xl3 + kk);
  if (kk < a.xw3) return load_straddle(a.xp3, a.xw3, a.xl3, nullptr, 0, 0, row, kk);
  return f32x4{0.f, 0.f, 0.f, 0.f};
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

struct LinArgs2 {
  LinArgs p0, p1;
  int blocks0;  // blocks [0, blocks0) work on p0, the rest on p1 (independent GEMMs sharing one launch)
};

// One 32 x (32 NW) output tile per workgroup; each of the four waves owns a quarter of K and walks it in blocks of 32.
//   DIRECT (K <= 512): operands go from global memory straight into MFMA operand registers - lane (i, h) fetches
//     k = 8 q + 4 h .. + 3 of row m0 + i of X and of row n0 + i of W with one 16-byte load each and feeds them to four
//     consecutive MFMA steps (which k a step multiplies is free as long as both operands agree, so the steps of a lane simply
//     walk its own four elements).  No LDS staging, no wave barrier.  NW = 2: the X fragments serve two column tiles, so a
//     launch that would need more workgroups than the chip holds at once gets by with half as many.
//   staged (K > 512: the policy's 1152-wide layers): row-contiguous 16-byte loads (8 lanes cover one 128-byte row segment,
//     every cache line is requested once), parked in a wave-private LDS strip (36-float pitch) and read back as operand
//     fragments; the direct form touches 32 lines per load instruction, which costs more than the staging once a wave walks
//     many blocks.
// No software prefetch in either: on gfx950 a wave that issues v_mfma_f32_32x32x2_f32 while its own global loads are in
// flight runs the matrix pipe at about half rate (scripts/ubench/mfma_loads.hip); the load latency of one wave is covered by
// the other waves of the SIMD.  Two k-blocks per round trip: these layers are latency-bound (a wave walks 2..9 blocks, a round
// trip to L2 / Infinity Cache costs ~2 us against 0.4 us of MFMAs per block), so the number of trips is what counts.
template <bool DIRECT, int NW, int TRIP = 2>
__global__ __launch_bounds__(256) void egx_linear_kernel(LinArgs2 two) {
  static_assert(DIRECT || NW == 1, "the staged form computes one column tile");
  const bool second = (int)blockIdx.x >= two.blocks0;
  const LinArgs& a = second ? two.p1 : two.p0;
  const int bid = second ? (int)blockIdx.x - two.blocks0 : (int)blockIdx.x;
  __shared__ __attribute__((aligned(16))) float stage[DIRECT ? NW * 4 * 16 * 64 : 4 * 2 * 32 * 36];
  float* red = stage;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 31, h = lane >> 5;
  // XCD-aware tile map (blocks are dealt round-robin to the 8 XCDs, each with a private 4 MiB L2): every XCD gets a
  // contiguous chunk of tiles along the LONGER tile axis and sweeps the other axis, so it touches 1/8 of one operand
  // and all of the other instead of everything.
  constexpr int TW = 32 * NW;
  int mt, nt;
  {
    const int MT = (a.M + 31) >> 5, NT = (a.N + TW - 1) / TW;
    const int xcd = bid & 7, local = bid >> 3;
    if (NT >= MT) {
      const int per = (NT + 7) >> 3;
      nt = xcd * per + local / MT;
      mt = local % MT;
      if (local >= per * MT || nt >= NT) return;
    } else {
      const int per = (MT + 7) >> 3;
      mt = xcd * per + local / NT;
      nt = local % NT;
      if (local >= per * NT || mt >= MT) return;
    }
  }
  const int m0 = mt * 32, n0 = nt * TW;
  const int nblk = (a.K + 31) >> 5;
  const int per = (nblk + 3) >> 2;
  const int b0 = wave * per, b1 = min(nblk, b0 + per);
  f32x16 acc[NW];
#pragma unroll
  for (int t = 0; t < NW; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  if (DIRECT) {
    const int xrow = min(m0 + i, a.M - 1);
    int wrow[NW];
#pragma unroll
    for (int t = 0; t < NW; ++t) wrow[t] = min(n0 + 32 * t + i, a.N - 1);
    for (int b = b0; b < b1; b += TRIP) {
      f32x4 gx[TRIP][4], gw[TRIP][NW][4];
      const int nbt = min(TRIP, b1 - b);
#pragma unroll
      for (int u = 0; u < TRIP; ++u) {
        if (u < nbt) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            gx[u][q] = load_x4(a, xrow, (b + u) * 32 + q * 8 + 4 * h);
#pragma unroll
            for (int t = 0; t < NW; ++t) gw[u][t][q] = load_w4(a, wrow[t], (b + u) * 32 + q * 8 + 4 * h);
          }
        }
      }
#pragma unroll
      for (int u = 0; u < TRIP; ++u) {
        if (u >= nbt) break;
        #pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int t = 0; t < NW; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(gx[u][q][e], gw[u][t][q][e], acc[t], 0, 0, 0);
      
      }
    }
  } else {
    const int lr = lane >> 3, lc = (lane & 7) * 4;  // loader role: row lr (+8q), k offset lc
    float* xs = stage + (wave * 2 + 0) * (32 * 36);
    float* ws = stage + (wave * 2 + 1) * (32 * 36);
    // the loads of the next trip are issued right after the MFMAs of this one (issuing them BEFORE the MFMAs, a register
    // double buffer, was measured slower: 35.4 against 28.7 us at 512 x 1152 x 1152)
    f32x4 gx[2][4], gw[2][4];
    auto fetch = [&](int b, f32x4 (&dx)[2][4], f32x4 (&dw)[2][4]) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        if (b + u < b1) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            dx[u][q] = load_x4(a, min(m0 + q * 8 + lr, a.M - 1), (b + u) * 32 + lc);
            dw[u][q] = load_w4(a, min(n0 + q * 8 + lr, a.N - 1), (b + u) * 32 + lc);
          }
        }
      }
    };
    if (b0 < b1) fetch(b0, gx, gw);
    for (int b = b0; b < b1; b += 2) {
      const bool two_blocks = b + 1 < b1;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        if (u == 1 && !two_blocks) break;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          *reinterpret_cast<f32x4*>(xs + (q * 8 + lr) * 36 + lc) = gx[u][q];
          *reinterpret_cast<f32x4*>(ws + (q * 8 + lr) * 36 + lc) = gw[u][q];
        }
        #pragma unroll
        for (int c = 0; c < 4; ++c) {
          const f32x4 xa = *reinterpret_cast<const f32x4*>(xs + i * 36 + c * 8 + 4 * h);
          const f32x4 wb = *reinterpret_cast<const f32x4*>(ws + i * 36 + c * 8 + 4 * h);
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[e], wb[e], acc[0], 0, 0, 0);
        }
      
      }
      if (b + 2 < b1) fetch(b + 2, gx, gw);
    }
  }
  // split-K reduction through LDS; every wave then finishes four of the lane's sixteen rows (bias, activation, residual)
  __syncthreads();
#pragma unroll
  for (int t = 0; t < NW; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) red[((t * 4 + wave) * 16 + r) * 64 + lane] = acc[t][r];
  __syncthreads();
#pragma unroll
  for (int t = 0; t < NW; ++t) {
    const int n = n0 + 32 * t + i;
    const float bsv = (a.bias && n < a.N) ? a.bias[n] : 0.f;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int r = wave * 4 + rr;
      const int m = m0 + (r & 3) + 8 * (r >> 2) + 4 * h;
      float v = ((red[((t * 4 + 0) * 16 + r) * 64 + lane] + red[((t * 4 + 1) * 16 + r) * 64 + lane]) +
                 red[((t * 4 + 2) * 16 + r) * 64 + lane]) + red[((t * 4 + 3) * 16 + r) * 64 + lane] + bsv;
      v = apply_act(v, a.act, a.slope);
      if (m < a.M && n < a.N) {
        if (a.res) v += a.res[(size_t)m * a.ldr + n];
        a.y[(size_t)m * a.ldy + n] = v;
      }
    }
  }
}

// Large-M variant (vposer encoder over 20 A rows, batched decoder output over 18 A rows): 64x64 output tile per workgroup,
// each wave one 32x32 quadrant over the FULL K.  A 32-wide k-block of X (64 rows) and of W (64 rows) is fetched once per
// workgroup (two 16-byte loads per thread and operand, waited for before anything else - no loads in flight under the
// MFMAs), parked in a double-buffered LDS tile (36-float pitch) and read by the two waves that share it: half the L2
// traffic per MFMA of the 32x32 split-K kernel, whose extra parallelism these shapes do not need.
__global__ __launch_bounds__(256) void egx_linear64_kernel(LinArgs a) {
  __shared__ __attribute__((aligned(16))) float tile[2][2][64 * 36];  // [buffer][X | W][row * 36 + k]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 31, h = lane >> 5;
  const int wr = wave >> 1, wc = wave & 1;
  int mt, nt;
  {
    const int MT = (a.M + 63) >> 6, NT = (a.N + 63) >> 6;
    const int bid = blockIdx.x, xcd = bid & 7, local = bid >> 3;
    const int per = (MT + 7) >> 3;          // M is the long axis here: every XCD owns a contiguous chunk of row tiles
    mt = xcd * per + local / NT;
    nt = local % NT;
    if (local >= per * NT || mt >= MT) return;
  }
  const int m0 = mt * 64, n0 = nt * 64;
  const int lrow = threadIdx.x >> 2, lk = (threadIdx.x & 3) * 8;  // loader role: row 0..63, k offset 0, 8, 16, 24
  const int xrow = min(m0 + lrow, a.M - 1), wrow = min(n0 + lrow, a.N - 1);
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const int nblk = (a.K + 31) >> 5;
  for (int b = 0; b < nblk; ++b) {
    const f32x4 x0 = load_x4(a, xrow, b * 32 + lk), x1 = load_x4(a, xrow, b * 32 + lk + 4);
    const f32x4 w0 = load_w4(a, wrow, b * 32 + lk), w1 = load_w4(a, wrow, b * 32 + lk + 4);
    float* xs = tile[b & 1][0];
    float* ws = tile[b & 1][1];
    *reinterpret_cast<f32x4*>(xs + lrow * 36 + lk) = x0;
    *reinterpret_cast<f32x4*>(xs + lrow * 36 + lk + 4) = x1;
    *reinterpret_cast<f32x4*>(ws + lrow * 36 + lk) = w0;
    *reinterpret_cast<f32x4*>(ws + lrow * 36 + lk + 4) = w1;
    __syncthreads();  // tile b visible; the other buffer (tile b-1) is free again once everyone has passed this point
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const f32x4 xa = *reinterpret_cast<const f32x4*>(xs + (wr * 32 + i) * 36 + c * 8 + 4 * h);
      const f32x4 wb = *reinterpret_cast<const f32x4*>(ws + (wc * 32 + i) * 36 + c * 8 + 4 * h);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[e], wb[e], acc, 0, 0, 0);
    }
  }
  const int n = n0 + wc * 32 + i;
  const float bsv = (a.bias && n < a.N) ? a.bias[n] : 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = m0 + wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
    float v = apply_act(acc[r] + bsv, a.act, a.slope);
    if (m < a.M && n < a.N) {
      if (a.res) v += a.res[(size_t)m * a.ldr + n];
      a.y[(size_t)m * a.ldy + n] = v;
    }
  }
}

// GRU gate math (torch.nn.GRU / GRUCell, gate order r,z,n): gi = x W_ih^T + b_ih, gh = h W_hh^T + b_hh
__global__ void egx_gru_pointwise_kernel(const float* __restrict__ gi, const float* __restrict__ gh,
                                         const float* __restrict__ gh_bias, const float* __restrict__ hprev, int ldh,
                                         float* __restrict__ hout, int ldo, int M, int H) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= M * H) return;
  const int m = idx / H, c = idx % H;
  const float* gim = gi + (size_t)m * 3 * H;
  const float* ghm = gh ? gh + (size_t)m * 3 * H : gh_bias;  // zero previous state: h W_hh^T + b_hh = b_hh
  const float r = 1.f / (1.f + expf(-(gim[c] + ghm[c])));
  const float z = 1.f / (1.f + expf(-(gim[H + c] + ghm[H + c])));
  const float nn = tanhf(gim[2 * H + c] + r * ghm[2 * H + c]);
  const float hp = hprev ? hprev[(size_t)m * ldh + c] : 0.f;
  hout[(size_t)m * ldo + c] = (1.f - z) * nn + z * hp;
}

__global__ void egx_cont6d_to_aa_kernel(const float* __restrict__ xb6, int n, float* __restrict__ out, int ldo) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n * 23) return;
  const int row = idx / 23, j = idx % 23;
  egx_cont6d_item(xb6 + (size_t)row * 159, out + (size_t)row * ldo, j);
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
static LinArgs make_lin_args(int M, int N, const EgxSeg* segs, int nseg, const float* W, int ldw, const float* b, int act,
                             float slope, const float* res, int ldr, float* out, int ldo) {
  LinArgs a;
  const float* ps[4] = {nullptr, nullptr, nullptr, nullptr};
  int ws[4] = {0, 0, 0, 0}, ls[4] = {0, 0, 0, 0};
  int K = 0;
  for (int s = 0; s < nseg; ++s) {
    ps[s] = segs[s].p; ws[s] = segs[s].w; ls[s] = segs[s].ld;
    K += segs[s].w;
  }
  a.xp0 = ps[0]; a.xp1 = ps[1]; a.xp2 = ps[2]; a.xp3 = ps[3];
  a.xw0 = ws[0]; a.xw1 = ws[1]; a.xw2 = ws[2]; a.xw3 = ws[3];
  a.xl0 = ls[0]; a.xl1 = ls[1]; a.xl2 = ls[2]; a.xl3 = ls[3];
  a.W = W; a.ldw = ldw > 0 ? ldw : K; a.bias = b; a.res = res; a.ldr = ldr; a.y = out; a.ldy = ldo;
  a.M = M; a.N = N; a.K = K; a.act = act; a.slope = slope;
  return a;
}
static int lin_blocks(const LinArgs& a, int nw) {
  const int MT = egx_ceil_div(a.M, 32), NT = egx_ceil_div(a.N, 32 * nw);
  return (NT >= MT) ? 8 * egx_ceil_div(NT, 8) * MT : 8 * egx_ceil_div(MT, 8) * NT;
}
// Form of a launch: direct operand loads up to K = 512 (one or two round trips per wave), LDS-staged coalesced loads above.
// (Two column tiles per workgroup - the NW = 2 instantiation - halve the workgroups of the 768-tile paired launches but
// double the MFMA chain of every wave: 17.7 -> 20.9 us at 512 x 1536 x 402, so one tile per workgroup it stays.  The direct
// form at K = 1152 with two / three blocks per trip: 39.7 / 40.3 us against 37.0 staged - that launch moves 170 MB through
// the L2s (576 tiles x two 32 x 1152 operand strips); a 64 x 64 tile per 8-wave workgroup (four quadrants x two halves of K)
// halves those bytes but makes every wave walk 18 blocks in 9 trips: 39.2 us against 28.9 - the trip count, not the byte
// count, is what such a launch pays for.)
static void launch_linear2(hipStream_t st, LinArgs2& two, bool pair) {
  const bool direct = std::max(two.p0.K, pair ? two.p1.K : 0) <= 512;
  two.blocks0 = lin_blocks(two.p0, 1);
  const int blocks = two.blocks0 + (pair ? lin_blocks(two.p1, 1) : 0);
  if (direct) hipLaunchKernelGGL((egx_linear_kernel<true, 1>), dim3(blocks), dim3(256), 0, st, two);
  else hipLaunchKernelGGL((egx_linear_kernel<false, 1>), dim3(blocks), dim3(256), 0, st, two);
}

int egx_launch_linear(hipStream_t st, int M, int N, const EgxSeg* segs, int nseg, const float* W, const float* b,
                      int act, float slope, const float* res, int ldr, float* out, int ldo) {
  LinArgs2 two;
  two.p0 = make_lin_args(M, N, segs, nseg, W, 0, b, act, slope, res, ldr, out, ldo);
  if (M >= 2048 && N >= 64) {  // enough row tiles to fill the chip without split-K: the 64x64 kernel
    const int MT = (M + 63) >> 6, NT = (N + 63) >> 6;
    const int blocks = 8 * ((MT + 7) >> 3) * NT;
    hipLaunchKernelGGL(egx_linear64_kernel, dim3(blocks), dim3(256), 0, st, two.p0);
    return EGX_OK;
  }
  two.p1 = two.p0;
  launch_linear2(st, two, false);
  return EGX_OK;
}

int egx_launch_gru_pointwise(hipStream_t st, const float* gi, const float* gh, const float* hprev, int ldh, float* hout,
                             int ldo, int M, int H) {
  hipLaunchKernelGGL(egx_gru_pointwise_kernel, dim3(egx_ceil_div(M * H, 256)), dim3(256), 0, st, gi, gh, nullptr, hprev, ldh,
                     hout, ldo, M, H);
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
    EGX_REQUIRE(segs[s].w >= 4 || s == d->num_segments - 1, "only the last input segment may be narrower than 4 columns");
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

__global__ void egx_frame_scan_kernel(float* __restrict__ y, const float* __restrict__ x_last, int x_ld, int A, int width, int T) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= A * width) return;
  const int a = idx / width, c = idx % width;
  float run = x_last[(size_t)a * x_ld + c];
  for (int t = 0; t < T; ++t) {
    float* p = y + ((size_t)t * A + a) * width + c;
    run = *p + run;  // d_t + y_(t-1), the order of `d_out(hfc) + y_p`
    *p = run;
  }
}

void egx_launch_frame_scan(hipStream_t st, float* y, const float* x_last, int x_ld, int A, int width, int T) {
  hipLaunchKernelGGL(egx_frame_scan_kernel, dim3(egx_ceil_div(A * width, 256)), dim3(256), 0, st, y, x_last, x_ld, A, width, T);
}
