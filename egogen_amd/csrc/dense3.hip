// Rollout dense layers on the bf16 matrix pipe with fp32-equivalent arithmetic ("dense3").
//
// The rollout networks (C-VAE decoder models/models_GAMMA_primitive.py:83-133, policy models/models_policy_ppo.py:24-39,
// 287-350) are chains of small dependent products (M = agents, K, N <= 1536): what a launch costs is its latency, and on
// gfx950 the fp32 MFMA runs at 1/16 of the bf16 rate.  Here every fp32 operand x is carried as three bf16 terms
// x = hi + mid + lo (24+ significant bits) and a product keeps the six partial products down to 2^-24 relative
// (mid.mid, hi.lo, lo.hi, hi.mid, mid.hi, hi.hi), accumulated in fp32 by v_mfma_f32_16x16x32_bf16 - the arithmetic of the
// LBS blend GEMM's three-plane mode (body_model.hip).  What makes it pay for latency-bound layers:
//   * operands live in HBM already split and in MFMA fragment order ("packed": [16-row tile][32-wide k-step][plane][lane]
//     16 bytes), so a wave's loads are whole contiguous KiB and no consumer spends VALU time on splitting;
//   * the PRODUCER of an activation writes that packed form from its epilogue (one split per element instead of one per
//     consuming workgroup), next to the fp32 row-major copy only where a non-GEMM consumer needs it;
//   * concatenated inputs ([hx | z], [x_enc | ego_enc | posenc]) are k-step ranges of one packed buffer: no copies;
//   * the GRU cell is ONE launch: a workgroup owns 32 rows x 16 hidden columns of all three gates on both sides
//     (x W_ih^T and h W_hh^T), so the gate math runs in its epilogue (was: paired GEMM launch + pointwise launch).
// Weights are packed once (motion prior: at load; policy: once per collect).
#include <mutex>
#include "egx_nets.h"

namespace {

typedef __bf16 bf16v8 __attribute__((ext_vector_type(8)));
typedef float f32x4a1 __attribute__((ext_vector_type(4), aligned(4)));

// x[8] -> three bf16 planes (v_cvt_pk_bf16_f32, round to nearest even; the residuals are exact in fp32)
__device__ __forceinline__ void d3_split(const float (&x)[8], bf16x8 (&pl)[3]) {
  float r[8];
#pragma unroll
  for (int p = 0; p < 3; ++p) {
    bf16v8 h;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float v = (p == 0) ? x[e] : r[e];
      h[e] = (__bf16)v;
      r[e] = v - (float)h[e];
    }
    pl[p] = __builtin_bit_cast(bf16x8, h);
  }
}

// acc += a . b with the six significant partial products, small ones first; `one`: the leading product only (operands
// rounded to bf16, fp32 accumulation - the "bf16 MFMA policy" of BASELINE config 5, egx_policy_set_precision)
__device__ __forceinline__ f32x4 d3_mma(const bf16x8 (&a)[3], const bf16x8 (&b)[3], f32x4 acc, bool one) {
  if (one) return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[2], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], b[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[0], acc, 0, 0, 0);
  return acc;
}

__device__ __forceinline__ float d3_act(float v, int act, float slope) {
  switch (act) {
    case 1: return tanhf(v);
    case 2: return fmaxf(v, 0.f);
    case 3: return v > 0.f ? v : v * slope;
    default: return v;
  }
}

// XCD-aware tile map (blocks are dealt round-robin to the 8 XCDs, each with a private L2): every XCD owns a contiguous chunk
// of column tiles - i.e. of the weights - and sweeps the row tiles.
__device__ __forceinline__ bool d3_tile(int bid, int MT, int NT, int& mt, int& nt) {
  const int xcd = bid & 7, local = bid >> 3;
  const int per = (NT + 7) >> 3;
  nt = xcd * per + local / MT;
  mt = local % MT;
  return local < per * MT && nt < NT;
}
__host__ __device__ inline int d3_blocks(int MT, int NT) { return 8 * ((NT + 7) / 8) * MT; }

}  // namespace

// ---------------------------------------------------------------------------------------------------------
// packing: fp32 rows [R, K] (leading dimension ld, starting at column col0) -> [2 ceil(R/32)][S][3][64] fragments at k-step
// offset s0 of a buffer with S_total k-steps per row tile.  Fragment lane l = (r & 15) + 16 ((k >> 3) & 3), element k & 7.
// Rows >= R and columns >= K are zero; the row-tile count is even so that a 32-row workgroup tile always finds both of its
// 16-row halves.  Up to four jobs per launch (the motion prior's x0 / x1 / z, the policy's two frames of state and egosensing).
// ---------------------------------------------------------------------------------------------------------
struct D3PackJob {
  const float* src;
  int R, K, ld, col0;
  bf16x8* dst;
  int S_total, s0;
};
struct D3PackJobs {
  D3PackJob j0, j1, j2, j3;
  int end0, end1, end2;   // running fragment counts: job i owns fragments [end(i-1), end(i))
};

__global__ __launch_bounds__(256) void egx_pack3_kernel(D3PackJobs jobs) {
  int frag = blockIdx.x * 4 + (threadIdx.x >> 6);   // (rt, s) of one of the jobs
  const int which = frag < jobs.end0 ? 0 : (frag < jobs.end1 ? 1 : (frag < jobs.end2 ? 2 : 3));
  const D3PackJob& j = which == 0 ? jobs.j0 : (which == 1 ? jobs.j1 : (which == 2 ? jobs.j2 : jobs.j3));
  frag -= which == 0 ? 0 : (which == 1 ? jobs.end0 : (which == 2 ? jobs.end1 : jobs.end2));
  const int lane = threadIdx.x & 63;
  const int RT = 2 * ((j.R + 31) >> 5), S = (j.K + 31) >> 5;
  if (!j.src || frag >= RT * S) return;
  const int rt = frag / S, s = frag % S;
  const int row = rt * 16 + (lane & 15), k0 = s * 32 + 8 * (lane >> 4);
  float x[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) x[e] = (row < j.R && k0 + e < j.K) ? j.src[(size_t)row * j.ld + j.col0 + k0 + e] : 0.f;
  bf16x8 pl[3];
  d3_split(x, pl);
  bf16x8* o = j.dst + ((size_t)rt * j.S_total + j.s0 + s) * 3 * 64 + lane;
#pragma unroll
  for (int p = 0; p < 3; ++p) o[p * 64] = pl[p];
}

static int d3_pack_frags(int R, int K) { return 2 * egx_ceil_div(R, 32) * egx_ceil_div(K, 32); }

void egx_launch_pack3(hipStream_t st, const D3Pack* jobs, int njobs) {
  D3PackJobs J;
  D3PackJob* dst[4] = {&J.j0, &J.j1, &J.j2, &J.j3};
  int frags[4] = {0, 0, 0, 0};
  for (int i = 0; i < 4; ++i) {
    D3PackJob& d = *dst[i];
    if (i < njobs) {
      d.src = jobs[i].src; d.R = jobs[i].R; d.K = jobs[i].K; d.ld = jobs[i].ld; d.col0 = jobs[i].col0;
      d.dst = static_cast<bf16x8*>(jobs[i].dst); d.S_total = jobs[i].S_total; d.s0 = jobs[i].s0;
      frags[i] = d3_pack_frags(d.R, d.K);
    } else {
      d.src = nullptr; d.R = d.K = d.ld = d.col0 = d.S_total = d.s0 = 0; d.dst = nullptr;
    }
  }
  J.end0 = frags[0]; J.end1 = J.end0 + frags[1]; J.end2 = J.end1 + frags[2];
  const int total = J.end2 + frags[3];
  hipLaunchKernelGGL(egx_pack3_kernel, dim3(egx_ceil_div(total, 4)), dim3(256), 0, st, J);
}

// ---------------------------------------------------------------------------------------------------------
// plain layer: out = act(A B^T + bias) + res for a 32 x 32 output tile per workgroup, the reduction split over the four
// waves; up to three independent layers may share a launch.
// ---------------------------------------------------------------------------------------------------------
struct D3Args3 {
  D3Plain p0, p1, p2;
  int end0, end1;   // blocks [0, end0) work on p0, [end0, end1) on p1, the rest on p2
};

template <int TRIP>
__global__ __launch_bounds__(256) void egx_dense3_kernel(D3Args3 three) {
  const int which = (int)blockIdx.x < three.end0 ? 0 : ((int)blockIdx.x < three.end1 ? 1 : 2);
  const D3Plain& a = which == 0 ? three.p0 : (which == 1 ? three.p1 : three.p2);
  const int bid = (int)blockIdx.x - (which == 0 ? 0 : (which == 1 ? three.end0 : three.end1));
  __shared__ __attribute__((aligned(16))) float red[4 * 16 * 64];
  __shared__ __attribute__((aligned(16))) float tile[32 * 36];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int MT = (a.M + 31) >> 5, NT = (a.N + 31) >> 5;
  const int per_batch = d3_blocks(MT, NT);
  const int batch = bid / per_batch;
  int mt, nt;
  if (!d3_tile(bid - batch * per_batch, MT, NT, mt, nt)) return;
  const bf16x8* Ab = a.A + (size_t)batch * a.batch_strideA;
  const int per = (a.S + 3) >> 2;
  const int s_lo = wave * per, s_hi = min(a.S, s_lo + per);
  f32x4 acc[2][2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
  // fragment streams of this tile: A row tiles 2 mt, 2 mt + 1 (k-steps sa0 + s of a buffer with SA per row tile);
  // B column tiles 2 nt, 2 nt + 1 (a column tile past N is all zeros in the packed weights: B is padded to 32 columns)
  const bf16x8* pa[2];
  const bf16x8* pb[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    pa[h] = Ab + ((size_t)(2 * mt + h) * a.SA + a.sa0) * 3 * 64 + lane;
    pb[h] = a.B + (size_t)(2 * nt + h) * a.S * 3 * 64 + lane;
  }
  for (int s = s_lo; s < s_hi; s += TRIP) {
    bf16x8 fa[TRIP][2][3], fb[TRIP][2][3];
#pragma unroll
    for (int u = 0; u < TRIP; ++u) {
      const int su = min(s + u, s_hi - 1);
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          fa[u][h][p] = pa[h][(size_t)(su * 3 + p) * 64];
          fb[u][h][p] = pb[h][(size_t)(su * 3 + p) * 64];
        }
    }
    // burst, wait, then only MFMAs: a wave that issues MFMAs with its own loads in flight runs the matrix pipe at about
    // half rate on this part (scripts/ubench/mfma_bf16.hip)
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < TRIP; ++u) {
      if (s + u >= s_hi) break;
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = d3_mma(fa[u][mi], fb[u][ni], acc[mi][ni], a.prec != 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  // split-K reduction through LDS; wave w then finishes MFMA tile (w >> 1, w & 1): bias, activation, residual
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[(wave * 16 + (mi * 2 + ni) * 4 + r) * 64 + lane] = acc[mi][ni][r];
  __syncthreads();
  {
    const int mi = wave >> 1, ni = wave & 1;
    const int col = 16 * ni + (lane & 15), n = nt * 32 + col;
    const float bsv = (a.bias && n < a.N) ? a.bias[n] : 0.f;
    const int row_base = batch * a.batch_rows_out;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int q = (mi * 2 + ni) * 4 + r;
      float v = ((red[(0 * 16 + q) * 64 + lane] + red[(1 * 16 + q) * 64 + lane]) + red[(2 * 16 + q) * 64 + lane]) +
                red[(3 * 16 + q) * 64 + lane] + bsv;
      v = d3_act(v, a.act, a.slope);
      const int row = 16 * mi + 4 * (lane >> 4) + r, m = mt * 32 + row;
      const bool live = m < a.M && n < a.N;
      if (live && a.res) v += a.res[(size_t)(row_base + m) * a.ldr + n];
      if (live && a.out) a.out[(size_t)(row_base + m) * a.ldo + n] = v;
      tile[row * 36 + col] = live ? v : 0.f;
    }
  }
  if (a.out3) {
    // the tile is k-step (s30 + nt) of the consumer's A operand: two 16-row fragments x three planes
    __syncthreads();
    if (wave < 2) {
      const int row = 16 * wave + (lane & 15), g = lane >> 4;
      const f32x4 x0 = *reinterpret_cast<const f32x4*>(&tile[row * 36 + 8 * g]), x1 = *reinterpret_cast<const f32x4*>(&tile[row * 36 + 8 * g + 4]);
      const float x[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
      bf16x8 pl[3];
      d3_split(x, pl);
      bf16x8* o = a.out3 + (size_t)batch * a.batch_stride3 + ((size_t)(2 * mt + wave) * a.S3 + a.s30 + nt) * 3 * 64 + lane;
#pragma unroll
      for (int p = 0; p < 3; ++p) o[p * 64] = pl[p];
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// GRU cell (torch.nn.GRU / GRUCell, gate order r, z, n) in one launch: workgroup = 32 rows x 16 hidden columns of all
// three gates on both sides,
//   gi = gi_in + Ai Bi^T + bias_i      (stored to gi_out when asked: the decoder keeps it as a running sum, prior.hip)
//   gh = Ah Bh^T + bias_h              (Ah null: zero previous state, gh = bias_h)
//   r = s(gi_r + gh_r), z = s(gi_z + gh_z), n = tanh(gi_n + r gh_n), h = (1 - z) n + z h_prev
// h is written fp32 row-major (the next cell's h_prev) and packed (the next products' A operand).
// ---------------------------------------------------------------------------------------------------------
template <int TRIP>
__global__ __launch_bounds__(256) void egx_gru3_kernel(D3Gru a) {
  extern __shared__ __attribute__((aligned(16))) float gsm[];
  float* red = gsm;                  // [4 waves][48][64]
  float* tile = gsm + 4 * 48 * 64;   // [32][20]: h of this workgroup's 32 x 16 block
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int MT = (a.M + 31) >> 5, CT = a.H >> 4;
  int mt, ct;
  if (!d3_tile(blockIdx.x, MT, CT, mt, ct)) return;
  f32x4 acc[2][3][2];   // [side][gate][row half]
#pragma unroll
  for (int sd = 0; sd < 2; ++sd)
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) acc[sd][g][mi] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int sd = 0; sd < 2; ++sd) {
    const bf16x8* A = sd ? a.Ah : a.Ai;
    if (!A) continue;
    const bf16x8* B = sd ? a.Bh : a.Bi;
    const int S = sd ? a.Sh : a.Si, SA = sd ? a.SAh : a.SAi, sa0 = sd ? a.sah0 : a.sai0;
    const int per = (S + 3) >> 2;
    const int s_lo = wave * per, s_hi = min(S, s_lo + per);
    const bf16x8* pa[2];
    const bf16x8* pb[3];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) pa[mi] = A + ((size_t)(2 * mt + mi) * SA + sa0) * 3 * 64 + lane;
#pragma unroll
    for (int g = 0; g < 3; ++g) pb[g] = B + (size_t)(g * CT + ct) * S * 3 * 64 + lane;
    for (int s = s_lo; s < s_hi; s += TRIP) {
      bf16x8 fa[TRIP][2][3], fb[TRIP][3][3];
#pragma unroll
      for (int u = 0; u < TRIP; ++u) {
        const int su = min(s + u, s_hi - 1);
#pragma unroll
        for (int p = 0; p < 3; ++p) {
#pragma unroll
          for (int mi = 0; mi < 2; ++mi) fa[u][mi][p] = pa[mi][(size_t)(su * 3 + p) * 64];
#pragma unroll
          for (int g = 0; g < 3; ++g) fb[u][g][p] = pb[g][(size_t)(su * 3 + p) * 64];
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < TRIP; ++u) {
        if (s + u >= s_hi) break;
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
          for (int mi = 0; mi < 2; ++mi) acc[sd][g][mi] = d3_mma(fa[u][mi], fb[u][g], acc[sd][g][mi], a.prec != 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#pragma unroll
  for (int sd = 0; sd < 2; ++sd)
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[(wave * 48 + ((sd * 3 + g) * 2 + mi) * 4 + r) * 64 + lane] = acc[sd][g][mi][r];
  __syncthreads();
  // wave w finishes positions (mi, r) = (w >> 1, 2 (w & 1) + {0, 1}) of every lane: all six gate values of an element in
  // one thread
  {
    const int mi = wave >> 1;
    const int col = lane & 15, c = ct * 16 + col;
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      const int r = 2 * (wave & 1) + rr;
      const int row = 16 * mi + 4 * (lane >> 4) + r, m = mt * 32 + row;
      float gv[2][3];
#pragma unroll
      for (int sd = 0; sd < 2; ++sd)
#pragma unroll
        for (int g = 0; g < 3; ++g) {
          const int q = ((sd * 3 + g) * 2 + mi) * 4 + r;
          gv[sd][g] = ((red[(0 * 48 + q) * 64 + lane] + red[(1 * 48 + q) * 64 + lane]) + red[(2 * 48 + q) * 64 + lane]) +
                      red[(3 * 48 + q) * 64 + lane];
        }
      float hv = 0.f;
      if (m < a.M) {
        float gi[3], gh[3];
#pragma unroll
        for (int g = 0; g < 3; ++g) {
          const int n = g * a.H + c;
          gi[g] = (a.gi_in ? a.gi_in[(size_t)m * 3 * a.H + n] : 0.f) + gv[0][g] + (a.bias_i ? a.bias_i[n] : 0.f);
          gh[g] = gv[1][g] + a.bias_h[n];
          if (a.gi_out) a.gi_out[(size_t)m * 3 * a.H + n] = gi[g];
        }
        const float rg = 1.f / (1.f + expf(-(gi[0] + gh[0])));
        const float zg = 1.f / (1.f + expf(-(gi[1] + gh[1])));
        const float ng = tanhf(gi[2] + rg * gh[2]);
        const float hp = a.h_prev ? a.h_prev[(size_t)m * a.ldh + c] : 0.f;
        hv = (1.f - zg) * ng + zg * hp;
        if (a.h_out) a.h_out[(size_t)m * a.ldo + c] = hv;
      }
      tile[row * 20 + col] = hv;
    }
  }
  if (a.h_out3) {
    // 16 columns = k groups 2 (ct & 1), 2 (ct & 1) + 1 of k-step s30 + ct / 2: half of the lanes of each fragment
    __syncthreads();
    if (wave < 2 && lane < 32) {
      const int row = 16 * wave + (lane & 15), g = lane >> 4;   // g in {0, 1}
      const f32x4 x0 = *reinterpret_cast<const f32x4*>(&tile[row * 20 + 8 * g]), x1 = *reinterpret_cast<const f32x4*>(&tile[row * 20 + 8 * g + 4]);
      const float x[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
      bf16x8 pl[3];
      d3_split(x, pl);
      bf16x8* o = a.h_out3 + ((size_t)(2 * mt + wave) * a.S3 + a.s30 + (ct >> 1)) * 3 * 64 + 32 * (ct & 1) + lane;
#pragma unroll
      for (int p = 0; p < 3; ++p) o[p * 64] = pl[p];
    }
  }
}

// positional_encoding (models_policy_ppo.py:276-285) of dist and time as the last 128 columns of the policy's [hx | he | pe]
// input: fp32 into `out` (row stride ld, the residual of the first MLP unit) and packed into k-steps s0 .. s0 + 3 of `out3`.
__global__ __launch_bounds__(256) void egx_posenc3_kernel(const float* __restrict__ dist, const float* __restrict__ time, int n,
                                                          float* __restrict__ out, int ld, bf16x8* __restrict__ out3, int S3, int s0) {
  const int frag = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int RT = 2 * ((n + 31) >> 5);
  if (frag >= RT * 4) return;
  const int rt = frag >> 2, s = frag & 3;
  const int row = rt * 16 + (lane & 15), c0 = s * 32 + 8 * (lane >> 4);
  float x[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = c0 + e;
    float v = 0.f;
    if (row < n) {
      const float f = ((c < 64) ? dist[row] : time[row]) * exp2f((float)((c & 63) >> 1));
      v = (c & 1) ? cosf(f) : sinf(f);
      out[(size_t)row * ld + c] = v;
    }
    x[e] = v;
  }
  bf16x8 pl[3];
  d3_split(x, pl);
  bf16x8* o = out3 + ((size_t)rt * S3 + s0 + s) * 3 * 64 + lane;
#pragma unroll
  for (int p = 0; p < 3; ++p) o[p * 64] = pl[p];
}
void egx_launch_posenc3(hipStream_t st, const float* dist, const float* time, int n, float* out, int ld, void* out3, int S3, int s0) {
  const int frags = 2 * egx_ceil_div(n, 32) * 4;
  hipLaunchKernelGGL(egx_posenc3_kernel, dim3(egx_ceil_div(frags, 4)), dim3(256), 0, st, dist, time, n, out, ld,
                     static_cast<bf16x8*>(out3), S3, s0);
}

// ---- launchers ------------------------------------------------------------------------------------------
static void d3_launch_plain(hipStream_t st, D3Args3& three, int n) {
  auto blocks = [](const D3Plain& p) { return d3_blocks((p.M + 31) >> 5, (p.N + 31) >> 5) * std::max(1, p.batches); };
  three.end0 = blocks(three.p0);
  three.end1 = three.end0 + (n > 1 ? blocks(three.p1) : 0);
  const int total = three.end1 + (n > 2 ? blocks(three.p2) : 0);
  const int smax = std::max(three.p0.S, std::max(n > 1 ? three.p1.S : 0, n > 2 ? three.p2.S : 0));
  if (smax > 16) hipLaunchKernelGGL(egx_dense3_kernel<3>, dim3(total), dim3(256), 0, st, three);
  else hipLaunchKernelGGL(egx_dense3_kernel<2>, dim3(total), dim3(256), 0, st, three);
}
void egx_launch_dense3(hipStream_t st, const D3Plain& p) {
  D3Args3 t;
  t.p0 = p; t.p1 = p; t.p2 = p;
  d3_launch_plain(st, t, 1);
}
void egx_launch_dense3_pair(hipStream_t st, const D3Plain& p, const D3Plain& q) {
  D3Args3 t;
  t.p0 = p; t.p1 = q; t.p2 = q;
  d3_launch_plain(st, t, 2);
}
void egx_launch_dense3_triple(hipStream_t st, const D3Plain& p, const D3Plain& q, const D3Plain& r) {
  D3Args3 t;
  t.p0 = p; t.p1 = q; t.p2 = r;
  d3_launch_plain(st, t, 3);
}
int egx_launch_gru3(hipStream_t st, const D3Gru& g) {
  constexpr size_t lds = (size_t)(4 * 48 * 64 + 32 * 20) * sizeof(float);   // 50.5 KiB: within the default dynamic-LDS cap
  const int blocks = d3_blocks((g.M + 31) >> 5, g.H >> 4);
  hipLaunchKernelGGL(egx_gru3_kernel<2>, dim3(blocks), dim3(256), lds, st, g);
  return EGX_OK;
}

// ---- C ABI: packing (weights once, raw network inputs per call) -------------------------------------------
extern "C" size_t egx_pack3_bytes(int num_rows, int num_cols) {
  if (num_rows <= 0 || num_cols <= 0) return 0;
  return (size_t)d3_pack_frags(num_rows, num_cols) * 3 * 64 * 16;
}
extern "C" int egx_pack3(const float* src, int num_rows, int num_cols, int src_ld, int src_col0, void* dst, int dst_ksteps,
                         int dst_kstep0, void* stream) {
  EGX_REQUIRE(src && dst && num_rows > 0 && num_cols > 0 && src_ld >= src_col0 + num_cols, "bad arguments");
  const int S = egx_ceil_div(num_cols, 32);
  EGX_REQUIRE(dst_kstep0 >= 0 && dst_ksteps >= dst_kstep0 + S, "destination k-step range too small");
  D3Pack job{src, num_rows, num_cols, src_ld, src_col0, dst, dst_ksteps, dst_kstep0};
  egx_launch_pack3(static_cast<hipStream_t>(stream), &job, 1);
  EGX_HIP_CHECK(hipGetLastError());
  return EGX_OK;
}
