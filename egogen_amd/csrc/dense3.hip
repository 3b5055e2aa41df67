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
// Weights are packed once (motion prior: at load; policy: after every optimiser step, update3.hip).
// Also here, built on the same packed operands: the fused body regressor (egx_regressor3_kernel) and the fused VPoser encoder
// (egx_vposer3_kernel) - whole networks per launch with their activations as packed planes in LDS.
#include <cstddef>
#include <cstdlib>
#include <mutex>
#include "egx_nets.h"

namespace {

typedef __bf16 bf16v8 __attribute__((ext_vector_type(8)));
typedef float f32x4a1 __attribute__((ext_vector_type(4), aligned(4)));

// Arithmetic of a product ("prec" of D3Plain / D3Gru) = how many of the bf16 planes of each operand take part:
//   prec 0: three planes, six partial products (2^-24 relative: fp32-equivalent)
//   prec 2: two planes (hi, mid: 16 significant bits per operand), three partial products - the LBS blend GEMM's default mode
//   prec 1: the leading plane only (operands rounded to bf16), one product - "bf16 MFMA, fp32 accumulate"
// Accumulation, biases, activations and every fp32 output are the same in all three.  Images always have room for three
// planes; a layer only READS the planes its mode uses and only WRITES those planes of the activation images it produces
// (weight images and raw-input images always carry all three: they are shared with launches of other modes).
__host__ __device__ constexpr int d3_planes(int prec) { return prec == 0 ? 3 : (prec == 2 ? 2 : 1); }

// x[8] -> NP bf16 planes (v_cvt_pk_bf16_f32, round to nearest even; the residuals are exact in fp32)
template <int NP = 3>
__device__ __forceinline__ void d3_split(const float (&x)[8], bf16x8 (&pl)[NP]) {
  float r[8];
#pragma unroll
  for (int p = 0; p < NP; ++p) {
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

// acc += a . b with the significant partial products of the mode, small ones first (NPL = 3: mid.mid, hi.lo, lo.hi, hi.mid,
// mid.hi, hi.hi; NPL = 2: hi.mid, mid.hi, hi.hi; NPL = 1: hi.hi).  For a block of MI x NI output tiles, product-major:
// consecutive MFMAs go to different accumulators, so none waits for the previous one's result.
template <int MI, int NI, int NPL>
__device__ __forceinline__ void d3_mma_tiles(const bf16x8 (&fa)[MI][NPL], const bf16x8 (&fb)[NI][NPL], f32x4 (&acc)[MI][NI]) {
  constexpr int NPROD = NPL == 3 ? 6 : (NPL == 2 ? 3 : 1);
#pragma unroll
  for (int pr = 0; pr < NPROD; ++pr) {
    int pa, pb;
    if (NPL == 3) {
      pa = (pr == 0) ? 1 : (pr == 1) ? 0 : (pr == 2) ? 2 : (pr == 3) ? 0 : (pr == 4) ? 1 : 0;
      pb = (pr == 0) ? 1 : (pr == 1) ? 2 : (pr == 2) ? 0 : (pr == 3) ? 1 : (pr == 4) ? 0 : 0;
    } else if (NPL == 2) {
      pa = (pr == 0) ? 0 : (pr == 1) ? 1 : 0;
      pb = (pr == 0) ? 1 : (pr == 1) ? 0 : 0;
    } else {
      pa = pb = 0;
    }
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[mi][pa], fb[ni][pb], acc[mi][ni], 0, 0, 0);
  }
}

__device__ __forceinline__ float d3_act(float v, int act, float slope) {
  switch (act) {
    case 1: return tanhf(v);
    case 2: return fmaxf(v, 0.f);
    case 3: return v > 0.f ? v : v * slope;
    default: return v;
  }
}

// derivative of the activation as a function of its OUTPUT a = act(z) (tanh: 1 - a^2; relu / leaky relu keep the sign of z)
__device__ __forceinline__ float d3_act_grad(float a, int act, float slope) {
  switch (act) {
    case 1: return 1.f - a * a;
    case 2: return a > 0.f ? 1.f : 0.f;
    case 3: return a > 0.f ? 1.f : slope;
    default: return 1.f;
  }
}

// XCD-aware tile map (blocks are dealt round-robin to the 8 XCDs, each with a private L2): every XCD owns a contiguous chunk
// of column tiles - i.e. of the weights - and sweeps the row tiles.
// With rowmap every XCD owns a chunk of ROW tiles - of the activations - and sweeps the weights instead.  An XCD's L2 is filled
// with all of the operand it sweeps and an eighth of the one it owns, so the map follows the larger operand: rows when M >= N
// (the decoder's 512-row layers: 8.1 -> 6.9 us), columns otherwise (the policy's 256 x 1152 layers: 18.4 against 19.8 us with
// rows).  mode < 0: that rule; 0 / 1: columns / rows always (EGX_D3_ROWMAP, development).
__host__ __device__ inline int d3_rowmap(int mode, int M, int N) { return mode < 0 ? (M >= N ? 1 : 0) : mode; }
__device__ __forceinline__ bool d3_tile(int bid, int MT, int NT, int& mt, int& nt, int rowmap = 0) {
  const int xcd = bid & 7, local = bid >> 3;
  if (rowmap) {
    const int per = (MT + 7) >> 3;
    mt = xcd * per + local / NT;
    nt = local % NT;
    return local < per * NT && mt < MT;
  }
  const int per = (NT + 7) >> 3;
  nt = xcd * per + local / MT;
  mt = local % MT;
  return local < per * MT && nt < NT;
}
__host__ __device__ inline int d3_blocks(int MT, int NT, int rowmap = 0) {
  return rowmap ? 8 * ((MT + 7) / 8) * NT : 8 * ((NT + 7) / 8) * MT;
}

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
  int transpose;   // 1: the source block is [K rows (the reduction index), R columns]: image rows = source columns
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
  if (!j.transpose) {
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = (row < j.R && k0 + e < j.K) ? j.src[(size_t)row * j.ld + j.col0 + k0 + e] : 0.f;
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = (row < j.R && k0 + e < j.K) ? j.src[(size_t)(k0 + e) * j.ld + j.col0 + row] : 0.f;
  }
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
      d.transpose = jobs[i].transpose;
      frags[i] = d3_pack_frags(d.R, d.K);
    } else {
      d.src = nullptr; d.R = d.K = d.ld = d.col0 = d.S_total = d.s0 = d.transpose = 0; d.dst = nullptr;
    }
  }
  J.end0 = frags[0]; J.end1 = J.end0 + frags[1]; J.end2 = J.end1 + frags[2];
  const int total = J.end2 + frags[3];
  hipLaunchKernelGGL(egx_pack3_kernel, dim3(egx_ceil_div(total, 4)), dim3(256), 0, st, J);
}

// One output element after the reduction (bias already added): activation, saved activation, residual, fp32 stores, and the
// value that goes into the packed images (0 outside the matrix).
__device__ __forceinline__ float d3_finish(const D3Plain& a, float v, int m, int n, int row_base) {
  v = d3_act(v, a.act, a.slope);
  const bool live = m < a.M && n < a.N;
  if (live && a.out_act) a.out_act[(size_t)m * a.ldact + n] = v;   // activation before the skip connection (saved for backward)
  if (live && a.res && !(a.n_split > 0 && n >= a.n_split)) v += a.res[(size_t)(row_base + m) * a.ldr + n];
  if (live && a.n_split > 0 && n >= a.n_split) {
    // weight-gradient launch: the B operand's extra row of ones makes column n_split the bias gradient
    if (n == a.n_split && a.bias_out) a.bias_out[m] = v;
  } else if (live && a.out) {
    a.out[(size_t)(row_base + m) * a.ldo + n] = v;
  }
  // gradient launches: what goes on to the next products is v x act'(saved activation of the layer below)
  if (live && a.dact) v *= d3_act_grad(a.dact[(size_t)m * a.lddact + n], a.dact_code, a.dact_slope);
  return live ? v : 0.f;
}

// Packed images of a finished tile (values in `tile`, pitch TN + 4).  Row-major image (the consumer's A operand): 16-row tiles
// MI mt + i, k-steps s30 + nt NI/2 + j.  Transposed image (rows = this layer's columns, reduction index = its rows - what a
// weight-gradient product reads): row tiles NI nt + j, k-steps s3T0 + mt MI/2 + i.  One wave per fragment.
template <int MI, int NI, int NW, int NPL>
__device__ __forceinline__ void d3_write_packed(const D3Plain& a, const float* tile, int mt, int nt, int batch, int wave, int lane) {
  constexpr int PITCH = 16 * NI + 4;
  constexpr int NR = MI * (NI / 2), NTT = NI * (MI / 2);
  const int ntask = (a.out3 ? NR : 0) + (a.out3T ? NTT : 0);
  const int m_ksteps = (a.M + 31) >> 5, n_ksteps = (a.N + 31) >> 5;   // extents of the images (even tile counts)
  for (int task = wave; task < ntask; task += NW) {
    float x[8];
    bf16x8* o;
    if (a.out3 && task < NR) {
      const int i = task / (NI / 2), j = task % (NI / 2);
      if (MI * mt + i >= 2 * m_ksteps || nt * (NI / 2) + j >= n_ksteps) continue;   // past the image (ragged last tile)
      const int row = 16 * i + (lane & 15), g = lane >> 4;
      const f32x4 x0 = *reinterpret_cast<const f32x4*>(&tile[row * PITCH + 32 * j + 8 * g]), x1 = *reinterpret_cast<const f32x4*>(&tile[row * PITCH + 32 * j + 8 * g + 4]);
      x[0] = x0[0]; x[1] = x0[1]; x[2] = x0[2]; x[3] = x0[3]; x[4] = x1[0]; x[5] = x1[1]; x[6] = x1[2]; x[7] = x1[3];
      o = a.out3 + (size_t)batch * a.batch_stride3 + ((size_t)(MI * mt + i) * a.S3 + a.s30 + nt * (NI / 2) + j) * 3 * 64 + lane;
    } else {
      const int tt = task - (a.out3 ? NR : 0);
      const int j = tt / (MI / 2), i = tt % (MI / 2);
      if (NI * nt + j >= 2 * n_ksteps || mt * (MI / 2) + i >= m_ksteps) continue;
      const int c = 16 * j + (lane & 15), kg = lane >> 4;
#pragma unroll
      for (int e = 0; e < 8; ++e) x[e] = tile[(32 * i + 8 * kg + e) * PITCH + c];
      o = a.out3T + ((size_t)(NI * nt + j) * a.S3T + a.s3T0 + mt * (MI / 2) + i) * 3 * 64 + lane;
    }
    bf16x8 pl[NPL];
    d3_split<NPL>(x, pl);
#pragma unroll
    for (int p = 0; p < NPL; ++p) o[p * 64] = pl[p];
  }
}

// ---------------------------------------------------------------------------------------------------------
// plain layer: out = act(A B^T + bias) + res for a 32 x 32 output tile per workgroup, the reduction split over the four
// waves; up to three independent layers may share a launch.
// ---------------------------------------------------------------------------------------------------------
struct D3Args4 {
  D3Plain p0, p1, p2, p3;
  int end0, end1, end2;   // blocks [0, end0) work on p0, [end0, end1) on p1, [end1, end2) on p2, the rest on p3
  int rowmap;
};

// The layer a block works on.  Picking one of the four structs by reference (`which == 0 ? four.p0 : ...`) makes the
// compiler copy all of the kernel arguments to scratch in every wave and read the fields back with vector loads; selecting
// field by field keeps them in SGPRs but loads all four structs.  Reading the one struct straight from the kernel-argument
// segment (constant address space, uniform offset: scalar loads) touches only what is used.  D3Args4 is the kernels' only
// argument, so p0 sits at offset 0 of the segment.
__device__ __forceinline__ D3Plain d3_pick(int which) {
  D3Plain a;
  static_assert(offsetof(D3Args4, p0) == 0 && offsetof(D3Args4, p1) == sizeof(D3Plain) && sizeof(D3Plain) % 4 == 0, "layout");
#if defined(__HIP_DEVICE_COMPILE__)
  typedef const __attribute__((address_space(4))) char* kptr;
  typedef const __attribute__((address_space(4))) int* iptr;
  iptr src = (iptr)((kptr)__builtin_amdgcn_kernarg_segment_ptr() + (size_t)which * sizeof(D3Plain));
  int* dst = reinterpret_cast<int*>(&a);
#pragma unroll
  for (int i = 0; i < (int)(sizeof(D3Plain) / 4); ++i) dst[i] = src[i];
#else
  (void)which;
#endif
  return a;
}

// MI x NI MFMA tiles of 16 x 16 per workgroup of NW waves (instantiated: 2 x 2 tiles; four waves, eight for 9..16 k-steps).
template <int TRIP, int MI, int NI, int NW, int NPL>
__global__ __launch_bounds__(64 * NW) void egx_dense3_kernel(D3Args4 four) {
  constexpr int TM = 16 * MI, TN = 16 * NI, NACC = MI * NI * 4, PITCH = TN + 4;
  const int bx = (int)blockIdx.x;
  const int which = bx < four.end0 ? 0 : (bx < four.end1 ? 1 : (bx < four.end2 ? 2 : 3));
  const D3Plain a = d3_pick(which);
  const int bid = bx - (which == 0 ? 0 : (which == 1 ? four.end0 : (which == 2 ? four.end1 : four.end2)));
  extern __shared__ __attribute__((aligned(16))) float d3_smem[];
  float* red = d3_smem;                      // [NW waves][NACC][64]
  float* tile = d3_smem + NW * NACC * 64;    // [TM][PITCH]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int MT = (a.M + TM - 1) / TM, NT = (a.N + TN - 1) / TN;
  const int rowmap = d3_rowmap(four.rowmap, a.M, a.N);
  const int per_batch = d3_blocks(MT, NT, rowmap);
  const int batch = bid / per_batch;
  int mt, nt;
  if (!d3_tile(bid - batch * per_batch, MT, NT, mt, nt, rowmap)) return;
  const bf16x8* Ab = a.A + (size_t)batch * a.batch_strideA;
  const int per = (a.S + NW - 1) / NW;
  const int s_lo = wave * per, s_hi = min(a.S, s_lo + per);
  f32x4 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
  // fragment streams of this tile: A row tiles MI mt + i (k-steps sa0 + s of a buffer with SA per row tile); B column tiles
  // NI nt + j, clamped to the image (packed images hold an even number of 16-row tiles; columns past N are never stored)
  const int a_tiles = 2 * ((a.M + 31) >> 5), b_tiles = 2 * ((a.N + 31) >> 5);
  const bf16x8* pa[MI];
  const bf16x8* pb[NI];
#pragma unroll
  for (int i = 0; i < MI; ++i) pa[i] = Ab + ((size_t)min(MI * mt + i, a_tiles - 1) * a.SA + a.sa0) * 3 * 64 + lane;
#pragma unroll
  for (int j = 0; j < NI; ++j) pb[j] = a.B + (size_t)min(NI * nt + j, b_tiles - 1) * a.S * 3 * 64 + lane;
  for (int s = s_lo; s < s_hi; s += TRIP) {
    bf16x8 fa[TRIP][MI][NPL], fb[TRIP][NI][NPL];
#pragma unroll
    for (int u = 0; u < TRIP; ++u) {
      const int su = min(s + u, s_hi - 1);
#pragma unroll
      for (int p = 0; p < NPL; ++p) {
#pragma unroll
        for (int i = 0; i < MI; ++i) fa[u][i][p] = pa[i][(size_t)(su * 3 + p) * 64];
#pragma unroll
        for (int j = 0; j < NI; ++j) fb[u][j][p] = pb[j][(size_t)(su * 3 + p) * 64];
      }
    }
    // burst, wait, then only MFMAs: a wave that issues MFMAs with its own loads in flight runs the matrix pipe at about
    // half rate on this part (scripts/ubench/mfma_bf16.hip)
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < TRIP; ++u)
      if (s + u < s_hi) d3_mma_tiles<MI, NI, NPL>(fa[u], fb[u], acc);
    __builtin_amdgcn_sched_barrier(0);
  }
  // split-K reduction through LDS; wave w then finishes MFMA tiles w, w + NW, ...: bias, activation, residual
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[(wave * NACC + (mi * NI + ni) * 4 + r) * 64 + lane] = acc[mi][ni][r];
  __syncthreads();
  const int row_base = batch * a.batch_rows_out;
#pragma unroll
  for (int t0 = 0; t0 < MI * NI; t0 += NW) {
    const int t = t0 + wave;
    if (t >= MI * NI) continue;
    const int mi = t / NI, ni = t % NI;
    const int col = 16 * ni + (lane & 15), n = nt * TN + col;
    const float bsv = (a.bias && n < a.N) ? a.bias[n] : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int q = t * 4 + r;
      float v = (red[(0 * NACC + q) * 64 + lane] + red[(1 * NACC + q) * 64 + lane]) + red[(2 * NACC + q) * 64 + lane];
#pragma unroll
      for (int w2 = 3; w2 < NW; ++w2) v += red[(w2 * NACC + q) * 64 + lane];
      const int row = 16 * mi + 4 * (lane >> 4) + r, m = mt * TM + row;
      tile[row * PITCH + col] = d3_finish(a, v + bsv, m, n, row_base);
    }
  }
  if (!a.out3 && !a.out3T) return;
  __syncthreads();
  d3_write_packed<MI, NI, NW, NPL>(a, tile, mt, nt, batch, wave, lane);
}

// ---------------------------------------------------------------------------------------------------------
// GRU cell (torch.nn.GRU / GRUCell, gate order r, z, n) in one launch: workgroup = 32 rows x 16 hidden columns of all
// three gates on both sides,
//   gi = gi_in + Ai Bi^T + bias_i      (stored to gi_out when asked: the decoder keeps it as a running sum, prior.hip)
//   gh = Ah Bh^T + bias_h              (Ah null: zero previous state, gh = bias_h)
//   r = s(gi_r + gh_r), z = s(gi_z + gh_z), n = tanh(gi_n + r gh_n), h = (1 - z) n + z h_prev
// h is written fp32 row-major (the next cell's h_prev) and packed (the next products' A operand).
// ---------------------------------------------------------------------------------------------------------
struct D3Gru2 {
  D3Gru g0, g1;
  int blocks0;   // blocks [0, blocks0) work on g0, the rest on g1 (the policy's two encoders)
  int rowmap;
};
template <int TRIP, int NPL>
__global__ __launch_bounds__(512) void egx_gru3_kernel(D3Gru2 two) {
  // eight waves: waves 0-3 split the reduction of the x side (x W_ih^T), waves 4-7 that of the h side (h W_hh^T) - the two
  // products are independent, so their operand bursts are in flight together and a cell whose sides are each <= 4 TRIP
  // k-steps deep (the decoder cell: 8 + 8) is ONE memory round trip instead of two
  const bool second = (int)blockIdx.x >= two.blocks0;
  const D3Gru& a = second ? two.g1 : two.g0;
  const int gru_bid = second ? (int)blockIdx.x - two.blocks0 : (int)blockIdx.x;
  extern __shared__ __attribute__((aligned(16))) float gsm[];
  float* red = gsm;                  // [8 waves][24][64]
  float* tile = gsm + 8 * 24 * 64;   // [32][20]: h of this workgroup's 32 x 16 block
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int MT = (a.M + 31) >> 5, CT = a.H >> 4;
  int mt, ct;
  if (!d3_tile(gru_bid, MT, CT, mt, ct, d3_rowmap(two.rowmap, a.M, 3 * a.H))) return;
  const int sd = wave >> 2, w4 = wave & 3;
  f32x4 acc[2][3];   // [row half][gate] of this wave's side
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int g = 0; g < 3; ++g) acc[mi][g] = f32x4{0.f, 0.f, 0.f, 0.f};
  {
    const bf16x8* A = sd ? a.Ah : a.Ai;
    const bf16x8* B = sd ? a.Bh : a.Bi;
    const int S = A ? (sd ? a.Sh : a.Si) : 0, SA = sd ? a.SAh : a.SAi, sa0 = sd ? a.sah0 : a.sai0;
    const int per = (S + 3) >> 2;
    const int s_lo = w4 * per, s_hi = min(S, s_lo + per);
    const bf16x8* pa[2];
    const bf16x8* pb[3];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) pa[mi] = A + ((size_t)(2 * mt + mi) * SA + sa0) * 3 * 64 + lane;
#pragma unroll
    for (int g = 0; g < 3; ++g) pb[g] = B + (size_t)(g * CT + ct) * S * 3 * 64 + lane;
    for (int s = s_lo; s < s_hi; s += TRIP) {
      bf16x8 fa[TRIP][2][NPL], fb[TRIP][3][NPL];
#pragma unroll
      for (int u = 0; u < TRIP; ++u) {
        const int su = min(s + u, s_hi - 1);
#pragma unroll
        for (int p = 0; p < NPL; ++p) {
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
        d3_mma_tiles<2, 3, NPL>(fa[u], fb[u], acc);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#pragma unroll
  for (int g = 0; g < 3; ++g)
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[(wave * 24 + (g * 2 + mi) * 4 + r) * 64 + lane] = acc[mi][g][r];
  __syncthreads();
  // wave w finishes position (mi, r) = (w >> 2, w & 3) of every lane: all six gate values of an element in one thread
  {
    const int mi = wave >> 2, r = wave & 3;
    const int col = lane & 15, c = ct * 16 + col;
    const int row = 16 * mi + 4 * (lane >> 4) + r, m = mt * 32 + row;
    float gv[2][3];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int g = 0; g < 3; ++g) {
        const int q = (g * 2 + mi) * 4 + r;
        gv[s2][g] = ((red[((s2 * 4 + 0) * 24 + q) * 64 + lane] + red[((s2 * 4 + 1) * 24 + q) * 64 + lane]) +
                     red[((s2 * 4 + 2) * 24 + q) * 64 + lane]) + red[((s2 * 4 + 3) * 24 + q) * 64 + lane];
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
        if (a.gh_out) a.gh_out[(size_t)m * 3 * a.H + n] = gh[g];
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
  if (a.h_out3 || a.h_out3T) __syncthreads();
  if (a.h_out3T && wave == 2) {
    // transposed image (rows = hidden columns, reduction index = batch rows): one whole fragment, row tile col0T / 16 + ct
    const int c = lane & 15, kg = lane >> 4;
    float x[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = tile[(8 * kg + e) * 20 + c];
    bf16x8 pl[NPL];
    d3_split<NPL>(x, pl);
    bf16x8* o = a.h_out3T + ((size_t)((a.col0T >> 4) + ct) * a.S3T + a.s3T0 + mt) * 3 * 64 + lane;
#pragma unroll
    for (int p = 0; p < NPL; ++p) o[p * 64] = pl[p];
  }
  if (a.h_out3) {
    // 16 columns = k groups 2 (ct & 1), 2 (ct & 1) + 1 of k-step s30 + ct / 2: half of the lanes of each fragment
    if (wave < 2 && lane < 32) {
      const int row = 16 * wave + (lane & 15), g = lane >> 4;   // g in {0, 1}
      const f32x4 x0 = *reinterpret_cast<const f32x4*>(&tile[row * 20 + 8 * g]), x1 = *reinterpret_cast<const f32x4*>(&tile[row * 20 + 8 * g + 4]);
      const float x[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
      bf16x8 pl[NPL];
      d3_split<NPL>(x, pl);
      bf16x8* o = a.h_out3 + ((size_t)(2 * mt + wave) * a.S3 + a.s30 + (ct >> 1)) * 3 * 64 + 32 * (ct & 1) + lane;
#pragma unroll
      for (int p = 0; p < NPL; ++p) o[p * 64] = pl[p];
    }
  }
}

// positional_encoding (models_policy_ppo.py:276-285) of dist and time as the last 128 columns of the policy's [hx | he | pe]
// input: fp32 into `out` (row stride ld, the residual of the first MLP unit) and packed into k-steps s0 .. s0 + 3 of `out3`.
__global__ __launch_bounds__(256) void egx_posenc3_kernel(const float* __restrict__ dist, const float* __restrict__ time, int n,
                                                          float* __restrict__ out, int ld, bf16x8* __restrict__ out3, int S3, int s0,
                                                          bf16x8* __restrict__ out3T, int S3T, int col0T, float* __restrict__ zero6) {
  // (the update chain's loss kernel accumulates six sums with atomics: cleared here, several launches ahead of it, instead of
  // by a launch of their own)
  if (zero6 && blockIdx.x == 0 && threadIdx.x < 6) zero6[threadIdx.x] = 0.f;
  int frag = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int RT = 2 * ((n + 31) >> 5);
  if (frag >= RT * 4) {
    // training: the same 128 columns as rows col0T .. col0T + 127 of the transposed image (reduction index = batch row)
    frag -= RT * 4;
    const int Sn = (n + 31) >> 5;
    if (!out3T || frag >= 8 * Sn) return;
    const int t = frag / Sn, s = frag % Sn;
    const int c = 16 * t + (lane & 15), m0 = 32 * s + 8 * (lane >> 4);
    float x[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int m = m0 + e;
      float v = 0.f;
      if (m < n) {
        const float f = ((c < 64) ? dist[m] : time[m]) * exp2f((float)((c & 63) >> 1));
        v = (c & 1) ? cosf(f) : sinf(f);
      }
      x[e] = v;
    }
    bf16x8 pl[3];
    d3_split(x, pl);
    bf16x8* o = out3T + ((size_t)((col0T >> 4) + t) * S3T + s) * 3 * 64 + lane;
#pragma unroll
    for (int p = 0; p < 3; ++p) o[p * 64] = pl[p];
    return;
  }
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
void egx_launch_posenc3(hipStream_t st, const float* dist, const float* time, int n, float* out, int ld, void* out3, int S3, int s0,
                        void* out3T, int S3T, int col0T, float* zero6) {
  const int frags = 2 * egx_ceil_div(n, 32) * 4 + (out3T ? 8 * egx_ceil_div(n, 32) : 0);
  hipLaunchKernelGGL(egx_posenc3_kernel, dim3(egx_ceil_div(frags, 4)), dim3(256), 0, st, dist, time, n, out, ld,
                     static_cast<bf16x8*>(out3), S3, s0, static_cast<bf16x8*>(out3T), S3T, col0T, zero6);
}

// ---------------------------------------------------------------------------------------------------------
// Fused body regressor on the bf16 matrix pipe: MoshRegressor.forward (models_GAMMA_primitive.py:222-301) for 16 NRT rows per
// workgroup - all 3 recurrences x (in_fc + 10 residual blocks + out_fc) and the 6D -> axis-angle tail in ONE launch.
//   * 48 rows (NRT = 3): 18 x 512 = 9216 rows give 192 workgroups, one per CU and all equally loaded (the fp32 kernel's 288
//     32-row workgroups put two on 32 of the 256 CUs, which then set the launch time); smaller batches take 32 or 16 rows per
//     workgroup so that more CUs work (a workgroup's time follows its own rows, not the chip's load);
//   * eight waves, wave w owns output columns 16 w .. 16 w + 15 of every 128-wide layer for all three 16-row tiles; the
//     activations live in LDS as packed planes (operand fragments: 16 rows x 32 reduction indices, eight consecutive indices per
//     lane).  Every product is taken TRANSPOSED, out^T = W act^T (the weight fragment is the MFMA's first operand, the
//     activation fragment its second): a lane's four accumulator registers are then FOUR CONSECUTIVE COLUMNS of ONE row - half
//     of the eight-index group of a fragment lane of the next layer - so the epilogue splits its own values and stores them
//     with one 8-byte LDS write per plane and row tile: no transposition buffer, no cross-lane traffic, all 64 lanes at work
//     (the row-major form needed a strip round trip through LDS and split on half the lanes);
//   * the residual stream h stays in registers (a lane owns the same elements in every layer);
//   * in_fc([markers | xb | betas]) = W_m markers + W_b betas + b (the same in all three recurrences: computed once, kept
//     in registers) + W_xb xb (159 of the 370 columns, zero in the first recurrence);
//   * weights are read from packed images (one contiguous KiB per fragment), the next layer's prefetched under the epilogue.
// ---------------------------------------------------------------------------------------------------------
namespace {
constexpr int R3_XBP = 164, R3_NOUT = 159;
// NRT 16-row tiles per workgroup: 3 (48 rows) when the rows fill the chip that way, fewer for small batches (see the launcher)
constexpr int r3_act_frags(int nrt) { return nrt * 4 * 3 * 64; }   // one (16 NRT) x 128 activation buffer, in bf16x8 fragments-lanes
constexpr size_t r3_lds(int nrt) { return (size_t)16 * nrt * R3_XBP * 4 + 2 * (size_t)r3_act_frags(nrt) * 16; }   // 102.75 KiB at NRT = 3

// this wave's weight fragments of one 128-deep layer: column tile `tile`, 4 k-steps x 3 planes
__device__ __forceinline__ void r3_load_w(const bf16x8* P, int tile, int S, int lane, bf16x8 (&wf)[4][3]) {
  const bf16x8* p = P + (size_t)tile * S * 3 * 64 + lane;
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) wf[s][pl] = p[(s * 3 + pl) * 64];
}
// acc[rt] += (a[rt] . w)^T for the three row tiles, product-major (no MFMA waits for the previous one's result): lane
// (m = lane & 15, g = lane >> 4) holds row 16 rt + m, columns 4 g .. 4 g + 3 of the wave's 16-column tile
template <int NRT>
__device__ __forceinline__ void r3_mma3(const bf16x8 (&a)[NRT][3], const bf16x8 (&wf)[3], f32x4 (&acc)[NRT]) {
#pragma unroll
  for (int pr = 0; pr < 6; ++pr) {
    const int pa = (pr == 0) ? 1 : (pr == 1) ? 0 : (pr == 2) ? 2 : (pr == 3) ? 0 : (pr == 4) ? 1 : 0;
    const int pb = (pr == 0) ? 1 : (pr == 1) ? 2 : (pr == 2) ? 0 : (pr == 3) ? 1 : (pr == 4) ? 0 : 0;
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt) acc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[pb], a[rt][pa], acc[rt], 0, 0, 0);
  }
}
// acc[rt] += act(48 x 128, packed in LDS) . w^T for the wave's 16 columns
template <int NRT>
__device__ __forceinline__ void r3_mma128(const bf16x8* act, int lane, const bf16x8 (&wf)[4][3], f32x4 (&acc)[NRT]) {
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    bf16x8 a[NRT][3];
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) a[rt][pl] = act[((rt * 4 + s) * 3 + pl) * 64 + lane];
    r3_mma3<NRT>(a, wf[s], acc);
  }
}
// v[rt][r] = element (row 16 rt + (lane & 15), column 16 wave + 4 (lane >> 4) + r) -> the packed planes of `dst`: k-step
// wave >> 1, fragment lane 16 (2 (wave & 1) + (g >> 1)) + m, elements 4 (g & 1) .. + 3 of its eight
template <int NRT>
__device__ __forceinline__ void r3_store_packed(const float (&v)[NRT][4], bf16x8* dst, int wave, int lane) {
  typedef __bf16 bf16v4 __attribute__((ext_vector_type(4)));
  const int g = lane >> 4;
  char* o = reinterpret_cast<char*>(dst + (size_t)((wave >> 1) * 3) * 64 + 16 * (2 * (wave & 1) + (g >> 1)) + (lane & 15)) + 8 * (g & 1);
#pragma unroll
  for (int rt = 0; rt < NRT; ++rt) {
    float r[4];
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      bf16v4 h;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float x = (p == 0) ? v[rt][e] : r[e];
        h[e] = (__bf16)x;
        r[e] = x - (float)h[e];
      }
      *reinterpret_cast<bf16v4*>(o + (size_t)((rt * 4) * 3 + p) * 64 * sizeof(bf16x8)) = h;
    }
  }
}
}  // namespace

template <int NRT>
__global__ __launch_bounds__(512) void egx_regressor3_kernel(RegWeights3 w, const float* __restrict__ Y,
                                                                const float* __restrict__ betas, int A, int M,
                                                                float* __restrict__ out_Yb) {
  constexpr int ROWS = 16 * NRT;
  extern __shared__ __attribute__((aligned(16))) char r3_smem[];
  float* xb = reinterpret_cast<float*>(r3_smem);                                      // [48][164] fp32: the running 6D parameters
  bf16x8* hb3 = reinterpret_cast<bf16x8*>(r3_smem + (size_t)ROWS * R3_XBP * 4);    // packed h (also: scratch of the prologue)
  bf16x8* tb3 = hb3 + r3_act_frags(NRT);                                                   // packed t
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = blockIdx.x * ROWS;
  const int col4 = 16 * wave + 4 * (lane >> 4);   // this lane's four output columns in every 128-wide layer
  for (int i = tid; i < ROWS * R3_XBP; i += 512) xb[i] = 0.f;
  // ---- prologue: [markers | betas] as packed planes in the (still unused) activation region: 3 row tiles x (7 + 1) k-steps
  bf16x8* in3 = hb3;
  for (int f = wave; f < NRT * 8; f += 8) {
    const int rt = f >> 3, s = f & 7;
    const int row = min(m0 + 16 * rt + (lane & 15), M - 1), k0 = 8 * (lane >> 4);
    float x[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = 32 * s + k0 + e;
      x[e] = (s < 7) ? (k < 201 ? Y[(size_t)row * 201 + k] : 0.f) : (k0 + e < 10 ? betas[(size_t)(row % A) * 10 + k0 + e] : 0.f);
    }
    bf16x8 pl[3];
    d3_split(x, pl);
#pragma unroll
    for (int p = 0; p < 3; ++p) in3[((rt * 8 + s) * 3 + p) * 64 + lane] = pl[p];
  }
  __syncthreads();
  f32x4 base[NRT];   // W_m markers + W_b betas + b_in for this wave's columns
  {
    const f32x4 b = *reinterpret_cast<const f32x4a1*>(w.in_b + col4);
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt) base[rt] = b;
    for (int s = 0; s < 8; ++s) {
      bf16x8 wf[3];
      const bf16x8* pw = (s < 7) ? w.in_m + ((size_t)(wave * 7 + s) * 3) * 64 + lane : w.in_b3 + ((size_t)wave * 3) * 64 + lane;
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) wf[pl] = pw[pl * 64];
      bf16x8 a[NRT][3];
#pragma unroll
      for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) a[rt][pl] = in3[((rt * 8 + s) * 3 + pl) * 64 + lane];
      r3_mma3<NRT>(a, wf, base);
    }
  }
  __syncthreads();
  bf16x8 wA[4][3], wB[4][3];
  float hres[NRT][4];   // residual stream h of this lane's elements
  f32x4 acc[NRT];
  // one 128 -> 128 layer of a residual block with the weights in `cur`; the NEXT layer's weights (or out_fc's tile `wave`) are
  // requested into `nxt` before the products start, so that their round trip to L2 runs under this layer's matrix work and
  // epilogue instead of in front of the next layer's
  auto layer = [&](int l, const bf16x8 (&cur)[4][3], bf16x8 (&nxt)[4][3]) __attribute__((always_inline)) {
    const bf16x8* src = (l & 1) ? tb3 : hb3;
    const f32x4 b = *reinterpret_cast<const f32x4a1*>(w.blk_b + l * 128 + col4);
    if (l + 1 < 20) r3_load_w(w.blk + (size_t)(l + 1) * 8 * 4 * 3 * 64, wave, 4, lane, nxt);
    else r3_load_w(w.out, wave, 4, lane, nxt);
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt) acc[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
    __builtin_amdgcn_sched_barrier(0);
    r3_mma128<NRT>(src, lane, cur, acc);
    __builtin_amdgcn_sched_barrier(0);
    float v[NRT][4];
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float y = fmaxf(acc[rt][r] + b[r], 0.f);
        if (l & 1) { y += hres[rt][r]; hres[rt][r] = y; }
        v[rt][r] = y;
      }
    r3_store_packed<NRT>(v, (l & 1) ? hb3 : tb3, wave, lane);
    __syncthreads();
  };
  r3_load_w(w.blk, wave, 4, lane, wA);   // first block layer's weights: independent of the activations
  for (int rc = 0; rc < 3; ++rc) {
    // ---- in_fc: h = base + W_xb xb (xb is zero in the first recurrence)
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt) acc[rt] = base[rt];
    if (rc > 0) {
      bf16x8* xb3 = hb3;   // 3 row tiles x 5 k-steps of packed xb, made by all waves (the activation region is free here)
      for (int f = wave; f < NRT * 5; f += 8) {
        const int rt = f / 5, s = f % 5;
        const float* sp = xb + (16 * rt + (lane & 15)) * R3_XBP + 32 * s + 8 * (lane >> 4);
        const f32x4 x0 = *reinterpret_cast<const f32x4*>(sp), x1 = *reinterpret_cast<const f32x4*>(sp + 4);
        const float x[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
        bf16x8 pl[3];
        d3_split(x, pl);
#pragma unroll
        for (int p = 0; p < 3; ++p) xb3[((rt * 5 + s) * 3 + p) * 64 + lane] = pl[p];
      }
      __syncthreads();
      for (int s = 0; s < 5; ++s) {
        bf16x8 wf[3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) wf[pl] = w.in_xb[((size_t)(wave * 5 + s) * 3 + pl) * 64 + lane];
        bf16x8 a[NRT][3];
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) a[rt][pl] = xb3[((rt * 5 + s) * 3 + pl) * 64 + lane];
        r3_mma3<NRT>(a, wf, acc);
      }
      __syncthreads();   // everyone is done reading xb3 before h overwrites the region
    }
    {
      float v[NRT][4];
#pragma unroll
      for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) { v[rt][r] = acc[rt][r]; hres[rt][r] = acc[rt][r]; }
      r3_store_packed<NRT>(v, hb3, wave, lane);
    }
    __syncthreads();
    // ---- 10 residual blocks: t = relu(W1 h + b1); h = relu(W2 t + b2) + h.  Even layers read wA, odd ones wB.
    for (int l = 0; l < 20; l += 2) {
      layer(l, wA, wB);
      layer(l + 1, wB, wA);
    }
    // ---- out_fc: N = 159 -> column tiles 0..9; wave w owns tile w (weights already in wA), waves 0 and 1 also tiles 8, 9
    if (rc + 1 < 3) r3_load_w(w.blk, wave, 4, lane, wB);   // the next recurrence's first block layer
    for (int tI = wave; tI < 10; tI += 8) {
      if (tI >= 8) r3_load_w(w.out, tI, 4, lane, wA);
#pragma unroll
      for (int rt = 0; rt < NRT; ++rt) acc[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
      r3_mma128<NRT>(hb3, lane, wA, acc);
      const int nn = tI * 16 + 4 * (lane >> 4);   // four columns of one row (column 159 is padding: it stays zero)
      float b[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) b[r] = (nn + r < R3_NOUT) ? w.out_b[nn + r] : 0.f;
#pragma unroll
      for (int rt = 0; rt < NRT; ++rt) {
        f32x4* xp = reinterpret_cast<f32x4*>(xb + (16 * rt + (lane & 15)) * R3_XBP + nn);
        f32x4 x = *xp;
#pragma unroll
        for (int r = 0; r < 4; ++r) x[r] += (nn + r < R3_NOUT) ? acc[rt][r] + b[r] : 0.f;
        *xp = x;
      }
    }
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) wA[s][pl] = wB[s][pl];
    __syncthreads();
  }
  // ---- 6D -> axis-angle tail straight from LDS
  for (int idx = tid; idx < ROWS * 23; idx += 512) {
    const int r = idx / 23, j = idx % 23;
    if (m0 + r < M) egx_cont6d_item(xb + r * R3_XBP, out_Yb + (size_t)(m0 + r) * 93, j);
  }
}

namespace {
template <int NRT>
int r3_launch(hipStream_t st, const RegWeights3& w, const float* Y, const float* betas, int A, int M, float* out_Yb) {
  if (r3_lds(NRT) > 64 * 1024) {  // dynamic LDS above the 64 KiB default cap: raised once per device
    static std::mutex mu;
    static bool attr_set[64] = {false};
    int dev = 0;
    EGX_HIP_CHECK(hipGetDevice(&dev));
    EGX_REQUIRE(dev >= 0 && dev < 64, "device ordinal out of range");
    std::lock_guard<std::mutex> lk(mu);
    if (!attr_set[dev]) {
      EGX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(egx_regressor3_kernel<NRT>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)r3_lds(NRT)));
      attr_set[dev] = true;
    }
  }
  hipLaunchKernelGGL(egx_regressor3_kernel<NRT>, dim3(egx_ceil_div(M, 16 * NRT)), dim3(512), r3_lds(NRT), st, w, Y, betas, A, M, out_Yb);
  return EGX_OK;
}
}  // namespace

// Rows per workgroup: a workgroup's time hardly depends on how many of the chip's CUs are busy, so a batch that does not fill
// 256 CUs with 48-row workgroups takes fewer rows per workgroup (9216 rows = 512 agents: 192 x 48; 4608: 144 x 32; <= 4096
// rows: 16 each).  EGX_R3_ROWTILES = 1..3 forces the tile count (tests run every variant on small batches).
int egx_launch_regressor3(hipStream_t st, const RegWeights3& w, const float* Y, const float* betas, int A, int M, float* out_Yb) {
  static const int forced = [] { const char* e = getenv("EGX_R3_ROWTILES"); return (e && *e) ? atoi(e) : 0; }();
  const int nrt = forced >= 1 && forced <= 3 ? forced : std::min(3, std::max(1, egx_ceil_div(M, 16 * 256)));
  switch (nrt) {
    case 1: return r3_launch<1>(st, w, Y, betas, A, M, out_Yb);
    case 2: return r3_launch<2>(st, w, Y, betas, A, M, out_Yb);
    default: return r3_launch<3>(st, w, Y, betas, A, M, out_Yb);
  }
}

// ---------------------------------------------------------------------------------------------------------
// Fused VPoser encoder mean on the bf16 matrix pipe (three planes: fp32-equivalent): out = mu(lrelu(fc2(lrelu(fc1 x)))) for 48
// rows per workgroup, BatchNorms folded into fc1 / fc2 by the host (egx_vposer_weights).  Same scheme as the regressor:
// transposed products, the epilogues write the next layer's operand fragments into LDS themselves.
//   * eight waves, wave w owns columns 64 w .. 64 w + 63 (four 16-column tiles) of the two 512-wide layers for all three
//     16-row tiles: 12 accumulators; fc2's weights (1.5 MB, L2-resident: every workgroup reads all of them) come as bursts of
//     TRIP k-steps x 12 one-KiB fragments per wave -> s_waitcnt -> 72 TRIP MFMAs, the two waves of a SIMD alternating;
//   * y1 = lrelu(fc1 x) lives in LDS as packed planes [3 row tiles][16 k-steps][3 planes] = 144 KiB - the whole LDS budget:
//     the packed input rows (18 KiB) borrow its head before y1 exists, y2 = lrelu(fc2 y1) overwrites it in place (a wave
//     writes the two k-steps it alone will read), the split-K partial sums of mu (wave w reduces over ITS 64 columns of y2)
//     borrow it once y2 is consumed.
// ---------------------------------------------------------------------------------------------------------
namespace {
constexpr size_t vp_lds(int nrt) { return (size_t)nrt * 16 * 3 * 64 * 16; }   // 48 KiB per 16-row tile: 147 456 B at NRT = 3

// values v[rt][r] of column tile `ct16` (16 columns: 32-wide k-step ct16 >> 1, half ct16 & 1) -> packed planes in `ybuf`
template <int NRT>
__device__ __forceinline__ void vp_store_packed(const float (&v)[NRT][4], bf16x8* ybuf, int ct16, int lane) {
  typedef __bf16 bf16v4 __attribute__((ext_vector_type(4)));
  const int g = lane >> 4;
  char* o = reinterpret_cast<char*>(ybuf + (size_t)((ct16 >> 1) * 3) * 64 + 16 * (2 * (ct16 & 1) + (g >> 1)) + (lane & 15)) + 8 * (g & 1);
#pragma unroll
  for (int rt = 0; rt < NRT; ++rt) {
    float r[4];
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      bf16v4 h;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float x = (p == 0) ? v[rt][e] : r[e];
        h[e] = (__bf16)x;
        r[e] = x - (float)h[e];
      }
      *reinterpret_cast<bf16v4*>(o + (size_t)((rt * 16) * 3 + p) * 64 * sizeof(bf16x8)) = h;
    }
  }
}
// acc[ct][rt] += (a[rt] . w[ct])^T, product-major
template <int NCT, int NRT>
__device__ __forceinline__ void vp_mma(const bf16x8 (&a)[NRT][3], const bf16x8 (&wf)[NCT][3], f32x4 (&acc)[NCT][NRT]) {
#pragma unroll
  for (int pr = 0; pr < 6; ++pr) {
    const int pa = (pr == 0) ? 1 : (pr == 1) ? 0 : (pr == 2) ? 2 : (pr == 3) ? 0 : (pr == 4) ? 1 : 0;
    const int pb = (pr == 0) ? 1 : (pr == 1) ? 2 : (pr == 2) ? 0 : (pr == 3) ? 1 : (pr == 4) ? 0 : 0;
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
      for (int rt = 0; rt < NRT; ++rt) acc[ct][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ct][pb], a[rt][pa], acc[ct][rt], 0, 0, 0);
  }
}
__device__ __forceinline__ float vp_lrelu(float v) { return v > 0.f ? v : 0.2f * v; }
}  // namespace

template <int TRIP, int NRT>
__global__ __launch_bounds__(512) void egx_vposer3_kernel(VpWeights3 w, const float* __restrict__ X, int x_ld, int n,
                                                          float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) char vp_smem[];
  bf16x8* ybuf = reinterpret_cast<bf16x8*>(vp_smem);   // [3 row tiles][16 k-steps][3 planes][64 lanes]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4;
  const int m0 = blockIdx.x * 16 * NRT;
  // ---- packed input rows: 3 row tiles x 2 k-steps (63 -> 64 columns), one fragment per wave 0..5, at the head of ybuf
  if (wave < 2 * NRT) {
    const int rt = wave >> 1, s = wave & 1;
    const int row = min(m0 + 16 * rt + (lane & 15), n - 1), k0 = 32 * s + 8 * g;
    float x[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = (k0 + e < 63) ? X[(size_t)row * x_ld + k0 + e] : 0.f;
    bf16x8 pl[3];
    d3_split(x, pl);
#pragma unroll
    for (int p = 0; p < 3; ++p) ybuf[(wave * 3 + p) * 64 + lane] = pl[p];
  }
  __syncthreads();
  // ---- fc1: 64 -> 512
  f32x4 acc[4][NRT];
#pragma unroll
  for (int ct = 0; ct < 4; ++ct)
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt) acc[ct][rt] = f32x4{0.f, 0.f, 0.f, 0.f};
  {
    bf16x8 wf[2][4][3];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int p = 0; p < 3; ++p) wf[s][ct][p] = w.fc1[((size_t)((4 * wave + ct) * 2 + s) * 3 + p) * 64 + lane];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 a[NRT][3];
#pragma unroll
      for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
        for (int p = 0; p < 3; ++p) a[rt][p] = ybuf[((rt * 2 + s) * 3 + p) * 64 + lane];
      vp_mma<4, NRT>(a, wf[s], acc);
    }
  }
  __syncthreads();   // the input fragments are consumed: y1 may overwrite them
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) {
    const f32x4 b = *reinterpret_cast<const f32x4a1*>(w.b1 + 64 * wave + 16 * ct + 4 * g);
    float v[NRT][4];
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) v[rt][r] = vp_lrelu(acc[ct][rt][r] + b[r]);
    vp_store_packed<NRT>(v, ybuf, 4 * wave + ct, lane);
  }
  __syncthreads();
  // ---- fc2: 512 -> 512, the weights in bursts of TRIP k-steps
#pragma unroll
  for (int ct = 0; ct < 4; ++ct)
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt) acc[ct][rt] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int s0 = 0; s0 < 16; s0 += TRIP) {
    bf16x8 wf[TRIP][4][3];
#pragma unroll
    for (int u = 0; u < TRIP; ++u) {
      const int su = min(s0 + u, 15);
#pragma unroll
      for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int p = 0; p < 3; ++p) wf[u][ct][p] = w.fc2[((size_t)((4 * wave + ct) * 16 + su) * 3 + p) * 64 + lane];
    }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < TRIP; ++u) {
      if (s0 + u < 16) {
        bf16x8 a[NRT][3];
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
          for (int p = 0; p < 3; ++p) a[rt][p] = ybuf[((rt * 16 + s0 + u) * 3 + p) * 64 + lane];
        vp_mma<4, NRT>(a, wf[u], acc);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  // mu's weights for this wave's two k-steps: their round trip runs under fc2's epilogue
  bf16x8 wmu[2][2][3];
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
      for (int p = 0; p < 3; ++p) wmu[u][jt][p] = w.mu[((size_t)(jt * 16 + 2 * wave + u) * 3 + p) * 64 + lane];
  __syncthreads();   // every wave is done with y1: y2 takes its place
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) {
    const f32x4 b = *reinterpret_cast<const f32x4a1*>(w.b2 + 64 * wave + 16 * ct + 4 * g);
    float v[NRT][4];
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) v[rt][r] = vp_lrelu(acc[ct][rt][r] + b[r]);
    vp_store_packed<NRT>(v, ybuf, 4 * wave + ct, lane);
  }
  __syncthreads();
  // ---- mu: 512 -> 32, wave w reduces over k-steps 2 w, 2 w + 1 (its own columns of y2)
  f32x4 am[2][NRT];
#pragma unroll
  for (int jt = 0; jt < 2; ++jt)
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt) am[jt][rt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    bf16x8 a[NRT][3];
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
      for (int p = 0; p < 3; ++p) a[rt][p] = ybuf[((rt * 16 + 2 * wave + u) * 3 + p) * 64 + lane];
    vp_mma<2, NRT>(a, wmu[u], am);
  }
  __syncthreads();   // y2 is consumed: the partial sums borrow the buffer
  float* red = reinterpret_cast<float*>(vp_smem);   // [8 waves][8 NRT][64]
  constexpr int NQ = 8 * NRT;
#pragma unroll
  for (int jt = 0; jt < 2; ++jt)
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[(wave * NQ + (jt * NRT + rt) * 4 + r) * 64 + lane] = am[jt][rt][r];
  __syncthreads();
  for (int idx = tid; idx < NQ * 64; idx += 512) {
    const int q = idx >> 6, l = idx & 63;
    float v = ((red[(0 * NQ + q) * 64 + l] + red[(1 * NQ + q) * 64 + l]) + (red[(2 * NQ + q) * 64 + l] + red[(3 * NQ + q) * 64 + l])) +
              ((red[(4 * NQ + q) * 64 + l] + red[(5 * NQ + q) * 64 + l]) + (red[(6 * NQ + q) * 64 + l] + red[(7 * NQ + q) * 64 + l]));
    const int jt = q / (4 * NRT), rt = (q >> 2) % NRT, r = q & 3;
    const int row = m0 + 16 * rt + (l & 15), j = 16 * jt + 4 * (l >> 4) + r;
    if (row < n) out[(size_t)row * 32 + j] = v + w.bmu[j];
  }
}

namespace {
template <int NRT>
int vp_launch(hipStream_t st, const VpWeights3& w, const float* x, int x_ld, int n, float* out) {
  if (vp_lds(NRT) > 64 * 1024) {  // dynamic LDS above the 64 KiB default cap: raised once per device
    static std::mutex mu;
    static bool attr_set[64] = {false};
    int dev = 0;
    EGX_HIP_CHECK(hipGetDevice(&dev));
    EGX_REQUIRE(dev >= 0 && dev < 64, "device ordinal out of range");
    std::lock_guard<std::mutex> lk(mu);
    if (!attr_set[dev]) {
      EGX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(egx_vposer3_kernel<2, NRT>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)vp_lds(NRT)));
      attr_set[dev] = true;
    }
  }
  hipLaunchKernelGGL((egx_vposer3_kernel<2, NRT>), dim3(egx_ceil_div(n, 16 * NRT)), dim3(512), vp_lds(NRT), st, w, x, x_ld, n, out);
  return EGX_OK;
}
}  // namespace

// bursts of 2 k-steps: 40.5 us for 10 240 rows (1: 41.6, 3: 42.1; the three fp32-MFMA launches this replaces: 126).  Rows per
// workgroup as for the regressor: 48 when that fills the chip, 32 / 16 for smaller batches (EGX_VP_ROWTILES forces 1..3).
int egx_launch_vposer3(hipStream_t st, const VpWeights3& w, const float* x, int x_ld, int n, float* out) {
  static const int forced = [] { const char* e = getenv("EGX_VP_ROWTILES"); return (e && *e) ? atoi(e) : 0; }();
  const int nrt = forced >= 1 && forced <= 3 ? forced : std::min(3, std::max(1, egx_ceil_div(n, 16 * 256)));
  switch (nrt) {
    case 1: return vp_launch<1>(st, w, x, x_ld, n, out);
    case 2: return vp_launch<2>(st, w, x, x_ld, n, out);
    default: return vp_launch<3>(st, w, x, x_ld, n, out);
  }
}

// ---- launchers ------------------------------------------------------------------------------------------
namespace {
// development knobs, read once
int d3_env_int(const char* name, int dflt) {
  const char* v = std::getenv(name);
  return (v && *v) ? std::atoi(v) : dflt;
}
template <int TRIP, int MI, int NI, int NW, int NPL>
void d3_launch_cfg(hipStream_t st, const D3Plain* ps, int n) {
  constexpr int TM = 16 * MI, TN = 16 * NI;
  constexpr size_t lds = (size_t)(NW * MI * NI * 4 * 64 + TM * (TN + 4)) * sizeof(float);
  static_assert(lds <= 160 * 1024, "LDS");
  if (lds > 64 * 1024) {   // above the default dynamic-LDS cap: raised once per device
    static std::mutex mu;
    static bool attr_set[64] = {false};
    int dev = 0;
    if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64) {
      std::lock_guard<std::mutex> lk(mu);
      if (!attr_set[dev]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&egx_dense3_kernel<TRIP, MI, NI, NW, NPL>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set[dev] = true;
      }
    }
  }
  D3Args4 f;
  static const int rowmap = d3_env_int("EGX_D3_ROWMAP", -1);
  f.rowmap = rowmap;
  D3Plain* dst[4] = {&f.p0, &f.p1, &f.p2, &f.p3};
  int ends[4] = {0, 0, 0, 0}, total = 0;
  for (int i = 0; i < 4; ++i) {
    *dst[i] = ps[i < n ? i : n - 1];
    if (i < n) total += d3_blocks(egx_ceil_div(ps[i].M, TM), egx_ceil_div(ps[i].N, TN), d3_rowmap(rowmap, ps[i].M, ps[i].N)) * std::max(1, ps[i].batches);
    ends[i] = total;
  }
  f.end0 = ends[0]; f.end1 = ends[1]; f.end2 = ends[2];
  hipLaunchKernelGGL((egx_dense3_kernel<TRIP, MI, NI, NW, NPL>), dim3(total), dim3(64 * NW), lds, st, f);
}

}  // namespace

// 32 x 32 tile per 4-wave workgroup, the reduction split over the waves; three k-steps per round trip for deep reductions.
// (Measured for the update's 1152-deep layers, profiles/r03_update_experiments.md: 64 x 64 and 64 x 32 tiles, eight-wave
// split-K, a software-pipelined loop and two k-steps per trip all land within a few per cent of this form or behind it.)
// All layers of a launch share the arithmetic mode of the first (the callers build them from one setting).
void egx_launch_dense3_n(hipStream_t st, const D3Plain* ps, int n) {
  int smax = 0;
  for (int i = 0; i < n; ++i) smax = std::max(smax, ps[i].S);
  const bool deep = smax > 16;
  // 9..16 k-steps: eight waves take two k-steps each, ONE operand round trip (four waves need two).  EGX_D3_NW8=0 turns it off.
  static const int nw8 = d3_env_int("EGX_D3_NW8", 1);
  int tiles = 0;
  for (int i = 0; i < n; ++i) tiles += egx_ceil_div(ps[i].M, 32) * egx_ceil_div(ps[i].N, 32) * std::max(1, ps[i].batches);
  const bool wide = nw8 && smax > 8 && !deep && (nw8 > 1 || tiles <= 512);   // a latency-bound launch: at most two workgroups per CU
  switch (ps[0].prec) {
    case 2: {
      static const int trip = d3_env_int("EGX_D3_TRIP_P2", 3);
      if (wide) d3_launch_cfg<2, 2, 2, 8, 2>(st, ps, n);
      else if (!deep) d3_launch_cfg<2, 2, 2, 4, 2>(st, ps, n);
      else if (trip >= 5) d3_launch_cfg<5, 2, 2, 4, 2>(st, ps, n);
      else d3_launch_cfg<3, 2, 2, 4, 2>(st, ps, n);
      break;
    }
    case 1: {
      static const int trip = d3_env_int("EGX_D3_TRIP_P1", 5);
      if (wide) d3_launch_cfg<2, 2, 2, 8, 1>(st, ps, n);
      else if (!deep) d3_launch_cfg<2, 2, 2, 4, 1>(st, ps, n);
      else if (trip >= 9) d3_launch_cfg<9, 2, 2, 4, 1>(st, ps, n);
      else if (trip >= 5) d3_launch_cfg<5, 2, 2, 4, 1>(st, ps, n);
      else d3_launch_cfg<3, 2, 2, 4, 1>(st, ps, n);
      break;
    }
    default:
      if (wide) d3_launch_cfg<2, 2, 2, 8, 3>(st, ps, n);
      else if (deep) d3_launch_cfg<3, 2, 2, 4, 3>(st, ps, n);
      else d3_launch_cfg<2, 2, 2, 4, 3>(st, ps, n);
  }
}
void egx_launch_dense3(hipStream_t st, const D3Plain& p) { egx_launch_dense3_n(st, &p, 1); }
void egx_launch_dense3_pair(hipStream_t st, const D3Plain& p, const D3Plain& q) {
  const D3Plain ps[2] = {p, q};
  egx_launch_dense3_n(st, ps, 2);
}
void egx_launch_dense3_triple(hipStream_t st, const D3Plain& p, const D3Plain& q, const D3Plain& r) {
  const D3Plain ps[3] = {p, q, r};
  egx_launch_dense3_n(st, ps, 3);
}
static void d3_launch_gru(hipStream_t st, const D3Gru& g0, const D3Gru* g1) {
  constexpr size_t lds = (size_t)(8 * 24 * 64 + 32 * 20) * sizeof(float);   // 50.5 KiB: within the default dynamic-LDS cap
  D3Gru2 two;
  two.g0 = g0; two.g1 = g1 ? *g1 : g0;
  static const int rowmap = d3_env_int("EGX_D3_ROWMAP", -1);
  two.rowmap = rowmap;
  two.blocks0 = d3_blocks((g0.M + 31) >> 5, g0.H >> 4, d3_rowmap(rowmap, g0.M, 3 * g0.H));
  const int total = two.blocks0 + (g1 ? d3_blocks((g1->M + 31) >> 5, g1->H >> 4, d3_rowmap(rowmap, g1->M, 3 * g1->H)) : 0);
  switch (g0.prec) {
    case 2: hipLaunchKernelGGL((egx_gru3_kernel<2, 2>), dim3(total), dim3(512), lds, st, two); break;
    case 1: hipLaunchKernelGGL((egx_gru3_kernel<4, 1>), dim3(total), dim3(512), lds, st, two); break;
    default: hipLaunchKernelGGL((egx_gru3_kernel<2, 3>), dim3(total), dim3(512), lds, st, two);
  }
}
int egx_launch_gru3(hipStream_t st, const D3Gru& g) {
  d3_launch_gru(st, g, nullptr);
  return EGX_OK;
}
int egx_launch_gru3_pair(hipStream_t st, const D3Gru& g0, const D3Gru& g1) {
  d3_launch_gru(st, g0, &g1);
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

// ---- one product on fp32 row-major operands (training-side autograd nodes: fused_ops.py) ---------------------------------
extern "C" size_t egx_gemm3_workspace_bytes(int M, int N, int K) { return egx_pack3_bytes(M, K) + egx_pack3_bytes(N, K); }

extern "C" int egx_gemm3(const float* A, int lda, int trans_a, const float* B, int ldb, int trans_b, int M, int N, int K,
                         const float* bias, int act, float slope, const float* res, int ldr, float* out, int ldo, float* out_act,
                         int ldact, void* workspace, size_t workspace_bytes, void* stream) {
  EGX_REQUIRE(A && B && out && M > 0 && N > 0 && K > 0 && workspace, "bad arguments");
  EGX_REQUIRE(lda >= (trans_a ? M : K) && ldb >= (trans_b ? N : K) && ldo >= N && (!res || ldr >= N) && (!out_act || ldact >= N),
              "leading dimension smaller than the row");
  EGX_REQUIRE(workspace_bytes >= egx_gemm3_workspace_bytes(M, N, K), "workspace too small (egx_gemm3_workspace_bytes)");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int S = egx_ceil_div(K, 32);
  char* ws = static_cast<char*>(workspace);
  void* img_a = ws;
  void* img_b = ws + egx_pack3_bytes(M, K);
  D3Pack jobs[2];
  jobs[0].src = A; jobs[0].R = M; jobs[0].K = K; jobs[0].ld = lda; jobs[0].col0 = 0; jobs[0].dst = img_a; jobs[0].S_total = S; jobs[0].s0 = 0;
  jobs[0].transpose = trans_a ? 1 : 0;
  jobs[1].src = B; jobs[1].R = N; jobs[1].K = K; jobs[1].ld = ldb; jobs[1].col0 = 0; jobs[1].dst = img_b; jobs[1].S_total = S; jobs[1].s0 = 0;
  jobs[1].transpose = trans_b ? 1 : 0;
  egx_launch_pack3(st, jobs, 2);
  D3Plain p;
  p.A = static_cast<const bf16x8*>(img_a); p.SA = S; p.sa0 = 0;
  p.B = static_cast<const bf16x8*>(img_b); p.S = S;
  p.bias = bias; p.res = res; p.ldr = ldr; p.out = out; p.ldo = ldo; p.M = M; p.N = N; p.act = act; p.slope = slope;
  p.out_act = out_act; p.ldact = ldact;
  egx_launch_dense3(st, p);
  EGX_HIP_CHECK(hipGetLastError());
  return EGX_OK;
}
