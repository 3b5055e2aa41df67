// One PPO minibatch - forward, clipped-PPO loss, backward - of the policy networks as a fixed chain of hand-written launches
// (crowd_ppo/ppo_policy.py:189-252 `learn`, through models/models_policy_ppo.py:24-39, 287-350).  No autograd graph, no library
// GEMM: every product runs on the dense3 kernels (bf16 matrix pipe, three-term splits = fp32-equivalent), with what used to be
// separate launches folded into their epilogues:
//   forward   activation, skip connection, the saved activation for backward, and the packed images the next products read
//             (row-major image for the next layer, transposed image for this layer's weight gradient);
//   backward  dX = G W^T-image with the skip gradient added and the result multiplied by act'(saved activation) on its way into
//             the packed images of the next gradient; dW = G^T-image x X^T-image, whose extra row of ones makes the bias
//             gradient one more output column; input- and weight-gradient products of the actor AND the critic share a launch;
//   GRU       cell forward = one launch per time step for both encoders (dense3 gru kernel), cell backward = one launch per
//             step producing the gate gradients directly as packed images.
// ~27 launches per minibatch instead of ~85.  Gradients are WRITTEN (not accumulated) into the flat gradient buffer - every
// parameter is used exactly once per minibatch - so the buffer needs no zero fill.
#include <algorithm>
#include <cstring>
#include <vector>
#include "egx_nets.h"

namespace {

typedef __bf16 bf16v8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void u3_split(const float (&x)[8], bf16x8 (&pl)[3]) {
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

// ---- table-driven packing: any number of matrices per launch, optionally transposed, optionally with a row of ones ----------
struct PackEntry {
  const float* src;
  int red, cols, ld, col0;   // source block: `red` rows x `cols` columns starting at column col0 (leading dimension ld)
  bf16x8* dst;
  int S_total, s0;           // destination: k-steps per row tile, first k-step
  int transpose;             // 0: image rows = source rows (reduction along the columns); 1: image rows = source columns;
                             // 2: BOTH images of the whole matrix from one read (dst: rows = source rows, dst2: rows = source
                             //    columns), a 32 x 32 source block per wave
  int ones_row;              // transposed images: image row that is all ones (-1: none)
  int frag_end;              // running fragment count (this entry owns [previous frag_end, frag_end)); blocks for transpose 2
  bf16x8* dst2;              // transpose 2: the transposed image
  int S_total2;              //              and its k-steps per row tile
};

__global__ __launch_bounds__(256) void egx_pack3_table_kernel(const PackEntry* __restrict__ tab, int n, int nplanes) {
  int frag = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (n <= 0 || frag >= tab[n - 1].frag_end) return;
  int lo = 0, hi = n - 1;   // first entry whose running count exceeds `frag`
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (frag >= tab[mid].frag_end) lo = mid + 1; else hi = mid;
  }
  const PackEntry e = tab[lo];
  frag -= lo > 0 ? tab[lo - 1].frag_end : 0;
  float x[8];
  int rt, s;
  if (e.transpose == 2) {
    // weight matrix W [red, cols] -> image of W (the forward products' operand) and image of W^T (the input-gradient
    // products'): the wave reads its 32 x 32 block once, writes W's two fragments from registers and W^T's two after a
    // transposition through its LDS strip
    __shared__ float sblk[4][32][33];
    float (*sb)[33] = sblk[threadIdx.x >> 6];
    const int BJ = (e.cols + 31) >> 5;
    const int bi = frag / BJ, bj = frag % BJ;
    const int r = lane & 15, kg = lane >> 4;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int row = 32 * bi + 16 * t + r, k0 = 32 * bj + 8 * kg;
      const float* sp = e.src + (size_t)row * e.ld + e.col0 + k0;
      if (row < e.red && k0 + 7 < e.cols && (reinterpret_cast<uintptr_t>(sp) & 15) == 0) {   // the interior: two 16-byte loads
        const f32x4 x0 = *reinterpret_cast<const f32x4*>(sp), x1 = *reinterpret_cast<const f32x4*>(sp + 4);
        x[0] = x0[0]; x[1] = x0[1]; x[2] = x0[2]; x[3] = x0[3]; x[4] = x1[0]; x[5] = x1[1]; x[6] = x1[2]; x[7] = x1[3];
      } else {
#pragma unroll
        for (int q = 0; q < 8; ++q) x[q] = (row < e.red && k0 + q < e.cols) ? sp[q] : 0.f;
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) sb[16 * t + r][8 * kg + q] = x[q];
      bf16x8 pl[3];
      u3_split(x, pl);
      bf16x8* o = e.dst + ((size_t)(2 * bi + t) * e.S_total + e.s0 + bj) * 3 * 64 + lane;
#pragma unroll
      for (int p = 0; p < 3; ++p)
        if (p < nplanes) o[p * 64] = pl[p];
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
      for (int q = 0; q < 8; ++q) x[q] = sb[8 * kg + q][16 * t + r];
      bf16x8 pl[3];
      u3_split(x, pl);
      bf16x8* o = e.dst2 + ((size_t)(2 * bj + t) * e.S_total2 + bi) * 3 * 64 + lane;
#pragma unroll
      for (int p = 0; p < 3; ++p)
        if (p < nplanes) o[p * 64] = pl[p];
    }
    return;
  }
  if (!e.transpose) {
    const int S = (e.cols + 31) >> 5;
    rt = frag / S; s = frag % S;
    const int row = rt * 16 + (lane & 15), k0 = s * 32 + 8 * (lane >> 4);
#pragma unroll
    for (int q = 0; q < 8; ++q) x[q] = (row < e.red && k0 + q < e.cols) ? e.src[(size_t)row * e.ld + e.col0 + k0 + q] : 0.f;
  } else {
    const int S = (e.red + 31) >> 5;
    rt = frag / S; s = frag % S;
    const int irow = rt * 16 + (lane & 15), k0 = s * 32 + 8 * (lane >> 4);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      float v = 0.f;
      if (k0 + q < e.red) {
        if (irow < e.cols) v = e.src[(size_t)(k0 + q) * e.ld + e.col0 + irow];
        else if (irow == e.ones_row) v = 1.f;
      }
      x[q] = v;
    }
  }
  bf16x8 pl[3];
  u3_split(x, pl);
  bf16x8* o = e.dst + ((size_t)rt * e.S_total + e.s0 + s) * 3 * 64 + lane;
#pragma unroll
  for (int p = 0; p < 3; ++p)
    if (p < nplanes) o[p * 64] = pl[p];   // planes no consumer reads are not written (a third of the image bytes per plane)
}

inline int img_tiles(int rows) { return 2 * egx_ceil_div(rows, 32); }   // 16-row tiles, even count
inline size_t img_frags(int rows, int red) { return (size_t)img_tiles(rows) * egx_ceil_div(red, 32); }
constexpr size_t FRAG_BYTES = 3 * 64 * 16;

// ---- GRU cell backward (torch.nn.GRU gate order r, z, n) with packed outputs ---------------------------------------------
//   r = s(gi_r + gh_r), z = s(gi_z + gh_z), nn = tanh(gi_n + r gh_n), h = (1 - z) nn + z hp
//   dnn = dh (1 - z), dz = dh (hp - nn), dhp = dh z, dpn = dnn (1 - nn^2), dgi_n = dpn, dgh_n = dpn r, dr = dpn gh_n,
//   dgi_r = dgh_r = dr r (1 - r), dgi_z = dgh_z = dz z (1 - z)
// Workgroup = 32 rows x 32 hidden columns.  Outputs: dgi / dgh as transposed images (rows = the 3H gate rows, reduction
// index = batch rows, k-step s0T + row tile / 2) for the weight-gradient products, dgh also as a row-major image (the product
// dh_prev += dgh W_hh) and dh z as fp32 - the last two only where asked.
struct GruBwd {
  const float* gi;        // [M, 3H]
  const float* gh;        // [M, 3H] or null: gh = bias_h (first step: zero previous state)
  const float* bias_h;
  const float* h_prev;    // [M, H] or null (zero)
  const float* dh0;       // gradient of h: dh0[m * ld0 + c] (+ dh1[m * ld1 + c] when dh1 != null)
  const float* dh1;
  int ld0, ld1;
  bf16x8* dgiT;           // images with ST k-steps per row tile
  bf16x8* dghT;
  int ST, s0T;
  bf16x8* dgh_r;          // [M rows, reduction 3H] or null
  float* dhp;             // [M, H] or null
  int M, H;
};
struct GruBwd2 {
  GruBwd a, b;
  int blocks0;
};

__global__ __launch_bounds__(256) void egx_gru_bwd3_kernel(GruBwd2 two) {
  const bool second = (int)blockIdx.x >= two.blocks0;
  const GruBwd& a = second ? two.b : two.a;
  const int bid = second ? (int)blockIdx.x - two.blocks0 : (int)blockIdx.x;
  __shared__ __attribute__((aligned(16))) float tile[6][32 * 36];   // [dgi r,z,n | dgh r,z,n]
  const int CT = a.H >> 5;
  const int mt = bid / CT, ct = bid % CT;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int H = a.H;
  // thread t: column c = t & 31, rows (t >> 5) + 8 j
  const int cl = threadIdx.x & 31, c = ct * 32 + cl;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int rl = (threadIdx.x >> 5) + 8 * j, m = mt * 32 + rl;
    float o[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (m < a.M) {
      const float* gim = a.gi + (size_t)m * 3 * H;
      const float* ghm = a.gh ? a.gh + (size_t)m * 3 * H : a.bias_h;
      const float ghn = ghm[2 * H + c];
      const float r = 1.f / (1.f + expf(-(gim[c] + ghm[c])));
      const float z = 1.f / (1.f + expf(-(gim[H + c] + ghm[H + c])));
      const float nn = tanhf(gim[2 * H + c] + r * ghn);
      const float hp = a.h_prev ? a.h_prev[(size_t)m * H + c] : 0.f;
      float dh = a.dh0[(size_t)m * a.ld0 + c];
      if (a.dh1) dh += a.dh1[(size_t)m * a.ld1 + c];
      const float dnn = dh * (1.f - z), dz = dh * (hp - nn);
      const float dpn = dnn * (1.f - nn * nn);
      const float dpr = dpn * ghn * r * (1.f - r), dpz = dz * z * (1.f - z);
      o[0] = dpr; o[1] = dpz; o[2] = dpn; o[3] = dpr; o[4] = dpz; o[5] = dpn * r;
      if (a.dhp) a.dhp[(size_t)m * H + c] = dh * z;
    }
#pragma unroll
    for (int q = 0; q < 6; ++q) tile[q][rl * 36 + cl] = o[q];
  }
  __syncthreads();
  // transposed fragments: 6 tiles x 2 column halves; row-major fragments of dgh: 3 tiles x 2 row halves
  const int ntask = 12 + (a.dgh_r ? 6 : 0);
  for (int task = wave; task < ntask; task += 4) {
    float x[8];
    bf16x8* o;
    if (task < 12) {
      const int q = task >> 1, half = task & 1;   // q: tile, gate g = q % 3
      const int cc = 16 * half + (lane & 15), kg = lane >> 4;
#pragma unroll
      for (int e = 0; e < 8; ++e) x[e] = tile[q][(8 * kg + e) * 36 + cc];
      bf16x8* img = q < 3 ? a.dgiT : a.dghT;
      const int row_tile = ((q % 3) * H + ct * 32) / 16 + half;
      o = img + ((size_t)row_tile * a.ST + a.s0T + mt) * 3 * 64 + lane;
    } else {
      const int g = (task - 12) >> 1, half = (task - 12) & 1;
      const int row = 16 * half + (lane & 15), kg = lane >> 4;
      const f32x4 x0 = *reinterpret_cast<const f32x4*>(&tile[3 + g][row * 36 + 8 * kg]), x1 = *reinterpret_cast<const f32x4*>(&tile[3 + g][row * 36 + 8 * kg + 4]);
      x[0] = x0[0]; x[1] = x0[1]; x[2] = x0[2]; x[3] = x0[3]; x[4] = x1[0]; x[5] = x1[1]; x[6] = x1[2]; x[7] = x1[3];
      const int S = (3 * H) >> 5;
      o = a.dgh_r + ((size_t)(2 * mt + half) * S + (g * H) / 32 + ct) * 3 * 64 + lane;
    }
    bf16x8 pl[3];
    u3_split(x, pl);
#pragma unroll
    for (int p = 0; p < 3; ++p) o[p * 64] = pl[p];
  }
}

// ---- device arena ---------------------------------------------------------------------------------------------------------
struct Arena {
  char* base = nullptr;
  size_t off = 0, cap = 0;
  void* take(size_t bytes) {
    void* p = base ? base + off : nullptr;
    off = egx_align_up(off + bytes, 256);
    return p;
  }
};

constexpr int HD = 512, CAT = 1152, ZP = 256, ST_DIM = 402, EGO_DIM = 32;

struct Branch {   // actor or critic MLP block
  const float* W[4]; const float* b[4]; const float* Wout; const float* bout;
  float* gW[4]; float* gb[4]; float* gWout; float* gbout;
  int nout;
  bf16x8 *W_r[4], *W_t[4], *Wout_r, *Wout_t;   // weight images
  float *a[4], *u1f, *du2, *du1, *dhx;         // saved activations (fp32), unit-1 output, fp32 gradients of u2 / u1 / hx
  bf16x8 *a1_r, *u1_r, *a3_r, *u2_r, *a1T, *u1T, *a3T, *u2T;
  bf16x8 *g_r[4], *gT[4];                      // images of the gradient w.r.t. the pre-activation of layer l (and their transposes)
  float* head;                                 // zp [n,256] / value [n]
  float* ghead;                                // loss gradient w.r.t. the head's output
  bf16x8 *ghead_r, *gheadT;
};
struct Encoder {   // one 2-step GRU
  const float *Wih, *Whh, *bih, *bhh;
  float *gWih, *gWhh, *gbih, *gbhh;
  int in_dim, cat_off;
  bf16x8 *Wih_r, *Whh_r, *Whh_t;
  bf16x8 *x0_r, *x1_r, *xT;        // inputs: the two frames, and [x0; x1]^T with a row of ones
  float *gi1, *gi2, *gh2, *h1f;
  bf16x8 *h1_r, *hprevT;           // h1 as an operand; [h0 = 0; h1]^T with a row of ones
  bf16x8 *dgiT, *dghT, *dgh_r;
  float *dhp, *dh1;
};

}  // namespace

struct egx_policy_train {
  int n = 0, Sn = 0;
  Arena ar;
  Encoder enc[2];
  Branch br[2];
  float* catf = nullptr;
  bf16x8 *cat_r = nullptr, *catT = nullptr;
  PackEntry* tab_weights = nullptr; int n_weights = 0, frags_weights = 0;
  PackEntry* tab_inputs = nullptr; int n_inputs = 0, frags_inputs = 0;
  PackEntry* tab_loss = nullptr; int n_loss = 0, frags_loss = 0;
  std::vector<PackEntry> h_inputs;   // host copy: the source pointers of the observation change per call
  const float* bound_state = nullptr; const float* bound_ego = nullptr;
  int prec = 0;   // arithmetic of every product of the chain (D3Plain::prec): egx_policy_train_set_precision
};

namespace {

int upload_table(std::vector<PackEntry>& v, PackEntry** dev, int* frags) {
  int run = 0;
  for (PackEntry& e : v) {
    if (e.transpose == 2) {
      run += egx_ceil_div(e.red, 32) * egx_ceil_div(e.cols, 32);
    } else {
      const int rows = e.transpose ? e.cols + (e.ones_row >= 0 ? 1 : 0) : e.red;
      const int red = e.transpose ? e.red : e.cols;
      run += (int)img_frags(rows, red);
    }
    e.frag_end = run;
  }
  *frags = run;
  if (!*dev) EGX_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(dev), v.size() * sizeof(PackEntry)));
  EGX_HIP_CHECK(hipMemcpy(*dev, v.data(), v.size() * sizeof(PackEntry), hipMemcpyHostToDevice));
  return EGX_OK;
}
void run_table(hipStream_t st, const PackEntry* tab, int n, int frags, int nplanes = 3) {
  hipLaunchKernelGGL(egx_pack3_table_kernel, dim3(egx_ceil_div(frags, 4)), dim3(256), 0, st, tab, n, nplanes);
}
inline int planes_of(int prec) { return prec == 0 ? 3 : (prec == 2 ? 2 : 1); }

// carve every buffer of a handle; with ar.base == nullptr this only measures
void layout(egx_policy_train* h) {
  const int n = h->n, Sn = h->Sn;
  Arena& ar = h->ar;
  ar.off = 0;
  auto img = [&](int rows, int red) { return static_cast<bf16x8*>(ar.take(img_frags(rows, red) * FRAG_BYTES)); };
  auto f32 = [&](size_t count) { return static_cast<float*>(ar.take(count * sizeof(float))); };
  for (int e = 0; e < 2; ++e) {
    Encoder& E = h->enc[e];
    E.Wih_r = img(3 * HD, E.in_dim); E.Whh_r = img(3 * HD, HD); E.Whh_t = img(HD, 3 * HD);
    E.x0_r = img(n, E.in_dim); E.x1_r = img(n, E.in_dim); E.xT = img(E.in_dim + 1, 2 * n);
    E.gi1 = f32((size_t)n * 3 * HD); E.gi2 = f32((size_t)n * 3 * HD); E.gh2 = f32((size_t)n * 3 * HD); E.h1f = f32((size_t)n * HD);
    E.h1_r = img(n, HD); E.hprevT = img(HD + 1, 2 * n);
    E.dgiT = img(3 * HD, 2 * n); E.dghT = img(3 * HD, 2 * n); E.dgh_r = img(n, 3 * HD);
    E.dhp = f32((size_t)n * HD); E.dh1 = f32((size_t)n * HD);
  }
  h->catf = f32((size_t)n * CAT); h->cat_r = img(n, CAT); h->catT = img(CAT + 1, n);
  for (int b = 0; b < 2; ++b) {
    Branch& B = h->br[b];
    for (int l = 0; l < 4; ++l) { B.W_r[l] = img(CAT, CAT); B.W_t[l] = img(CAT, CAT); B.a[l] = f32((size_t)n * CAT); }
    B.Wout_r = img(B.nout, CAT); B.Wout_t = img(CAT, B.nout);
    B.u1f = f32((size_t)n * CAT); B.du2 = f32((size_t)n * CAT); B.du1 = f32((size_t)n * CAT); B.dhx = f32((size_t)n * CAT);
    B.a1_r = img(n, CAT); B.u1_r = img(n, CAT); B.a3_r = img(n, CAT); B.u2_r = img(n, CAT);
    B.a1T = img(CAT + 1, n); B.u1T = img(CAT + 1, n); B.a3T = img(CAT + 1, n); B.u2T = img(CAT + 1, n);
    for (int l = 0; l < 4; ++l) { B.g_r[l] = img(n, CAT); B.gT[l] = img(CAT, n); }
    B.head = f32((size_t)n * B.nout); B.ghead = f32((size_t)n * B.nout);
    B.ghead_r = img(n, B.nout); B.gheadT = img(B.nout, n);
  }
  (void)Sn;
}

// the all-ones row of a transposed activation image whose other rows are written by kernel epilogues
__global__ void egx_ones_row_kernel(bf16x8* img, int row, int S_total, int s_lo, int s_hi) {
  const int s = s_lo + blockIdx.x, kg = threadIdx.x;   // 4 threads: the k-groups of one fragment
  if (s >= s_hi) return;
  const int lane = (row & 15) + 16 * kg;
  bf16x8 one;
#pragma unroll
  for (int e = 0; e < 8; ++e) one[e] = (short)0x3F80;   // bf16 1.0
  img[((size_t)(row >> 4) * S_total + s) * 3 * 64 + lane] = one;   // plane 0; planes 1, 2 stay zero
}

}  // namespace

extern "C" int egx_policy_train_create(const egx_policy_weights* w, const egx_policy_grads* g, int num_rows, egx_policy_train** out) {
  EGX_REQUIRE(w && g && out, "null argument");
  EGX_REQUIRE(num_rows > 0 && num_rows % 32 == 0, "the minibatch must be a multiple of 32 rows");
  auto* h = new egx_policy_train();
  h->n = num_rows; h->Sn = num_rows / 32;
  Encoder& X = h->enc[0];
  X.Wih = w->x_enc_w_ih; X.Whh = w->x_enc_w_hh; X.bih = w->x_enc_b_ih; X.bhh = w->x_enc_b_hh;
  X.gWih = g->x_enc_w_ih; X.gWhh = g->x_enc_w_hh; X.gbih = g->x_enc_b_ih; X.gbhh = g->x_enc_b_hh;
  X.in_dim = ST_DIM; X.cat_off = 0;
  Encoder& E = h->enc[1];
  E.Wih = w->ego_enc_w_ih; E.Whh = w->ego_enc_w_hh; E.bih = w->ego_enc_b_ih; E.bhh = w->ego_enc_b_hh;
  E.gWih = g->ego_enc_w_ih; E.gWhh = g->ego_enc_w_hh; E.gbih = g->ego_enc_b_ih; E.gbhh = g->ego_enc_b_hh;
  E.in_dim = EGO_DIM; E.cat_off = HD;
  for (int l = 0; l < 4; ++l) {
    h->br[0].W[l] = w->actor_w[l]; h->br[0].b[l] = w->actor_b[l]; h->br[0].gW[l] = g->actor_w[l]; h->br[0].gb[l] = g->actor_b[l];
    h->br[1].W[l] = w->critic_w[l]; h->br[1].b[l] = w->critic_b[l]; h->br[1].gW[l] = g->critic_w[l]; h->br[1].gb[l] = g->critic_b[l];
  }
  h->br[0].Wout = w->actor_out_w; h->br[0].bout = w->actor_out_b; h->br[0].gWout = g->actor_out_w; h->br[0].gbout = g->actor_out_b;
  h->br[0].nout = ZP;
  h->br[1].Wout = w->critic_out_w; h->br[1].bout = w->critic_out_b; h->br[1].gWout = g->critic_out_w; h->br[1].gbout = g->critic_out_b;
  h->br[1].nout = 1;
  layout(h);
  h->ar.cap = h->ar.off;
  if (hipMalloc(reinterpret_cast<void**>(&h->ar.base), h->ar.cap) != hipSuccess) {
    delete h;
    egx_set_error("egx_policy_train_create: device allocation failed");
    return EGX_ERR_HIP;
  }
  // every failure below releases the handle, its arena and whatever tables were uploaded (the caller never sees `h`)
  auto fail = [&](int rc_) { egx_policy_train_destroy(h); return rc_; };
#define U3_CHECK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { egx_set_error(std::string("egx_policy_train_create: ") + hipGetErrorString(e_)); return fail(EGX_ERR_HIP); } } while (0)
  U3_CHECK(hipMemset(h->ar.base, 0, h->ar.cap));
  layout(h);
  const int n = h->n, Sn = h->Sn;
  // rows of ones of the transposed activation images the epilogues fill (the tables write their own)
  auto ones = [&](bf16x8* img, int row, int S_total, int s_lo, int s_hi) {
    hipLaunchKernelGGL(egx_ones_row_kernel, dim3(s_hi - s_lo), dim3(4), 0, nullptr, img, row, S_total, s_lo, s_hi);
  };
  for (int e = 0; e < 2; ++e) ones(h->enc[e].hprevT, HD, 2 * Sn, 0, 2 * Sn);
  ones(h->catT, CAT, Sn, 0, Sn);
  for (int b = 0; b < 2; ++b)
    for (bf16x8* img : {h->br[b].a1T, h->br[b].u1T, h->br[b].a3T, h->br[b].u2T}) ones(img, CAT, Sn, 0, Sn);
  U3_CHECK(hipGetLastError());
  // ---- pack tables
  std::vector<PackEntry> tw;
  auto add = [](std::vector<PackEntry>& v, const float* src, int red, int cols, int ld, int col0, bf16x8* dst, int S_total, int s0,
                int transpose, int ones_row) {
    PackEntry e;
    e.src = src; e.red = red; e.cols = cols; e.ld = ld; e.col0 = col0; e.dst = dst; e.S_total = S_total; e.s0 = s0;
    e.transpose = transpose; e.ones_row = ones_row; e.frag_end = 0; e.dst2 = nullptr; e.S_total2 = 0;
    v.push_back(e);
  };
  // W and W^T images of a whole weight matrix [rows, cols] from one read of it
  auto add_both = [](std::vector<PackEntry>& v, const float* src, int rows, int cols, bf16x8* img, bf16x8* imgT) {
    PackEntry e;
    e.src = src; e.red = rows; e.cols = cols; e.ld = cols; e.col0 = 0; e.dst = img; e.S_total = egx_ceil_div(cols, 32); e.s0 = 0;
    e.transpose = 2; e.ones_row = -1; e.frag_end = 0; e.dst2 = imgT; e.S_total2 = egx_ceil_div(rows, 32);
    v.push_back(e);
  };
  for (int e = 0; e < 2; ++e) {
    Encoder& En = h->enc[e];
    add(tw, En.Wih, 3 * HD, En.in_dim, En.in_dim, 0, En.Wih_r, egx_ceil_div(En.in_dim, 32), 0, 0, -1);
    add_both(tw, En.Whh, 3 * HD, HD, En.Whh_r, En.Whh_t);
  }
  for (int b = 0; b < 2; ++b) {
    Branch& B = h->br[b];
    for (int l = 0; l < 4; ++l) {
      add_both(tw, B.W[l], CAT, CAT, B.W_r[l], B.W_t[l]);
    }
    add_both(tw, B.Wout, B.nout, CAT, B.Wout_r, B.Wout_t);
  }
  h->n_weights = (int)tw.size();
  int rc = upload_table(tw, &h->tab_weights, &h->frags_weights);
  if (rc) return fail(rc);
  // inputs: the two frames of every encoder as operands, and [frame 0; frame 1]^T (+ ones) for the weight gradient of W_ih
  for (int e = 0; e < 2; ++e) {
    Encoder& En = h->enc[e];
    const int d = En.in_dim, S = egx_ceil_div(d, 32);
    add(h->h_inputs, nullptr, n, d, 2 * d, 0, En.x0_r, S, 0, 0, -1);
    add(h->h_inputs, nullptr, n, d, 2 * d, d, En.x1_r, S, 0, 0, -1);
    add(h->h_inputs, nullptr, n, d, 2 * d, 0, En.xT, 2 * Sn, 0, 1, d);
    add(h->h_inputs, nullptr, n, d, 2 * d, d, En.xT, 2 * Sn, Sn, 1, d);
  }
  h->n_inputs = (int)h->h_inputs.size();
  std::vector<PackEntry> tl;
  for (int b = 0; b < 2; ++b) {
    Branch& B = h->br[b];
    add(tl, B.ghead, n, B.nout, B.nout, 0, B.ghead_r, egx_ceil_div(B.nout, 32), 0, 0, -1);
    add(tl, B.ghead, n, B.nout, B.nout, 0, B.gheadT, Sn, 0, 1, -1);
  }
  h->n_loss = (int)tl.size();
  if ((rc = upload_table(tl, &h->tab_loss, &h->frags_loss))) return fail(rc);
  U3_CHECK(hipDeviceSynchronize());
#undef U3_CHECK
  *out = h;
  return EGX_OK;
}

extern "C" void egx_policy_train_destroy(egx_policy_train* h) {
  if (!h) return;
  (void)hipFree(h->ar.base); (void)hipFree(h->tab_weights); (void)hipFree(h->tab_inputs); (void)hipFree(h->tab_loss);
  delete h;
}

extern "C" int egx_policy_train_refresh(egx_policy_train* h, void* stream) {
  EGX_REQUIRE(h, "null handle");
  // the weight images are read by this chain (h->prec) and by the rollout forward (egx_policy_get_precision): only the planes
  // one of them uses are written
  const int planes = std::max(planes_of(h->prec), planes_of(egx_policy_get_precision()));
  run_table(static_cast<hipStream_t>(stream), h->tab_weights, h->n_weights, h->frags_weights, planes);
  EGX_HIP_CHECK(hipGetLastError());
  return EGX_OK;
}

extern "C" int egx_policy_train_set_precision(egx_policy_train* h, int prec) {
  EGX_REQUIRE(h, "null handle");
  EGX_REQUIRE(prec == 0 || prec == 1 || prec == 2, "precision must be 0 (three bf16 planes: fp32-equivalent), 2 (two planes) or 1 (bf16)");
  h->prec = prec;
  return EGX_OK;
}

extern "C" int egx_policy_train_packed(const egx_policy_train* h, egx_policy_packed3* out) {
  EGX_REQUIRE(h && out, "null argument");
  out->x_enc_w_ih = h->enc[0].Wih_r; out->x_enc_w_hh = h->enc[0].Whh_r;
  out->ego_enc_w_ih = h->enc[1].Wih_r; out->ego_enc_w_hh = h->enc[1].Whh_r;
  for (int l = 0; l < 4; ++l) { out->actor_w[l] = h->br[0].W_r[l]; out->critic_w[l] = h->br[1].W_r[l]; }
  out->actor_out_w = h->br[0].Wout_r; out->critic_out_w = h->br[1].Wout_r;
  return EGX_OK;
}

extern "C" int egx_policy_train_bind(egx_policy_train* h, const float* state, const float* egosensing) {
  EGX_REQUIRE(h && state && egosensing, "null argument");
  if (h->bound_state == state && h->bound_ego == egosensing && h->tab_inputs) return EGX_OK;
  for (int i = 0; i < 4; ++i) h->h_inputs[i].src = state;
  for (int i = 4; i < 8; ++i) h->h_inputs[i].src = egosensing;
  int rc = upload_table(h->h_inputs, &h->tab_inputs, &h->frags_inputs);
  if (rc) return rc;
  h->bound_state = state; h->bound_ego = egosensing;
  return EGX_OK;
}

static int train_step_parts(egx_policy_train* h, const float* dist, const float* time, const float* act, const float* adv,
                            const float* ret, const float* logp_old, const float* adv_stats, const float* scale, float adv_eps,
                            float min_logvar, float max_logvar, float eps_clip, float vf_coef, float ent_coef, float* out_terms,
                            void* stream_, int parts);

extern "C" int egx_policy_train_step(egx_policy_train* h, const float* dist, const float* time, const float* act, const float* adv,
                                     const float* ret, const float* logp_old, const float* adv_stats, const float* scale,
                                     float adv_eps, float min_logvar, float max_logvar, float eps_clip, float vf_coef, float ent_coef,
                                     float* out_terms, void* stream_) {
  return train_step_parts(h, dist, time, act, adv, ret, logp_old, adv_stats, scale, adv_eps, min_logvar, max_logvar, eps_clip, vf_coef,
                          ent_coef, out_terms, stream_, 3);
}

// The same chain in two halves, for data-parallel training: `egx_policy_train_step_heads` ends when the last weight gradient of
// the actor and critic blocks has been enqueued (forward, loss, their whole backward: the 10.3 M-parameter prefix of the flat
// gradient is final), `egx_policy_train_step_encoders` runs the rest (the two GRU encoders' backward).  The caller all-reduces
// the first bucket on a side stream while the second half runs (egogen_amd/ppo_policy.py).
extern "C" int egx_policy_train_step_heads(egx_policy_train* h, const float* dist, const float* time, const float* act, const float* adv,
                                           const float* ret, const float* logp_old, const float* adv_stats, const float* scale,
                                           float adv_eps, float min_logvar, float max_logvar, float eps_clip, float vf_coef,
                                           float ent_coef, float* out_terms, void* stream_) {
  return train_step_parts(h, dist, time, act, adv, ret, logp_old, adv_stats, scale, adv_eps, min_logvar, max_logvar, eps_clip, vf_coef,
                          ent_coef, out_terms, stream_, 1);
}
extern "C" int egx_policy_train_step_encoders(egx_policy_train* h, void* stream_) {
  EGX_REQUIRE(h, "null handle");
  return train_step_parts(h, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, nullptr,
                          stream_, 2);
}

static int train_step_parts(egx_policy_train* h, const float* dist, const float* time, const float* act, const float* adv,
                            const float* ret, const float* logp_old, const float* adv_stats, const float* scale, float adv_eps,
                            float min_logvar, float max_logvar, float eps_clip, float vf_coef, float ent_coef, float* out_terms,
                            void* stream_, int parts) {
  EGX_REQUIRE(h && (!(parts & 1) || (dist && time && act && adv && ret && logp_old && scale && out_terms)), "null argument");
  EGX_REQUIRE(h->tab_inputs, "egx_policy_train_bind has not been called");
  hipStream_t st = static_cast<hipStream_t>(stream_);
  const int n = h->n, Sn = h->Sn;
  constexpr int S_HD = HD / 32, S_CAT = CAT / 32, S_G = 3 * HD / 32;
  const float slope = 0.01f;   // torch.nn.LeakyReLU() default (baseops.py:627-628)
  const int prec = h->prec;
  auto launch_n = [prec](hipStream_t s_, D3Plain* ps, int cnt) {
    for (int i = 0; i < cnt; ++i) ps[i].prec = prec;
    egx_launch_dense3_n(s_, ps, cnt);
  };
  auto launch_gru = [prec](hipStream_t s_, D3Gru& g0, D3Gru& g1) {
    g0.prec = g1.prec = prec;
    return egx_launch_gru3_pair(s_, g0, g1);
  };

  int rc = EGX_OK;
  if (parts & 1) {
  // ================= forward =================
  run_table(st, h->tab_inputs, h->n_inputs, h->frags_inputs, planes_of(h->prec));
  egx_launch_posenc3(st, dist, time, n, h->catf + 2 * HD, CAT, h->cat_r, S_CAT, 2 * S_HD, h->catT, Sn, 2 * HD, out_terms);   // also clears the loss sums
  {
    D3Gru g[2];
    for (int e = 0; e < 2; ++e) {   // step 1: zero previous state
      Encoder& E = h->enc[e];
      const int Sx = egx_ceil_div(E.in_dim, 32);
      D3Gru& q = g[e];
      q.M = n; q.H = HD;
      q.Ai = E.x0_r; q.SAi = Sx; q.Bi = E.Wih_r; q.Si = Sx; q.bias_i = E.bih; q.bias_h = E.bhh;
      q.gi_out = E.gi1; q.h_out = E.h1f; q.ldo = HD; q.h_out3 = E.h1_r; q.S3 = S_HD;
      q.h_out3T = E.hprevT; q.S3T = 2 * Sn; q.s3T0 = Sn; q.col0T = 0;
    }
    launch_gru(st, g[0], g[1]);
    for (int e = 0; e < 2; ++e) {   // step 2 -> [hx | he | pe]
      Encoder& E = h->enc[e];
      D3Gru& q = g[e];
      q.Ai = E.x1_r; q.gi_out = E.gi2; q.gh_out = E.gh2;
      q.Ah = E.h1_r; q.SAh = S_HD; q.Bh = E.Whh_r; q.Sh = S_HD; q.h_prev = E.h1f; q.ldh = HD;
      q.h_out = h->catf + E.cat_off; q.ldo = CAT; q.h_out3 = h->cat_r; q.S3 = S_CAT; q.s30 = E.cat_off / 32;
      q.h_out3T = h->catT; q.S3T = Sn; q.s3T0 = 0; q.col0T = E.cat_off;
    }
    launch_gru(st, g[0], g[1]);
  }
  {
    D3Plain L[2];
    // one dense layer of both MLP blocks: in -> lrelu(W_l in + b_l) (+ res); saves the activation, writes the images of its output
    auto fwd = [&](int l, const bf16x8* in_a, const bf16x8* in_c, const float* res_a, const float* res_c, float* outf_a, float* outf_c,
                   bf16x8* o3_a, bf16x8* o3_c, bf16x8* o3T_a, bf16x8* o3T_c) {
      const bf16x8* in[2] = {in_a, in_c};
      const float* res[2] = {res_a, res_c};
      float* outf[2] = {outf_a, outf_c};
      bf16x8* o3[2] = {o3_a, o3_c};
      bf16x8* o3T[2] = {o3T_a, o3T_c};
      for (int b = 0; b < 2; ++b) {
        Branch& B = h->br[b];
        D3Plain& q = L[b];
        q = D3Plain();
        q.M = n; q.N = CAT; q.A = in[b]; q.SA = S_CAT; q.S = S_CAT; q.B = B.W_r[l]; q.bias = B.b[l]; q.act = 3; q.slope = slope;
        q.out_act = B.a[l]; q.ldact = CAT;
        q.res = res[b]; q.ldr = CAT; q.out = outf[b]; q.ldo = CAT;
        q.out3 = o3[b]; q.S3 = S_CAT; q.out3T = o3T[b]; q.S3T = Sn;
      }
      launch_n(st, L, 2);
    };
    Branch &Ba = h->br[0], &Bc = h->br[1];
    // h = hx; per unit: h = lrelu(fc2(lrelu(fc1(h)))) + h   (models_policy_ppo.py:24-39, baseops.py:615-641)
    fwd(0, h->cat_r, h->cat_r, nullptr, nullptr, nullptr, nullptr, Ba.a1_r, Bc.a1_r, Ba.a1T, Bc.a1T);
    fwd(1, Ba.a1_r, Bc.a1_r, h->catf, h->catf, Ba.u1f, Bc.u1f, Ba.u1_r, Bc.u1_r, Ba.u1T, Bc.u1T);
    fwd(2, Ba.u1_r, Bc.u1_r, nullptr, nullptr, nullptr, nullptr, Ba.a3_r, Bc.a3_r, Ba.a3T, Bc.a3T);
    fwd(3, Ba.a3_r, Bc.a3_r, Ba.u1f, Bc.u1f, nullptr, nullptr, Ba.u2_r, Bc.u2_r, Ba.u2T, Bc.u2T);
    for (int b = 0; b < 2; ++b) {   // heads: zp = [mu | logvar] and the value
      Branch& B = h->br[b];
      D3Plain& q = L[b];
      q = D3Plain();
      q.M = n; q.N = B.nout; q.A = B.u2_r; q.SA = S_CAT; q.S = S_CAT; q.B = B.Wout_r; q.bias = B.bout; q.out = B.head; q.ldo = B.nout;
    }
    launch_n(st, L, 2);
  }
  // ================= loss and its gradient w.r.t. the two heads (ppo_policy.py:189-241) =================
  rc = egx_ppo_loss_packed_precleared(h->br[0].head, h->br[1].head, act, adv, ret, logp_old, adv_stats, scale, adv_eps, min_logvar, max_logvar,
                               eps_clip, vf_coef, ent_coef, n, h->br[0].ghead, h->br[1].ghead, out_terms, st);
  if (rc) return rc;
  run_table(st, h->tab_loss, h->n_loss, h->frags_loss, planes_of(h->prec));
  // ================= backward =================
  // Input-gradient products form the dependent chain (out_fc -> unit 2 -> unit 1 -> GRU cells); each layer's weight-gradient
  // product follows the launch that left its gradient image behind.  (Weight gradients on a second stream beside the chain,
  // and fused into the input-gradient launches, were both measured and are not faster: profiles/r03_update_experiments.md.)
  hipStream_t sw = st;
  auto wgrad = [&](D3Plain& q, const bf16x8* gT, const bf16x8* xT, int rows, float* gW, float* gb) {
    q = D3Plain();
    q.M = rows; q.N = CAT + 1; q.A = gT; q.SA = Sn; q.S = Sn; q.B = xT; q.out = gW; q.ldo = CAT; q.n_split = CAT; q.bias_out = gb;
  };
  Branch &Ba = h->br[0], &Bc = h->br[1];
  D3Plain W[4], D[2];
  for (int b = 0; b < 2; ++b) wgrad(W[b], h->br[b].gheadT, h->br[b].u2T, h->br[b].nout, h->br[b].gWout, h->br[b].gbout);
  launch_n(sw, W, 2);
  // input gradient of one layer for both blocks: d = g_in W^T-image (+ skip) -> fp32 `out` raw, images gated by act'(a[gate])
  auto dgrad = [&](const bf16x8* const (&gin)[2], const int (&S_in)[2], const bf16x8* const (&Wt)[2], float* const (&res)[2],
                   float* const (&outf)[2], int gate, int lout) {
    for (int b = 0; b < 2; ++b) {
      Branch& B = h->br[b];
      D3Plain& d = D[b];
      d = D3Plain();
      d.M = n; d.N = CAT; d.A = gin[b]; d.SA = S_in[b]; d.S = S_in[b]; d.B = Wt[b];
      d.res = res[b]; d.ldr = CAT; d.out = outf[b]; d.ldo = CAT;
      if (gate >= 0) {
        d.dact = B.a[gate]; d.lddact = CAT; d.dact_code = 3; d.dact_slope = slope;
        d.out3 = B.g_r[lout]; d.S3 = S_CAT; d.out3T = B.gT[lout]; d.S3T = Sn;
      }
    }
    launch_n(st, D, 2);
  };
  const int S_head[2] = {egx_ceil_div(Ba.nout, 32), egx_ceil_div(Bc.nout, 32)}, S_full[2] = {S_CAT, S_CAT};
  float* const none[2] = {nullptr, nullptr};
  {   // out_fc: du2 = g W_out -> g4 = du2 x lrelu'(a4)
    const bf16x8* const gin[2] = {Ba.ghead_r, Bc.ghead_r};
    const bf16x8* const Wt[2] = {Ba.Wout_t, Bc.Wout_t};
    float* const outf[2] = {Ba.du2, Bc.du2};
    dgrad(gin, S_head, Wt, none, outf, 3, 3);
  }
  // layer l of both blocks: the input gradient now (the dependent chain); its weight gradient dW_l = g_l^T x_l only needs the
  // images this launch leaves behind, so the eight weight-gradient products of the four layers go out afterwards as two
  // four-product launches (fewer, fuller launches: ~10 us of every launch is fixed cost, profiles/r03_update_experiments.md)
  D3Plain WL[8];
  auto layer_bwd = [&](int l, bf16x8* const (&xT)[2], float* const (&res)[2], float* const (&outf)[2], int gate) {
    for (int b = 0; b < 2; ++b) wgrad(WL[2 * l + b], h->br[b].gT[l], xT[b], CAT, h->br[b].gW[l], h->br[b].gb[l]);
    const bf16x8* const gin[2] = {Ba.g_r[l], Bc.g_r[l]};
    const bf16x8* const Wt[2] = {Ba.W_t[l], Bc.W_t[l]};
    dgrad(gin, S_full, Wt, res, outf, gate, gate);
    return EGX_OK;
  };
  {
    bf16x8* const x3[2] = {Ba.a3T, Bc.a3T};
    if ((rc = layer_bwd(3, x3, none, none, 2))) return rc;                    // d(a3) = g4 W4 -> g3
    bf16x8* const x2[2] = {Ba.u1T, Bc.u1T};
    float* const du2[2] = {Ba.du2, Bc.du2};
    float* const du1[2] = {Ba.du1, Bc.du1};
    if ((rc = layer_bwd(2, x2, du2, du1, 1))) return rc;                      // du1 = g3 W3 + du2 -> g2
    bf16x8* const x1[2] = {Ba.a1T, Bc.a1T};
    if ((rc = layer_bwd(1, x1, none, none, 0))) return rc;                    // d(a1) = g2 W2 -> g1
    bf16x8* const x0[2] = {h->catT, h->catT};
    float* const dhx[2] = {Ba.dhx, Bc.dhx};
    if ((rc = layer_bwd(0, x0, du1, dhx, -1))) return rc;                     // dhx = g1 W1 + du1
    launch_n(sw, WL, 4);
    launch_n(sw, WL + 4, 4);
  }
  }   // parts & 1: every gradient of the actor and critic blocks is enqueued
  // ---- the two GRU encoders (dhx = actor's + critic's)
  if (parts & 2) {
    hipStream_t sw = st;
    GruBwd2 two;
    const int blocks = (n / 32) * (HD / 32);
    two.blocks0 = blocks;
    GruBwd* gb[2] = {&two.a, &two.b};
    for (int e = 0; e < 2; ++e) {   // step 2
      Encoder& E = h->enc[e];
      GruBwd& q = *gb[e];
      q.gi = E.gi2; q.gh = E.gh2; q.bias_h = E.bhh; q.h_prev = E.h1f;
      q.dh0 = h->br[0].dhx + E.cat_off; q.ld0 = CAT; q.dh1 = h->br[1].dhx + E.cat_off; q.ld1 = CAT;
      q.dgiT = E.dgiT; q.dghT = E.dghT; q.ST = 2 * Sn; q.s0T = Sn; q.dgh_r = E.dgh_r; q.dhp = E.dhp; q.M = n; q.H = HD;
    }
    hipLaunchKernelGGL(egx_gru_bwd3_kernel, dim3(2 * blocks), dim3(256), 0, st, two);
    D3Plain P[4];
    for (int e = 0; e < 2; ++e) {   // dh1 = dh z + dgh2 W_hh
      Encoder& E = h->enc[e];
      D3Plain& d = P[e];
      d = D3Plain();
      d.M = n; d.N = HD; d.A = E.dgh_r; d.SA = S_G; d.S = S_G; d.B = E.Whh_t; d.res = E.dhp; d.ldr = HD; d.out = E.dh1; d.ldo = HD;
    }
    launch_n(st, P, 2);
    for (int e = 0; e < 2; ++e) {   // step 1 (h0 = 0: gh = b_hh)
      Encoder& E = h->enc[e];
      GruBwd& q = *gb[e];
      q.gi = E.gi1; q.gh = nullptr; q.h_prev = nullptr; q.dh0 = E.dh1; q.ld0 = HD; q.dh1 = nullptr; q.ld1 = 0;
      q.s0T = 0; q.dgh_r = nullptr; q.dhp = nullptr;
    }
    hipLaunchKernelGGL(egx_gru_bwd3_kernel, dim3(2 * blocks), dim3(256), 0, st, two);
      for (int e = 0; e < 2; ++e) {   // dW_ih = [dgi1; dgi2]^T [x0; x1], dW_hh = [dgh1; dgh2]^T [0; h1]; the ones rows give the biases
      Encoder& E = h->enc[e];
      D3Plain& a = P[2 * e];
      a = D3Plain();
      a.M = 3 * HD; a.N = E.in_dim + 1; a.A = E.dgiT; a.SA = 2 * Sn; a.S = 2 * Sn; a.B = E.xT; a.out = E.gWih; a.ldo = E.in_dim;
      a.n_split = E.in_dim; a.bias_out = E.gbih;
      D3Plain& c = P[2 * e + 1];
      c = D3Plain();
      c.M = 3 * HD; c.N = HD + 1; c.A = E.dghT; c.SA = 2 * Sn; c.S = 2 * Sn; c.B = E.hprevT; c.out = E.gWhh; c.ldo = HD;
      c.n_split = HD; c.bias_out = E.gbhh;
    }
    launch_n(sw, P, 4);
  }
  EGX_HIP_CHECK(hipGetLastError());
  return EGX_OK;
}
