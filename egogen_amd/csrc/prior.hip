// Whole-network entry points: the motion prior (C-VAE decode + body regressor), the policy networks and
// the VPoser encoder, each as one C call that enqueues its chain of fused dense-layer kernels.
#include <algorithm>
#include <atomic>
#include "egx_nets.h"

namespace {
constexpr int H = 256, ZD = 128, MK = 201, T_PRED = 18;

struct Carver {
  char* base;
  size_t off = 0, cap;
  Carver(void* p, size_t c) : base(static_cast<char*>(p)), cap(c) {}
  float* take(size_t nfloat) {
    float* r = reinterpret_cast<float*>(base + off);
    off = egx_align_up(off + nfloat * sizeof(float), 256);
    return r;
  }
};
size_t carve_bytes(std::initializer_list<size_t> nfloats) {
  size_t off = 0;
  for (size_t n : nfloats) off = egx_align_up(off + n * sizeof(float), 256);
  return off;
}
}  // namespace

// packed-activation buffers of the dense3 decoder path: row tiles (of 16) x k-steps x 3072 bytes
constexpr size_t FRAG_FLOATS = 3 * 64 * 4;   // one (row tile, k-step): three planes of 64 lanes x 16 bytes, in floats
inline size_t rt16(int A) { return 2 * (size_t)egx_ceil_div(A, 32); }
constexpr int S_MK = 7, S_H = 8, S_HZ = 12, S_512 = 16;   // k-steps of 201 / 256 / 384 / 512 columns

extern "C" size_t egx_sample_prior_workspace_bytes(int A) {
  if (A <= 0) return 0;
  const size_t a = A;
  const size_t rt = rt16(A);
  return carve_bytes({rt * S_MK * FRAG_FLOATS, rt * S_MK * FRAG_FLOATS, rt * S_HZ * FRAG_FLOATS,
                      rt * S_H * FRAG_FLOATS, rt * S_512 * FRAG_FLOATS, rt * S_H * FRAG_FLOATS,
                      rt * S_H * FRAG_FLOATS, rt * S_H * FRAG_FLOATS, T_PRED * rt * S_H * FRAG_FLOATS,
                      a * 3 * H, a * 3 * H, a * H, a * H, a * H});
}

// The decoder on packed operands (dense3.hip): every activation between two products lives as three bf16 planes in MFMA
// fragment order, written by its producer; the GRU cell is one launch.
static int sample_prior_packed(const egx_prior_weights* w, const float* x0, const float* x1, int x_ld, const float* z, int A,
                               float* out_Y, void* workspace, size_t workspace_bytes, hipStream_t st) {
  const egx_prior_packed3& P = *w->packed3;
  Carver cv(workspace, workspace_bytes);
  const size_t rt = rt16(A);
  auto take3 = [&](size_t ksteps) { return reinterpret_cast<bf16x8*>(cv.take(rt * ksteps * FRAG_FLOATS)); };
  bf16x8* x0p = take3(S_MK);
  bf16x8* x1p = take3(S_MK);
  bf16x8* hz3 = take3(S_HZ);      // [hx | z]
  bf16x8* hB3 = take3(S_H);
  bf16x8* t512 = take3(S_512);
  bf16x8* t256 = take3(S_H);
  bf16x8* h3[2] = {take3(S_H), take3(S_H)};
  bf16x8* hfc_all = reinterpret_cast<bf16x8*>(cv.take(T_PRED * rt * S_H * FRAG_FLOATS));
  const size_t hfc_stride = rt * S_H * 3 * 64;   // fragments per decode step
  float* gi = cv.take((size_t)A * 3 * H);
  float* gconst = cv.take((size_t)A * 3 * H);
  float* hBf = cv.take((size_t)A * H);
  float* hf[2] = {cv.take((size_t)A * H), cv.take((size_t)A * H)};
  auto B3 = [](const void* p) { return static_cast<const bf16x8*>(p); };
  {
    const D3Pack jobs[3] = {{x0, A, MK, x_ld, 0, x0p, S_MK, 0}, {x1, A, MK, x_ld, 0, x1p, S_MK, 0}, {z, A, ZD, ZD, 0, hz3, S_HZ, S_H}};
    egx_launch_pack3(st, jobs, 3);
  }
  // ---- x_enc GRU over the two history frames (zero initial state) -> hx = k-steps 0..7 of [hx | z]
  {
    D3Gru g;
    g.M = A; g.H = H;
    g.Ai = x0p; g.SAi = S_MK; g.Bi = B3(P.x_enc_w_ih); g.Si = S_MK; g.bias_i = w->x_enc_b_ih;
    g.bias_h = w->x_enc_b_hh;
    g.h_out = hBf; g.ldo = H; g.h_out3 = hB3; g.S3 = S_H;
    egx_launch_gru3(st, g);
    g.Ai = x1p;
    g.Ah = hB3; g.SAh = S_H; g.Bh = B3(P.x_enc_w_hh); g.Sh = S_H; g.h_prev = hBf; g.ldh = H;
    g.h_out = nullptr; g.h_out3 = hz3; g.S3 = S_HZ;
    egx_launch_gru3(st, g);
  }
  // ---- h_rnn = drnn_mlp(hx) (tanh after every layer); beside its first layer: the part of the decoder cell's input product
  // that does not change over the 18 steps, gconst = [hx | z] W_ih[:, :384]^T + b_ih
  {
    D3Plain l0, gc;
    l0.M = A; l0.N = 512; l0.A = hz3; l0.SA = S_HZ; l0.S = S_H; l0.B = B3(P.drnn_w[0]); l0.bias = w->drnn_b[0]; l0.act = 1;
    l0.out3 = t512; l0.S3 = S_512;
    gc.M = A; gc.N = 3 * H; gc.A = hz3; gc.SA = S_HZ; gc.S = S_HZ; gc.B = B3(P.d_rnn_w_hz); gc.bias = w->d_rnn_b_ih;
    gc.out = gconst; gc.ldo = 3 * H;
    egx_launch_dense3_pair(st, l0, gc);
    D3Plain l1;
    l1.M = A; l1.N = H; l1.A = t512; l1.SA = S_512; l1.S = S_512; l1.B = B3(P.drnn_w[1]); l1.bias = w->drnn_b[1]; l1.act = 1;
    l1.out3 = t256; l1.S3 = S_H;
    egx_launch_dense3(st, l1);
    D3Plain l2;
    l2.M = A; l2.N = H; l2.A = t256; l2.SA = S_H; l2.S = S_H; l2.B = B3(P.drnn_w[2]); l2.bias = w->drnn_b[2]; l2.act = 1;
    l2.out = hf[0]; l2.ldo = H; l2.out3 = h3[0]; l2.S3 = S_H;
    egx_launch_dense3(st, l2);
  }
  // ---- 18 decode steps of three launches: GRU cell (input product as the running sum gi, see d_comb_w), two MLP layers
  int cur = 0;
  for (int i = 0; i < T_PRED; ++i) {
    D3Gru g;
    g.M = A; g.H = H;
    if (i == 0) {   // y_p = x1: gi = gconst + y_p W_ih[:, 384:]^T
      g.Ai = x1p; g.SAi = S_MK; g.Bi = B3(P.d_rnn_w_y); g.Si = S_MK; g.gi_in = gconst;
    } else {        // gi += hfc_(i-1) d_comb_w^T + d_comb_b
      g.Ai = hfc_all + (size_t)(i - 1) * hfc_stride; g.SAi = S_H; g.Bi = B3(P.d_comb_w); g.Si = S_H; g.bias_i = w->d_comb_b; g.gi_in = gi;
    }
    g.gi_out = gi;
    g.Ah = h3[cur]; g.SAh = S_H; g.Bh = B3(P.d_rnn_w_hh); g.Sh = S_H; g.bias_h = w->d_rnn_b_hh;
    g.h_prev = hf[cur]; g.ldh = H; g.h_out = hf[cur ^ 1]; g.ldo = H; g.h_out3 = h3[cur ^ 1]; g.S3 = S_H;
    egx_launch_gru3(st, g);
    cur ^= 1;
    D3Plain m0;
    m0.M = A; m0.N = 512; m0.A = h3[cur]; m0.SA = S_H; m0.S = S_H; m0.B = B3(P.d_mlp_w[0]); m0.bias = w->d_mlp_b[0]; m0.act = 1;
    m0.out3 = t512; m0.S3 = S_512;
    egx_launch_dense3(st, m0);
    D3Plain m1;
    m1.M = A; m1.N = H; m1.A = t512; m1.SA = S_512; m1.S = S_512; m1.B = B3(P.d_mlp_w[1]); m1.bias = w->d_mlp_b[1]; m1.act = 1;
    m1.out3 = hfc_all + (size_t)i * hfc_stride; m1.S3 = S_H;
    egx_launch_dense3(st, m1);
  }
  // ---- all 18 output layers as one launch (d_i = d_out(hfc_i)), then the residual chain y_i = d_i + y_(i-1) as a scan
  {
    D3Plain o;
    o.M = A; o.N = MK; o.A = hfc_all; o.SA = S_H; o.S = S_H; o.B = B3(P.d_out_w); o.bias = w->d_out_b;
    o.out = out_Y; o.ldo = MK; o.batches = T_PRED; o.batch_strideA = hfc_stride; o.batch_rows_out = A;
    egx_launch_dense3(st, o);
    egx_launch_frame_scan(st, out_Y, x1, x_ld, A, MK, T_PRED);
  }
  return EGX_OK;
}

extern "C" int egx_sample_prior(const egx_prior_weights* w, const float* x0, const float* x1, int x_ld,
                                const float* betas, const float* z, int A, float* out_Y, float* out_Yb,
                                void* workspace, size_t workspace_bytes, void* stream_) {
  EGX_REQUIRE(w && x0 && x1 && betas && z && out_Y && out_Yb, "null argument");
  EGX_REQUIRE(A > 0 && x_ld >= MK, "bad sizes");
  if (!workspace || workspace_bytes < egx_sample_prior_workspace_bytes(A)) {
    egx_set_error("egx_sample_prior: workspace too small");
    return EGX_ERR_WORKSPACE;
  }
  hipStream_t st = static_cast<hipStream_t>(stream_);
  const int M = A * T_PRED;
  EGX_REQUIRE(w->packed3, "egx_sample_prior needs the packed weight images (egx_prior_weights.packed3, built with egx_pack3)");
  EGX_REQUIRE(w->d_comb_w && w->d_comb_b, "egx_sample_prior needs the folded output layer (d_comb_w / d_comb_b)");
  const egx_prior_packed3& P = *w->packed3;
  EGX_REQUIRE(P.reg_in_m && P.reg_in_xb && P.reg_in_betas && P.reg_blk && P.reg_out && P.reg_blk_b, "packed regressor weights missing");
  int rc = sample_prior_packed(w, x0, x1, x_ld, z, A, out_Y, workspace, workspace_bytes, st);
  if (rc) return rc;
  // regressor on all 18*A frames (rows ordered [t][a] like Y_gen.view(nt*nb,-1); betas row = a): one fused launch,
  // 66 dense layers + the 6D -> axis-angle tail
  RegWeights3 r3;
  r3.in_m = static_cast<const bf16x8*>(P.reg_in_m); r3.in_xb = static_cast<const bf16x8*>(P.reg_in_xb);
  r3.in_b3 = static_cast<const bf16x8*>(P.reg_in_betas); r3.in_b = w->reg_in_b;
  r3.blk = static_cast<const bf16x8*>(P.reg_blk); r3.blk_b = P.reg_blk_b;
  r3.out = static_cast<const bf16x8*>(P.reg_out); r3.out_b = w->reg_out_b;
  if ((rc = egx_launch_regressor3(st, r3, out_Y, betas, A, M, out_Yb))) return rc;
  EGX_HIP_CHECK(hipGetLastError());
  return EGX_OK;
}

// ---------------------------------------------------------------------------------------------------------
namespace {
std::atomic<int> g_policy_bf16{0};
}
constexpr int PS_ST = 13, PS_EGO = 1, PS_HD = 16, PS_CAT = 36;   // k-steps of 402 / 32 / 512 / 1152 columns

extern "C" size_t egx_policy_workspace_bytes(int n) {
  if (n <= 0) return 0;
  const size_t m = n;
  const size_t rt = rt16(n);
  return carve_bytes({rt * PS_ST * FRAG_FLOATS, rt * PS_ST * FRAG_FLOATS, rt * PS_EGO * FRAG_FLOATS,
                                          rt * PS_EGO * FRAG_FLOATS, rt * PS_HD * FRAG_FLOATS, rt * PS_HD * FRAG_FLOATS,
                                          rt * PS_CAT * FRAG_FLOATS, rt * PS_CAT * FRAG_FLOATS, rt * PS_CAT * FRAG_FLOATS,
                                          rt * PS_CAT * FRAG_FLOATS, rt * PS_CAT * FRAG_FLOATS, m * 512, m * 512, m * 1152,
                                          m * 1152, m * 1152});
}

extern "C" int egx_policy_set_precision(int bf16) {
  EGX_REQUIRE(bf16 == 0 || bf16 == 1 || bf16 == 2,
              "precision must be 0 (fp32-equivalent), 2 (operands as two bf16 terms) or 1 (bf16 operands), fp32 accumulate in all");
  g_policy_bf16.store(bf16);
  return EGX_OK;
}
extern "C" int egx_policy_get_precision(void) { return g_policy_bf16.load(); }

// The policy networks on packed operands (dense3.hip): GAMMAPolicyBase (two 2-step GRUs + positional encoding, outputs written
// side by side into one [hx | he | pe] buffer), then the actor and critic MLP blocks layer by layer in shared launches.
static int policy_forward_packed(const egx_policy_weights* w, const float* state, const float* ego, const float* dist,
                                 const float* time, int n, float* out_mu, float* out_logvar, float* out_value, void* workspace,
                                 size_t workspace_bytes, hipStream_t st) {
  const egx_policy_packed3& P = *w->packed3;
  const int prec = g_policy_bf16.load();
  Carver cv(workspace, workspace_bytes);
  const size_t rt = rt16(n), m = n;
  auto take3 = [&](size_t ksteps) { return reinterpret_cast<bf16x8*>(cv.take(rt * ksteps * FRAG_FLOATS)); };
  bf16x8* s0p = take3(PS_ST);
  bf16x8* s1p = take3(PS_ST);
  bf16x8* e0p = take3(PS_EGO);
  bf16x8* e1p = take3(PS_EGO);
  bf16x8* hx1 = take3(PS_HD);
  bf16x8* he1 = take3(PS_HD);
  bf16x8* cat3 = take3(PS_CAT);
  bf16x8* a1 = take3(PS_CAT);
  bf16x8* a2 = take3(PS_CAT);
  bf16x8* c1 = take3(PS_CAT);
  bf16x8* c2 = take3(PS_CAT);
  float* hx1f = cv.take(m * 512);
  float* he1f = cv.take(m * 512);
  float* catf = cv.take(m * 1152);   // [hx | he | pe] fp32: residual of the first MLP unit
  float* a2f = cv.take(m * 1152);    // first unit's output, fp32: residual of the second
  float* c2f = cv.take(m * 1152);
  constexpr int HD = 512;
  auto B3 = [](const void* p) { return static_cast<const bf16x8*>(p); };
  {
    const D3Pack jobs[4] = {{state, n, 402, 804, 0, s0p, PS_ST, 0}, {state, n, 402, 804, 402, s1p, PS_ST, 0},
                            {ego, n, 32, 64, 0, e0p, PS_EGO, 0}, {ego, n, 32, 64, 32, e1p, PS_EGO, 0}};
    egx_launch_pack3(st, jobs, 4);
  }
  egx_launch_posenc3(st, dist, time, n, catf + 2 * HD, 1152, cat3, PS_CAT, 2 * PS_HD);
  {   // the two 2-step GRU encoders, both cells of a step in one launch
    struct Enc { const bf16x8 *x0, *x1; int Sx; const void *w_ih, *w_hh; const float *b_ih, *b_hh; bf16x8* h1; float* h1f; int col0, s30; };
    const Enc enc[2] = {{s0p, s1p, PS_ST, P.x_enc_w_ih, P.x_enc_w_hh, w->x_enc_b_ih, w->x_enc_b_hh, hx1, hx1f, 0, 0},
                        {e0p, e1p, PS_EGO, P.ego_enc_w_ih, P.ego_enc_w_hh, w->ego_enc_b_ih, w->ego_enc_b_hh, he1, he1f, HD, PS_HD}};
    D3Gru g[2];
    for (int e = 0; e < 2; ++e) {
      const Enc& E = enc[e];
      g[e].M = n; g[e].H = HD; g[e].prec = prec;
      g[e].Ai = E.x0; g[e].SAi = E.Sx; g[e].Bi = B3(E.w_ih); g[e].Si = E.Sx; g[e].bias_i = E.b_ih; g[e].bias_h = E.b_hh;
      g[e].h_out = E.h1f; g[e].ldo = HD; g[e].h_out3 = E.h1; g[e].S3 = PS_HD;
    }
    egx_launch_gru3_pair(st, g[0], g[1]);
    for (int e = 0; e < 2; ++e) {
      const Enc& E = enc[e];
      g[e].Ai = E.x1;
      g[e].Ah = E.h1; g[e].SAh = PS_HD; g[e].Bh = B3(E.w_hh); g[e].Sh = PS_HD; g[e].h_prev = E.h1f; g[e].ldh = HD;
      g[e].h_out = catf + E.col0; g[e].ldo = 1152; g[e].h_out3 = cat3; g[e].S3 = PS_CAT; g[e].s30 = E.s30;
    }
    egx_launch_gru3_pair(st, g[0], g[1]);
  }
  const float slope = 0.01f;  // torch.nn.LeakyReLU() default (baseops.py:627-628)
  const bool do_a = out_mu != nullptr, do_c = out_value != nullptr;
  auto layer = [&](const bf16x8* xin, const void* W, const float* b, const float* res, float* outf, bf16x8* out3) {
    D3Plain l;
    l.M = n; l.N = 1152; l.A = xin; l.SA = PS_CAT; l.S = PS_CAT; l.B = B3(W); l.bias = b; l.act = 3; l.slope = slope;
    l.res = res; l.ldr = 1152; l.out = outf; l.ldo = 1152; l.out3 = out3; l.S3 = PS_CAT; l.prec = prec;
    return l;
  };
  auto run = [&](const D3Plain& la, const D3Plain& lc) {
    if (do_a && do_c) egx_launch_dense3_pair(st, la, lc);
    else egx_launch_dense3(st, do_a ? la : lc);
  };
  // h = hx; for unit: h = lrelu(fc2(lrelu(fc1(h)))) + h; y = out_fc(h)  - actor and critic layer i share a launch
  run(layer(cat3, P.actor_w[0], w->actor_b[0], nullptr, nullptr, a1), layer(cat3, P.critic_w[0], w->critic_b[0], nullptr, nullptr, c1));
  run(layer(a1, P.actor_w[1], w->actor_b[1], catf, a2f, a2), layer(c1, P.critic_w[1], w->critic_b[1], catf, c2f, c2));
  run(layer(a2, P.actor_w[2], w->actor_b[2], nullptr, nullptr, a1), layer(c2, P.critic_w[2], w->critic_b[2], nullptr, nullptr, c1));
  run(layer(a1, P.actor_w[3], w->actor_b[3], a2f, nullptr, a2), layer(c1, P.critic_w[3], w->critic_b[3], c2f, nullptr, c2));
  // heads: mu = rows 0..127 and logvar = rows 128..255 of actor.out_fc go straight to their own outputs
  D3Plain hm, hl, hv;
  hm.M = n; hm.N = 128; hm.A = a2; hm.SA = PS_CAT; hm.S = PS_CAT; hm.B = B3(P.actor_out_w); hm.bias = w->actor_out_b;
  hm.out = out_mu; hm.ldo = 128; hm.prec = prec;
  hl = hm;
  hl.B = B3(P.actor_out_w) + (size_t)8 * PS_CAT * 3 * 64;   // row tiles 8..15 of the packed [256,1152] image
  hl.bias = w->actor_out_b + 128; hl.out = out_logvar;
  hv.M = n; hv.N = 1; hv.A = c2; hv.SA = PS_CAT; hv.S = PS_CAT; hv.B = B3(P.critic_out_w); hv.bias = w->critic_out_b;
  hv.out = out_value; hv.ldo = 1; hv.prec = prec;
  if (do_a && do_c) egx_launch_dense3_triple(st, hm, hl, hv);
  else if (do_a) egx_launch_dense3_pair(st, hm, hl);
  else egx_launch_dense3(st, hv);
  return EGX_OK;
}

extern "C" int egx_policy_forward(const egx_policy_weights* w, const float* state, const float* ego, const float* dist,
                                  const float* time, int n, float* out_mu, float* out_logvar, float* out_value,
                                  void* workspace, size_t workspace_bytes, void* stream_) {
  EGX_REQUIRE(w && state && ego && dist && time, "null argument");
  EGX_REQUIRE(n > 0, "num_rows must be positive");
  EGX_REQUIRE((out_mu == nullptr) == (out_logvar == nullptr), "mu and logvar come together");
  if (!workspace || workspace_bytes < egx_policy_workspace_bytes(n)) {
    egx_set_error("egx_policy_forward: workspace too small");
    return EGX_ERR_WORKSPACE;
  }
  hipStream_t st = static_cast<hipStream_t>(stream_);
  EGX_REQUIRE(w->packed3, "egx_policy_forward needs the packed weight images (egx_policy_weights.packed3: egx_pack3 or egx_policy_train_packed)");
  const int rc = policy_forward_packed(w, state, ego, dist, time, n, out_mu, out_logvar, out_value, workspace, workspace_bytes, st);
  if (rc) return rc;
  EGX_HIP_CHECK(hipGetLastError());
  return EGX_OK;
}

// ---------------------------------------------------------------------------------------------------------
extern "C" size_t egx_vposer_workspace_bytes(int n) {
  (void)n;
  return 0;   // the fused encoder keeps its intermediates in LDS (the argument pair stays in the ABI)
}

extern "C" int egx_vposer_encode(const egx_vposer_weights* w, const float* x, int x_ld, int n, float* out,
                                 void* workspace, size_t workspace_bytes, void* stream_) {
  (void)workspace; (void)workspace_bytes;
  EGX_REQUIRE(w && x && out && n > 0 && x_ld >= 63, "bad arguments");
  EGX_REQUIRE(w->fc1_w3 && w->fc2_w3 && w->mu_w3, "egx_vposer_encode needs the packed weight images (egx_vposer_weights.fc1_w3 / fc2_w3 / mu_w3, built with egx_pack3)");
  EGX_REQUIRE(w->fc1_b && w->fc2_b && w->mu_b, "null bias");
  VpWeights3 v;
  v.fc1 = static_cast<const bf16x8*>(w->fc1_w3); v.fc2 = static_cast<const bf16x8*>(w->fc2_w3); v.mu = static_cast<const bf16x8*>(w->mu_w3);
  v.b1 = w->fc1_b; v.b2 = w->fc2_b; v.bmu = w->mu_b;
  const int rc = egx_launch_vposer3(static_cast<hipStream_t>(stream_), v, x, x_ld, n, out);
  if (rc) return rc;
  EGX_HIP_CHECK(hipGetLastError());
  return EGX_OK;
}
