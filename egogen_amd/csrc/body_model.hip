// SMPL-X forward for gfx950: pose/chain kernel + fused blend-shape GEMM (fp32 MFMA) / skinning /
// SDF-count / vertex-pick kernel + joints/markers gather kernel, behind the C ABI of include/egogen_hip.h.
//
// What the reference does (models/baseops.py:338-398 -> smplx.SMPLX.forward [upstream]):
//   v_posed = v_template + [betas | vec(R_1..54 - I)] @ [shapedirs ; posedirs]      (B x 496 x 3V GEMM)
//   A       = rigid chain over 55 joints (relative to the rest joints)
//   verts   = (sum_j W[v,j] A_j) [v_posed;1] + transl
// and then only ever consumes markers, joints and (crowd_env_2f.py:163-175) the per-frame count of
// vertices whose scene SDF is negative.  Layout decisions for MI355X:
//   * the blend GEMM runs on v_mfma_f32_32x32x2_f32 with rows = vertices, cols = bodies and three
//     accumulator sets (x,y,z) so each lane ends up owning complete (vertex, body) points;
//   * both operands are pre-packed in exactly the lane order of the MFMA operands, 4 k-steps per
//     16-byte load, so every global load is a fully coalesced 1 KiB wave access;
//   * skinning weights are ELL-packed (nnz per vertex found at load time), joint transforms are
//     written body-minor ([tile][joint][row][32 bodies] float4) so the epilogue reads are coalesced;
//   * the vertex tensor never has to exist: SDF counting and the ~240 picked vertices are epilogues.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>

#include <atomic>
#include "egx_common.h"

static thread_local std::string g_last_error;
void egx_set_error(const std::string& msg) { g_last_error = msg; }
extern "C" const char* egx_last_error(void) { return g_last_error.c_str(); }
extern "C" int egx_version(void) { return 1; }

static thread_local hipEvent_t g_prof_start = nullptr, g_prof_stop = nullptr;
extern "C" int egx_event_create(void** out_event) {
  EGX_REQUIRE(out_event, "null argument");
  hipEvent_t e;
  EGX_HIP_CHECK(hipEventCreate(&e));
  *out_event = e;
  return EGX_OK;
}
extern "C" int egx_event_destroy(void* event) {
  if (event) EGX_HIP_CHECK(hipEventDestroy(static_cast<hipEvent_t>(event)));
  return EGX_OK;
}
extern "C" int egx_event_elapsed_ms(void* start_event, void* stop_event, float* out_ms) {
  EGX_REQUIRE(start_event && stop_event && out_ms, "null argument");
  EGX_HIP_CHECK(hipEventSynchronize(static_cast<hipEvent_t>(stop_event)));
  EGX_HIP_CHECK(hipEventElapsedTime(out_ms, static_cast<hipEvent_t>(start_event), static_cast<hipEvent_t>(stop_event)));
  return EGX_OK;
}
// Streams restricted to part of the device, for running a throughput-bound launch (the fused LBS kernel) beside a chain of
// latency-bound launches of another shard of agents: bit i of `mask` (num_words x 32 bits) enables compute unit i.
extern "C" int egx_profile_next_lbs(void* start_event, void* stop_event) {
  g_prof_start = static_cast<hipEvent_t>(start_event);
  g_prof_stop = static_cast<hipEvent_t>(stop_event);
  return EGX_OK;
}

namespace {

constexpr int NJ = EGX_NUM_JOINTS;
// GEMM K axis: 10 betas + 9 rotation features of the 51 joints that can move through this API (global orient is not a
// blend feature; jaw and both eyes have no field in xb[93], their R - I is exactly 0 and their 27 columns are dropped)
constexpr int KACT = 10 + 51 * 9;      // 469 live columns
constexpr int KDIM = EGX_BLEND_K;      // 472 = 469 padded to a multiple of 8
constexpr int KSTEPS = KDIM / 2;       // 236 MFMA k-steps (32x32x2)
constexpr int KGROUPS = KSTEPS / 4;    // 59 float4 groups
static_assert(KGROUPS >= 2, "the operand ring preloads two k-groups");
// 3-term bf16 split of the blend GEMM (LBS blend mode 1): every fp32 operand x = hi + mid + lo with three bf16 terms
// (24+ significant bits, i.e. the whole fp32 mantissa); the product keeps the six partial products down to 2^-24 relative
// (hi.hi, hi.mid, mid.hi, hi.lo, lo.hi, mid.mid), accumulated in fp32 by v_mfma_f32_32x32x16_bf16 - 6 MFMAs of 32 cycles
// per 16 k instead of 8 fp32 MFMAs of 64 cycles.  K is padded to 480 = 30 steps of 16.
constexpr int KS3 = 30;
__host__ __device__ inline void egx_bf16_split3(float x, unsigned short* h) {
  h[0] = egx_bf16_rne(x);
  const float r1 = x - egx_bf16_to_f32(h[0]);   // exact
  h[1] = egx_bf16_rne(r1);
  const float r2 = r1 - egx_bf16_to_f32(h[1]);  // exact
  h[2] = egx_bf16_rne(r2);
}
// Mixed blend (LBS blend mode 3): the two k-steps that hold metre-scale or shape columns - k-step 0 (10 betas + 6 pose
// columns) and k-step 29 (pose columns, template, template residual) - keep the two-plane bf16 split (three products); the 28
// k-steps in between hold pose-corrective columns only (centimetre-scale offsets) and run as ONE v_mfma_f32_32x32x16_f16 product
// on operands rounded to fp16 (11 significant bits: 2^-12 per operand; fp16's range covers both the bases, |x| < 1, and the
// features R - I in [-2, 2]).  Against float64 that moves a vertex by ~4 um rms / ~22 um worst case on the synthetic body
// (offsets 1.3 cm rms, 7 cm max) - 2e-5 of a metre-scale coordinate, a fifth of north_star's 1e-4 - for 204 instead of 540
// MFMAs per wave and item and a third of the operand bytes.  Operand images are sequences of 1 KiB PIECES:
//   bases    [vt][96 pieces][64 lanes] 8 x 16 bit: k-step 0 (plane, coord) 0..5 | k-steps 1..28 coord 6..89 | k-step 29 (plane, coord) 90..95
//   features [bt][32 pieces][64 lanes]            : k-step 0 planes 0, 1         | k-steps 1..28 2..29        | k-step 29 planes 30, 31
constexpr int M4_BASE_PIECES = 96, M4_FEAT_PIECES = 32;
constexpr int M4_FKS = 4;                       // fp16 k-steps per stage (7 stages) between the two precise stages
__host__ __device__ inline int egx_m4_feat_piece(int s, int pl) { return s == 0 ? pl : (s <= 28 ? s + 1 : 30 + pl); }
__host__ __device__ inline int egx_m4_base_piece(int s, int pl, int c) { return s == 0 ? pl * 3 + c : (s <= 28 ? 6 + (s - 1) * 3 + c : 90 + pl * 3 + c); }
__host__ __device__ inline unsigned short egx_f16_rne(float x) {
  const _Float16 h = (_Float16)x;
  unsigned short u;
  __builtin_memcpy(&u, &h, 2);
  return u;
}
__host__ __device__ inline int egx_compact_joint(int j) { return (j - 1) - (j > 24 ? 3 : 0); }  // j in 1..54, j != 22..24
constexpr int NLMK = 51, NEXTRA = 21;
constexpr int BODY_PAD = 256;          // bodies per workgroup of the fused kernel

struct PoseConsts {
  int parents[NJ];
  int depth[NJ];
  int max_depth;
  float J_template[NJ * 3];
  float J_shapedirs[NJ * 3 * 10];
  float hand_comps[2 * 12 * 45];
  float hand_mean[2 * 45];
  float fix_c[NJ];   // per joint: largest |pose-corrective base column| (3-vector norm) over its 9 columns and all vertices
  float v_norm_max;  // largest |v_template| + 0.25 m for the blend offsets: bound of a vertex before skinning
};

// Fix-up of the mixed blend (mode 3).  The count-only tiles evaluate the pose-corrective columns as ONE fp16 product: a vertex moves by
// up to ~2e-5 m against the fp32 chain, and one that lies that close to the scene surface may be counted differently from the
// reference's fp32 evaluation (crowd_env_2f.py:169-177).  So the cheap evaluation only CLASSIFIES: a vertex whose interpolated SDF
// value is further from zero than the value change its position error can cause keeps the cheap decision; the few inside that band
// are re-evaluated by their wave in fp32 (three-plane operand images, fp32 skinning: lbs_fix_process) and counted from that.
// Position error of a body: the rounding errors of the 448 x 3 products are independent, std ~0.2 x
//   bound(body) = 2^-11 sqrt(sum_j ||R_j - I||_F^2 C_j^2),  C_j = PoseConsts::fix_c
// (scripts/emulate_lbs_fixup.py: error / bound rms 0.20, max 0.66 over 5e5 vertex samples, ordinary and wild poses); the band is
// LBS_FIX_KAPPA x bound = ten standard deviations, plus a constant for the fp32 round-off of the rest of the chain.
constexpr float LBS_FIX_KAPPA = 2.0f;
constexpr float LBS_FIX_SLACK_M = 3e-6f;
// Matrix-pipe skinning of the count-only tiles (lbs_skin_cell): weights and transforms as two bf16 planes, products hi.hi + hi.mid +
// mid.hi.  Each operand is off by <= 2^-18 relative and the dropped mid.mid term is <= 2^-18, so a coordinate moves by at most
// 3 x 2^-18 (|v| + |t_j|) and the position by sqrt(3) times that = 2.0e-5 (|v| + max_j |t_j|) in the worst case of every rounding
// pointing the same way.  The roundings are independent: over 3.4e5 vertex samples of ordinary and wild poses the worst error is
// 0.27 of that bound (scripts/emulate_lbs_fixup.py skin); the band allows half of it, twice the worst case seen.
constexpr float LBS_SKIN_ERR = 1.0e-5f;
// The fix-up queue is LBS_FIX_NQ sub-queues, a workgroup appends to sub-queue blockIdx % NQ: one counter for the whole launch made
// every append (and, in the first version, every processed vertex) an atomic on ONE address - ~12 ns each at the L2, 190 us for
// 16 000 vertices.  Counters sit 128 bytes apart: fix_stats[LBS_FIX_CNT0 + 32 q]; fix_stats[0] counts the vertices re-evaluated
// inside the fused kernel (a full sub-queue).
constexpr int LBS_FIX_NQ = 64;
constexpr int LBS_FIX_CNT0 = 32;
constexpr int LBS_FIX_STATS_INTS = LBS_FIX_CNT0 + 32 * LBS_FIX_NQ;

}  // namespace

// Skinning of the count-only tiles of the mixed blend on the matrix pipe (see lbs_epilogue_cell): T = W x A', W = the tile's skinning
// weights [32 vertices x the joints of the tile's list], A' = the bodies' joint transforms premultiplied by the agent's
// canonical-frame -> SDF-cell map, both as two bf16 planes.  One v_mfma_f32_32x32x16_bf16 k-step covers EIGHT joints of the list
// with both planes of A' folded into K: lane half 0 holds (W_hi | A'_hi), lane half 1 (W_hi | A'_mid), so one MFMA yields
// W_hi A'_hi + W_hi A'_mid; a second one with (W_mid | 0) on the same A' registers adds W_mid A'_hi.
//   skinB  [bt][joint][plane][n] 8 x bf16 = entries (a, c) of rows a = 0, 1, then [bt][joint][plane][n] 4 x bf16 = row a = 2
//   skinW  [ks_off[vt] + ks][operand 0 | 1][64 lanes] 8 x bf16, lane (h, row): operand 0 = W_hi[row][list[8 ks + e]] in both halves,
//          operand 1 = W_mid in half 0, zero in half 1
constexpr int SKIN_BT_A = NJ * 2 * 32;            // 16-byte records of rows a = 0, 1 per 32-body tile
constexpr int SKIN_BT_BYTES = NJ * 2 * 32 * 24;   // both parts

struct egx_body_model {
  int V = 0, NVT = 0, NW = 0, M = 0, NP = 0;
  float* dirs_rm = nullptr;    // [NVT*32 rows][3 coords][KDIM] fp32, vertex-major: what the fp32 re-evaluation of single vertices reads (lbs_fix_one)
  bf16x8* skinW = nullptr;     // matrix-pipe skinning weights (see SKIN_BT_BYTES)
  int* skin_ks_off = nullptr;  // [NVT+1] k-steps (8 joints of the tile's list each) before tile vt
  f32x4* dirs = nullptr;       // [NVT][59][3][64] float4 (fp32 blend)
  bf16x8* dirs3 = nullptr;     // [NVT][30 k-steps][3 planes][3 coords][64 lanes] 8 x bf16 (bf16x3 blend)
  bf16x8* dirs4 = nullptr;     // [NVT][96 pieces][64 lanes] 8 x 16 bit (mixed blend, mode 3: see M4_BASE_PIECES)
  int* tj_off = nullptr;       // [NVT+1] offsets into the per-tile joint lists
  int* tj_idx = nullptr;       // [tj_off[NVT]] joints with a non-zero skinning weight on some vertex of the tile
  float* tj_w = nullptr;       // [tj_off[NVT]][32] dense weights of the tile's 32 vertices for that joint
  int* pick_slot = nullptr;    // [NVT*32], -1 = not picked
  int* pick_tiles = nullptr;   // [n_pick_tiles] vertex tiles that hold a picked vertex (all a markers-and-joints-only call needs)
  int n_pick_tiles = 0;
  int* sdf_tiles = nullptr;    // [n_sdf_tiles] tiles that hold a picked vertex or a vertex of the penetration count (non-feet)
  int n_sdf_tiles = 0;
  int verts_pick_tiles = 0, verts_sdf_tiles = 0;   // real vertices inside the two tile lists (work accounting)
  uint8_t* vflags = nullptr;   // [NVT*32] bit0 feet, bit1 valid
  int* vorig = nullptr;        // [NVT*32] original vertex id of every (sorted) row, -1 = padding
  PoseConsts* pc = nullptr;
  int* marker_slot = nullptr;  // [M]
  int* extra_slot = nullptr;   // [21]
  int* lmk_slot = nullptr;     // [153]
  float* lmk_bary = nullptr;   // [153]
  // free-space culling of SDF work items (see egx_lbs_cull_kernel): per-tile bounds of how far a posed vertex can be from the
  // posed joints it is bound to
  float* cull_E = nullptr;     // [NVT][64]: [0,10) shape terms, [10,61) pose terms per movable joint (compact order), rest 0
  float* cull_D0 = nullptr;    // [tj_off[NVT]]: rest distance bound per (tile, joint of its list)
  int cull_ok = 0;             // skinning weights are a convex combination (>= 0, rows sum to 1): the bound holds
  int sdf_lead_picks = 0;      // sdf_tiles starts with pick_tiles (in the same order)
  float rest_pelvis[3] = {0.f, 0.f, 0.f};   // root joint of the mean shape (host copy)
  float cull_ref_margin = 0.f;              // blend-shape margin of the median tile at the reference pose (metres)
};

// Rotation matrix of joint j of one body from its parameter row x[93] (transl 3 | global orient 3 | body pose 63 | hand PCA 12 + 12;
// jaw and eyes - joints 22..24 - have no field: identity): axis-angle -> smplx batch_rodrigues (angle = ||a + 1e-8||).
__device__ __forceinline__ void lbs_joint_rotation(const PoseConsts* __restrict__ pc, const float* __restrict__ x, int j, float (&R)[9]) {
  float a[3] = {0.f, 0.f, 0.f};
  if (j == 0) {
    a[0] = x[3]; a[1] = x[4]; a[2] = x[5];
  } else if (j <= 21) {
    a[0] = x[6 + 3 * (j - 1)]; a[1] = x[7 + 3 * (j - 1)]; a[2] = x[8 + 3 * (j - 1)];
  } else if (j >= 25) {
    const int side = (j >= 40) ? 1 : 0;
    const int o = 3 * (j - (side ? 40 : 25));
    const float* comps = pc->hand_comps + side * 12 * 45;
    const float* pca = x + 69 + side * 12;
    for (int c = 0; c < 3; ++c) {
      float s = 0.f;
      for (int k = 0; k < 12; ++k) s += pca[k] * comps[k * 45 + o + c];
      a[c] = s + pc->hand_mean[side * 45 + o + c];
    }
  }
  const float ex = a[0] + 1e-8f, ey = a[1] + 1e-8f, ez = a[2] + 1e-8f;
  const float angle = sqrtf(ex * ex + ey * ey + ez * ez);
  const float rx = a[0] / angle, ry = a[1] / angle, rz = a[2] / angle;
  const float sn = sinf(angle), cs = 1.f - cosf(angle);
  R[0] = 1.f + cs * (-(ry * ry + rz * rz)); R[1] = -sn * rz + cs * (rx * ry);     R[2] = sn * ry + cs * (rx * rz);
  R[3] = sn * rz + cs * (rx * ry);          R[4] = 1.f + cs * (-(rx * rx + rz * rz)); R[5] = -sn * rx + cs * (ry * rz);
  R[6] = -sn * ry + cs * (rx * rz);         R[7] = sn * rx + cs * (ry * rz);      R[8] = 1.f + cs * (-(rx * rx + ry * ry));
}

// ------------------------------------------------------------------------------------------------
// kernel 1: per-body pose features, rigid chain, joint transforms
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void egx_pose_chain_kernel(const PoseConsts* __restrict__ pc,
                                                             const float* __restrict__ xb,
                                                             const float* __restrict__ betas, int B, int fpa,
                                                             float* __restrict__ feat,   // packed B operand (fp32 blend) or null
                                                             unsigned short* __restrict__ feat3,  // bf16x3 planes or null
                                                             f32x4* __restrict__ A4,     // [bt][55][3][32]
                                                             float* __restrict__ out_joints, int joints_ld,
                                                             float template_lo_feat /* 1 in the two-plane blend mode */,
                                                             unsigned short* __restrict__ feat4 /* mixed-blend image (mode 3) or null */,
                                                             int* __restrict__ zero_counts /* [B] cleared here, or null */,
                                                             const int* __restrict__ agent_of_slot /* culled launches: slot order */,
                                                             float* __restrict__ fvec /* [64][Bp] |beta|, ||R_j - I||_F or null */,
                                                             float* __restrict__ jpos /* [55][3][Bp] posed joints + transl or null */,
                                                             int Bp,
                                                             float* __restrict__ fix_e /* [Bp] per slot: position error bound of the mixed blend (metres) or null */,
                                                             int* __restrict__ fix_stats /* [1] cleared here, or null */,
                                                             unsigned short* __restrict__ skinB /* matrix-pipe skinning operands (see SKIN_BT_BYTES) or null */,
                                                             f32x4* __restrict__ cinit /* [Bp] per slot: cell coordinates of the body's translation */,
                                                             const float* __restrict__ R0, const float* __restrict__ T0, SdfDev sdf) {
  __shared__ float sR[4][NJ][9];
  __shared__ float sJ[4][NJ][3];
  __shared__ float sG[4][NJ][12];
  __shared__ __attribute__((aligned(16))) unsigned short sF3[4][3][KS3 * 16];  // bf16x3 planes of the 4 bodies of the block
  __shared__ __attribute__((aligned(16))) unsigned short sF4[4][KS3 * 16];     // fp16 values (mixed blend, k-steps 1..28)
  const int w = threadIdx.x >> 6, j = threadIdx.x & 63;
  // a block works on four SLOTS of the operand buffers; slot s holds body agent_of_slot[s / fpa] * fpa + s % fpa (identity
  // without the table): inputs and per-body outputs are addressed by body, the GEMM operands by slot
  const int slot = blockIdx.x * 4 + w;
  const bool live = slot < B;
  const int ss = live ? slot : B - 1;
  const int b = agent_of_slot ? agent_of_slot[ss / fpa] * fpa + ss % fpa : ss;
  if (zero_counts && live && j == 0) zero_counts[b] = 0;   // the SDF epilogue of the skinning kernel adds to these
  if (fix_stats && blockIdx.x == 0 && threadIdx.x <= LBS_FIX_NQ) fix_stats[threadIdx.x == 0 ? 0 : LBS_FIX_CNT0 + 32 * (threadIdx.x - 1)] = 0;
  const int bb = b;
  const float* x = xb + (size_t)bb * EGX_XB_DIM;
  const float* be = betas + (size_t)(bb / fpa) * 10;
  const int bt = ss >> 5, n = ss & 31;
  float* featb = feat ? feat + (size_t)bt * KGROUPS * 64 * 4 : nullptr;  // tile base
  unsigned short* feat3b = feat3 ? feat3 + (size_t)bt * KS3 * 3 * 64 * 8 : nullptr;
  auto feat_store = [&](int k, float v) {
    if (featb && k < KDIM) {
      const int s = k >> 1, kk = k & 1;
      featb[((s >> 2) * 64 + (kk * 32 + n)) * 4 + (s & 3)] = v;
    }
    if (feat3b) {  // staged in LDS; written out below as whole 16-byte operand fragments
      unsigned short h[3];
      egx_bf16_split3(v, h);
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) sF3[w][pl][k] = h[pl];
      if (feat4) sF4[w][k] = egx_f16_rne(v);   // the mixed blend reads k-steps 1..28 as one fp16 plane
    }
  };
  float R[9], Jr[3];
  if (j < NJ) {
    {
    // (lbs_joint_rotation, spelled out: through the shared function the compiler contracts these products differently - last-bit
    // changes of R that the egosensing rays, aimed by eye landmarks centimetres apart, amplify past the parity floor)
    float a[3] = {0.f, 0.f, 0.f};
    if (j == 0) {
      a[0] = x[3]; a[1] = x[4]; a[2] = x[5];
    } else if (j <= 21) {
      a[0] = x[6 + 3 * (j - 1)]; a[1] = x[7 + 3 * (j - 1)]; a[2] = x[8 + 3 * (j - 1)];
    } else if (j >= 25) {
      const int side = (j >= 40) ? 1 : 0;
      const int o = 3 * (j - (side ? 40 : 25));
      const float* comps = pc->hand_comps + side * 12 * 45;
      const float* pca = x + 69 + side * 12;
      for (int c = 0; c < 3; ++c) {
        float s = 0.f;
        for (int k = 0; k < 12; ++k) s += pca[k] * comps[k * 45 + o + c];
        a[c] = s + pc->hand_mean[side * 45 + o + c];
      }
    }
    const float ex = a[0] + 1e-8f, ey = a[1] + 1e-8f, ez = a[2] + 1e-8f;
    const float angle = sqrtf(ex * ex + ey * ey + ez * ez);
    const float rx = a[0] / angle, ry = a[1] / angle, rz = a[2] / angle;
    const float sn = sinf(angle), cs = 1.f - cosf(angle);
    R[0] = 1.f + cs * (-(ry * ry + rz * rz)); R[1] = -sn * rz + cs * (rx * ry);     R[2] = sn * ry + cs * (rx * rz);
    R[3] = sn * rz + cs * (rx * ry);          R[4] = 1.f + cs * (-(rx * rx + rz * rz)); R[5] = -sn * rx + cs * (ry * rz);
    R[6] = -sn * ry + cs * (rx * rz);         R[7] = sn * rx + cs * (ry * rz);      R[8] = 1.f + cs * (-(rx * rx + ry * ry));
    }
    for (int c = 0; c < 3; ++c) {
      float s = pc->J_template[j * 3 + c];
      for (int k = 0; k < 10; ++k) s += be[k] * pc->J_shapedirs[(j * 3 + c) * 10 + k];
      Jr[c] = s;
      sJ[w][j][c] = s;
    }
    for (int e = 0; e < 9; ++e) sR[w][j][e] = R[e];
  }
  if (fix_e) {   // wave-uniform: every lane of the body's wave takes part in the reduction
    float q = 0.f;
    if (j >= 1 && j < NJ) {
      for (int e = 0; e < 9; ++e) {
        const float dlt = R[e] - ((e == 0 || e == 4 || e == 8) ? 1.f : 0.f);
        q += dlt * dlt;
      }
      q *= pc->fix_c[j] * pc->fix_c[j];
    }
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
    if (live && j == 0) fix_e[slot] = LBS_FIX_KAPPA * 0.00048828125f * sqrtf(q) + LBS_FIX_SLACK_M;
  }
  if (j < NJ) {
    if (live && fvec) {
      if (j < 10) fvec[(size_t)j * Bp + slot] = fabsf(be[j]);
      if (j >= 1 && (j < 22 || j > 24)) {
        float q = 0.f;
        for (int e = 0; e < 9; ++e) {
          const float dlt = R[e] - ((e == 0 || e == 4 || e == 8) ? 1.f : 0.f);
          q += dlt * dlt;
        }
        fvec[(size_t)(10 + egx_compact_joint(j)) * Bp + slot] = sqrtf(q) * 1.000001f;
      }
    }
    if (live) {
      if (j < 10) feat_store(j, be[j]);
      if (j >= 1 && (j < 22 || j > 24)) {
        const int k0 = 10 + egx_compact_joint(j) * 9;
        for (int e = 0; e < 9; ++e) feat_store(k0 + e, R[e] - ((e == 0 || e == 4 || e == 8) ? 1.f : 0.f));
      }
      // column 469 multiplies the template column of the bases (acc = v_template + offsets); column 470 multiplies the
      // third bf16 term of the template (bf16x3 bases only): switched on in the two-plane blend mode, where the product keeps
      // 16 bits per operand - enough for the centimetre-scale offsets, not for the metre-scale template; 471 is padding
      if (j >= 22 && j <= 24) feat_store(KACT + (j - 22), j == 22 ? 1.f : (j == 23 ? template_lo_feat : 0.f));
      if (feat3b && j >= 22 && j <= 24) {  // bf16x3 pads K to 480: columns 472..479
        for (int k = KDIM + (j - 22); k < KS3 * 16; k += 3) feat_store(k, 0.f);
      }
    }
  }
  __syncthreads();
  if (feat3) {
    // k = 16 s + 8 half + e  ->  [bt][s][plane][half*32 + n][e]: one 16-byte fragment per (s, plane, half, body); the four
    // bodies of the block are neighbours in n, so a quarter-wave writes 64 contiguous bytes
    for (int c = threadIdx.x; c < KS3 * 3 * 2 * 4; c += 256) {
      const int wb = c & 3, hf = (c >> 2) & 1, pl = (c >> 3) % 3, sidx = c / 24;
      const int body = blockIdx.x * 4 + wb;   // slot
      if (body < B) {
        const int4 frag = *reinterpret_cast<const int4*>(&sF3[wb][pl][sidx * 16 + hf * 8]);
        unsigned short* dst = feat3 + ((((size_t)(body >> 5) * KS3 + sidx) * 3 + pl) * 64 + hf * 32 + (body & 31)) * 8;
        *reinterpret_cast<int4*>(dst) = frag;
      }
    }
  }
  if (feat3 && feat4) {
    // the mixed image: 32 pieces per body tile (two bf16 planes of k-steps 0 and 29, one fp16 plane of k-steps 1..28)
    for (int c = threadIdx.x; c < M4_FEAT_PIECES * 2 * 4; c += 256) {
      const int wb = c & 3, hf = (c >> 2) & 1, piece = c >> 3;
      const int sidx = piece < 2 ? 0 : (piece < 30 ? piece - 1 : 29), pl = piece < 2 ? piece : (piece < 30 ? 0 : piece - 30);
      const int body = blockIdx.x * 4 + wb;   // slot
      if (body < B) {
        const unsigned short* src = (piece >= 2 && piece < 30) ? &sF4[wb][sidx * 16 + hf * 8] : &sF3[wb][pl][sidx * 16 + hf * 8];
        const int4 frag = *reinterpret_cast<const int4*>(src);
        unsigned short* dst = feat4 + (((size_t)(body >> 5) * M4_FEAT_PIECES + piece) * 64 + hf * 32 + (body & 31)) * 8;
        *reinterpret_cast<int4*>(dst) = frag;
      }
    }
  }

  const int par = (j < NJ) ? pc->parents[j] : -1;
  const int dep = (j < NJ) ? pc->depth[j] : -1;
  float rel[3] = {0.f, 0.f, 0.f};
  if (j < NJ) {
    for (int c = 0; c < 3; ++c) rel[c] = Jr[c] - (par >= 0 ? sJ[w][par][c] : 0.f);
  }
  float G[12];
  const int max_depth = pc->max_depth;
  for (int d = 0; d <= max_depth; ++d) {
    if (dep == d) {
      if (par < 0) {
        for (int r = 0; r < 3; ++r) {
          G[r * 4 + 0] = R[r * 3 + 0]; G[r * 4 + 1] = R[r * 3 + 1]; G[r * 4 + 2] = R[r * 3 + 2]; G[r * 4 + 3] = rel[r];
        }
      } else {
        const float* P = sG[w][par];
        for (int r = 0; r < 3; ++r) {
          for (int c = 0; c < 3; ++c)
            G[r * 4 + c] = P[r * 4 + 0] * R[0 * 3 + c] + P[r * 4 + 1] * R[1 * 3 + c] + P[r * 4 + 2] * R[2 * 3 + c];
          G[r * 4 + 3] = P[r * 4 + 0] * rel[0] + P[r * 4 + 1] * rel[1] + P[r * 4 + 2] * rel[2] + P[r * 4 + 3];
        }
      }
      for (int e = 0; e < 12; ++e) sG[w][j][e] = G[e];
    }
    // sG[w] is private to this wave (one body per wave) and a wave's LDS operations complete in order: the next level's reads
    // only have to stay behind these writes in program order - no workgroup barrier per level of the tree
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  if (skinB) {   // block-uniform
    // A'_j = Mc A_j: the joint transform followed by the agent's canonical frame -> SDF-cell map (the folded affine map of the
    // fused kernel's epilogue scaled by 1/4, egx_sdf_coarse_at_cell); the constant part Mc transl + tc stays in fp32 (cinit)
    const int ag = bb / fpa;
    float Mc[9], tcv[3];
    {
      const float kk[3] = {sdf.scale * (float)sdf.d0 * 0.5f, sdf.scale * (float)sdf.d1 * 0.5f, sdf.scale * (float)sdf.d2 * 0.5f};
      const float cc[3] = {sdf.cx, sdf.cy, sdf.cz};
      const float dd[3] = {(float)sdf.d0, (float)sdf.d1, (float)sdf.d2};
      for (int a = 0; a < 3; ++a) {
        for (int e = 0; e < 3; ++e) Mc[a * 3 + e] = 0.25f * (kk[a] * (R0 ? R0[(size_t)ag * 9 + a * 3 + e] : ((a == e) ? 1.f : 0.f)));
        const float tw = kk[a] * ((T0 ? T0[(size_t)ag * 3 + a] : 0.f) - cc[a]) + (dd[a] - 1.f) * 0.5f;
        tcv[a] = fmaf(0.25f, tw, 1.f);
      }
    }
    float tn = 0.f;
    if (j < NJ) {
      float Arow[3][4];
      for (int r = 0; r < 3; ++r) {
        Arow[r][0] = G[r * 4 + 0]; Arow[r][1] = G[r * 4 + 1]; Arow[r][2] = G[r * 4 + 2];
        Arow[r][3] = G[r * 4 + 3] - (G[r * 4 + 0] * Jr[0] + G[r * 4 + 1] * Jr[1] + G[r * 4 + 2] * Jr[2]);
      }
      tn = sqrtf(Arow[0][3] * Arow[0][3] + Arow[1][3] * Arow[1][3] + Arow[2][3] * Arow[2][3]);
      // the joint's two records (hi and mid plane: 12 entries each), packed in registers and stored straight to the image:
      // 16 + 8 bytes per plane at [joint][plane][n] (the four bodies of the block are neighbours in n: 64-byte runs)
      unsigned pk[2][6];
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        unsigned short hh[2][3];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int a = (2 * i + u) >> 2, c = (2 * i + u) & 3;
          egx_bf16_split3(Mc[a * 3 + 0] * Arow[0][c] + Mc[a * 3 + 1] * Arow[1][c] + Mc[a * 3 + 2] * Arow[2][c], hh[u]);
        }
        pk[0][i] = (unsigned)hh[0][0] | ((unsigned)hh[1][0] << 16);
        pk[1][i] = (unsigned)hh[0][1] | ((unsigned)hh[1][1] << 16);
      }
      if (live) {
        unsigned short* base = skinB + (size_t)bt * (SKIN_BT_BYTES / 2);
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
          const size_t rec = ((size_t)j * 2 + pl) * 32 + n;
          *reinterpret_cast<uint4*>(base + rec * 8) = make_uint4(pk[pl][0], pk[pl][1], pk[pl][2], pk[pl][3]);
          *reinterpret_cast<uint2*>(base + (size_t)SKIN_BT_A * 8 + rec * 4) = make_uint2(pk[pl][4], pk[pl][5]);
        }
      }
    }
    for (int o = 32; o > 0; o >>= 1) tn = fmaxf(tn, __shfl_xor(tn, o));
    if (live && j == 0) {
      f32x4 ci;
      for (int a = 0; a < 3; ++a) ci[a] = fmaf(Mc[a * 3 + 0], x[0], fmaf(Mc[a * 3 + 1], x[1], fmaf(Mc[a * 3 + 2], x[2], tcv[a])));
      ci[3] = 0.f;
      cinit[slot] = ci;
      // the two-plane products of the skinning add to the body's position error bound (see LBS_SKIN_ERR)
      if (fix_e) fix_e[slot] += LBS_SKIN_ERR * (pc->v_norm_max + tn);
    }
  }
  if (j < NJ && live) {
    // relative transform: translation column minus R_g * rest joint (smplx batch_rigid_transform)
    for (int r = 0; r < 3; ++r) {
      const float t = G[r * 4 + 3] - (G[r * 4 + 0] * Jr[0] + G[r * 4 + 1] * Jr[1] + G[r * 4 + 2] * Jr[2]);
      f32x4 row = {G[r * 4 + 0], G[r * 4 + 1], G[r * 4 + 2], t};
      A4[(((size_t)bt * NJ + j) * 3 + r) * 32 + n] = row;
    }
    if (out_joints) {
      float* o = out_joints + ((size_t)b * joints_ld + j) * 3;
      o[0] = G[3] + x[0]; o[1] = G[7] + x[1]; o[2] = G[11] + x[2];
    }
    if (jpos) {
      jpos[(size_t)(j * 3 + 0) * Bp + slot] = G[3] + x[0];
      jpos[(size_t)(j * 3 + 1) * Bp + slot] = G[7] + x[1];
      jpos[(size_t)(j * 3 + 2) * Bp + slot] = G[11] + x[2];
    }
  }
}

// ------------------------------------------------------------------------------------------------
// kernel 2: fused blend GEMM + skinning + (SDF count) + (vertex picks) + (vertex write)
// ------------------------------------------------------------------------------------------------
struct LbsParams {
  const f32x4* dirs;
  const int* tj_off;
  const int* tj_idx;
  const float* tj_w;
  const int* pick_slot;
  const int* tiles;    // vertex tiles to compute (null = all NVT); n_tiles of them
  int n_tiles;
  const uint8_t* vflags;
  const int* vorig;    // original vertex id per sorted row
  const bf16x8* dirs3; // bf16x3 bases (blend mode 1)
  const bf16x8* feat3; // [bt][30][3 planes][64] 8 x bf16
  const bf16x8* dirs4; // mixed-blend bases (mode 3), [vt][96 pieces][64]
  const bf16x8* feat4; // mixed-blend features, [bt][32 pieces][64]
  int n_precise;       // mode 3: the first n_precise entries of `tiles` (the tiles that hold picked vertices) use the two-plane split
  const f32x4* feat;   // [bt][59][64] float4
  const f32x4* A4;     // [bt][55][3][32] float4
  const float* xb;     // transl = xb[b*93 + 0..2]
  int B, V, NVT, NW, NP, fpa;
  int nbg;             // body groups (256 bodies each)
  int bg_block;        // body groups per L2 block of the item order (bf16x3 kernel)
  int dbg;             // development ablations (EGX_LBS_DBG): 1 = skip the epilogue, 2 = skip the MFMA loop
  float* verts;        // [B][V][3] or null
  float* picked;       // [B][NP][3] or null
  SdfDev sdf;
  const float* R0;     // [A][9] or null
  const float* T0;     // [A][3] or null
  int* pene;           // [B]
  // culled launches (egx_lbs_cull_kernel): operand slot -> body order, and per-XCD lists of the active work items
  const int* agent_of_slot;   // [B / fpa] or null (identity)
  const int* items;           // [8][items_stride] codes tile_index * nbg + body_group, or null (walk every item)
  const int* item_counts;     // [8]
  int items_stride;
  // fix-up of the mixed blend (see LBS_FIX_KAPPA)
  const float* fix_e;         // [Bp] per slot: position error bound (metres)
  const float* sdf_aux;       // aux floats of the SDF's bracket table (egx_sdf_aux_offset): [0..2] largest sample step per axis
  int* fix_stats;             // [0] vertices re-evaluated inside the fused kernel, [LBS_FIX_CNT0 + 32 q] fill of sub-queue q (cleared by the pose kernel)
  const float* dirs_rm;       // vertex-major fp32 bases (fix-up)
  const PoseConsts* pc;       // pose constants (fix-up: the body's rotation features are recomputed from xb)
  const float* betas;         // [A][10]
  int2* fixq;                 // fix-up queue: LBS_FIX_NQ sub-queues of fixq_cap entries (vertex tile * 32 + row, operand slot)
  int fixq_cap;
  // matrix-pipe skinning of the count-only tiles (lbs_epilogue_cell)
  const bf16x8* skinW;        // [k-step][2][64] (see SKIN_BT_BYTES)
  const int* skin_ks_off;     // [NVT+1]
  const bf16x8* skinB;        // [bt] SKIN_BT_BYTES each
  const f32x4* cinit;         // [Bp]
};

// one v_fma_f32, opaque to the SLP vectoriser (which would pair adjacent rows into v_pk_fma_f32 again)
__device__ __forceinline__ float lbs_fma(float a, float b, float c) {
  float d;
  asm("v_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}

// fp32-MFMA variant (blend mode 0 and every vertex-writing call): persistent workgroups, one per CU, eight waves = two
// SETS of four waves; each set walks its own stream of work items (vertex tile x 256 bodies: 4 waves x 64 bodies), so
// one wave of a SIMD can wait for its operand burst while the other issues MFMAs.  Nothing synchronises across waves
// (per-wave LDS metadata, no barriers).  The fp32 MFMA shares the fp32 VALU lanes (scripts/ubench/mfma_valu.hip), so here
// the epilogue's VALU work adds to the MFMA time whatever the relative phase of the two sets (a phase offset between
// them was tried and changes nothing).
constexpr int LBS_META_BYTES = 7680;                 // s_W[55*32] f32, s_jl[56], s_slot[32], masks[4], s_cnt[64] (16-byte multiple)
constexpr int LBS_VERT_BYTES = 32 * 97 * 4;          // per-wave transpose buffer of the vertex-writing variants
constexpr int LBS_QCAP = 640;                        // entries of the per-wave queue of undecided SDF points (>= 512 + 64)
constexpr int LBS_THREADS = 512;

// Per-wave state of the fused kernels: lane coordinates and the wave's private LDS regions (metadata of the current
// vertex tile, penetration counters, queue of undecided SDF points / vertex transpose buffer).
struct LbsWave {
  int lane, n, half;
  float* s_W;          // [jj][row] dense skinning weights of the tile's joint list
  int* s_jl;           // [jj] joint ids
  int* s_slot;         // [row] pick slot or -1
  unsigned* s_masks;   // [0] rows with a pick slot, [1] rows in the SDF count
  int* s_cnt;          // [q*32 + n] penetration count of this item's 64 bodies
  unsigned* s_fixmap;  // [q*32 + n] bit r = vertex row r of that body awaits the fp32 re-evaluation (mixed blend only)
  float* s_thr;        // [q*32 + n] |SDF value| below which the cheap evaluation does not decide (mixed blend only)
  float* lds;          // vertex transpose buffer (vertex-writing variants)
  f32x4* s_queue;      // undecided SDF points (voxel x, y, z, counter slot)
  int qn;              // queued points (wave-uniform)
#ifdef EGX_LBS_TIMING
  unsigned long long et[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // per-wave cycle totals, written out once when the kernel ends
#endif
};
constexpr int LBS_NB = 2;  // 32-body MFMA column tiles per wave
#ifndef EGX_LBS_PF
#define EGX_LBS_PF 1       // joint transforms fetched this many joints ahead in the skinning loop
#endif

template <bool WRITE_VERTS, bool DO_SDF>
__device__ __forceinline__ LbsWave lbs_wave_init(char* my, int lane) {
  LbsWave w;
  w.lane = lane; w.n = lane & 31; w.half = lane >> 5;
  w.s_W = reinterpret_cast<float*>(my);
  w.s_jl = reinterpret_cast<int*>(my + NJ * 32 * 4);
  w.s_slot = w.s_jl + 56;
  w.s_masks = reinterpret_cast<unsigned*>(w.s_slot + 32);
  w.s_cnt = reinterpret_cast<int*>(w.s_masks + 4);
  w.lds = reinterpret_cast<float*>(my + LBS_META_BYTES);
  w.s_queue = reinterpret_cast<f32x4*>(my + LBS_META_BYTES);
  w.s_fixmap = nullptr; w.s_thr = nullptr;
  w.qn = 0;
  w.s_cnt[lane] = 0;
  return w;
}

// per-tile metadata, private to the wave (DS operations of one wave execute in order: no barrier)
__device__ __forceinline__ int lbs_load_meta(const LbsParams& p, LbsWave& w, int vt) {
  const int lane = w.lane;
  float* s_W = w.s_W; int* s_jl = w.s_jl; int* s_slot = w.s_slot; unsigned* s_masks = w.s_masks;
  const int j_lo = p.tj_off[vt];
  const int JT = p.tj_off[vt + 1] - j_lo;
  for (int idx = lane * 4; idx < JT * 32; idx += 256)
    *reinterpret_cast<f32x4*>(&s_W[idx]) = *reinterpret_cast<const f32x4*>(&p.tj_w[(size_t)j_lo * 32 + idx]);
  if (lane < JT) s_jl[lane] = p.tj_idx[j_lo + lane];
  {
    const int sl = (lane < 32) ? p.pick_slot[vt * 32 + lane] : -1;
    const int fl = (lane < 32) ? p.vflags[vt * 32 + lane] : 0;
    if (lane < 32) s_slot[lane] = sl;
    const unsigned long long mp = __ballot(sl >= 0), ms = __ballot((fl & 3) == 2);
    if (lane == 0) { s_masks[0] = (unsigned)mp; s_masks[1] = (unsigned)ms; }
  }
  __builtin_amdgcn_wave_barrier();

  return JT;
}

#ifdef EGX_LBS_TIMING
// development build only (make CXXFLAGS+=-DEGX_LBS_TIMING): cycle totals of the phases of the bf16x3 stage loop
__device__ unsigned long long g_lbs_t[16];
#define LBS_T(i, v) do { tacc[i] += (unsigned long long)(v); } while (0)
#define LBS_NOW() __builtin_readcyclecounter()
#else
#define LBS_T(i, v) do { } while (0)
#define LBS_NOW() 0ull
#endif

// Epilogue of one work item: each lane owns 16 vertices (rows) x 2 bodies (column n of tiles bt0, bt0+1).
__device__ __forceinline__ float lbs_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// v_posed of one vertex from the MFMA-ordered three-plane operand images (hi + mid + lo = the fp32 value exactly): 60 lanes take one
// 8-column fragment each.  720 scattered cache lines per vertex - only the overflow path of lbs_fix_process (a full fix-up queue)
// uses it, because it is compact code inside the fused kernel; egx_lbs_fix_kernel reads the vertex-major copy instead.
__device__ __forceinline__ void lbs_fix_blend_planes(const LbsParams& p, int lane, int vt, int row, int bt, int n, float (&v)[3]) {
  v[0] = v[1] = v[2] = 0.f;
  if (lane < 2 * KS3) {
    const int sidx = lane >> 1, hf = lane & 1;
    float f[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = 0.f;
#pragma unroll
    for (int pl = 2; pl >= 0; --pl) {   // lo + mid first: their sum is exact, then + hi = the fp32 value
      const bf16x8 fr = p.feat3[(((size_t)bt * KS3 + sidx) * 3 + pl) * 64 + hf * 32 + n];
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] += egx_bf16_to_f32((unsigned short)fr[e]);
    }
    // column 470 (the template's third bf16 term, switched on for the two-plane product) is part of column 469 here
    if (sidx == KS3 - 1 && hf == 0) f[6] = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float b[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) b[e] = 0.f;
#pragma unroll
      for (int pl = 2; pl >= 0; --pl) {
        const bf16x8 br = p.dirs3[((((size_t)vt * KS3 + sidx) * 3 + pl) * 3 + c) * 64 + hf * 32 + row];
#pragma unroll
        for (int e = 0; e < 8; ++e) b[e] += egx_bf16_to_f32((unsigned short)br[e]);
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) v[c] = fmaf(f[e], b[e], v[c]);
    }
  }
}

// fp32 re-evaluation of ONE vertex the cheap evaluation of the mixed blend could not decide (see LBS_FIX_KAPPA): row `row` of vertex
// tile vt for the body in operand slot `slot`.  The whole wave works on it: lane j takes joint j's share of the blend product
// (below), the sums are reduced across the wave, lane jj < JT applies joint jl[jj] of the tile's list (weight Wt[jj * 32 + row]),
// and the trilinear sample decides.  ~70 cache lines per vertex (the first version read the MFMA-ordered operand images: 720
// lines, and 22 000 vertices of a launch with every body inside the obstacle took 330 us).
// Returns -trilinear (wave-uniform); negative = the vertex counts.
template <bool VERTEX_MAJOR>
__device__ __forceinline__ float lbs_fix_one(const LbsParams& p, int lane, int vt, int row, int slot, int JT, const int* jl, const float* Wt) {
  const int bt = slot >> 5, n = slot & 31;
  const int body = p.agent_of_slot ? p.agent_of_slot[slot / p.fpa] * p.fpa + slot % p.fpa : slot;
  const int ag = body / p.fpa;
  // v_posed = v_template + shape offsets + pose correctives, in fp32 from the vertex-major bases: lane j owns joint j - it
  // recomputes the joint's rotation from the body's parameter row (the pose kernel's formulas) and multiplies its nine R - I
  // entries with the vertex's nine columns of that joint (36 contiguous bytes per coordinate); lane 0 (the global orientation is
  // not a blend feature) takes the ten shape columns and the template
  float v[3] = {0.f, 0.f, 0.f};
  if constexpr (!VERTEX_MAJOR) {
    lbs_fix_blend_planes(p, lane, vt, row, bt, n, v);
  } else {
    const float* x = p.xb + (size_t)body * EGX_XB_DIM;
    const float* base = p.dirs_rm + ((size_t)vt * 32 + row) * 3 * KDIM;
    if (lane == 0) {
      const float* be = p.betas + (size_t)ag * 10;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        float acc = base[c * KDIM + KACT];
        for (int k = 0; k < 10; ++k) acc = fmaf(be[k], base[c * KDIM + k], acc);
        v[c] = acc;
      }
    } else if (lane < NJ && (lane < 22 || lane > 24)) {
      float R[9];
      lbs_joint_rotation(p.pc, x, lane, R);
      const int k0 = 10 + egx_compact_joint(lane) * 9;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        float acc = 0.f;
#pragma unroll
        for (int e = 0; e < 9; ++e) acc = fmaf(R[e] - ((e == 0 || e == 4 || e == 8) ? 1.f : 0.f), base[c * KDIM + k0 + e], acc);
        v[c] = acc;
      }
    }
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) v[c] = lbs_wave_sum(v[c]);
  // skinning: one joint of the tile's list per lane
  float o[3] = {0.f, 0.f, 0.f};
  if (lane < JT) {   // JT <= 55 < 64
    const int j = jl[lane] & 0xff;
    const float wv = Wt[lane * 32 + row];
    const f32x4* Aq = p.A4 + ((size_t)bt * NJ + j) * 3 * 32 + n;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const f32x4 ar = Aq[a * 32];
      o[a] = wv * fmaf(ar[0], v[0], fmaf(ar[1], v[1], fmaf(ar[2], v[2], ar[3])));
    }
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) o[a] = lbs_wave_sum(o[a]) + p.xb[(size_t)body * EGX_XB_DIM + a];
  // canonical frame -> world -> voxel coordinates: the folded affine map of the epilogue
  const float kk[3] = {p.sdf.scale * (float)p.sdf.d0 * 0.5f, p.sdf.scale * (float)p.sdf.d1 * 0.5f, p.sdf.scale * (float)p.sdf.d2 * 0.5f};
  const float cc[3] = {p.sdf.cx, p.sdf.cy, p.sdf.cz};
  const float dd[3] = {(float)p.sdf.d0, (float)p.sdf.d1, (float)p.sdf.d2};
  float vox[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    float Mw[3];
#pragma unroll
    for (int e = 0; e < 3; ++e) Mw[e] = kk[a] * (p.R0 ? p.R0[(size_t)ag * 9 + a * 3 + e] : ((a == e) ? 1.f : 0.f));
    const float tw = kk[a] * ((p.T0 ? p.T0[(size_t)ag * 3 + a] : 0.f) - cc[a]) + (dd[a] - 1.f) * 0.5f;
    vox[a] = fmaf(Mw[0], o[0], fmaf(Mw[1], o[1], fmaf(Mw[2], o[2], tw)));
  }
  return egx_sdf_neg_trilinear_at(p.sdf, __builtin_amdgcn_fmed3f(vox[0], 0.f, (float)(p.sdf.d0 - 1)), __builtin_amdgcn_fmed3f(vox[1], 0.f, (float)(p.sdf.d1 - 1)),
                                  __builtin_amdgcn_fmed3f(vox[2], 0.f, (float)(p.sdf.d2 - 1)));
}

// What a wave does with the vertices of its item that fell into the band (bit r of s_fixmap[q*32 + n] = row r of the item's
// vertex tile for body (q, n)): they go to the launch's fix-up queue - (vertex tile, row, operand slot) - which
// egx_lbs_fix_kernel works through after the fused kernel, one wave per vertex, thousands of them side by side.  Doing it here
// instead (a whole wave busy for several dependent round trips per vertex while the three other waves of its workgroup wait at
// the next item's barrier) cost 55 us of a 700 us launch for 3 300 vertices; the queue costs one atomic per wave and item
// that has any.  Only when the queue is full are they re-evaluated on the spot.
template <int NB>
__device__ __forceinline__ void lbs_fix_process(const LbsParams& p, const LbsWave& w, int vt, int bt0, int JT) {
  const int lane = w.lane;
  unsigned mybits = lane < 32 * NB ? w.s_fixmap[lane] : 0u;
  if (mybits != 0u) w.s_fixmap[lane] = 0u;
  unsigned long long pend = __ballot(mybits != 0u);
  // exclusive prefix of the per-lane counts over the (few) lanes that hold any
  const int mine = __popc(mybits);
  int total = 0, my_off = 0;
  for (unsigned long long m = pend; m != 0ull; m &= m - 1) {
    const int sl = __builtin_ctzll(m);
    if (lane == sl) my_off = total;
    total += __builtin_amdgcn_readlane(mine, sl);
  }
  int base = 0;
#ifndef EGX_LBS_FIX_IN_KERNEL   // development builds (A/B timing): never queue
  const int sq = blockIdx.x % LBS_FIX_NQ;
  int2* q = p.fixq + (size_t)sq * p.fixq_cap;
  if (lane == 0) base = atomicAdd(p.fix_stats + LBS_FIX_CNT0 + 32 * sq, total);
  base = __builtin_amdgcn_readfirstlane(base);
  if (base + total <= p.fixq_cap) {
    const int slot = min((bt0 + (lane >> 5)) * 32 + (lane & 31), p.B - 1);
    for (int k = 0; mybits != 0u; ++k) {
      const int row = __builtin_ctz(mybits);
      mybits &= mybits - 1;
      q[base + my_off + k] = make_int2(vt * 32 + row, slot);
    }
    return;
  }
  // full: what this wave reserved below the capacity is marked void (egx_lbs_fix_kernel skips it), its vertices are done here
  for (int i = base + lane; i < min(base + total, p.fixq_cap); i += 64) q[i] = make_int2(-1, 0);
#endif
  while (pend != 0ull) {
    const int sl = __builtin_ctzll(pend);
    pend &= pend - 1;
    unsigned bits = (unsigned)__builtin_amdgcn_readlane((int)mybits, sl);
    const int slot = min((bt0 + (sl >> 5)) * 32 + (sl & 31), p.B - 1);
    while (bits != 0u) {
      const int row = __builtin_ctz(bits);
      bits &= bits - 1;
      const float sv = lbs_fix_one<false>(p, lane, vt, row, slot, JT, w.s_jl, w.s_W);
      if (lane == 0) {
        if (sv < 0.f) atomicAdd(&w.s_cnt[sl], 1);
        atomicAdd(p.fix_stats, 1);
      }
    }
  }
}

template <bool WRITE_VERTS, bool DO_SDF, int RB, int QCAP, int NB = LBS_NB, bool FIX = false>
__device__ __forceinline__ void lbs_epilogue(const LbsParams& p, LbsWave& w, f32x16 (&acc)[3][NB], int vt, int bt0, int JT, bool fix_on = false) {
  const int lane = w.lane, n = w.n, half = w.half;
  float* s_W = w.s_W; int* s_jl = w.s_jl; int* s_slot = w.s_slot; unsigned* s_masks = w.s_masks; int* s_cnt = w.s_cnt;
  float* lds = w.lds; f32x4* s_queue = w.s_queue;
  const int num_bt = (p.B + 31) >> 5;
  int qn = w.qn;
  float tr[NB][3];
  int body[NB];
  bool bvalid[NB];
#pragma unroll
  for (int q = 0; q < NB; ++q) {
    const int slot = (bt0 + q) * 32 + n;     // operand slot; the body it holds (culled launches re-order the agents):
    bvalid[q] = slot < p.B;
    const int sl = bvalid[q] ? slot : p.B - 1;
    body[q] = p.agent_of_slot ? p.agent_of_slot[sl / p.fpa] * p.fpa + sl % p.fpa : sl;
    const int bb = body[q];
    tr[q][0] = p.xb[(size_t)bb * EGX_XB_DIM + 0];
    tr[q][1] = p.xb[(size_t)bb * EGX_XB_DIM + 1];
    tr[q][2] = p.xb[(size_t)bb * EGX_XB_DIM + 2];
  }
  // mixed blend, count-only tiles: |SDF value| below which the cheap evaluation of a body's vertices does not decide = the
  // body's position error bound (pose kernel) x the most the interpolated value can change per metre (aux[3] of the table)
  [[maybe_unused]] float thr[NB];
#pragma unroll
  for (int q = 0; q < NB; ++q) thr[q] = 0.f;
  if constexpr (FIX && DO_SDF) {
    if (fix_on) {
      const float lip = p.sdf_aux[3];   // steepest slope of the interpolated field, value per metre
#pragma unroll
      for (int q = 0; q < NB; ++q) {
        thr[q] = p.fix_e[min((bt0 + q) * 32 + n, p.B - 1)] * lip;
        w.s_thr[q * 32 + n] = thr[q];   // both lane halves write the same value
      }
    }
  }
  // Skinning walks the tile's joint list: one transform fetch per (joint, body) - prefetched one joint ahead - applied
  // to the lane's 16 vertices with their weights from LDS (o = sum_j w_j (A_j v + t_j); rows whose weights are all zero
  // are skipped in groups of four).  The accumulators already hold v_template + offsets (template column of the GEMM).
  auto sdf_flush = [&](int count) {
    __builtin_amdgcn_wave_barrier();
    for (int base = 0; base < count; base += 64) {
      const int idx = base + lane;
      if (idx < count) {
        const f32x4 e = s_queue[idx];
        const int code = __float_as_int(e[3]);   // counter slot | vertex row << 8
        const float sv = egx_sdf_neg_trilinear_at(p.sdf, e[0], e[1], e[2]);
        if constexpr (FIX) {
          const float t = fix_on ? w.s_thr[code & 63] : 0.f;
          if (sv < -t) atomicAdd(&s_cnt[code & 63], 1);
          else if (fix_on && sv <= t) {
            const int rr = (code >> 8) & 15;   // accumulator row -> row of the vertex tile
            atomicOr(&w.s_fixmap[code & 63], 1u << ((rr & 3) + 8 * (rr >> 2) + 4 * (code >> 12)));
          }
        } else {
          if (sv < 0.f) atomicAdd(&s_cnt[code & 63], 1);
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
  };
  // both masks are properties of the TILE: wave-uniform, kept in SGPRs (readfirstlane), so "does this tile hold picked
  // vertices" (8 of 328 tiles) is a scalar branch; the lane's share is one shift by 4 * half, after which every row test is a
  // compile-time bit position.  (Round 4 tested `mask >> row` with row = f(r, half): the compiler hoisted sixteen per-lane
  // `1 << row` constants out of the persistent loop and spilled eleven of them - scratch reloads, each with a vmcnt(0) that
  // drained the transform prefetch.)
  const unsigned pick_mask = p.picked ? (unsigned)__builtin_amdgcn_readfirstlane((int)s_masks[0]) : 0u;
  const unsigned sdf_mask = (unsigned)__builtin_amdgcn_readfirstlane((int)s_masks[1]);
#ifdef EGX_LBS_TIMING
  unsigned long long (&et)[8] = w.et;
#endif
#pragma unroll
  for (int q = 0; q < NB; ++q) {
    [[maybe_unused]] const unsigned long long q0 = LBS_NOW();
    const f32x4* Aq = p.A4 + (size_t)min(bt0 + q, num_bt - 1) * NJ * 3 * 32 + n;
    // rows are handled in adjacent pairs (r, r+1): the accumulator registers, weights and outputs of a pair are
    // neighbours, so the nine transform FMAs and three weight FMAs map onto packed fp32 instructions
    float o[16][3];
#pragma unroll
    for (int r = 0; r < 16; ++r) { o[r][0] = tr[q][0]; o[r][1] = tr[q][1]; o[r][2] = tr[q][2]; }
    f32x4 a0, a1, a2;
    {
      const int j = s_jl[0] & 0xff;   // entries: joint | row-group mask << 8
      a0 = Aq[(j * 3 + 0) * 32]; a1 = Aq[(j * 3 + 1) * 32]; a2 = Aq[(j * 3 + 2) * 32];
    }
#if EGX_LBS_PF >= 2
    f32x4 b0, b1, b2;   // the joint after next: two transform fetches in flight (development variant)
    {
      const int j = s_jl[min(1, JT - 1)] & 0xff;
      b0 = Aq[(j * 3 + 0) * 32]; b1 = Aq[(j * 3 + 1) * 32]; b2 = Aq[(j * 3 + 2) * 32];
    }
#endif
    for (int jj = 0; jj < JT; ++jj) {
#if EGX_LBS_PF >= 2
      const int jn = s_jl[min(jj + 2, JT - 1)] & 0xff;
#else
      const int jn = s_jl[min(jj + 1, JT - 1)] & 0xff;
#endif
      const f32x4 n0 = Aq[(jn * 3 + 0) * 32], n1 = Aq[(jn * 3 + 1) * 32], n2 = Aq[(jn * 3 + 2) * 32];
      // which row groups this joint touches: a property of the tile, precomputed at load (round 4 derived it from the weights
      // with four compares, three ORs and a ballot per group, joint and body tile)
      const int gmask = __builtin_amdgcn_readfirstlane(s_jl[jj]) >> 8;
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        if (!((gmask >> rg) & 1)) continue;   // scalar branch
        const f32x4 w4 = *reinterpret_cast<const f32x4*>(&s_W[jj * 32 + 8 * rg + 4 * half]);
        // plain v_fma_f32 on purpose (lbs_fma): packed fp32 FMAs beside another wave's MFMAs cost more than they save on
        // gfx950 (MI355X_MICROARCH.md, price of a filler), and the row pairs they need cost two v_mov per operand
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = rg * 4 + e;
          const float vx = acc[0][q][r], vy = acc[1][q][r], vz = acc[2][q][r], wv = w4[e];
          const float px = lbs_fma(a0[0], vx, lbs_fma(a0[1], vy, lbs_fma(a0[2], vz, a0[3])));
          const float py = lbs_fma(a1[0], vx, lbs_fma(a1[1], vy, lbs_fma(a1[2], vz, a1[3])));
          const float pz = lbs_fma(a2[0], vx, lbs_fma(a2[1], vy, lbs_fma(a2[2], vz, a2[3])));
          o[r][0] = lbs_fma(wv, px, o[r][0]);
          o[r][1] = lbs_fma(wv, py, o[r][1]);
          o[r][2] = lbs_fma(wv, pz, o[r][2]);
        }
      }
#if EGX_LBS_PF >= 2
      a0 = b0; a1 = b1; a2 = b2;
      b0 = n0; b1 = n1; b2 = n2;
#else
      a0 = n0; a1 = n1; a2 = n2;
#endif
    }
#ifdef EGX_LBS_TIMING
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long q1 = LBS_NOW();
    et[0] += q1 - q0;
#endif
    if (DO_SDF) {
      // Bracket table first (eight independent 8-byte loads per batch).  A vertex the brackets cannot decide needs the
      // eight-corner interpolation; executed in place that would run for the whole wave whenever ONE lane needs it, and
      // with 64 different bodies across the lanes that is almost every row.  Undecided points are therefore appended
      // to a wave-private LDS queue and evaluated densely (64 queued points per pass) by sdf_flush().
      const int ag = body[q] / p.fpa;
      // canonical frame -> world (R0, T0) -> unclamped voxel coordinates ((w - c) scale + 1) d / 2 - 1 / 2 folded into one
      // affine map per body (align_corners=False, utils.py:58-68); the clamp (padding "border") happens in the lookup /
      // before the exact evaluation.  The folded rounding differs from the reference's chain by ~1e-7 relative - far
      // inside the level-set band the counts are compared in.
      float Mw[9], tw[3];
      {
        const float kx = p.sdf.scale * (float)p.sdf.d0 * 0.5f, ky = p.sdf.scale * (float)p.sdf.d1 * 0.5f,
                    kz = p.sdf.scale * (float)p.sdf.d2 * 0.5f;
        const float kk[3] = {kx, ky, kz};
        const float cc[3] = {p.sdf.cx, p.sdf.cy, p.sdf.cz};
        const float dd[3] = {(float)p.sdf.d0, (float)p.sdf.d1, (float)p.sdf.d2};
#pragma unroll
        for (int a = 0; a < 3; ++a) {
#pragma unroll
          for (int e = 0; e < 3; ++e) Mw[a * 3 + e] = kk[a] * (p.R0 ? p.R0[(size_t)ag * 9 + a * 3 + e] : ((a == e) ? 1.f : 0.f));
          tw[a] = kk[a] * ((p.T0 ? p.T0[(size_t)ag * 3 + a] : 0.f) - cc[a]) + (dd[a] - 1.f) * 0.5f;
        }
      }
      const float hx = (float)(p.sdf.d0 - 1), hy = (float)(p.sdf.d1 - 1), hz = (float)(p.sdf.d2 - 1);
      const unsigned mine = bvalid[q] ? (sdf_mask >> (4 * half)) : 0u;  // bit (r&3)+8(r>>2) = this lane's row r
      int cnt = 0;
      // all sixteen bracket lookups of the lane's rows are issued before the first one is used: one L2 round trip per body
      // tile instead of one per batch of RB rows (round 4; the epilogue is a latency chain - two waves per SIMD - and these
      // gathers were four of its eight round trips per item).  The world coordinates are not kept: the rare undecided point
      // recomputes its own (12 FMAs) inside the queue branch.
      auto world = [&](int r, int a) { return fmaf(Mw[a * 3 + 0], o[r][0], fmaf(Mw[a * 3 + 1], o[r][1], fmaf(Mw[a * 3 + 2], o[r][2], tw[a]))); };
      // the bracket lookup wants CELL coordinates r / 4 + 1: the same affine map scaled by 1/4 (exact) with the +1 folded into
      // its constant - three FMAs per point instead of six; a point within round-off of a cell border may land in the
      // neighbouring cell, which egx_sdf_coarse_at_raw already allows for
      float Mc[9], tc[3];
#pragma unroll
      for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int e = 0; e < 3; ++e) Mc[a * 3 + e] = 0.25f * Mw[a * 3 + e];
        tc[a] = fmaf(0.25f, tw[a], 1.f);
      }
      auto cell = [&](int r, int a) { return fmaf(Mc[a * 3 + 0], o[r][0], fmaf(Mc[a * 3 + 1], o[r][1], fmaf(Mc[a * 3 + 2], o[r][2], tc[a]))); };
      constexpr int LB = WRITE_VERTS ? RB : 16;   // rows per lookup burst (the vertex-writing variant has no registers to spare)
#pragma unroll
      for (int rb0 = 0; rb0 < 16; rb0 += LB) {
      float2 mm[LB];
#pragma unroll
      for (int r = rb0; r < rb0 + LB; ++r) mm[r - rb0] = egx_sdf_coarse_at_cell(p.sdf, cell(r, 0), cell(r, 1), cell(r, 2));
#pragma unroll
      for (int r0 = rb0; r0 < rb0 + LB; r0 += RB) {
        if (qn + RB * 64 > QCAP) { sdf_flush(qn); qn = 0; }  // room for one batch: RB rows x 64 lanes
#pragma unroll
        for (int r = r0; r < r0 + RB; ++r) {
          const bool on = (mine >> ((r & 3) + 8 * (r >> 2))) & 1u;
          const bool inside = mm[r - rb0].x > thr[q];
          cnt += (on && inside) ? 1 : 0;
          const bool und = on && !inside && !(mm[r - rb0].y < -thr[q]);
          const unsigned long long bm = __ballot(und);
          if (bm != 0) {
            const int pos = qn + __builtin_amdgcn_mbcnt_hi((unsigned)(bm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bm, 0u));
            if (und) {
              // counter slot | accumulator row r << 8 | lane half << 12, formed here (as loop invariants the sixteen per-lane
              // codes of a body tile would be kept alive across the whole item)
              int ln = lane;
              asm volatile("" : "+v"(ln));
              s_queue[pos] = f32x4{__builtin_amdgcn_fmed3f(world(r, 0), 0.f, hx), __builtin_amdgcn_fmed3f(world(r, 1), 0.f, hy),
                                   __builtin_amdgcn_fmed3f(world(r, 2), 0.f, hz),
                                   __int_as_float((q * 32 + (ln & 31)) | (r << 8) | ((ln >> 5) << 12))};
            }
            qn += __popcll(bm);
          }
        }
      }
      }
      if (cnt != 0) {
        int nn = n;                       // address formed here (see run_item: no loop-invariant per-lane address to keep alive)
        asm volatile("" : "+v"(nn));
        atomicAdd(&s_cnt[q * 32 + nn], cnt);
      }
      if (WRITE_VERTS) { sdf_flush(qn); qn = 0; }  // the queue shares its LDS with the vertex transpose buffer
    }
#ifdef EGX_LBS_TIMING
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long q2 = LBS_NOW();
    et[1] += q2 - q1;
#endif
    if (pick_mask != 0) {   // scalar branch
      const unsigned pmine = bvalid[q] ? (pick_mask >> (4 * half)) : 0u;   // bit (r&3)+8(r>>2) = this lane's row r
      const int* slot_h = s_slot + 4 * half;
      float* pbase = p.picked + (size_t)body[q] * p.NP * 3;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if ((pmine >> ((r & 3) + 8 * (r >> 2))) & 1u) {
          float* op = pbase + slot_h[(r & 3) + 8 * (r >> 2)] * 3;
          op[0] = o[r][0]; op[1] = o[r][1]; op[2] = o[r][2];
        }
      }
    }
    if (WRITE_VERTS) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
        lds[n * 97 + row * 3 + 0] = o[r][0];
        lds[n * 97 + row * 3 + 1] = o[r][1];
        lds[n * 97 + row * 3 + 2] = o[r][2];
      }
    }
    if (WRITE_VERTS) {
      // transpose through LDS so that one wave instruction writes whole vertices of ONE body (the rows of a tile are
      // in the joint-sorted order: each lands at its original vertex id; wave-private LDS region, DS ops of one wave
      // execute in order, no barrier needed)
      int dst[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int f = lane + 64 * u;  // f = row * 3 + coordinate, 96 values per body
        const int vo = (f < 96) ? p.vorig[vt * 32 + f / 3] : -1;
        dst[u] = (vo >= 0) ? vo * 3 + f % 3 : -1;
      }
      for (int bi = 0; bi < 32; ++bi) {
        const int bd = (bt0 + q) * 32 + bi;
        if (bd >= p.B) break;
        float* o = p.verts + (size_t)bd * p.V * 3;
#pragma unroll
        for (int u = 0; u < 2; ++u)
          if (dst[u] >= 0) o[dst[u]] = lds[bi * 97 + lane + 64 * u];
      }
    }
  }
#ifdef EGX_LBS_TIMING
  const unsigned long long f0 = LBS_NOW();
#endif
  if (DO_SDF) {
    if (!WRITE_VERTS) { sdf_flush(qn); qn = 0; }
    __builtin_amdgcn_wave_barrier();
    if constexpr (FIX) {
      if (fix_on) {   // wave-uniform
        const unsigned fb = lane < 32 * NB ? w.s_fixmap[lane] : 0u;
        if (__ballot(fb != 0u) != 0ull) lbs_fix_process<NB>(p, w, vt, bt0, JT);
        __builtin_amdgcn_wave_barrier();
      }
    }
    const int c = lane < 32 * NB ? s_cnt[lane] : 0;   // lane = q*32 + n: one global atomic per body and item
    if (lane < 32 * NB) s_cnt[lane] = 0;
    const int sd = (bt0 + (lane >> 5)) * 32 + (lane & 31);
    if (c != 0 && sd < p.B && lane < 32 * NB) {
      const int bd = p.agent_of_slot ? p.agent_of_slot[sd / p.fpa] * p.fpa + sd % p.fpa : sd;
      atomicAdd(p.pene + bd, c);
    }
    __builtin_amdgcn_wave_barrier();
  }
#ifdef EGX_LBS_TIMING
  __builtin_amdgcn_sched_barrier(0);
  et[2] += LBS_NOW() - f0;
  et[3] += 1;
#endif
  w.qn = qn;
}

#define LBS_MFMA_RESULT_WAIT()                   \
  do {                                           \
    __builtin_amdgcn_sched_barrier(0);           \
    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory"); \
    __builtin_amdgcn_sched_barrier(0);           \
  } while (0)

// Epilogue of a COUNT-ONLY tile of the mixed blend (mode 3; a tile after the ones that hold picked vertices, joint list of at most
// eight): skinning on the matrix pipe, straight into SDF-cell coordinates.
//   o_cell[v, body] = cinit[body] + sum_j W[v, j] (A'_j [v_posed; 1]),   A'_j = Mc A_j   (pose kernel: skinB, cinit)
// is evaluated as T = W x A' - twelve 32 x 32 outputs per 32-body tile, one per entry (a, c) of the 3 x 4 transform, K = the eight
// joints of the tile's list x the two planes of A' (see SKIN_BT_BYTES), two MFMAs each - followed by
// o[a] = T[a][0] x + T[a][1] y + T[a][2] z + T[a][3] on the accumulators of the blend GEMM, which already hold (x, y, z) in the
// same lane layout: 9 FMAs per (vertex, body) instead of 12 per (vertex, body, joint), and no canonical -> cell map (9 more)
// afterwards.  The operands of a body tile arrive in ONE burst (24 bytes per joint and lane, joint-major) and are turned into
// MFMA operands (entry-major, eight joints each) by 48 v_perm_b32: one L2 round trip per body tile instead of one per joint.
// What it costs: 24 MFMAs per body tile on a matrix pipe that was 23 % busy, and a position error of up to LBS_SKIN_ERR (|v| + |t|),
// which the fix-up band absorbs - the result only classifies, lbs_fix_process decides the close calls.
template <int RB, int QCAP, int NB>
__device__ __forceinline__ void lbs_epilogue_cell(const LbsParams& p, LbsWave& w, f32x16 (&acc)[3][NB], int vt, int bt0, int JT) {
  int lane = w.lane;
  asm volatile("" : "+v"(lane));   // per-lane operand addresses are formed per item (not kept alive as invariants of the persistent loop)
  const int n = lane & 31, half = lane >> 5;
  int* s_cnt = w.s_cnt;
  f32x4* s_queue = w.s_queue;
  const int num_bt = (p.B + 31) >> 5;
  int qn = w.qn;
  const unsigned sdf_mask = (unsigned)__builtin_amdgcn_readfirstlane((int)w.s_masks[1]);
  const int ks0 = __builtin_amdgcn_readfirstlane(p.skin_ks_off[vt]);   // JT <= 8 here: one k-step (the caller sends longer lists to the VALU epilogue)
  const float lip = p.sdf_aux[3];   // steepest slope of the interpolated field, value per metre
  const float hx = (float)(p.sdf.d0 - 1), hy = (float)(p.sdf.d1 - 1), hz = (float)(p.sdf.d2 - 1);
  auto sdf_flush = [&](int count) {
    __builtin_amdgcn_wave_barrier();
    for (int base = 0; base < count; base += 64) {
      const int idx = base + lane;
      if (idx < count) {
        const f32x4 e = s_queue[idx];
        const int code = __float_as_int(e[3]);   // counter slot | accumulator row << 8 | lane half << 12
        const float sv = egx_sdf_neg_trilinear_at(p.sdf, e[0], e[1], e[2]);
        const float t = w.s_thr[code & 63];
        if (sv < -t) atomicAdd(&s_cnt[code & 63], 1);
        else if (sv <= t) {
          const int rr = (code >> 8) & 15;
          atomicOr(&w.s_fixmap[code & 63], 1u << ((rr & 3) + 8 * (rr >> 2) + 4 * (code >> 12)));
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
  };
  // two 16-bit entries of neighbouring joints -> one operand register: v_perm_b32 picks the low (even entry) or high halves
  auto pack2 = [](unsigned hi_joint, unsigned lo_joint, int odd) {
    return odd ? __builtin_amdgcn_perm(hi_joint, lo_joint, 0x07060302u) : __builtin_amdgcn_perm(hi_joint, lo_joint, 0x05040100u);
  };
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  // the tile's joint list (LDS, published by the blend's barriers) -> one register, read out lane by lane below: a load per
  // entry in front of its records would make every body tile a chain of JT round trips again
  const int jl_v = w.s_jl[min(lane, JT - 1)] & 0xff;
  // the tile's weight operands: once per item
  const bf16x8 W0 = p.skinW[((size_t)ks0 * 2 + 0) * 64 + lane], W1 = p.skinW[((size_t)ks0 * 2 + 1) * 64 + lane];
#pragma unroll
  for (int q = 0; q < NB; ++q) {
    const int slot = (bt0 + q) * 32 + n;
    const bool bvalid = slot < p.B;
    const f32x4 ci = p.cinit[bvalid ? slot : p.B - 1];
    const float fe = p.fix_e[bvalid ? slot : p.B - 1];   // requested here, used after the skinning
    // this lane's records: plane = lane half, body column n; record of joint j at index j * 64
    const char* tile_base = reinterpret_cast<const char*>(p.skinB) + (size_t)min(bt0 + q, num_bt - 1) * SKIN_BT_BYTES;
    const u32x4* recA = reinterpret_cast<const u32x4*>(tile_base) + half * 32 + n;
    const u32x2* recB = reinterpret_cast<const u32x2*>(tile_base + (size_t)SKIN_BT_A * 16) + half * 32 + n;
    float o[16][3];
    {
      // all twelve entries of the (up to) eight joints in ONE burst: 16 + 8 bytes per joint and lane
      u32x4 RA[8];
      u32x2 RC[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        if (e < JT) {   // scalar branch
          const int j = __builtin_amdgcn_readlane(jl_v, e);
          RA[e] = recA[j * 64];
          RC[e] = recB[j * 64];
        } else {        // the weights of the unused slots are zero: any finite operand does
          RA[e] = u32x4{0u, 0u, 0u, 0u};
          RC[e] = u32x2{0u, 0u};
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      // records (joint-major) -> operands (entry-major, eight joints per operand): all twelve now, so that the 48 record registers
      // are free before the first accumulators are
      bf16x8 Bop[12];
#pragma unroll
      for (int cc = 0; cc < 12; ++cc) {
        u32x4 v;
#pragma unroll
        for (int i = 0; i < 4; ++i)
          v[i] = cc < 8 ? pack2(RA[2 * i + 1][cc >> 1], RA[2 * i][cc >> 1], cc & 1) : pack2(RC[2 * i + 1][(cc - 8) >> 1], RC[2 * i][(cc - 8) >> 1], cc & 1);
        Bop[cc] = __builtin_bit_cast(bf16x8, v);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        // two entries of the row at a time (32 accumulator registers live instead of 64): (translation, x) then (y, z);
        // W_mid A'_hi first (the small term), then W_hi (A'_hi + A'_mid)
        f32x16 Ta, Tb;
#pragma unroll
        for (int r = 0; r < 16; ++r) { Ta[r] = ci[a]; Tb[r] = 0.f; }
        Ta = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W1, Bop[a * 4 + 3], Ta, 0, 0, 0);
        Tb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W1, Bop[a * 4 + 0], Tb, 0, 0, 0);
        Ta = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W0, Bop[a * 4 + 3], Ta, 0, 0, 0);
        Tb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W0, Bop[a * 4 + 0], Tb, 0, 0, 0);
        float oa[16];
#pragma unroll
        // The FMAs below are inline asm (lbs_fma: see there), which the compiler's hazard recogniser does not look into: the wait
        // states between an MFMA and a VALU read of its result (software-managed on CDNA: up to 19 for a 16-pass MFMA) are put
        // here by hand.  Without them the FMAs read accumulators the matrix pipe is still writing.
        LBS_MFMA_RESULT_WAIT();
        for (int r = 0; r < 16; ++r) oa[r] = lbs_fma(Tb[r], acc[0][q][r], Ta[r]);
#pragma unroll
        for (int r = 0; r < 16; ++r) { Ta[r] = 0.f; Tb[r] = 0.f; }
        Ta = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W1, Bop[a * 4 + 1], Ta, 0, 0, 0);
        Tb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W1, Bop[a * 4 + 2], Tb, 0, 0, 0);
        Ta = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W0, Bop[a * 4 + 1], Ta, 0, 0, 0);
        Tb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W0, Bop[a * 4 + 2], Tb, 0, 0, 0);
#pragma unroll
        LBS_MFMA_RESULT_WAIT();
#pragma unroll
        for (int r = 0; r < 16; ++r) o[r][a] = lbs_fma(Ta[r], acc[1][q][r], lbs_fma(Tb[r], acc[2][q][r], oa[r]));
      }
    }
    // SDF: bracket lookups of all sixteen rows in one burst (cell coordinates are what the skinning produced), decisions with the
    // body's band, undecided points to the wave's queue as clamped voxel coordinates 4 (cell - 1)
    const unsigned mine = bvalid ? (sdf_mask >> (4 * half)) : 0u;
    int cnt = 0;
    float2 mm[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) mm[r] = egx_sdf_coarse_at_cell(p.sdf, o[r][0], o[r][1], o[r][2]);
#ifdef EGX_LBS_THR0   // development builds (A/B timing): no band - the cheap evaluation decides everything
    const float thr = 0.f * fe;
#else
    const float thr = fe * lip;
#endif
    w.s_thr[q * 32 + n] = thr;   // both lane halves write the same value
#pragma unroll
    for (int r0 = 0; r0 < 16; r0 += RB) {
      if (qn + RB * 64 > QCAP) { sdf_flush(qn); qn = 0; }
#pragma unroll
      for (int r = r0; r < r0 + RB; ++r) {
        const bool on = (mine >> ((r & 3) + 8 * (r >> 2))) & 1u;
        const bool inside = mm[r].x > thr;
        cnt += (on && inside) ? 1 : 0;
        const bool und = on && !inside && !(mm[r].y < -thr);
        const unsigned long long bm = __ballot(und);
        if (bm != 0) {
          const int pos = qn + __builtin_amdgcn_mbcnt_hi((unsigned)(bm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bm, 0u));
          if (und) {
            int ln = lane;
            asm volatile("" : "+v"(ln));
            s_queue[pos] = f32x4{__builtin_amdgcn_fmed3f(fmaf(4.f, o[r][0], -4.f), 0.f, hx), __builtin_amdgcn_fmed3f(fmaf(4.f, o[r][1], -4.f), 0.f, hy),
                                 __builtin_amdgcn_fmed3f(fmaf(4.f, o[r][2], -4.f), 0.f, hz),
                                 __int_as_float((q * 32 + (ln & 31)) | (r << 8) | ((ln >> 5) << 12))};
          }
          qn += __popcll(bm);
        }
      }
    }
    if (cnt != 0) {
      int nn = n;
      asm volatile("" : "+v"(nn));
      atomicAdd(&s_cnt[q * 32 + nn], cnt);
    }
  }
  sdf_flush(qn);
  qn = 0;
  __builtin_amdgcn_wave_barrier();
  {
    const unsigned fb = lane < 32 * NB ? w.s_fixmap[lane] : 0u;
#ifdef EGX_LBS_NOFIXPROC   // development builds (A/B timing): the band is kept, what falls into it is dropped
    if (fb != 0u) w.s_fixmap[lane] = 0u;
#else
    if (__ballot(fb != 0u) != 0ull) lbs_fix_process<NB>(p, w, vt, bt0, JT);
#endif
    __builtin_amdgcn_wave_barrier();
  }
  const int c = lane < 32 * NB ? s_cnt[lane] : 0;   // lane = q*32 + n: one global atomic per body and item
  if (lane < 32 * NB) s_cnt[lane] = 0;
  const int sd = (bt0 + (lane >> 5)) * 32 + (lane & 31);
  if (c != 0 && sd < p.B && lane < 32 * NB) {
    const int bd = p.agent_of_slot ? p.agent_of_slot[sd / p.fpa] * p.fpa + sd % p.fpa : sd;
    atomicAdd(p.pene + bd, c);
  }
  __builtin_amdgcn_wave_barrier();
  w.qn = qn;
}

// fp32 blend GEMM of one work item on v_mfma_f32_32x32x2_f32: acc = [v_template | bases] x [1 | features]
__device__ __forceinline__ void lbs_blend_f32(const LbsParams& p, f32x16 (&acc)[3][LBS_NB], int vt, int bt0, int lane) {
  constexpr int NB = LBS_NB;
  const int num_bt = (p.B + 31) >> 5;
  const f32x4* dp = p.dirs + (size_t)vt * KGROUPS * 3 * 64 + lane;
  const f32x4* fp[NB];
#pragma unroll
  for (int q = 0; q < NB; ++q) fp[q] = p.feat + (size_t)min(bt0 + q, num_bt - 1) * KGROUPS * 64 + lane;

  // Operand bursts.  Measured on gfx950 (scripts/ubench/mfma_loads.hip): a wave that issues v_mfma_f32_32x32x2_f32
  // while its own global loads are still in flight runs the matrix pipe at about half rate (72 vs 136 TFLOP/s
  // chip-wide for this exact loop), whereas "load a burst, s_waitcnt vmcnt(0), then only MFMAs" keeps 98 % of the
  // load-free rate - the exposed load latency is covered by the other wave of the SIMD, whose MFMAs are not affected
  // by this wave's returning data.  So: no software prefetch; LBS_BURST k-groups of operands per burst.
  constexpr int LBS_BURST = 2;
  if (!(p.dbg & 2)) {
    f32x4 a_st[LBS_BURST][3], b_st[LBS_BURST][NB];
    constexpr int KMAIN = KGROUPS / LBS_BURST * LBS_BURST;
    for (int g0 = 0; g0 < KMAIN; g0 += LBS_BURST) {
#pragma unroll
      for (int u = 0; u < LBS_BURST; ++u) {
#pragma unroll
        for (int c = 0; c < 3; ++c) a_st[u][c] = dp[((g0 + u) * 3 + c) * 64];
#pragma unroll
        for (int q = 0; q < NB; ++q) b_st[u][q] = fp[q][(g0 + u) * 64];
      }
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < LBS_BURST; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int q = 0; q < NB; ++q)
              acc[c][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_st[u][c][e], b_st[u][q][e], acc[c][q], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int g = KMAIN; g < KGROUPS; ++g) {  // tail groups, one at a time
#pragma unroll
      for (int c = 0; c < 3; ++c) a_st[0][c] = dp[(g * 3 + c) * 64];
#pragma unroll
      for (int q = 0; q < NB; ++q) b_st[0][q] = fp[q][g * 64];
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
          for (int q = 0; q < NB; ++q)
            acc[c][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_st[0][c][e], b_st[0][q][e], acc[c][q], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }

}

template <bool WRITE_VERTS, bool DO_SDF>
__global__ __launch_bounds__(LBS_THREADS, 1) void egx_lbs_fused_kernel(LbsParams p) {
  constexpr int NB = LBS_NB;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int set = wave >> 2, w4 = wave & 3;  // two independent sets of four waves, each walking its own item stream
  char* my = smem_raw + wave * (LBS_META_BYTES + (WRITE_VERTS ? LBS_VERT_BYTES : (DO_SDF ? LBS_QCAP * 16 : 0)));
  LbsWave w = lbs_wave_init<WRITE_VERTS, DO_SDF>(my, lane);
  // work streams.  With >= 8 body groups every XCD (block id % 8) owns a contiguous chunk of body groups, so their packed
  // features / transforms stay in that XCD's L2 while the blend bases stream through once per XCD; the streams of an XCD
  // walk its (vertex tile, body group) list vertex-tile-major, i.e. at any time they share a dozen consecutive tiles.
  int bg_lo, nper, n_streams, stream;
  if (p.nbg >= 8 && (gridDim.x & 7) == 0) {
    const int per = (p.nbg + 7) / 8, xcd = blockIdx.x & 7;
    bg_lo = xcd * per;
    nper = max(0, min(per, p.nbg - bg_lo));
    n_streams = (gridDim.x >> 3) * 2;
    stream = (blockIdx.x >> 3) * 2 + set;
  } else {
    bg_lo = 0; nper = p.nbg;
    n_streams = gridDim.x * 2;
    stream = blockIdx.x * 2 + set;
  }
  const int n_items = p.n_tiles * nper;
  for (int item = stream; item < n_items; item += n_streams) {
    const int vti = item / nper, bg = bg_lo + item % nper;
    const int vt = p.tiles ? p.tiles[vti] : vti;
    const int bt0 = bg * 8 + w4 * NB;  // first 32-body tile of this wave
    const int JT = lbs_load_meta(p, w, vt);
    f32x16 acc[3][NB];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int q = 0; q < NB; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][q][r] = 0.f;
    if (!(p.dbg & 2)) lbs_blend_f32(p, acc, vt, bt0, lane);
    if (p.dbg & 1) {
      float sum = 0.f;
#pragma unroll
      for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int q = 0; q < NB; ++q)
#pragma unroll
          for (int r = 0; r < 16; ++r) sum += acc[c][q][r];
      if (sum == 123.456f) p.pene[0] = 1;
      continue;
    }
    lbs_epilogue<WRITE_VERTS, DO_SDF, 8, LBS_QCAP>(p, w, acc, vt, bt0, JT);
  }
}

// ------------------------------------------------------------------------------------------------
// kernel 2b: the same work item with the blend GEMM as a 3-term bf16 split (see KS3 above).
// Workgroup = 4 waves = one vertex tile x 256 bodies; the bases of a stage (2 k-steps x 3 planes x 3 coordinates = 18
// pieces of 1 KiB) are fetched once per workgroup, parked in LDS (double buffered, one barrier per stage) and read back
// by all four waves; each wave fetches its own feature pieces (12 KiB per stage) into registers.  Loads are issued as a
// burst and waited for before the stage's 72 MFMAs (no VMEM in flight under MFMA, scripts/ubench/mfma_bf16.hip); the
// second workgroup of the CU covers the gap, and - unlike the fp32 MFMA, which shares the fp32 VALU lanes - the bf16
// matrix pipe runs concurrently with the other workgroup's VALU epilogue.
// ------------------------------------------------------------------------------------------------
template <int NPL>
struct Wg4Cfg {
  static constexpr int STAGE_KS = NPL == 3 ? 2 : 3;          // k-steps per stage: 72 / 54 MFMAs per wave and stage
  static constexpr int STAGE_PIECES = STAGE_KS * NPL * 3;    // 1 KiB base pieces per stage
  static constexpr int STAGES = KS3 / STAGE_KS;
  static_assert(KS3 % STAGE_KS == 0, "stages cover K exactly");
};
constexpr int LBS3_SHARED_BYTES = 2 * 18 * 1024 + 7424;                // stage ring (18 pieces in either mode) + tile metadata
static_assert(Wg4Cfg<3>::STAGE_PIECES <= 18 && Wg4Cfg<2>::STAGE_PIECES <= 18, "stage ring");
constexpr int LBS3_RB = 4;                                             // SDF rows per bracket batch
constexpr int LBS3_QCAP = LBS3_RB * 64 + 64;
// the small wave tile runs THREE workgroups per CU and must stay below round 5's 53.5 KB of LDS per workgroup to do so (with the
// fix-up bitmap and thresholds added, 54.0 KB, the launch lost the third workgroup: +45 % at every size): its queue gives up 32 entries
template <int NBW> constexpr int lbs3_qcap() { return NBW == 1 ? LBS3_RB * 64 + 32 : LBS3_QCAP; }
// per-wave LDS of the fused3 kernels: penetration counters, fix-up bitmap, fix-up thresholds (32 x NB entries each), queue
template <int NBW> constexpr int lbs3_wave_bytes() { return 3 * 128 * NBW + lbs3_qcap<NBW>() * 16; }

template <int NPL>
__device__ __forceinline__ void lbs_blend_split(const LbsParams& p, f32x16 (&acc)[3][LBS_NB], int vt, int bt0, int lane, int wave,
                                                bf16x8* sA, unsigned long long* tacc) {
  using Cfg = Wg4Cfg<NPL>;
  constexpr int NB = LBS_NB, SKS = Cfg::STAGE_KS, SP = Cfg::STAGE_PIECES;
  const int num_bt = (p.B + 31) >> 5;
  asm volatile("" : "+v"(lane));   // per-lane operand addresses are formed per item (not kept alive as invariants of the persistent loop)
  const bf16x8* dpv = p.dirs3 + (size_t)vt * KS3 * 9 * 64 + lane;  // piece (s, plane, coord) at ((s*3 + plane)*3 + coord)*64
  const bf16x8* fq[NB];
#pragma unroll
  for (int q = 0; q < NB; ++q) fq[q] = p.feat3 + (size_t)min(bt0 + q, num_bt - 1) * KS3 * 3 * 64 + lane;
  for (int st = 0; st < Cfg::STAGES; ++st) {
    // burst: this wave's share of the stage's base pieces + its own feature pieces
    constexpr int NGA = (SP + 3) / 4;
    bf16x8 ga[NGA], b[SKS][NPL][NB];
    [[maybe_unused]] const unsigned long long t0 = LBS_NOW();
#pragma unroll
    for (int i = 0; i < NGA; ++i) {
      const int piece = wave + 4 * i;                 // (ks, plane, coord) of the stage, planes 0..NPL-1 only
      if (piece < SP) ga[i] = dpv[(size_t)((st * SKS + piece / (NPL * 3)) * 9 + piece % (NPL * 3)) * 64];
    }
#pragma unroll
    for (int ks = 0; ks < SKS; ++ks)
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
        for (int q = 0; q < NB; ++q) b[ks][pl][q] = fq[q][((st * SKS + ks) * 3 + pl) * 64];
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    [[maybe_unused]] const unsigned long long t1 = LBS_NOW();
    bf16x8* buf = sA + (st & 1) * 18 * 64;
#pragma unroll
    for (int i = 0; i < NGA; ++i) {
      const int piece = wave + 4 * i;
      if (piece < SP) buf[piece * 64 + lane] = ga[i];
    }
    __syncthreads();  // stage visible; also: everyone is done reading the other buffer's previous contents
    [[maybe_unused]] const unsigned long long t2 = LBS_NOW();
#pragma unroll
    for (int ks = 0; ks < SKS; ++ks) {
      bf16x8 a[NPL][3];  // [plane][coord]
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
        for (int c = 0; c < 3; ++c) a[pl][c] = buf[((ks * NPL + pl) * 3 + c) * 64 + lane];
      // product-major order: consecutive MFMAs go to different accumulator tuples, so no MFMA waits for the previous
      // one's result (small partial products first)
      constexpr int NPROD = NPL == 3 ? 6 : 3;
#pragma unroll
      for (int pr = 0; pr < NPROD; ++pr) {
        int pa, pb;
        if (NPL == 3) {
          pa = (pr == 0) ? 1 : (pr == 1) ? 0 : (pr == 2) ? 2 : (pr == 3) ? 0 : (pr == 4) ? 1 : 0;
          pb = (pr == 0) ? 1 : (pr == 1) ? 2 : (pr == 2) ? 0 : (pr == 3) ? 1 : (pr == 4) ? 0 : 0;
        } else {
          pa = (pr == 0) ? 0 : (pr == 1) ? 1 : 0;
          pb = (pr == 0) ? 1 : (pr == 1) ? 0 : 0;
        }
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
          for (int q = 0; q < NB; ++q)
            acc[c][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[pa][c], b[ks][pb][q], acc[c][q], 0, 0, 0);
      }
    }
#ifdef EGX_LBS_TIMING
    {
      __builtin_amdgcn_sched_barrier(0);
      const unsigned long long t3 = LBS_NOW();
      LBS_T(0, t1 - t0); LBS_T(1, t2 - t1); LBS_T(2, t3 - t2); LBS_T(3, 1);
    }
#endif
  }
}

// Mixed blend (mode 3, see M4_BASE_PIECES): nine stages per item - the precise k-step 0, seven stages of four fp16 k-steps, the
// precise k-step 29 - each a burst (this wave's share of the stage's base pieces + its own feature pieces), s_waitcnt, the base
// pieces through the two-deep LDS ring, one barrier, then only MFMAs (the in-flight-load hazard of lbs_blend_f32).
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
constexpr int M4_RING_PIECES = 3 * M4_FKS;      // 12 KiB per ring slot (a precise stage uses 6)
static_assert(28 % M4_FKS == 0, "fp16 stages cover k-steps 1..28 exactly");

// Operand registers of one stage: this wave's share of the stage's base pieces (on their way to the LDS ring) and its own
// feature pieces.
template <int NB>
struct M4Regs {
  bf16x8 ga[(3 * M4_FKS + 3) / 4];
  bf16x8 b[M4_FKS][NB];
};
constexpr int M4_STAGES = 2 + 28 / M4_FKS;     // precise k-step 0, the fp16 stages, precise k-step 29
__device__ __forceinline__ constexpr bool m4_precise(int st) { return st == 0 || st == M4_STAGES - 1; }
__device__ __forceinline__ constexpr int m4_base0(int st) { return st == 0 ? 0 : (st == M4_STAGES - 1 ? 90 : 6 + (st - 1) * 3 * M4_FKS); }
__device__ __forceinline__ constexpr int m4_feat0(int st) { return st == 0 ? 0 : (st == M4_STAGES - 1 ? 30 : 2 + (st - 1) * M4_FKS); }

// The nine stages of an item as a software pipeline of depth one: the burst of stage st + 1 is issued as soon as stage st's
// operands have arrived - before stage st's barrier and MFMAs - so a stage costs max(operand latency, LDS + barrier + MFMAs)
// instead of their sum.  (The in-flight-load hazard of lbs_blend_f32 halves the MFMA rate of this wave meanwhile; here the
// MFMAs are a quarter of the GEMM half - 204 per item - and the operand latency, bases streaming from the Infinity Cache, is
// what the half waits for: 0.414 ms with the epilogue skipped against 0.10 ms of matrix time, profiles/r05_lbs_mixed.md.)
template <int NB>
__device__ __forceinline__ void lbs_blend_mixed(const LbsParams& p, f32x16 (&acc)[3][NB], int vt, int bt0, int lane, int wave, bf16x8* sA,
                                                unsigned long long* tacc) {
  const int num_bt = (p.B + 31) >> 5;
  asm volatile("" : "+v"(lane));   // per-lane operand addresses are formed per item
  const bf16x8* dpv = p.dirs4 + (size_t)vt * M4_BASE_PIECES * 64 + lane;
  const bf16x8* fq[NB];
#pragma unroll
  for (int q = 0; q < NB; ++q) fq[q] = p.feat4 + (size_t)min(bt0 + q, num_bt - 1) * M4_FEAT_PIECES * 64 + lane;
  M4Regs<NB> R[2];
  auto issue = [&](M4Regs<NB>& r, int st) {
    const int np = m4_precise(st) ? 6 : 3 * M4_FKS, nf = m4_precise(st) ? 2 : M4_FKS;
#pragma unroll
    for (int i = 0; i < (3 * M4_FKS + 3) / 4; ++i) {
      const int piece = wave + 4 * i;
      if (piece < np) r.ga[i] = dpv[(size_t)(m4_base0(st) + piece) * 64];
    }
#pragma unroll
    for (int f = 0; f < M4_FKS; ++f)
      if (f < nf) {
#pragma unroll
        for (int q = 0; q < NB; ++q) r.b[f][q] = fq[q][(size_t)(m4_feat0(st) + f) * 64];
      }
  };
  issue(R[0], 0);
#pragma unroll
  for (int st = 0; st < M4_STAGES; ++st) {
    M4Regs<NB>& r = R[st & 1];
    bf16x8* buf = sA + (st & 1) * M4_RING_PIECES * 64;
    [[maybe_unused]] const unsigned long long t0 = LBS_NOW();
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // stage st's operands (issued a stage ago)
    __builtin_amdgcn_sched_barrier(0);
    [[maybe_unused]] const unsigned long long t1 = LBS_NOW();
    const int np = m4_precise(st) ? 6 : 3 * M4_FKS;
#pragma unroll
    for (int i = 0; i < (3 * M4_FKS + 3) / 4; ++i) {
      const int piece = wave + 4 * i;
      if (piece < np) buf[piece * 64 + lane] = r.ga[i];
    }
    if (st + 1 < M4_STAGES) issue(R[(st + 1) & 1], st + 1);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();  // stage visible; also: everyone is done reading the other ring slot's previous contents
    [[maybe_unused]] const unsigned long long t2 = LBS_NOW();
    if (m4_precise(st)) {
      bf16x8 a[2][3];   // [plane][coord]
#pragma unroll
      for (int pl = 0; pl < 2; ++pl)
#pragma unroll
        for (int c = 0; c < 3; ++c) a[pl][c] = buf[(pl * 3 + c) * 64 + lane];
#pragma unroll
      for (int pr = 0; pr < 3; ++pr) {   // hi.mid, mid.hi, hi.hi: small partial products first, product-major
        const int pa = (pr == 1) ? 1 : 0, pb = (pr == 0) ? 1 : 0;
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
          for (int q = 0; q < NB; ++q)
            acc[c][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[pa][c], r.b[pb][q], acc[c][q], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int ks = 0; ks < M4_FKS; ++ks) {
        bf16x8 a[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) a[c] = buf[(ks * 3 + c) * 64 + lane];
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
          for (int q = 0; q < NB; ++q)
            acc[c][q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[c]), __builtin_bit_cast(f16x8, r.b[ks][q]),
                                                               acc[c][q], 0, 0, 0);
      }
    }
#ifdef EGX_LBS_TIMING
    {
      __builtin_amdgcn_sched_barrier(0);
      const unsigned long long t3 = LBS_NOW();
      LBS_T(0, t1 - t0); LBS_T(1, t2 - t1); LBS_T(2, t3 - t2); LBS_T(3, 1);
    }
#endif
  }
}

// The two-plane split (mode 2 arithmetic) for the tiles that hold picked vertices, on the small LDS ring of the three-workgroups-
// per-CU kernel: stages of two k-steps (12 base pieces: 2 k-steps x 2 planes x 3 coordinates), burst -> wait -> ring -> barrier ->
// 3 products per k-step.  8 of ~320 tiles: simple, not pipelined.
template <int NB>
__device__ __forceinline__ void lbs_blend_split2_small(const LbsParams& p, f32x16 (&acc)[3][NB], int vt, int bt0, int lane, int wave,
                                                       bf16x8* sA) {
  const int num_bt = (p.B + 31) >> 5;
  asm volatile("" : "+v"(lane));
  const bf16x8* dpv = p.dirs3 + (size_t)vt * KS3 * 9 * 64 + lane;  // piece (s, plane, coord) at ((s*3 + plane)*3 + coord)*64
  const bf16x8* fq[NB];
#pragma unroll
  for (int q = 0; q < NB; ++q) fq[q] = p.feat3 + (size_t)min(bt0 + q, num_bt - 1) * KS3 * 3 * 64 + lane;
  static_assert(KS3 % 2 == 0, "stages of two k-steps");
  for (int st = 0; st < KS3 / 2; ++st) {
    bf16x8 ga[3], b[2][2][NB];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int piece = wave + 4 * i;                 // (ks, plane, coord) of the stage: ks = piece / 6, plane = piece % 6 / 3
      ga[i] = dpv[(size_t)((st * 2 + piece / 6) * 9 + (piece % 6)) * 64];
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl)
#pragma unroll
        for (int q = 0; q < NB; ++q) b[ks][pl][q] = fq[q][((st * 2 + ks) * 3 + pl) * 64];
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    bf16x8* buf = sA + (st & 1) * M4_RING_PIECES * 64;
#pragma unroll
    for (int i = 0; i < 3; ++i) buf[(wave + 4 * i) * 64 + lane] = ga[i];
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 a[2][3];
#pragma unroll
      for (int pl = 0; pl < 2; ++pl)
#pragma unroll
        for (int c = 0; c < 3; ++c) a[pl][c] = buf[((ks * 2 + pl) * 3 + c) * 64 + lane];
#pragma unroll
      for (int pr = 0; pr < 3; ++pr) {
        const int pa = (pr == 1) ? 1 : 0, pb = (pr == 0) ? 1 : 0;
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
          for (int q = 0; q < NB; ++q)
            acc[c][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[pa][c], b[ks][pb][q], acc[c][q], 0, 0, 0);
      }
    }
  }
}

// LDS of the fused3 kernels: [operand ring][tile metadata 7424 B][4 x (counters 256 B + queue)].  NBW = 32-body tiles per wave:
// 2 = the round-2..4 shape (a workgroup item = 32 vertices x 256 bodies, two workgroups per CU, 256 registers per wave);
// 1 (mixed blend only, round 5) = 32 vertices x 128 bodies, THREE workgroups per CU: 48 accumulators instead of 96 fit a wave in
// 168 registers, and the third wave per SIMD is what the latency chain of this kernel was missing - one workgroup per CU runs
// the launch in 1.07 ms, two in 0.70 (profiles/r05_lbs_mixed.md section 5).
template <int NBW> constexpr int lbs3_ring_bytes() { return NBW == 1 ? 2 * M4_RING_PIECES * 1024 : 2 * 18 * 1024; }
template <int NBW> constexpr size_t lbs3_lds_bytes() { return (size_t)lbs3_ring_bytes<NBW>() + 7424 + 4 * lbs3_wave_bytes<NBW>(); }
static_assert(lbs3_lds_bytes<1>() <= 53504, "three workgroups of the small wave tile share a CU's LDS: not above round 5's size");

template <int NPL, bool DO_SDF, int NBW = LBS_NB>
__global__ __launch_bounds__(256, NBW == 1 ? 3 : 2) void egx_lbs_fused3_kernel(LbsParams p) {
  constexpr int NB = NBW;
  static_assert(NBW == LBS_NB || NPL == 4, "the small wave tile exists for the mixed blend only");
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // wave id: an SGPR
  bf16x8* sA = reinterpret_cast<bf16x8*>(smem_raw);
  char* meta = smem_raw + lbs3_ring_bytes<NBW>();
  char* my = meta + 7424 + wave * lbs3_wave_bytes<NBW>();
  LbsWave w;
  w.lane = lane; w.n = lane & 31; w.half = lane >> 5;
  w.s_W = reinterpret_cast<float*>(meta);                    // tile metadata is shared by the four waves here
  w.s_jl = reinterpret_cast<int*>(meta + NJ * 32 * 4);
  w.s_slot = w.s_jl + 56;
  w.s_masks = reinterpret_cast<unsigned*>(w.s_slot + 32);
  w.s_cnt = reinterpret_cast<int*>(my);
  w.s_fixmap = reinterpret_cast<unsigned*>(my + 128 * NBW);
  w.s_thr = reinterpret_cast<float*>(my + 256 * NBW);
  w.s_queue = reinterpret_cast<f32x4*>(my + 384 * NBW);
  w.lds = nullptr;
  w.qn = 0;
  if (lane < 32 * NBW) { w.s_cnt[lane] = 0; w.s_fixmap[lane] = 0u; }
  // Work partition over the XCDs (blocks are dealt to them round-robin).  Either every XCD owns a chunk of BODY GROUPS and
  // all vertex tiles (its bodies' features / transforms stay in its L2 and the bases stream through once per block of groups)
  // or a chunk of VERTEX TILES and all groups (it streams an eighth of the bases once per block; every XCD reads all
  // features).  The bases traffic is the same either way; what differs is the balance: 20 groups (256 agents) deal 3 / 2 over
  // the XCDs, 10 groups (128) deal 2 / 1, 5 groups (64 agents, the 8-way split) leave three XCDs idle - so the partition with
  // the shorter per-workgroup item count is taken, body groups on a tie (less feature traffic).
  int bg_lo, nper, n_streams, stream, vt_lo = 0, nvt = p.n_tiles;
  if ((gridDim.x & 7) == 0) {
    const int xcd = blockIdx.x & 7;
    n_streams = gridDim.x >> 3;
    stream = blockIdx.x >> 3;
    const int per_g = (p.nbg + 7) / 8, per_t = (p.n_tiles + 7) / 8;
    const int span_g = (per_g * p.n_tiles + n_streams - 1) / n_streams, span_t = (per_t * p.nbg + n_streams - 1) / n_streams;
    if (span_g <= span_t) {
      bg_lo = xcd * per_g;
      nper = max(0, min(per_g, p.nbg - bg_lo));
    } else {
      vt_lo = xcd * per_t;
      nvt = max(0, min(per_t, p.n_tiles - vt_lo));
      bg_lo = 0; nper = p.nbg;
    }
  } else {
    bg_lo = 0; nper = p.nbg;
    n_streams = gridDim.x;
    stream = blockIdx.x;
  }
  const int n_items = nvt * nper;
  // item order: blocks of bg_block body groups, vertex-tile-major inside a block - the features / joint transforms of
  // a block (1.4 MB per group) stay in the XCD's 4 MiB L2 while the bases stream through once per block
  const int PB = max(1, min(p.bg_block, max(nper, 1)));
  unsigned long long tacc[4] = {0, 0, 0, 0};
  (void)tacc;
  auto run_item = [&](int vti, int bg) {
    [[maybe_unused]] const unsigned long long item_t0 = LBS_NOW();
    const int vt = p.tiles ? p.tiles[vti] : vti;
    const int bt0 = bg * (4 * NB) + wave * NB;   // a body group of this kernel = 4 waves x NB tiles of 32 bodies
    __syncthreads();  // previous item: every wave is done with the metadata and with the stage ring
    const int j_lo = p.tj_off[vt];
    const int JT = p.tj_off[vt + 1] - j_lo;
    // (the thread index goes through an empty asm so that per-lane addresses derived from it are formed here, per item: as
    // invariants of the persistent loop they were kept alive across the whole item and spilled to scratch)
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    for (int idx = tid * 4; idx < JT * 32; idx += 1024)
      *reinterpret_cast<f32x4*>(&w.s_W[idx]) = *reinterpret_cast<const f32x4*>(&p.tj_w[(size_t)j_lo * 32 + idx]);
    if (wave == 0) {
      if (lane < JT) w.s_jl[lane] = p.tj_idx[j_lo + lane];
      const int sl = (lane < 32) ? p.pick_slot[vt * 32 + lane] : -1;
      const int fl = (lane < 32) ? p.vflags[vt * 32 + lane] : 0;
      if (lane < 32) w.s_slot[lane] = sl;
      const unsigned long long mp = __ballot(sl >= 0), ms = __ballot((fl & 3) == 2);
      if (lane == 0) { w.s_masks[0] = (unsigned)mp; w.s_masks[1] = (unsigned)ms; }
    }
    f32x16 acc[3][NB];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int q = 0; q < NB; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][q][r] = 0.f;
    if (!(p.dbg & 2)) {
      if constexpr (NPL == 4) {      // NPL 4 = the mixed blend (mode 3)
        // the tiles that hold the PICKED vertices (markers, vertex joints, landmark corners: the 8 leading tiles of 328) keep the
        // two-plane split: what the environment reads as positions - and differentiates into directions (the eye landmarks are
        // centimetres apart and aim 7 m rays) - stays at the 1e-6 m level; the fp16 product only feeds the penetration COUNT
        if (vti < p.n_precise) {
          if constexpr (NBW == LBS_NB) lbs_blend_split<2>(p, acc, vt, bt0, lane, wave, sA, tacc);
          else lbs_blend_split2_small<NB>(p, acc, vt, bt0, lane, wave, sA);
        } else lbs_blend_mixed<NB>(p, acc, vt, bt0, lane, wave, sA, tacc);
      } else if constexpr (NBW == LBS_NB) lbs_blend_split<NPL>(p, acc, vt, bt0, lane, wave, sA, tacc);
    } else __syncthreads();  // the blend's barriers also publish the metadata
    if (p.dbg & 1) {
      float sum = 0.f;
#pragma unroll
      for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int q = 0; q < NB; ++q)
#pragma unroll
          for (int r = 0; r < 16; ++r) sum += acc[c][q][r];
      if (sum == 123.456f) p.pene[0] = 1;
      return;
    }
    // mixed blend: the tiles that only feed the count (everything after the picked tiles) classify with the cheap product and
    // re-evaluate what it cannot decide
#ifdef EGX_LBS_NOFIX   // development builds: the cheap product decides alone (the round-5 kernel), for A/B timing
    constexpr bool FIX = false;
#else
    constexpr bool FIX = NPL == 4 && DO_SDF;
#endif
#ifdef EGX_LBS_VALU_SKIN   // development builds: the count-only tiles skinned on the VALU as well (fix-up only), for A/B timing
    lbs_epilogue<false, DO_SDF, LBS3_RB, lbs3_qcap<NB>(), NB, FIX>(p, w, acc, vt, bt0, JT, FIX && vti >= p.n_precise);
#else
    // count-only tiles whose joint list fits one k-step (eight joints: 309 of the 328 tiles of the synthetic body) are skinned on
    // the matrix pipe; the tiles with picked vertices (exact positions) and the long lists take the VALU epilogue - the latter with
    // the fix-up band as well, since their blend product is the cheap one too.  The small wave tile (three workgroups per CU, 168
    // registers) has no room for the twelve operands: VALU epilogue throughout.
    if (FIX && NB == LBS_NB && vti >= p.n_precise && JT <= 8) lbs_epilogue_cell<LBS3_RB, lbs3_qcap<NB>(), NB>(p, w, acc, vt, bt0, JT);
    else lbs_epilogue<false, DO_SDF, LBS3_RB, lbs3_qcap<NB>(), NB, FIX>(p, w, acc, vt, bt0, JT, FIX && vti >= p.n_precise);
#endif
#ifdef EGX_LBS_TIMING
    w.et[4] += LBS_NOW() - item_t0; w.et[5] += 1;
#endif
  };
  // one loop for both item sources (the body is inlined once): a culled launch walks this XCD's list of active items
  // (egx_lbs_compact_kernel: an item whose 256 bodies are provably in free space for the whole vertex tile is not on it), dealt
  // round-robin to the XCD's workgroups; otherwise the blocked (tile, body group) order above
  const int* list = p.items ? p.items + (size_t)(blockIdx.x & 7) * p.items_stride : nullptr;
  const int i_lo = list ? (int)(blockIdx.x >> 3) : stream, i_step = list ? (int)(gridDim.x >> 3) : n_streams;
  const int i_hi = list ? p.item_counts[blockIdx.x & 7] : n_items;
  for (int item = i_lo; item < i_hi; item += i_step) {
    int vti, bg;
    if (list) {
      const int code = list[item];
      vti = code / p.nbg;
      bg = code - vti * p.nbg;
    } else {
      const int blk = item / (nvt * PB);
      const int pb = min(PB, nper - blk * PB);
      const int r = item - blk * nvt * PB;
      vti = vt_lo + r / pb;
      bg = bg_lo + blk * PB + r % pb;
    }
    run_item(vti, bg);
  }
#ifdef EGX_LBS_TIMING
  if (lane == 0) {
    for (int i = 0; i < 4; ++i) atomicAdd(&g_lbs_t[i], tacc[i]);
    atomicAdd(&g_lbs_t[9], w.et[0]); atomicAdd(&g_lbs_t[10], w.et[1]); atomicAdd(&g_lbs_t[11], w.et[2]); atomicAdd(&g_lbs_t[12], w.et[3]);
    atomicAdd(&g_lbs_t[13], w.et[4]); atomicAdd(&g_lbs_t[14], w.et[5]);
  }
#endif
}

// The fix-up queue of a mixed-blend launch (lbs_fix_process): one wave per queued vertex, re-evaluated in fp32 and counted.
constexpr int LBS_FIXQ_CAP = 1 << 12;   // entries per sub-queue: 64 x 4096 x 8 bytes = 2 MB of workspace, 25 vertices per body at 10 240 bodies
constexpr int LBS_FIX_BLOCKS = 1024;
static_assert((LBS_FIX_BLOCKS * 4) % LBS_FIX_NQ == 0, "waves of the fix-up kernel per sub-queue");
__global__ __launch_bounds__(256) void egx_lbs_fix_kernel(LbsParams p) {
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = gridDim.x * 4;
  const int sq = wave % LBS_FIX_NQ;   // n_waves is a multiple of LBS_FIX_NQ: a wave stays with one sub-queue
  const int count = min(p.fix_stats[LBS_FIX_CNT0 + 32 * sq], p.fixq_cap);
  for (int i = wave / LBS_FIX_NQ; i < count; i += n_waves / LBS_FIX_NQ) {
    const int2 e = p.fixq[(size_t)sq * p.fixq_cap + i];
    if (e.x < 0) continue;   // wave-uniform
    const int vt = e.x >> 5, row = e.x & 31, slot = e.y;
    const int j_lo = p.tj_off[vt], JT = p.tj_off[vt + 1] - j_lo;
    const float sv = lbs_fix_one<true>(p, lane, vt, row, slot, JT, p.tj_idx + j_lo, p.tj_w + (size_t)j_lo * 32);
    if (lane == 0 && sv < 0.f) {
      const int body = p.agent_of_slot ? p.agent_of_slot[slot / p.fpa] * p.fpa + slot % p.fpa : slot;
      atomicAdd(p.pene + body, 1);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Free-space culling of the SDF work items (training path: split blend modes, picks + penetration counts, no vertex output).
//
// The penetration count needs EVERY vertex of every body (crowd_env_2f.py:165-175), which is what makes the blend GEMM the
// dominant cost - but a vertex can only count where the scene has geometry.  A posed vertex lies in the convex hull of balls
// around the posed joints it is bound to (radii bounded per tile at load, egx_body_model_create), so a (vertex tile, body)
// pair whose hull's bounding box - mapped to voxel coordinates - only covers cells of the free-space pyramid with max < 0
// cannot contribute to the count: trilinear interpolation is a convex combination of the samples a cell's bracket covers.
// A work item (tile x 256 bodies) all of whose bodies pass that test is skipped ENTIRELY (GEMM, skinning, SDF) unless the tile
// holds picked vertices.  The result is bit-identical to the unculled launch (tests/test_lbs_gpu.py); what changes is how much
// of the scene-independent work is done.  Three small launches in front of the fused kernel:
//   egx_lbs_agent_order_kernel  agents whose neighbourhood is free first: bodies near geometry share body groups
//   egx_lbs_cull_kernel         the test per (tile, body), OR-reduced per item
//   egx_lbs_compact_kernel      per-XCD item lists (non-picked tiles dealt by tile chunk: an XCD streams an eighth of the bases)
// ------------------------------------------------------------------------------------------------
namespace {
constexpr int CULL_TILES_PER_BLOCK = 16;
constexpr float CULL_SLACK_M = 2e-3f;        // metres added to every radius: covers the fp32 / bf16x2 evaluation of the vertex
constexpr float CULL_SLACK_VOX = 0.02f;      // voxels added to the box: covers the rounding of the affine map

// raw (unclamped) voxel-coordinate box [lo, hi] -> true if every point in it interpolates to a value < 0 (free space)
__device__ __forceinline__ bool cull_box_free(const SdfDev& s, const float* __restrict__ mips, const float (&lo)[3], const float (&hi)[3]) {
  if (!(lo[0] <= hi[0] && lo[1] <= hi[1] && lo[2] <= hi[2])) return false;   // NaN / inverted: not provable
  const int cdim[3] = {s.c0, s.c1, s.c2};
  int jl[3], jh[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    jl[a] = (int)__builtin_amdgcn_fmed3f(floorf(fmaf(lo[a], 0.25f, 1.f)), 0.f, (float)(cdim[a] + 1));
    jh[a] = (int)__builtin_amdgcn_fmed3f(floorf(fmaf(hi[a], 0.25f, 1.f)), 0.f, (float)(cdim[a] + 1));
  }
  for (int l = 0; l <= EGX_SDF_MIP_LEVELS; ++l) {
    if ((jh[0] >> l) - (jl[0] >> l) > 1 || (jh[1] >> l) - (jl[1] >> l) > 1 || (jh[2] >> l) - (jl[2] >> l) > 1) continue;
    const int e1 = l == 0 ? s.c1 + 2 : egx_sdf_mip_dim(s.c1, l), e2 = l == 0 ? s.c2 + 2 : egx_sdf_mip_dim(s.c2, l);
    const float* mp = l == 0 ? nullptr : mips + egx_sdf_mip_offset(s.c0, s.c1, s.c2, l);
    float mx = -3.4e38f;
    for (int x = jl[0] >> l; x <= jh[0] >> l; ++x)
      for (int y = jl[1] >> l; y <= jh[1] >> l; ++y)
        for (int z = jl[2] >> l; z <= jh[2] >> l; ++z) {
          const size_t idx = ((size_t)x * e1 + y) * e2 + z;
          mx = fmaxf(mx, l == 0 ? s.coarse[idx].y : mp[idx]);
        }
    return mx < 0.f;
  }
  return false;   // larger than two cells of the coarsest level
}

// canonical frame -> raw voxel coordinates of an agent: r = Mw x + tw (the affine map of the SDF epilogue)
__device__ __forceinline__ void cull_agent_map(const SdfDev& s, const float* R0, const float* T0, int ag, float (&Mw)[9], float (&tw)[3], float (&kk)[3]) {
  kk[0] = s.scale * (float)s.d0 * 0.5f; kk[1] = s.scale * (float)s.d1 * 0.5f; kk[2] = s.scale * (float)s.d2 * 0.5f;
  const float cc[3] = {s.cx, s.cy, s.cz};
  const float dd[3] = {(float)s.d0, (float)s.d1, (float)s.d2};
#pragma unroll
  for (int a = 0; a < 3; ++a) {
#pragma unroll
    for (int e = 0; e < 3; ++e) Mw[a * 3 + e] = kk[a] * (R0 ? R0[(size_t)ag * 9 + a * 3 + e] : ((a == e) ? 1.f : 0.f));
    tw[a] = kk[a] * ((T0 ? T0[(size_t)ag * 3 + a] : 0.f) - cc[a]) + (dd[a] - 1.f) * 0.5f;
  }
}

// One block.  (1) clears the item flags and counters of this launch; (2) classifies every agent: "far" = the 1 m cube around
// the pelvis of its first and last frame is free space; (3) slot order = far agents, then near agents (stable).
__global__ __launch_bounds__(256) void egx_lbs_agent_order_kernel(const float* __restrict__ xb, const float* __restrict__ R0,
                                                                  const float* __restrict__ T0, SdfDev sdf, const float* __restrict__ mips,
                                                                  int A, int fpa, float px, float py, float pz,
                                                                  int* __restrict__ agent_of_slot, int* __restrict__ flags, int n_flags,
                                                                  int* __restrict__ counts) {
  extern __shared__ int s_key[];   // [A]
  const int tid = threadIdx.x;
  for (int i = tid; i < n_flags; i += 256) flags[i] = 0;
  if (tid < 16) counts[tid] = tid == 15 ? 0x43554c4c : 0;   // [15]: marks the workspace as holding a culled launch's counters
  for (int a = tid; a < A; a += 256) {
    float Mw[9], tw[3], kk[3];
    cull_agent_map(sdf, R0, T0, a, Mw, tw, kk);
    bool far = true;
    for (int f = 0; f < fpa; f += max(1, fpa - 1)) {   // first and last frame
      const float* x = xb + ((size_t)a * fpa + f) * EGX_XB_DIM;
      const float c[3] = {x[0] + px, x[1] + py, x[2] + pz};
      float lo[3], hi[3];
#pragma unroll
      for (int ax = 0; ax < 3; ++ax) {
        const float r = fmaf(Mw[ax * 3 + 0], c[0], fmaf(Mw[ax * 3 + 1], c[1], fmaf(Mw[ax * 3 + 2], c[2], tw[ax])));
        lo[ax] = r - 1.0f * kk[ax]; hi[ax] = r + 1.0f * kk[ax];
      }
      far = far && cull_box_free(sdf, mips, lo, hi);
    }
    s_key[a] = far ? 0 : 1;
  }
  __syncthreads();
  if (tid < 64) {   // stable partition by one wave: 64 agents per step
    int base = 0;
    for (int pass = 0; pass < 2; ++pass)
      for (int a0 = 0; a0 < A; a0 += 64) {
        const int a = a0 + tid;
        const bool mine = a < A && s_key[a] == pass;
        const unsigned long long bm = __ballot(mine);
        if (mine) agent_of_slot[base + __builtin_amdgcn_mbcnt_hi((unsigned)(bm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bm, 0u))] = a;
        base += __popcll(bm);
      }
  }
}

// grid (nbg, tile chunks), 256 threads = the 256 slots of a body group.  flags[ti * nbg + bg] = 1 where some body of the
// group cannot be proven clear of geometry for tile ti of the launch's tile list (ti >= first_tile: the picked tiles in
// front are always evaluated).
__global__ __launch_bounds__(256) void egx_lbs_cull_kernel(const int* __restrict__ tiles, int first_tile, int n_tiles,
                                                           const int* __restrict__ tj_off, const int* __restrict__ tj_idx,
                                                           const float* __restrict__ D0, const float* __restrict__ E,
                                                           const float* __restrict__ fvec, const float* __restrict__ jpos, int Bp,
                                                           const int* __restrict__ agent_of_slot, int B, int fpa, int nbg,
                                                           const float* __restrict__ R0, const float* __restrict__ T0, SdfDev sdf,
                                                           const float* __restrict__ mips, int* __restrict__ flags) {
  const int bg = blockIdx.x, tid = threadIdx.x;
  const int slot = bg * 256 + tid;
  const bool valid = slot < B;
  const int ss = valid ? slot : B - 1;
  const int body = agent_of_slot ? agent_of_slot[ss / fpa] * fpa + ss % fpa : ss;
  float Mw[9], tw[3], kk[3];
  cull_agent_map(sdf, R0, T0, body / fpa, Mw, tw, kk);
  float f[61];
#pragma unroll
  for (int i = 0; i < 61; ++i) f[i] = fvec[(size_t)i * Bp + ss];
  const int t_lo = first_tile + blockIdx.y * CULL_TILES_PER_BLOCK, t_hi = min(n_tiles, t_lo + CULL_TILES_PER_BLOCK);
  for (int ti = t_lo; ti < t_hi; ++ti) {
    const int vt = __builtin_amdgcn_readfirstlane(tiles ? tiles[ti] : ti);   // uniform: the tables below are read with scalar loads
    const float* Et = E + (size_t)vt * 64;
    float margin = CULL_SLACK_M;
#pragma unroll
    for (int i = 0; i < 61; ++i) margin = fmaf(f[i], Et[i], margin);
    float lo[3] = {3.4e38f, 3.4e38f, 3.4e38f}, hi[3] = {-3.4e38f, -3.4e38f, -3.4e38f};
    const int jj_lo = __builtin_amdgcn_readfirstlane(tj_off[vt]), jj_hi = __builtin_amdgcn_readfirstlane(tj_off[vt + 1]);
    for (int jj = jj_lo; jj < jj_hi; ++jj) {
      const int j = __builtin_amdgcn_readfirstlane(tj_idx[jj]) & 0xff;
      const float rho = (D0[jj] + margin) * 1.0001f;
      const float c0 = jpos[(size_t)(j * 3 + 0) * Bp + ss], c1 = jpos[(size_t)(j * 3 + 1) * Bp + ss], c2 = jpos[(size_t)(j * 3 + 2) * Bp + ss];
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const float r = fmaf(Mw[a * 3 + 0], c0, fmaf(Mw[a * 3 + 1], c1, fmaf(Mw[a * 3 + 2], c2, tw[a])));
        const float ext = rho * kk[a] + CULL_SLACK_VOX + 1e-5f * fabsf(r);
        lo[a] = fminf(lo[a], r - ext); hi[a] = fmaxf(hi[a], r + ext);
      }
    }
    const bool active = valid && !cull_box_free(sdf, mips, lo, hi);
    if (__ballot(active) != 0ull && (tid & 63) == 0) flags[(size_t)ti * nbg + bg] = 1;
  }
}

// 8 blocks of one wave: block x builds the item list of XCD x.  Picked tiles (ti < first_tile, always active) go to XCD
// bg % 8; the other tiles are dealt round-robin (an XCD streams only an eighth of the bases), in the order
// "block of bg_block body groups, tile, group of the block" so that the features of a block stay in the XCD's L2.
__global__ __launch_bounds__(64) void egx_lbs_compact_kernel(const int* __restrict__ flags, int first_tile, int n_tiles, int nbg, int bg_block,
                                                             int* __restrict__ items, int items_stride, int* __restrict__ counts) {
  const int x = blockIdx.x, lane = threadIdx.x;
  int* list = items + (size_t)x * items_stride;
  int n = 0;
  auto append = [&](bool on, int code) {
    const unsigned long long bm = __ballot(on);
    if (on) list[n + __builtin_amdgcn_mbcnt_hi((unsigned)(bm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bm, 0u))] = code;
    n += __popcll(bm);
  };
  const int n_mine = (nbg - x + 7) / 8;   // body groups x, x + 8, ...
  for (int i0 = 0; i0 < n_mine * first_tile; i0 += 64) {
    const int i = i0 + lane;
    const bool on = i < n_mine * first_tile;
    const int bg = x + 8 * (on ? i / first_tile : 0), ti = on ? i % first_tile : 0;
    append(on, ti * nbg + bg);
  }
  // tiles first_tile + x, + 8, ...: the active tiles of standing bodies are neighbours in the joint-sorted tile order (the
  // legs), so contiguous chunks would leave most XCDs idle
  const int n_np = n_tiles - first_tile;
  const int t_n = max(0, (n_np - x + 7) / 8);
  const int PB = max(1, bg_block), n_blk = (nbg + PB - 1) / PB;
  const int total = n_blk * t_n * PB;
  for (int i0 = 0; i0 < total; i0 += 64) {
    const int i = i0 + lane;
    bool on = i < total;
    const int blk = on ? i / (t_n * PB) : 0, r = on ? i % (t_n * PB) : 0;
    const int ti = first_tile + x + 8 * (r / PB), bg = blk * PB + r % PB;
    on = on && bg < nbg && flags[(size_t)ti * nbg + bg] != 0;
    append(on, ti * nbg + bg);
  }
  if (lane == 0) { counts[x] = n; atomicAdd(&counts[8], n); }
}
}  // namespace

// ------------------------------------------------------------------------------------------------
// kernel 3: assemble joints[55..126] and markers from the picked vertices
// ------------------------------------------------------------------------------------------------
__global__ void egx_gather_kernel(const float* __restrict__ picked, int B, int NP, int M,
                                  const int* __restrict__ marker_slot, const int* __restrict__ extra_slot,
                                  const int* __restrict__ lmk_slot, const float* __restrict__ lmk_bary,
                                  float* __restrict__ out_joints, float* __restrict__ out_markers) {
  const int b = blockIdx.x;
  const float* pk = picked + (size_t)b * NP * 3;
  for (int i = threadIdx.x; i < M + NEXTRA + NLMK; i += blockDim.x) {
    if (i < M) {
      if (out_markers) {
        const float* s = pk + marker_slot[i] * 3;
        float* o = out_markers + ((size_t)b * M + i) * 3;
        o[0] = s[0]; o[1] = s[1]; o[2] = s[2];
      }
    } else if (out_joints) {
      float* o = out_joints + ((size_t)b * EGX_NUM_JOINTS_OUT + NJ + (i - M)) * 3;
      if (i < M + NEXTRA) {
        const float* s = pk + extra_slot[i - M] * 3;
        o[0] = s[0]; o[1] = s[1]; o[2] = s[2];
      } else {
        const int l = i - M - NEXTRA;
        float acc[3] = {0.f, 0.f, 0.f};
        for (int k = 0; k < 3; ++k) {
          const float* s = pk + lmk_slot[l * 3 + k] * 3;
          const float w = lmk_bary[l * 3 + k];
          acc[0] += s[0] * w; acc[1] += s[1] * w; acc[2] += s[2] * w;
        }
        o[0] = acc[0]; o[1] = acc[1]; o[2] = acc[2];
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// host side: packing + C ABI
// ------------------------------------------------------------------------------------------------
template <typename T>
static int upload(T** dptr, const std::vector<T>& h) {
  EGX_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(dptr), std::max<size_t>(h.size(), 1) * sizeof(T)));
  if (!h.empty()) EGX_HIP_CHECK(hipMemcpy(*dptr, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
  return EGX_OK;
}

extern "C" int egx_body_model_create(const egx_body_model_host* d, egx_body_model** out) {
  EGX_REQUIRE(d && out, "null argument");
  EGX_REQUIRE(d->num_verts > 0 && d->num_markers >= 0, "bad sizes");
  EGX_REQUIRE(d->v_template_host && d->shapedirs_host && d->posedirs_host && d->J_regressor_host && d->parents_host &&
                  d->lbs_weights_host && d->hand_comps_l_host && d->hand_comps_r_host && d->hand_mean_l_host &&
                  d->hand_mean_r_host && d->extra_vids_host && d->lmk_vids_host && d->lmk_bary_host,
              "null model array");
  const int V = d->num_verts, NVT = egx_ceil_div(V, 32), VP = NVT * 32;
  auto* m = new egx_body_model();
  m->V = V; m->NVT = NVT; m->M = d->num_markers;

  // Joint-coherent vertex order: vertices are sorted by the set of joints they are bound to, so that the 32 vertices
  // of a tile share few joints and the skinning loop of the epilogue (one pass per joint of the tile) stays short
  // (synthetic body: 9.8 -> 4.7 joints per tile).  perm[new] = original vertex id (-1 for the padding rows); every
  // table below is laid out in the new order, outputs are addressed through `vorig` / pick slots.
  std::vector<int> perm(VP, -1);
  {
    std::vector<std::vector<int>> key(V);
    for (int v = 0; v < V; ++v)
      for (int j = 0; j < NJ; ++j)
        if (d->lbs_weights_host[(size_t)v * NJ + j] != 0.f) key[v].push_back(j);
    std::vector<int> order(V);
    for (int v = 0; v < V; ++v) order[v] = v;
    bool natural = false;
#ifdef EGX_LBS_DEVELOPMENT
    if (const char* e = getenv("EGX_LBS_VERTEX_ORDER")) natural = std::string(e) == "natural";
#endif
    // The vertices the environment reads (markers, vertex joints, landmark corners: 241 of 10 475) come FIRST, packed into
    // ceil(241 / 32) = 8 tiles: a call that asks for neither all vertices nor SDF counts (box / crowd / EgoBody scenes, reset
    // tables, get_jts / get_markers) then evaluates 8 vertex tiles instead of every tile that happens to hold one of them
    // (~half of the 328 under the joint order alone).  Both parts keep the joint-coherent order among themselves.
    std::vector<char> is_pick(V, 0);
    auto mark = [&](const int* ids, int n) {
      for (int i = 0; i < n; ++i)
        if (ids[i] >= 0 && ids[i] < V) is_pick[ids[i]] = 1;   // out-of-range ids are reported below (slot_of)
    };
    if (d->marker_vids_host) mark(d->marker_vids_host, d->num_markers);
    mark(d->extra_vids_host, NEXTRA);
    mark(d->lmk_vids_host, NLMK * 3);
    if (!natural)
      std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
        if (is_pick[a] != is_pick[b]) return is_pick[a] > is_pick[b];
        return key[a] < key[b];
      });
    for (int v = 0; v < V; ++v) perm[v] = order[v];
  }

  // blend bases in MFMA A-operand order: [vt][g][c][lane] float4, element e <-> k = 2*(4g+e) + (lane>>5)
  std::vector<f32x4> dirs((size_t)NVT * KGROUPS * 3 * 64);
  for (int vt = 0; vt < NVT; ++vt)
    for (int g = 0; g < KGROUPS; ++g)
      for (int c = 0; c < 3; ++c)
        for (int l = 0; l < 64; ++l) {
          const int v = perm[vt * 32 + (l & 31)];
          f32x4 val = {0.f, 0.f, 0.f, 0.f};
          if (v >= 0)
            for (int e = 0; e < 4; ++e) {
              const int k = 2 * (4 * g + e) + (l >> 5);
              if (k < 10) {
                val[e] = d->shapedirs_host[((size_t)v * 3 + c) * 10 + k];
              } else if (k < KACT) {
                const int jc = (k - 10) / 9, e9 = (k - 10) % 9;
                const int j = jc + 1 + (jc >= 21 ? 3 : 0);        // inverse of egx_compact_joint
                val[e] = d->posedirs_host[(size_t)((j - 1) * 9 + e9) * 3 * V + (size_t)v * 3 + c];
              } else if (k == KACT) {
                val[e] = d->v_template_host[(size_t)v * 3 + c];  // multiplied by the constant-1 feature
              }
            }
          dirs[(((size_t)vt * KGROUPS + g) * 3 + c) * 64 + l] = val;
        }

  // the same bases vertex-major (one vertex's 3 x 472 columns contiguous): the single-vertex fp32 re-evaluation of the mixed blend
  // reads 5.7 KB per vertex from here instead of 720 scattered cache lines of the MFMA-ordered images
  std::vector<float> dirs_rm((size_t)VP * 3 * KDIM, 0.f);
  for (int vn = 0; vn < VP; ++vn) {
    const int v = perm[vn];
    if (v < 0) continue;
    for (int c = 0; c < 3; ++c) {
      float* dst = &dirs_rm[((size_t)vn * 3 + c) * KDIM];
      for (int k = 0; k < 10; ++k) dst[k] = d->shapedirs_host[((size_t)v * 3 + c) * 10 + k];
      for (int k = 10; k < KACT; ++k) {
        const int jc = (k - 10) / 9, e9 = (k - 10) % 9;
        const int j = jc + 1 + (jc >= 21 ? 3 : 0);
        dst[k] = d->posedirs_host[(size_t)((j - 1) * 9 + e9) * 3 * V + (size_t)v * 3 + c];
      }
      dst[KACT] = d->v_template_host[(size_t)v * 3 + c];
    }
  }

  // the same bases as three bf16 planes in the A-operand order of v_mfma_f32_32x32x16_bf16:
  // [vt][s][plane][c][lane] 8 x bf16, element e <-> k = 16 s + 8 (lane>>5) + e, row = lane & 31
  std::vector<unsigned short> dirs3((size_t)NVT * KS3 * 9 * 64 * 8, 0);
  for (int vt = 0; vt < NVT; ++vt)
    for (int sidx = 0; sidx < KS3; ++sidx)
      for (int c = 0; c < 3; ++c)
        for (int l = 0; l < 64; ++l) {
          const int v = perm[vt * 32 + (l & 31)];
          if (v < 0) continue;
          for (int e = 0; e < 8; ++e) {
            const int k = 16 * sidx + 8 * (l >> 5) + e;
            float val = 0.f;
            if (k < 10) {
              val = d->shapedirs_host[((size_t)v * 3 + c) * 10 + k];
            } else if (k < KACT) {
              const int jc = (k - 10) / 9, e9 = (k - 10) % 9;
              const int j = jc + 1 + (jc >= 21 ? 3 : 0);
              val = d->posedirs_host[(size_t)((j - 1) * 9 + e9) * 3 * V + (size_t)v * 3 + c];
            } else if (k == KACT) {
              val = d->v_template_host[(size_t)v * 3 + c];
            } else if (k == KACT + 1) {  // what two bf16 terms of the template leave over (feature 470 selects it)
              const float t = d->v_template_host[(size_t)v * 3 + c];
              unsigned short ht[3];
              egx_bf16_split3(t, ht);
              val = (t - egx_bf16_to_f32(ht[0])) - egx_bf16_to_f32(ht[1]);
            }
            unsigned short h[3];
            egx_bf16_split3(val, h);
            for (int pl = 0; pl < 3; ++pl)
              dirs3[(((((size_t)vt * KS3 + sidx) * 3 + pl) * 3 + c) * 64 + l) * 8 + e] = h[pl];
          }
        }

  // the mixed-blend image (mode 3): the same values, k-steps 0 and 29 as two bf16 planes, k-steps 1..28 as one fp16 plane
  std::vector<unsigned short> dirs4((size_t)NVT * M4_BASE_PIECES * 64 * 8, 0);
  for (int vt = 0; vt < NVT; ++vt)
    for (int sidx = 0; sidx < KS3; ++sidx)
      for (int c = 0; c < 3; ++c)
        for (int l = 0; l < 64; ++l)
          for (int e = 0; e < 8; ++e) {
            const size_t src = (((((size_t)vt * KS3 + sidx) * 3 + 0) * 3 + c) * 64 + l) * 8 + e;      // plane 0 of dirs3
            const size_t pstride = (size_t)3 * 64 * 8;                                               // plane stride in dirs3
            if (sidx == 0 || sidx == KS3 - 1) {
              for (int pl = 0; pl < 2; ++pl)
                dirs4[(((size_t)vt * M4_BASE_PIECES + egx_m4_base_piece(sidx, pl, c)) * 64 + l) * 8 + e] = dirs3[src + pl * pstride];
            } else {
              const float val = egx_bf16_to_f32(dirs3[src]) + egx_bf16_to_f32(dirs3[src + pstride]) + egx_bf16_to_f32(dirs3[src + 2 * pstride]);
              dirs4[(((size_t)vt * M4_BASE_PIECES + egx_m4_base_piece(sidx, 0, c)) * 64 + l) * 8 + e] = egx_f16_rne(val);
            }
          }

  // skinning weights -> per-tile joint lists: the joints any of the tile's 32 vertices is bound to, with the dense
  // [32] weight column of each.  The epilogue walks this list once per body tile (one transform fetch per joint
  // instead of one per vertex and weight).
  int NW = 1;
  for (int v = 0; v < V; ++v) {
    int c = 0;
    for (int j = 0; j < NJ; ++j) c += d->lbs_weights_host[(size_t)v * NJ + j] != 0.f;
    NW = std::max(NW, c);
  }
  m->NW = NW;
  std::vector<int> tj_off(NVT + 1, 0), tj_idx;
  std::vector<float> tj_w;
  for (int vt = 0; vt < NVT; ++vt) {
    for (int j = 0; j < NJ; ++j) {
      float col[32];
      bool any = false;
      for (int r = 0; r < 32; ++r) {
        const int v = perm[vt * 32 + r];
        col[r] = (v >= 0) ? d->lbs_weights_host[(size_t)v * NJ + j] : 0.f;
        any |= col[r] != 0.f;
      }
      if (any) {
        // bits 8..11: which groups of eight rows (rows 8g .. 8g+7 = the four rows of row group g in both lane halves of the
        // epilogue) hold a non-zero weight of this joint - the epilogue skips the others with a scalar test
        int gmask = 0;
        for (int r = 0; r < 32; ++r) gmask |= (col[r] != 0.f) ? (1 << (r >> 3)) : 0;
        tj_idx.push_back(j | (gmask << 8));
        tj_w.insert(tj_w.end(), col, col + 32);
      }
    }
    if ((int)tj_idx.size() == tj_off[vt]) {  // a tile without any weight still needs one (zero) entry
      tj_idx.push_back(0);
      tj_w.insert(tj_w.end(), 32, 0.f);
    }
    tj_off[vt + 1] = (int)tj_idx.size();
  }

  // matrix-pipe skinning weights of every tile (lbs_epilogue_cell): k-step ks covers joints list[8 ks .. 8 ks + 7] of the tile
  std::vector<int> skin_ks_off(NVT + 1, 0);
  for (int vt = 0; vt < NVT; ++vt) skin_ks_off[vt + 1] = skin_ks_off[vt] + (tj_off[vt + 1] - tj_off[vt] + 7) / 8;
  std::vector<unsigned short> skinW((size_t)skin_ks_off[NVT] * 2 * 64 * 8, 0);
  for (int vt = 0; vt < NVT; ++vt) {
    const int JT = tj_off[vt + 1] - tj_off[vt];
    for (int ks = 0; ks < (JT + 7) / 8; ++ks)
      for (int l = 0; l < 64; ++l)
        for (int e = 0; e < 8; ++e) {
          const int jj = 8 * ks + e;
          if (jj >= JT) continue;
          unsigned short hh[3];
          egx_bf16_split3(tj_w[(size_t)(tj_off[vt] + jj) * 32 + (l & 31)], hh);
          const size_t base = ((size_t)(skin_ks_off[vt] + ks) * 2) * 64 * 8;
          skinW[base + (size_t)l * 8 + e] = hh[0];                                   // operand 0: W_hi in both lane halves
          if (l < 32) skinW[base + (size_t)(64 + l) * 8 + e] = hh[1];                // operand 1: W_mid in lane half 0, zero in half 1
        }
  }

  // picked vertices (markers, vertex joints, landmark corners)
  std::vector<int> pick_slot(VP, -1), marker_slot(d->num_markers), extra_slot(NEXTRA), lmk_slot(NLMK * 3);
  int NP = 0;
  std::vector<int> inv(V);  // original vertex id -> position in the sorted order
  for (int vn = 0; vn < V; ++vn) inv[perm[vn]] = vn;
  auto slot_of = [&](int v) -> int {
    if (v < 0 || v >= V) return -1;
    if (pick_slot[inv[v]] < 0) pick_slot[inv[v]] = NP++;
    return pick_slot[inv[v]];
  };
  bool ok = true;
  for (int i = 0; i < d->num_markers; ++i) ok &= (marker_slot[i] = slot_of(d->marker_vids_host[i])) >= 0;
  for (int i = 0; i < NEXTRA; ++i) ok &= (extra_slot[i] = slot_of(d->extra_vids_host[i])) >= 0;
  for (int i = 0; i < NLMK * 3; ++i) ok &= (lmk_slot[i] = slot_of(d->lmk_vids_host[i])) >= 0;
  if (!ok) { delete m; egx_set_error("vertex id out of range in marker/extra/landmark tables"); return EGX_ERR_ARG; }
  m->NP = NP;
  std::vector<int> pick_tiles;
  for (int vt = 0; vt < NVT; ++vt) {
    bool any = false;
    for (int r = 0; r < 32; ++r) any |= pick_slot[vt * 32 + r] >= 0;
    if (any) pick_tiles.push_back(vt);
  }
  m->n_pick_tiles = (int)pick_tiles.size();
  std::vector<uint8_t> vflags(VP, 0);
  for (int v = 0; v < V; ++v) vflags[v] = 2;
  for (int i = 0; i < d->num_feet; ++i) {
    const int v = d->feet_vids_host[i];
    if (v < 0 || v >= V) { delete m; egx_set_error("feet vertex id out of range"); return EGX_ERR_ARG; }
    vflags[inv[v]] |= 1;
  }
  std::vector<int> sdf_tiles;   // tiles of feet vertices only (and no pick) contribute nothing to a picks + SDF-count call
  for (int vt = 0; vt < NVT; ++vt) {
    bool any = false;
    for (int r = 0; r < 32; ++r) any |= pick_slot[vt * 32 + r] >= 0 || (vflags[vt * 32 + r] & 3) == 2;
    if (any) sdf_tiles.push_back(vt);
  }
  m->n_sdf_tiles = (int)sdf_tiles.size();
  for (int vt : sdf_tiles)
    for (int r = 0; r < 32; ++r) m->verts_sdf_tiles += (vflags[vt * 32 + r] & 2) ? 1 : 0;
  for (int vt : pick_tiles)
    for (int r = 0; r < 32; ++r) m->verts_pick_tiles += (vflags[vt * 32 + r] & 2) ? 1 : 0;
  std::vector<float> lmk_bary(d->lmk_bary_host, d->lmk_bary_host + NLMK * 3);

  // pose constants; joint regression folded through the shape space in double precision:
  //   J(betas) = J_regressor (v_template + shapedirs betas) = J_template + J_shapedirs betas
  std::vector<PoseConsts> pcv(1);
  PoseConsts& pc = pcv[0];
  std::memset(&pc, 0, sizeof(pc));
  pc.max_depth = 0;
  for (int j = 0; j < NJ; ++j) {
    pc.parents[j] = d->parents_host[j];
    if (j > 0 && (pc.parents[j] < 0 || pc.parents[j] >= j)) { delete m; egx_set_error("parents must be topologically ordered"); return EGX_ERR_ARG; }
    pc.depth[j] = (j == 0) ? 0 : pc.depth[pc.parents[j]] + 1;
    pc.max_depth = std::max(pc.max_depth, pc.depth[j]);
  }
  pc.parents[0] = -1;
  for (int j = 0; j < NJ; ++j)
    for (int c = 0; c < 3; ++c) {
      double s = 0.0;
      double sd[10] = {0};
      for (int v = 0; v < V; ++v) {
        const double w = d->J_regressor_host[(size_t)j * V + v];
        if (w == 0.0) continue;
        s += w * d->v_template_host[(size_t)v * 3 + c];
        for (int k = 0; k < 10; ++k) sd[k] += w * d->shapedirs_host[((size_t)v * 3 + c) * 10 + k];
      }
      pc.J_template[j * 3 + c] = (float)s;
      for (int k = 0; k < 10; ++k) pc.J_shapedirs[(j * 3 + c) * 10 + k] = (float)sd[k];
    }
  for (int j = 1; j < NJ; ++j) {   // fix-up threshold of the mixed blend: largest pose-corrective column (3-vector norm) per joint
    double mx = 0.0;
    for (int e9 = 0; e9 < 9; ++e9)
      for (int v = 0; v < V; ++v) {
        double q = 0.0;
        for (int c = 0; c < 3; ++c) {
          const double e = d->posedirs_host[(size_t)((j - 1) * 9 + e9) * 3 * V + (size_t)v * 3 + c];
          q += e * e;
        }
        mx = std::max(mx, q);
      }
    pc.fix_c[j] = (float)(std::sqrt(mx) * (1.0 + 1e-6));
  }
  {
    double mx = 0.0;
    for (int v = 0; v < V; ++v) {
      double q = 0.0;
      for (int c = 0; c < 3; ++c) q += (double)d->v_template_host[(size_t)v * 3 + c] * d->v_template_host[(size_t)v * 3 + c];
      mx = std::max(mx, q);
    }
    pc.v_norm_max = (float)std::sqrt(mx) + 0.25f;
  }
  std::memcpy(pc.hand_comps, d->hand_comps_l_host, 12 * 45 * sizeof(float));
  std::memcpy(pc.hand_comps + 12 * 45, d->hand_comps_r_host, 12 * 45 * sizeof(float));
  std::memcpy(pc.hand_mean, d->hand_mean_l_host, 45 * sizeof(float));
  std::memcpy(pc.hand_mean + 45, d->hand_mean_r_host, 45 * sizeof(float));

  // ---- bounds for the free-space culling of SDF work items.  A posed vertex is a convex combination over its joints j of
  // R_j (v~ - J_j) + p_j (p_j: posed joint, J_j: shaped rest joint, v~: template + shape and pose offsets), so it lies in the
  // convex hull of the balls B(p_j, |v~ - J_j|), and
  //   |v~ - J_j| <= |v_t - J_t,j| + sum_k |beta_k| |S_k,v - JS_k,j| + sum_j' ||R_j' - I||_F ||P_j',v||_F
  // (Cauchy-Schwarz per joint block of the pose blend shapes).  Per tile: D0 = max of the first term over the vertices bound
  // to the joint, E = the maxima of the coefficient norms over the tile's vertices (and its joints, for the shape term).
  std::vector<float> cull_E((size_t)NVT * 64, 0.f), cull_D0(tj_idx.size(), 0.f);
  bool convex = true;
  for (int v = 0; v < V && convex; ++v) {
    double sum = 0.0;
    for (int j = 0; j < NJ; ++j) {
      const float wv = d->lbs_weights_host[(size_t)v * NJ + j];
      if (wv < 0.f) convex = false;
      sum += wv;
    }
    if (std::fabs(sum - 1.0) > 1e-4) convex = false;
  }
  {
    bool lead = pick_tiles.size() <= sdf_tiles.size();
    for (size_t i = 0; i < pick_tiles.size() && lead; ++i) lead = sdf_tiles[i] == pick_tiles[i];
    m->sdf_lead_picks = lead ? 1 : 0;
  }
  // the culled launch treats the first n_pick_tiles entries of the SDF tile list as "always evaluated"
  for (size_t i = 0; i < pick_tiles.size() && convex; ++i) convex = i < sdf_tiles.size() && sdf_tiles[i] == pick_tiles[i];
  m->cull_ok = convex ? 1 : 0;
  for (int c = 0; c < 3; ++c) m->rest_pelvis[c] = pc.J_template[c];
  for (int vt = 0; vt < NVT && convex; ++vt) {
    float* E = &cull_E[(size_t)vt * 64];
    for (int r = 0; r < 32; ++r) {
      const int v = perm[vt * 32 + r];
      if (v < 0) continue;
      for (int jj = tj_off[vt]; jj < tj_off[vt + 1]; ++jj) {
        const int j = tj_idx[jj] & 0xff;
        if (d->lbs_weights_host[(size_t)v * NJ + j] != 0.f) {
          double q = 0.0;
          for (int c = 0; c < 3; ++c) {
            const double e = (double)d->v_template_host[(size_t)v * 3 + c] - (double)pc.J_template[j * 3 + c];
            q += e * e;
          }
          cull_D0[jj] = std::max(cull_D0[jj], (float)(std::sqrt(q) * (1.0 + 1e-6)));
        }
        for (int k = 0; k < 10; ++k) {   // shape term, over every joint of the tile's list (a superset of the vertex's own)
          double q = 0.0;
          for (int c = 0; c < 3; ++c) {
            const double e = (double)d->shapedirs_host[((size_t)v * 3 + c) * 10 + k] - (double)pc.J_shapedirs[(j * 3 + c) * 10 + k];
            q += e * e;
          }
          E[k] = std::max(E[k], (float)(std::sqrt(q) * (1.0 + 1e-6)));
        }
      }
      for (int jc = 0; jc < 51; ++jc) {
        const int j = jc + 1 + (jc >= 21 ? 3 : 0);
        double q = 0.0;
        for (int e9 = 0; e9 < 9; ++e9)
          for (int c = 0; c < 3; ++c) {
            const double e = d->posedirs_host[(size_t)((j - 1) * 9 + e9) * 3 * V + (size_t)v * 3 + c];
            q += e * e;
          }
        E[10 + jc] = std::max(E[10 + jc], (float)(std::sqrt(q) * (1.0 + 1e-6)));
      }
    }
  }

  // The bound is only worth evaluating where it is tight: at a reference pose (|beta_k| = 0.8, ||R_j - I||_F = 0.42, i.e.
  // 0.3 rad at every joint) the blend-shape margin of the median tile must stay below 15 cm.  Learned body models pass (shape
  // directions are smooth fields, pose correctives act near their joint: centimetres); a model whose blend shapes are
  // i.i.d. noise in all 469 x 3V entries - the synthetic benchmark body of SURVEY 8(d) - does not (0.56 m: no box is ever
  // free), and its launches skip the three culling kernels altogether.
  if (convex) {
    std::vector<float> ref(NVT);
    for (int vt = 0; vt < NVT; ++vt) {
      float mg = 0.f;
      for (int i = 0; i < 10; ++i) mg += 0.8f * cull_E[(size_t)vt * 64 + i];
      for (int i = 10; i < 61; ++i) mg += 0.42f * cull_E[(size_t)vt * 64 + i];
      ref[vt] = mg;
    }
    std::nth_element(ref.begin(), ref.begin() + NVT / 2, ref.end());
    m->cull_ref_margin = ref[NVT / 2];
    if (m->cull_ref_margin > 0.15f) m->cull_ok = 0;
  }

  int rc = EGX_OK;
  if ((rc = upload(&m->cull_E, cull_E)) || (rc = upload(&m->cull_D0, cull_D0))) { egx_body_model_destroy(m); return rc; }
  {
    unsigned short* d3 = nullptr;
    if ((rc = upload(&d3, dirs3))) { egx_body_model_destroy(m); return rc; }
    m->dirs3 = reinterpret_cast<bf16x8*>(d3);
    unsigned short* d4 = nullptr;
    if ((rc = upload(&d4, dirs4))) { egx_body_model_destroy(m); return rc; }
    m->dirs4 = reinterpret_cast<bf16x8*>(d4);
  }
  {
    unsigned short* sw = nullptr;
    if ((rc = upload(&sw, skinW)) || (rc = upload(&m->skin_ks_off, skin_ks_off)) || (rc = upload(&m->dirs_rm, dirs_rm))) { egx_body_model_destroy(m); return rc; }
    m->skinW = reinterpret_cast<bf16x8*>(sw);
  }
  if ((rc = upload(&m->vorig, perm)) || (rc = upload(&m->pick_tiles, pick_tiles)) ||
      (rc = upload(&m->sdf_tiles, sdf_tiles))) { egx_body_model_destroy(m); return rc; }
  if ((rc = upload(&m->dirs, dirs)) || (rc = upload(&m->tj_off, tj_off)) || (rc = upload(&m->tj_idx, tj_idx)) ||
      (rc = upload(&m->tj_w, tj_w)) || (rc = upload(&m->pick_slot, pick_slot)) || (rc = upload(&m->vflags, vflags)) ||
      (rc = upload(&m->pc, pcv)) || (rc = upload(&m->marker_slot, marker_slot)) || (rc = upload(&m->extra_slot, extra_slot)) ||
      (rc = upload(&m->lmk_slot, lmk_slot)) || (rc = upload(&m->lmk_bary, lmk_bary))) {
    egx_body_model_destroy(m);
    return rc;
  }
  *out = m;
  return EGX_OK;
}

extern "C" void egx_body_model_destroy(egx_body_model* m) {
  if (!m) return;
  (void)hipFree(m->dirs); (void)hipFree(m->dirs3); (void)hipFree(m->dirs4); (void)hipFree(m->tj_off); (void)hipFree(m->tj_idx); (void)hipFree(m->tj_w);
  (void)hipFree(m->pick_slot); (void)hipFree(m->pick_tiles); (void)hipFree(m->sdf_tiles); (void)hipFree(m->vflags); (void)hipFree(m->vorig); (void)hipFree(m->pc); (void)hipFree(m->marker_slot);
  (void)hipFree(m->extra_slot); (void)hipFree(m->lmk_slot); (void)hipFree(m->lmk_bary);
  (void)hipFree(m->cull_E); (void)hipFree(m->cull_D0);
  (void)hipFree(m->skinW); (void)hipFree(m->skin_ks_off); (void)hipFree(m->dirs_rm);
  delete m;
}

extern "C" int egx_body_model_num_verts(const egx_body_model* m) { return m ? m->V : 0; }
extern "C" int egx_body_model_nnz(const egx_body_model* m) { return m ? m->NW : 0; }
extern "C" int egx_body_model_lbs_vertices(const egx_body_model* m, int with_sdf) {
  return m ? (with_sdf ? m->verts_sdf_tiles : m->verts_pick_tiles) : 0;
}

namespace {
constexpr int kMaxDevices = 64;
struct LbsDeviceInfo {
  std::mutex mu;
  int num_cu = 0;
};
LbsDeviceInfo g_lbs_dev[kMaxDevices];
// blend mode of the fused kernel: 0 = fp32 MFMA, 1 = 3-term bf16 split, 2 = 2-term bf16 split (default); vertex-writing
// calls always use 0
std::atomic<int> g_blend_mode{-1};
int blend_mode() {
  int m = g_blend_mode.load();
  if (m < 0) {
    const char* e = getenv("EGX_LBS_BLEND");
    const std::string v = e ? e : "";
    // unset = f16mix (mode 3: counts re-evaluated in fp32 where the cheap product cannot decide); an unknown string is an
    // error of the call that reads it (-2), not a silent default
    m = (v == "f32" || v == "0") ? 0 : ((v == "bf16x3" || v == "1") ? 1 : ((v == "bf16x2" || v == "2") ? 2 : ((v.empty() || v == "f16mix" || v == "3") ? 3 : -2)));
    if (m >= 0) g_blend_mode.store(m);
  }
  return m;
}
struct WsLayout {
  size_t feat, feat4, A4, picked, total;
  // culled SDF launches
  size_t fvec, jpos, order, flags, items, counts;
  size_t fix_e, fix_stats;   // fix-up of the mixed blend: per-slot error bound, counter of re-evaluated vertices
  size_t skinB, cinit;       // matrix-pipe skinning operands of the mixed blend
  size_t fixq;               // fix-up queue
  size_t Bp;
  int items_stride;
};
WsLayout ws_layout(const egx_body_model* m, int B) {
  const size_t Bp = egx_align_up((size_t)B, BODY_PAD);
  WsLayout w;
  w.Bp = Bp;
  w.feat = 0;
  w.feat4 = egx_align_up(w.feat + Bp * std::max<size_t>(KDIM * sizeof(float), (size_t)KS3 * 16 * 3 * 2), 256);   // mixed-blend features (mode 3)
  w.A4 = egx_align_up(w.feat4 + Bp * ((size_t)M4_FEAT_PIECES * 64 * 16 / 32), 256);   // 32 KiB per 32-body tile
  w.picked = egx_align_up(w.A4 + Bp * NJ * 12 * sizeof(float), 256);
  w.fvec = egx_align_up(w.picked + (size_t)B * m->NP * 3 * sizeof(float), 256);
  w.jpos = egx_align_up(w.fvec + 64 * Bp * sizeof(float), 256);
  w.order = egx_align_up(w.jpos + (size_t)NJ * 3 * Bp * sizeof(float), 256);
  w.flags = egx_align_up(w.order + (size_t)B * sizeof(int), 256);
  const size_t n_items = (size_t)m->n_sdf_tiles * (Bp / BODY_PAD);
  w.items_stride = (int)n_items;
  w.items = egx_align_up(w.flags + n_items * sizeof(int), 256);
  w.counts = egx_align_up(w.items + 8 * n_items * sizeof(int), 256);
  w.fix_e = egx_align_up(w.counts + 16 * sizeof(int), 256);
  w.fix_stats = egx_align_up(w.fix_e + Bp * sizeof(float), 256);
  w.skinB = egx_align_up(w.fix_stats + LBS_FIX_STATS_INTS * sizeof(int), 256);
  w.cinit = egx_align_up(w.skinB + (Bp / 32) * (size_t)SKIN_BT_BYTES, 256);
  w.fixq = egx_align_up(w.cinit + Bp * sizeof(f32x4), 256);
  w.total = egx_align_up(w.fixq + (size_t)LBS_FIX_NQ * LBS_FIXQ_CAP * sizeof(int2), 256);
  return w;
}
// free-space culling of SDF work items: OPT-IN (EGX_LBS_CULL=1 / egx_lbs_set_culling(1)).  Measured on MI355X, 10 240 bodies,
// structured body (profiles/r04_lbs_culling.md): freshly reset agents standing in the room - 60 % of the items evaluated, fused
// kernel 1.13 -> 0.77 ms, the three culling kernels + 0.11 ms; bodies inside geometry or outside the room (what a random-init
// policy produces, i.e. the benchmark loop): nothing to skip, + 0.11 ms.  A win for trained policies on learned body models, a
// loss in the benchmark loop, hence not the default.
std::atomic<int> g_cull{-1};
int culling_on() {
  int c = g_cull.load();
  if (c < 0) {
    const char* e = getenv("EGX_LBS_CULL");
    c = (e && std::string(e) == "1") ? 1 : 0;
    g_cull.store(c);
  }
  return c;
}
}  // namespace

namespace {
// wave tile of the mixed-blend kernel: 0 = by launch size (see egx_lbs_forward), 1 = 32 x 32 (three workgroups per CU), 2 = 32 x 64
std::atomic<int> g_wave_tile{-1};
int wave_tile() {
  int t = g_wave_tile.load();
  if (t < 0) {
    const char* e = getenv("EGX_LBS_WAVE_TILE");
    t = e ? atoi(e) : 0;
    g_wave_tile.store(t);
  }
  return t;
}
}  // namespace

namespace {
std::atomic<int> g_fixq_cap{0};   // 0 = LBS_FIXQ_CAP
}
extern "C" int egx_lbs_set_fix_queue_capacity(int entries) {
  EGX_REQUIRE(entries >= 0 && entries <= LBS_FIXQ_CAP, "capacity must be 0 (default) .. the workspace's queue size");
  g_fixq_cap.store(entries);
  return EGX_OK;
}

extern "C" int egx_lbs_set_wave_tile(int tile) {
  EGX_REQUIRE(tile >= 0 && tile <= 2, "wave tile must be 0 (by launch size), 1 (32 x 32) or 2 (32 x 64)");
  g_wave_tile.store(tile);
  return EGX_OK;
}
extern "C" int egx_lbs_get_wave_tile(void) { return wave_tile(); }

extern "C" int egx_lbs_set_culling(int on) {
  g_cull.store(on ? 1 : 0);
  return EGX_OK;
}
extern "C" int egx_lbs_get_culling(void) { return culling_on(); }
extern "C" int egx_body_model_culls(const egx_body_model* m, float* out_reference_margin_m) {
  if (!m) return 0;
  if (out_reference_margin_m) *out_reference_margin_m = m->cull_ref_margin;
  return m->cull_ok;
}

extern "C" int egx_lbs_cull_stats(const egx_body_model* m, const void* workspace, int num_bodies, int32_t* out_active_items,
                                  int32_t* out_total_items) {
  EGX_REQUIRE(m && workspace && num_bodies > 0 && out_active_items && out_total_items, "bad arguments");
  const WsLayout wl = ws_layout(m, num_bodies);
  int c[16];
  EGX_HIP_CHECK(hipMemcpy(c, static_cast<const char*>(workspace) + wl.counts, sizeof(c), hipMemcpyDeviceToHost));
  if (c[15] != 0x43554c4c) {
    egx_set_error("egx_lbs_cull_stats: no culled launch has run on this workspace (culling off, a model whose bound is not tight, "
                  "a call without SDF counts, or the fp32 blend mode)");
    return EGX_ERR_ARG;
  }
  *out_active_items = c[8];
  *out_total_items = wl.items_stride;
  return EGX_OK;
}

extern "C" int egx_lbs_fix_stats(const egx_body_model* m, const void* workspace, int num_bodies, int32_t* out_reevaluated) {
  EGX_REQUIRE(m && workspace && num_bodies > 0 && out_reevaluated, "bad arguments");
  const WsLayout wl = ws_layout(m, num_bodies);
  std::vector<int32_t> st(LBS_FIX_STATS_INTS);
  EGX_HIP_CHECK(hipMemcpy(st.data(), static_cast<const char*>(workspace) + wl.fix_stats, st.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
  const int cap = g_fixq_cap.load() > 0 ? g_fixq_cap.load() : LBS_FIXQ_CAP;   // the capacity in force now (= at the launch, unless changed since)
  int64_t n = st[0];                                                            // re-evaluated inside the fused kernel (full sub-queue)
  for (int q = 0; q < LBS_FIX_NQ; ++q) n += std::min(st[LBS_FIX_CNT0 + 32 * q], cap);
  // (entries a wave reserved and then marked void are counted twice: only when a sub-queue overflowed)
  *out_reevaluated = (int32_t)std::min<int64_t>(n, INT32_MAX);
  return EGX_OK;
}

#ifdef EGX_LBS_TIMING
extern "C" int egx_lbs_timing_read(unsigned long long* out16, int reset) {
  EGX_HIP_CHECK(hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_lbs_t), 16 * sizeof(unsigned long long)));
  if (reset) {
    unsigned long long z[16] = {0};
    EGX_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_lbs_t), z, sizeof(z)));
  }
  return EGX_OK;
}
#endif

extern "C" int egx_lbs_set_blend_mode(int mode) {
  EGX_REQUIRE(mode >= 0 && mode <= 3, "blend mode must be 0 (fp32 MFMA), 1 (bf16x3 split), 2 (bf16x2 split) or 3 (bf16x2 shape / template + fp16 pose correctives)");
  g_blend_mode.store(mode);
  return EGX_OK;
}
extern "C" int egx_lbs_get_blend_mode(void) { return blend_mode(); }   // < 0: EGX_LBS_BLEND holds an unknown string

extern "C" size_t egx_lbs_workspace_bytes(const egx_body_model* m, int num_bodies) {
  if (!m || num_bodies <= 0) return 0;
  return ws_layout(m, num_bodies).total;
}

extern "C" int egx_lbs_joints(const egx_body_model* m, const float* xb, const float* betas, int B, int fpa, float* out_joints55,
                              void* workspace, size_t workspace_bytes, void* stream_) {
  EGX_REQUIRE(m && xb && betas && out_joints55, "null model/xb/betas/out");
  EGX_REQUIRE(B > 0 && fpa > 0, "num_bodies and frames_per_agent must be positive");
  const WsLayout wl = ws_layout(m, B);
  if (!workspace || workspace_bytes < wl.total) {
    egx_set_error("workspace too small: need " + std::to_string(wl.total) + " bytes");
    return EGX_ERR_WORKSPACE;
  }
  char* ws = static_cast<char*>(workspace);
  hipLaunchKernelGGL(egx_pose_chain_kernel, dim3(egx_ceil_div(B, 4)), dim3(256), 0, static_cast<hipStream_t>(stream_), m->pc, xb,
                     betas, B, fpa, static_cast<float*>(nullptr), static_cast<unsigned short*>(nullptr),
                     reinterpret_cast<f32x4*>(ws + wl.A4), out_joints55, NJ, 0.f, static_cast<unsigned short*>(nullptr), static_cast<int*>(nullptr),
                     static_cast<const int*>(nullptr), static_cast<float*>(nullptr), static_cast<float*>(nullptr), 0, static_cast<float*>(nullptr),
                     static_cast<int*>(nullptr), static_cast<unsigned short*>(nullptr), static_cast<f32x4*>(nullptr),
                     static_cast<const float*>(nullptr), static_cast<const float*>(nullptr), SdfDev{});
  EGX_HIP_CHECK(hipGetLastError());
  return EGX_OK;
}

extern "C" int egx_lbs_forward(const egx_body_model* m, const float* xb, const float* betas, int B, int fpa,
                               float* out_verts, float* out_joints, float* out_markers, const egx_sdf_grid* sdf,
                               const float* R0, const float* T0, int32_t* out_pene_count, void* workspace,
                               size_t workspace_bytes, void* stream_) {
  EGX_REQUIRE(m && xb && betas, "null model/xb/betas");
  EGX_REQUIRE(B > 0 && fpa > 0, "num_bodies and frames_per_agent must be positive");
  EGX_REQUIRE(!sdf || (sdf->grid && out_pene_count && sdf->d0 > 0 && sdf->d1 > 0 && sdf->d2 > 0), "sdf needs grid + out_pene_count");
  EGX_REQUIRE(!sdf || sdf->coarse_minmax, "sdf needs its bracket table: call egx_sdf_build_coarse once per grid");
  EGX_REQUIRE(!sdf || egx_sdf_dims_ok(sdf->d0, sdf->d1, sdf->d2), "sdf grid needs d2 >= 2 and fewer than 2^32 samples");
  const WsLayout wl = ws_layout(m, B);
  if (!workspace || workspace_bytes < wl.total) {
    egx_set_error("workspace too small: need " + std::to_string(wl.total) + " bytes");
    return EGX_ERR_WORKSPACE;
  }
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  char* ws = static_cast<char*>(workspace);
  float* feat = reinterpret_cast<float*>(ws + wl.feat);
  unsigned short* feat4 = reinterpret_cast<unsigned short*>(ws + wl.feat4);
  f32x4* A4 = reinterpret_cast<f32x4*>(ws + wl.A4);
  const bool need_picks = out_joints || out_markers;
  float* picked = need_picks ? reinterpret_cast<float*>(ws + wl.picked) : nullptr;

  const int mode = blend_mode();   // read ONCE per call: the feature flag of column 470 and the kernel choice must agree
  EGX_REQUIRE(mode >= 0, "EGX_LBS_BLEND must be one of f32 | bf16x3 | bf16x2 | f16mix");
  const bool split3 = mode >= 1 && !out_verts;
  const int nbg_all = egx_ceil_div(B, BODY_PAD);
  // free-space culling: SDF counts on the split kernels, a convex-weight model, whole agents, enough items to deal to 8 XCDs
  const bool cull = split3 && sdf && m->cull_ok && culling_on() && B % fpa == 0 && m->n_sdf_tiles > m->n_pick_tiles &&
                    (size_t)m->n_sdf_tiles * nbg_all >= 8;
  SdfDev sd;
  std::memset(&sd, 0, sizeof(sd));
  const float* mips = nullptr;
  if (sdf) {
    sd.grid = sdf->grid; sd.d0 = sdf->d0; sd.d1 = sdf->d1; sd.d2 = sdf->d2;
    sd.cx = sdf->center[0]; sd.cy = sdf->center[1]; sd.cz = sdf->center[2]; sd.scale = sdf->scale;
    sd.coarse = static_cast<const float2*>(sdf->coarse_minmax);
    sd.c0 = egx_ceil_div(sdf->d0, 4); sd.c1 = egx_ceil_div(sdf->d1, 4); sd.c2 = egx_ceil_div(sdf->d2, 4);
    mips = reinterpret_cast<const float*>(static_cast<const char*>(sdf->coarse_minmax) + egx_sdf_table_bytes(sd.c0, sd.c1, sd.c2));
  }
  const bool fix = split3 && mode == 3 && sdf;   // mixed blend with counts: the pose kernel also writes the per-body error bound
  int* order = cull ? reinterpret_cast<int*>(ws + wl.order) : nullptr;
  int* flags = reinterpret_cast<int*>(ws + wl.flags);
  int* items = reinterpret_cast<int*>(ws + wl.items);
  int* counts = reinterpret_cast<int*>(ws + wl.counts);
  float* fvec = cull ? reinterpret_cast<float*>(ws + wl.fvec) : nullptr;
  float* jpos = cull ? reinterpret_cast<float*>(ws + wl.jpos) : nullptr;
  if (cull) {
    const int A = B / fpa;
    // rest pelvis of the mean shape: the classification of an agent only steers the slot order, it decides nothing
    const float* pel = m->rest_pelvis;
    hipLaunchKernelGGL(egx_lbs_agent_order_kernel, dim3(1), dim3(256), (size_t)A * sizeof(int), stream, xb, R0, T0, sd, mips, A, fpa,
                       pel[0], pel[1], pel[2], order, flags, m->n_sdf_tiles * nbg_all, counts);
  }
  hipLaunchKernelGGL(egx_pose_chain_kernel, dim3(egx_ceil_div(B, 4)), dim3(256), 0, stream, m->pc, xb, betas, B, fpa,
                     split3 ? nullptr : feat, split3 ? reinterpret_cast<unsigned short*>(feat) : nullptr, A4, out_joints,
                     EGX_NUM_JOINTS_OUT, (split3 && mode >= 2) ? 1.f : 0.f, (split3 && mode == 3) ? feat4 : nullptr, sdf ? out_pene_count : nullptr,
                     static_cast<const int*>(order), fvec, jpos, (int)wl.Bp, fix ? reinterpret_cast<float*>(ws + wl.fix_e) : nullptr,
                     reinterpret_cast<int*>(ws + wl.fix_stats), fix ? reinterpret_cast<unsigned short*>(ws + wl.skinB) : nullptr,
                     reinterpret_cast<f32x4*>(ws + wl.cinit), R0, T0, sd);
  if (cull) {
    const int n_np = m->n_sdf_tiles - m->n_pick_tiles;
    hipLaunchKernelGGL(egx_lbs_cull_kernel, dim3(nbg_all, egx_ceil_div(n_np, CULL_TILES_PER_BLOCK)), dim3(256), 0, stream, m->sdf_tiles,
                       m->n_pick_tiles, m->n_sdf_tiles, m->tj_off, m->tj_idx, m->cull_D0, m->cull_E, fvec, jpos, (int)wl.Bp,
                       static_cast<const int*>(order), B, fpa, nbg_all, R0, T0, sd, mips, flags);
    hipLaunchKernelGGL(egx_lbs_compact_kernel, dim3(8), dim3(64), 0, stream, flags, m->n_pick_tiles, m->n_sdf_tiles, nbg_all, 2, items,
                       wl.items_stride, counts);
  }
  if (out_verts || need_picks || sdf) {
    LbsParams p;
    p.dirs = m->dirs; p.tj_off = m->tj_off; p.tj_idx = m->tj_idx; p.tj_w = m->tj_w; p.pick_slot = m->pick_slot;
    p.dirs3 = m->dirs3; p.feat3 = reinterpret_cast<const bf16x8*>(feat);
    p.dirs4 = m->dirs4; p.feat4 = reinterpret_cast<const bf16x8*>(feat4);
    p.vorig = m->vorig;
    p.vflags = m->vflags; p.feat = reinterpret_cast<const f32x4*>(feat); p.A4 = A4; p.xb = xb;
    p.B = B; p.V = m->V; p.NVT = m->NVT; p.NW = m->NW; p.NP = m->NP; p.fpa = fpa;
    p.nbg = egx_ceil_div(B, BODY_PAD);
#ifdef EGX_LBS_DEVELOPMENT   // ablation switches of development builds only (make CXXFLAGS+=-DEGX_LBS_DEVELOPMENT)
    {
      const char* e = getenv("EGX_LBS_DBG");
      p.dbg = e ? atoi(e) : 0;
    }
#else
    p.dbg = 0;
#endif
    p.verts = out_verts; p.picked = picked; p.R0 = R0; p.T0 = T0; p.pene = out_pene_count;
    p.agent_of_slot = order; p.items = cull ? items : nullptr; p.item_counts = counts; p.items_stride = wl.items_stride;
    // markers and joints only: the vertex tiles without a picked vertex are never looked at
    // (with SDF counts: nor the tiles made of feet vertices only, which the count excludes)
    p.tiles = out_verts ? nullptr : (sdf ? m->sdf_tiles : m->pick_tiles);
    p.n_tiles = out_verts ? m->NVT : (sdf ? m->n_sdf_tiles : m->n_pick_tiles);
    // mode 3: the tile lists start with the tiles that hold picked vertices (checked at load: sdf_lead_picks)
    p.n_precise = sdf ? (m->sdf_lead_picks ? m->n_pick_tiles : m->n_sdf_tiles) : m->n_pick_tiles;
#ifdef EGX_LBS_DEVELOPMENT
    if (const char* e = getenv("EGX_LBS_PRECISE")) p.n_precise = atoi(e);
#endif
    p.sdf = sd;   // out_pene_count was cleared by the pose kernel above
    p.fix_e = reinterpret_cast<const float*>(ws + wl.fix_e);
    p.fix_stats = reinterpret_cast<int*>(ws + wl.fix_stats);
    p.skinW = m->skinW; p.skin_ks_off = m->skin_ks_off;
    p.fixq = reinterpret_cast<int2*>(ws + wl.fixq); p.fixq_cap = g_fixq_cap.load() > 0 ? g_fixq_cap.load() : LBS_FIXQ_CAP;
    p.dirs_rm = m->dirs_rm; p.pc = m->pc; p.betas = betas;
    p.skinB = reinterpret_cast<const bf16x8*>(ws + wl.skinB); p.cinit = reinterpret_cast<const f32x4*>(ws + wl.cinit);
    p.sdf_aux = sdf ? reinterpret_cast<const float*>(static_cast<const char*>(sdf->coarse_minmax) + egx_sdf_aux_offset(sd.c0, sd.c1, sd.c2)) : nullptr;
    // one persistent workgroup per CU; per-device launch facts (CU count, raised dynamic-LDS caps) are set up once per device
    constexpr size_t lds_meta = (size_t)8 * LBS_META_BYTES, lds_verts = (size_t)8 * (LBS_META_BYTES + LBS_VERT_BYTES),
                     lds_sdf = (size_t)8 * (LBS_META_BYTES + LBS_QCAP * 16);
    int dev = 0;
    EGX_HIP_CHECK(hipGetDevice(&dev));
    EGX_REQUIRE(dev >= 0 && dev < kMaxDevices, "device ordinal out of range");
    LbsDeviceInfo& di = g_lbs_dev[dev];
    {
      std::lock_guard<std::mutex> lk(di.mu);
      if (di.num_cu == 0) {
        hipDeviceProp_t prop;
        EGX_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
        auto raise = [](const void* fn, size_t bytes) { return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes); };
        EGX_HIP_CHECK(raise(reinterpret_cast<const void*>(&egx_lbs_fused_kernel<true, true>), lds_verts));
        EGX_HIP_CHECK(raise(reinterpret_cast<const void*>(&egx_lbs_fused_kernel<true, false>), lds_verts));
        EGX_HIP_CHECK(raise(reinterpret_cast<const void*>(&egx_lbs_fused_kernel<false, true>), lds_sdf));
        EGX_HIP_CHECK(raise(reinterpret_cast<const void*>(&egx_lbs_fused_kernel<false, false>), lds_meta));
        constexpr size_t lds3a = (size_t)LBS3_SHARED_BYTES + 4 * lbs3_wave_bytes<LBS_NB>();
        EGX_HIP_CHECK(raise(reinterpret_cast<const void*>(&egx_lbs_fused3_kernel<3, true>), lds3a));
        EGX_HIP_CHECK(raise(reinterpret_cast<const void*>(&egx_lbs_fused3_kernel<3, false>), lds3a));
        EGX_HIP_CHECK(raise(reinterpret_cast<const void*>(&egx_lbs_fused3_kernel<2, true>), lds3a));
        EGX_HIP_CHECK(raise(reinterpret_cast<const void*>(&egx_lbs_fused3_kernel<2, false>), lds3a));
        EGX_HIP_CHECK(raise(reinterpret_cast<const void*>(&egx_lbs_fused3_kernel<4, true>), lds3a));
        EGX_HIP_CHECK(raise(reinterpret_cast<const void*>(&egx_lbs_fused3_kernel<4, false>), lds3a));
        EGX_HIP_CHECK(raise(reinterpret_cast<const void*>(&egx_lbs_fused3_kernel<4, true, 1>), lbs3_lds_bytes<1>()));
        EGX_HIP_CHECK(raise(reinterpret_cast<const void*>(&egx_lbs_fused3_kernel<4, false, 1>), lbs3_lds_bytes<1>()));
        di.num_cu = prop.multiProcessorCount;
      }
    }
    const int num_cu = di.num_cu;
    // three body groups per block: features + joint transforms of a block (3 x 1.2 MB) still live in the XCD's 4 MiB L2 and the
    // bases stream through twice per XCD instead of three times (5 groups per XCD at 10 240 bodies): fabric-side traffic 1.77 ->
    // 1.43 GB per launch at the same launch time (1: 2.6 GB, 5: 1.65 GB and +4 % time; profiles/r04_lbs_traffic.md)
    p.bg_block = 3;
#ifdef EGX_LBS_DEVELOPMENT
    if (const char* e = getenv("EGX_LBS_BG_BLOCK")) p.bg_block = atoi(e);
#endif
    const int n_items = p.nbg * p.n_tiles;
    const int grid = std::max(1, std::min(num_cu, (n_items + 1) / 2));
    const size_t lds = out_verts ? lds_verts : (sdf ? lds_sdf : lds_meta);
    hipEvent_t ev0 = g_prof_start, ev1 = g_prof_stop;
    g_prof_start = g_prof_stop = nullptr;
    if (ev0) EGX_HIP_CHECK(hipEventRecord(ev0, stream));
    if (split3) {
      // two persistent 4-wave workgroups per CU: one's VALU epilogue runs under the other's MFMA stages
      constexpr size_t lds3 = (size_t)LBS3_SHARED_BYTES + 4 * lbs3_wave_bytes<LBS_NB>();
      int wg_per_cu = 2;
#ifdef EGX_LBS_DEVELOPMENT
      if (const char* e = getenv("EGX_LBS_WG_PER_CU")) wg_per_cu = std::max(1, atoi(e));   // occupancy sensitivity (1 = one wave per SIMD)
#endif
      int grid3 = std::max(1, std::min(wg_per_cu * num_cu, n_items));
      if (grid3 >= 8) grid3 &= ~7;   // a multiple of 8: the kernel's XCD partition (body groups, or vertex tiles when groups are few)
      // mixed blend without culling, launches of at most 20 body groups of 256 (<= 256 agents x 20 frames): the small wave tile
      // (32 vertices x 32 bodies per wave, 128 bodies per workgroup item, three workgroups per CU) - finer items balance the
      // XCDs better and a third wave per SIMD helps where the launch is short: 640 bodies 0.086 -> 0.073 ms, 1 280 0.150 -> 0.116,
      // 2 560 0.250 -> 0.209, 5 120 0.461 -> 0.355; at 10 240 bodies the larger tile wins (0.686 against 0.734: the halved
      // item repeats the bases traffic and the barriers), profiles/r05_lbs_mixed.md section 5.  EGX_LBS_WAVE_TILE=1 | 2 forces one.
      const int forced_tile = wave_tile();
      const bool small_tile = forced_tile == 1 || (forced_tile != 2 && p.nbg <= 20);
      if (mode == 3 && small_tile && !p.items) {
        LbsParams q = p;
        q.nbg = egx_ceil_div(B, 128);
        q.bg_block = 2 * p.bg_block;
        const int n_items1 = q.nbg * q.n_tiles;
        int g1 = std::max(1, std::min(3 * num_cu, n_items1));
        if (g1 >= 8) g1 &= ~7;
        if (sdf) hipLaunchKernelGGL((egx_lbs_fused3_kernel<4, true, 1>), dim3(g1), dim3(256), lbs3_lds_bytes<1>(), stream, q);
        else hipLaunchKernelGGL((egx_lbs_fused3_kernel<4, false, 1>), dim3(g1), dim3(256), lbs3_lds_bytes<1>(), stream, q);
      } else if (mode == 3) {
        if (sdf) hipLaunchKernelGGL((egx_lbs_fused3_kernel<4, true>), dim3(grid3), dim3(256), lds3, stream, p);
        else hipLaunchKernelGGL((egx_lbs_fused3_kernel<4, false>), dim3(grid3), dim3(256), lds3, stream, p);
      } else if (mode == 2) {
        if (sdf) hipLaunchKernelGGL((egx_lbs_fused3_kernel<2, true>), dim3(grid3), dim3(256), lds3, stream, p);
        else hipLaunchKernelGGL((egx_lbs_fused3_kernel<2, false>), dim3(grid3), dim3(256), lds3, stream, p);
      } else {
        if (sdf) hipLaunchKernelGGL((egx_lbs_fused3_kernel<3, true>), dim3(grid3), dim3(256), lds3, stream, p);
        else hipLaunchKernelGGL((egx_lbs_fused3_kernel<3, false>), dim3(grid3), dim3(256), lds3, stream, p);
      }
    } else if (out_verts && sdf)
      hipLaunchKernelGGL((egx_lbs_fused_kernel<true, true>), dim3(grid), dim3(LBS_THREADS), lds, stream, p);
    else if (out_verts)
      hipLaunchKernelGGL((egx_lbs_fused_kernel<true, false>), dim3(grid), dim3(LBS_THREADS), lds, stream, p);
    else if (sdf)
      hipLaunchKernelGGL((egx_lbs_fused_kernel<false, true>), dim3(grid), dim3(LBS_THREADS), lds, stream, p);
    else
      hipLaunchKernelGGL((egx_lbs_fused_kernel<false, false>), dim3(grid), dim3(LBS_THREADS), lds, stream, p);
    // mixed blend with counts: the vertices the fused kernel queued for the fp32 re-evaluation (inside the profiled interval: it is
    // part of what the mode costs)
    if (fix) hipLaunchKernelGGL(egx_lbs_fix_kernel, dim3(LBS_FIX_BLOCKS), dim3(256), 0, stream, p);
    if (ev1) EGX_HIP_CHECK(hipEventRecord(ev1, stream));
  }
  if (need_picks)
    hipLaunchKernelGGL(egx_gather_kernel, dim3(B), dim3(256), 0, stream, picked, B, m->NP, m->M, m->marker_slot,
                       m->extra_slot, m->lmk_slot, m->lmk_bary, out_joints, out_markers);
  EGX_HIP_CHECK(hipGetLastError());
  return EGX_OK;
}
