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
__host__ __device__ inline int egx_compact_joint(int j) { return (j - 1) - (j > 24 ? 3 : 0); }  // j in 1..54, j != 22..24
constexpr int NLMK = 51, NEXTRA = 21;
constexpr int BODY_PAD = 256;          // bodies per workgroup of the fused kernel
constexpr int LBS_NW_MAX = 16;         // skinning weights per vertex held in LDS (ELL width, multiple of 4)

struct PoseConsts {
  int parents[NJ];
  int depth[NJ];
  int max_depth;
  float J_template[NJ * 3];
  float J_shapedirs[NJ * 3 * 10];
  float hand_comps[2 * 12 * 45];
  float hand_mean[2 * 45];
};

}  // namespace

struct egx_body_model {
  int V = 0, NVT = 0, NW = 0, M = 0, NP = 0;
  f32x4* dirs = nullptr;       // [NVT][62][3][64] float4
  float* vtemp = nullptr;      // [NVT][3][32]
  int* widx = nullptr;         // [NVT*32][NW]
  float* wval = nullptr;       // [NVT*32][NW]
  int* pick_slot = nullptr;    // [NVT*32], -1 = not picked
  uint8_t* vflags = nullptr;   // [NVT*32] bit0 feet, bit1 valid
  PoseConsts* pc = nullptr;
  int* marker_slot = nullptr;  // [M]
  int* extra_slot = nullptr;   // [21]
  int* lmk_slot = nullptr;     // [153]
  float* lmk_bary = nullptr;   // [153]
};

// ------------------------------------------------------------------------------------------------
// kernel 1: per-body pose features, rigid chain, joint transforms
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void egx_pose_chain_kernel(const PoseConsts* __restrict__ pc,
                                                             const float* __restrict__ xb,
                                                             const float* __restrict__ betas, int B, int fpa,
                                                             float* __restrict__ feat,   // packed B operand
                                                             f32x4* __restrict__ A4,     // [bt][55][3][32]
                                                             float* __restrict__ out_joints) {
  __shared__ float sR[4][NJ][9];
  __shared__ float sJ[4][NJ][3];
  __shared__ float sG[4][NJ][12];
  const int w = threadIdx.x >> 6, j = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + w;
  const bool live = b < B;
  const int bb = live ? b : B - 1;
  const float* x = xb + (size_t)bb * EGX_XB_DIM;
  const float* be = betas + (size_t)(bb / fpa) * 10;
  const int bt = bb >> 5, n = bb & 31;
  float* featb = feat + (size_t)bt * KGROUPS * 64 * 4;  // tile base
  auto feat_store = [&](int k, float v) {
    const int s = k >> 1, kk = k & 1;
    featb[((s >> 2) * 64 + (kk * 32 + n)) * 4 + (s & 3)] = v;
  };
  float R[9], Jr[3];
  if (j < NJ) {
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
    // smplx batch_rodrigues: angle = ||a + 1e-8||
    const float ex = a[0] + 1e-8f, ey = a[1] + 1e-8f, ez = a[2] + 1e-8f;
    const float angle = sqrtf(ex * ex + ey * ey + ez * ez);
    const float rx = a[0] / angle, ry = a[1] / angle, rz = a[2] / angle;
    const float sn = sinf(angle), cs = 1.f - cosf(angle);
    R[0] = 1.f + cs * (-(ry * ry + rz * rz)); R[1] = -sn * rz + cs * (rx * ry);     R[2] = sn * ry + cs * (rx * rz);
    R[3] = sn * rz + cs * (rx * ry);          R[4] = 1.f + cs * (-(rx * rx + rz * rz)); R[5] = -sn * rx + cs * (ry * rz);
    R[6] = -sn * ry + cs * (rx * rz);         R[7] = sn * rx + cs * (ry * rz);      R[8] = 1.f + cs * (-(rx * rx + ry * ry));
    for (int c = 0; c < 3; ++c) {
      float s = pc->J_template[j * 3 + c];
      for (int k = 0; k < 10; ++k) s += be[k] * pc->J_shapedirs[(j * 3 + c) * 10 + k];
      Jr[c] = s;
      sJ[w][j][c] = s;
    }
    for (int e = 0; e < 9; ++e) sR[w][j][e] = R[e];
    if (live) {
      if (j < 10) feat_store(j, be[j]);
      if (j >= 1 && (j < 22 || j > 24)) {
        const int k0 = 10 + egx_compact_joint(j) * 9;
        for (int e = 0; e < 9; ++e) feat_store(k0 + e, R[e] - ((e == 0 || e == 4 || e == 8) ? 1.f : 0.f));
      }
      if (j >= 22 && j <= 24) feat_store(KACT + (j - 22), 0.f);  // zero padding columns 469..471
    }
  }
  __syncthreads();
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
    __syncthreads();
  }
  if (j < NJ && live) {
    // relative transform: translation column minus R_g * rest joint (smplx batch_rigid_transform)
    for (int r = 0; r < 3; ++r) {
      const float t = G[r * 4 + 3] - (G[r * 4 + 0] * Jr[0] + G[r * 4 + 1] * Jr[1] + G[r * 4 + 2] * Jr[2]);
      f32x4 row = {G[r * 4 + 0], G[r * 4 + 1], G[r * 4 + 2], t};
      A4[(((size_t)bt * NJ + j) * 3 + r) * 32 + n] = row;
    }
    if (out_joints) {
      float* o = out_joints + ((size_t)b * EGX_NUM_JOINTS_OUT + j) * 3;
      o[0] = G[3] + x[0]; o[1] = G[7] + x[1]; o[2] = G[11] + x[2];
    }
  }
}

// ------------------------------------------------------------------------------------------------
// kernel 2: fused blend GEMM + skinning + (SDF count) + (vertex picks) + (vertex write)
// ------------------------------------------------------------------------------------------------
struct LbsParams {
  const f32x4* dirs;
  const float* vtemp;
  const int* widx;
  const float* wval;
  const int* pick_slot;
  const uint8_t* vflags;
  const f32x4* feat;   // [bt][62][64] float4
  const f32x4* A4;     // [bt][55][3][32] float4
  const float* xb;     // transl = xb[b*93 + 0..2]
  int B, V, NVT, NW, NP, fpa;
  int nbg;             // body groups (256 bodies each)
  int stagger_lo, stagger_hi;  // block-id range delayed at start (second residency slot of every CU)
  long long stagger_cycles;
  float* verts;        // [B][V][3] or null
  float* picked;       // [B][NP][3] or null
  SdfDev sdf;
  const float* R0;     // [A][9] or null
  const float* T0;     // [A][3] or null
  int* pene;           // [B]
};

template <bool WRITE_VERTS, bool DO_SDF>
__global__ __launch_bounds__(256, 2) void egx_lbs_fused_kernel(LbsParams p) {
  constexpr int NB = 2;  // 32-body MFMA column tiles per wave
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = lane & 31, half = lane >> 5;
  // block -> (vertex tile, body group).  With >= 8 body groups each XCD (block id % 8) owns a
  // contiguous chunk of body groups so their packed features / transforms stay in that XCD's L2
  // while the 62 MB of blend bases stream through once per XCD.
  int vt, bg;
  {
    const int id = blockIdx.x;
    if (p.nbg >= 8) {
      const int per = (p.nbg + 7) / 8;
      const int xcd = id & 7, local = id >> 3;
      bg = xcd * per + local % per;
      vt = local / per;
      if (bg >= p.nbg || vt >= p.NVT) return;
    } else {
      bg = id % p.nbg;
      vt = id / p.nbg;
    }
  }
  const int bt0 = bg * 8 + wave * NB;  // first 32-body tile of this wave
  const int num_bt = (p.B + 31) >> 5;
  // Phase stagger: two workgroups share each CU (2 waves per SIMD).  Launched together they run their MFMA loops at the
  // same time (sharing the matrix pipe) and then their epilogues at the same time (matrix pipe idle).  Delaying the
  // second-slot workgroups of the FIRST generation by about half a tile puts the pairs in anti-phase - one wave's
  // epilogue under the other's MFMA loop - and the offset persists because successors start when a slot frees up.
  if (p.stagger_cycles > 0 && blockIdx.x >= (unsigned)p.stagger_lo && blockIdx.x < (unsigned)p.stagger_hi) {
    const long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < p.stagger_cycles) __builtin_amdgcn_s_sleep(32);
  }
  // per-vertex metadata of this tile (shared by the 4 waves): skinning weights, template, pick slot, flags
  __shared__ int s_widx[32][LBS_NW_MAX];
  __shared__ float s_wval[32][LBS_NW_MAX];
  __shared__ float s_vt[3][32];
  __shared__ int s_slot[32];
  __shared__ int s_flag[32];
  for (int idx = threadIdx.x; idx < 32 * p.NW; idx += 256) {
    const int row = idx / p.NW, k = idx % p.NW;
    s_widx[row][k] = p.widx[(size_t)(vt * 32 + row) * p.NW + k];
    s_wval[row][k] = p.wval[(size_t)(vt * 32 + row) * p.NW + k];
  }
  if (threadIdx.x < 96) s_vt[threadIdx.x >> 5][threadIdx.x & 31] = p.vtemp[(vt * 3 + (threadIdx.x >> 5)) * 32 + (threadIdx.x & 31)];
  if (threadIdx.x >= 128 && threadIdx.x < 160) {
    s_slot[threadIdx.x - 128] = p.pick_slot[vt * 32 + threadIdx.x - 128];
    s_flag[threadIdx.x - 128] = p.vflags[vt * 32 + threadIdx.x - 128];
  }
  __syncthreads();

  f32x16 acc[3][NB];
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int q = 0; q < NB; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[c][q][r] = 0.f;

  const f32x4* dp = p.dirs + (size_t)vt * KGROUPS * 3 * 64 + lane;
  const f32x4* fp[NB];
#pragma unroll
  for (int q = 0; q < NB; ++q) fp[q] = p.feat + (size_t)min(bt0 + q, num_bt - 1) * KGROUPS * 64 + lane;

  f32x4 a_cur[3], b_cur[NB];
#pragma unroll
  for (int c = 0; c < 3; ++c) a_cur[c] = dp[c * 64];
#pragma unroll
  for (int q = 0; q < NB; ++q) b_cur[q] = fp[q][0];

  for (int g = 0; g < KGROUPS; ++g) {
    f32x4 a_nxt[3], b_nxt[NB];
    const int gn = (g + 1 < KGROUPS) ? g + 1 : g;
#pragma unroll
    for (int c = 0; c < 3; ++c) a_nxt[c] = dp[(gn * 3 + c) * 64];
#pragma unroll
    for (int q = 0; q < NB; ++q) b_nxt[q] = fp[q][gn * 64];
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int q = 0; q < NB; ++q)
          acc[c][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[c][e], b_cur[q][e], acc[c][q], 0, 0, 0);
#pragma unroll
    for (int c = 0; c < 3; ++c) a_cur[c] = a_nxt[c];
#pragma unroll
    for (int q = 0; q < NB; ++q) b_cur[q] = b_nxt[q];
  }

  // ---- epilogue: each lane owns 16 vertices (rows) x NB bodies (col n of tiles bt0+q) ----------
  // Organised in passes with many independent loads in flight (the first version walked the vertices one dependent
  // global round trip after the other and spent more cycles here than in the 1488 MFMAs).
  float tr[NB][3];
  int body[NB];
  bool bvalid[NB];
#pragma unroll
  for (int q = 0; q < NB; ++q) {
    body[q] = (bt0 + q) * 32 + n;
    bvalid[q] = body[q] < p.B;
    const int bb = bvalid[q] ? body[q] : p.B - 1;
    tr[q][0] = p.xb[(size_t)bb * EGX_XB_DIM + 0];
    tr[q][1] = p.xb[(size_t)bb * EGX_XB_DIM + 1];
    tr[q][2] = p.xb[(size_t)bb * EGX_XB_DIM + 2];
  }
  // One pass per (body tile, vertex): all per-vertex metadata comes from LDS, the 12 transform rows of a vertex are
  // fetched together, and the only dependent global access left is the SDF bracket lookup.
  float* lds = reinterpret_cast<float*>(smem_raw) + wave * (32 * 97);
#pragma unroll
  for (int q = 0; q < NB; ++q) {
    const f32x4* Aq = p.A4 + (size_t)min(bt0 + q, num_bt - 1) * NJ * 3 * 32 + n;
    float Rw[9], Tw[3];
    if (DO_SDF) {
      const int ag = (bvalid[q] ? body[q] : p.B - 1) / p.fpa;
#pragma unroll
      for (int e = 0; e < 9; ++e) Rw[e] = p.R0 ? p.R0[(size_t)ag * 9 + e] : ((e % 4 == 0) ? 1.f : 0.f);
#pragma unroll
      for (int e = 0; e < 3; ++e) Tw[e] = p.T0 ? p.T0[(size_t)ag * 3 + e] : 0.f;
    }
    int cnt = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
      f32x4 t0 = {0.f, 0.f, 0.f, 0.f}, t1 = t0, t2 = t0;
      for (int k0 = 0; k0 < p.NW; k0 += 4) {  // NW is padded to a multiple of 4 at load time (zero weights)
        f32x4 ld[4][3];
        float wv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {  // 12 independent 16-byte loads in flight
          const int jn = s_widx[row][k0 + k];
          wv[k] = s_wval[row][k0 + k];
#pragma unroll
          for (int c = 0; c < 3; ++c) ld[k][c] = Aq[(jn * 3 + c) * 32];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          t0 += wv[k] * ld[k][0];
          t1 += wv[k] * ld[k][1];
          t2 += wv[k] * ld[k][2];
        }
      }
      const float vx = acc[0][q][r] + s_vt[0][row], vy = acc[1][q][r] + s_vt[1][row], vz = acc[2][q][r] + s_vt[2][row];
      const float ox = t0[0] * vx + t0[1] * vy + t0[2] * vz + t0[3] + tr[q][0];
      const float oy = t1[0] * vx + t1[1] * vy + t1[2] * vz + t1[3] + tr[q][1];
      const float oz = t2[0] * vx + t2[1] * vy + t2[2] * vz + t2[3] + tr[q][2];
      if (DO_SDF) {
        if ((s_flag[row] & 3) == 2) {  // valid, not a feet vertex
          const float wx = Rw[0] * ox + Rw[1] * oy + Rw[2] * oz + Tw[0];
          const float wy = Rw[3] * ox + Rw[4] * oy + Rw[5] * oz + Tw[1];
          const float wz = Rw[6] * ox + Rw[7] * oy + Rw[8] * oz + Tw[2];
          int sg = p.sdf.coarse ? egx_sdf_coarse_sign(p.sdf, wx, wy, wz) : 0;
          if (sg == 0) sg = (egx_sdf_neg_trilinear(p.sdf, wx, wy, wz) < 0.f) ? 1 : -1;
          cnt += (sg > 0) ? 1 : 0;
        }
      }
      if (p.picked) {
        const int slot = s_slot[row];
        if (slot >= 0 && bvalid[q]) {
          float* o = p.picked + ((size_t)body[q] * p.NP + slot) * 3;
          o[0] = ox; o[1] = oy; o[2] = oz;
        }
      }
      if (WRITE_VERTS) {
        lds[n * 97 + row * 3 + 0] = ox;
        lds[n * 97 + row * 3 + 1] = oy;
        lds[n * 97 + row * 3 + 2] = oz;
      }
      asm volatile("" ::: "memory");
    }
    if (DO_SDF) {
      const int c2 = cnt + __shfl_xor(cnt, 32);
      if (half == 0 && bvalid[q] && c2 != 0) atomicAdd(p.pene + body[q], c2);
    }
    if (WRITE_VERTS) {
      // transpose through LDS: every body row is 32 vertices x 3 = 96 contiguous floats in HBM
      // (wave-private LDS region: DS ops of one wave execute in order, no barrier needed)
      const int vbase = vt * 32;
      const int nv = min(32, p.V - vbase);
      for (int bi = 0; bi < 32; ++bi) {
        const int bd = (bt0 + q) * 32 + bi;
        if (bd >= p.B) break;
        float* o = p.verts + ((size_t)bd * p.V + vbase) * 3;
        for (int f = lane; f < nv * 3; f += 64) o[f] = lds[bi * 97 + f];
      }
    }
  }
}

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

  // blend bases in MFMA A-operand order: [vt][g][c][lane] float4, element e <-> k = 2*(4g+e) + (lane>>5)
  std::vector<f32x4> dirs((size_t)NVT * KGROUPS * 3 * 64);
  for (int vt = 0; vt < NVT; ++vt)
    for (int g = 0; g < KGROUPS; ++g)
      for (int c = 0; c < 3; ++c)
        for (int l = 0; l < 64; ++l) {
          const int v = vt * 32 + (l & 31);
          f32x4 val = {0.f, 0.f, 0.f, 0.f};
          if (v < V)
            for (int e = 0; e < 4; ++e) {
              const int k = 2 * (4 * g + e) + (l >> 5);
              if (k < 10) {
                val[e] = d->shapedirs_host[((size_t)v * 3 + c) * 10 + k];
              } else if (k < KACT) {
                const int jc = (k - 10) / 9, e9 = (k - 10) % 9;
                const int j = jc + 1 + (jc >= 21 ? 3 : 0);        // inverse of egx_compact_joint
                val[e] = d->posedirs_host[(size_t)((j - 1) * 9 + e9) * 3 * V + (size_t)v * 3 + c];
              }
            }
          dirs[(((size_t)vt * KGROUPS + g) * 3 + c) * 64 + l] = val;
        }
  std::vector<float> vtemp((size_t)NVT * 3 * 32, 0.f);
  for (int v = 0; v < V; ++v)
    for (int c = 0; c < 3; ++c) vtemp[((size_t)(v / 32) * 3 + c) * 32 + (v % 32)] = d->v_template_host[(size_t)v * 3 + c];

  // skinning weights -> ELL
  int NW = 1;
  for (int v = 0; v < V; ++v) {
    int c = 0;
    for (int j = 0; j < NJ; ++j) c += d->lbs_weights_host[(size_t)v * NJ + j] != 0.f;
    NW = std::max(NW, c);
  }
  NW = (NW + 3) / 4 * 4;  // zero-weight padding: the skinning loop works in groups of 4
  if (NW > LBS_NW_MAX) {
    delete m;
    egx_set_error("more than 16 non-zero skinning weights on one vertex are not supported");
    return EGX_ERR_ARG;
  }
  m->NW = NW;
  std::vector<int> widx((size_t)VP * NW, 0);
  std::vector<float> wval((size_t)VP * NW, 0.f);
  for (int v = 0; v < V; ++v) {
    int c = 0;
    for (int j = 0; j < NJ; ++j) {
      const float w = d->lbs_weights_host[(size_t)v * NJ + j];
      if (w != 0.f) { widx[(size_t)v * NW + c] = j; wval[(size_t)v * NW + c] = w; ++c; }
    }
  }

  // picked vertices (markers, vertex joints, landmark corners)
  std::vector<int> pick_slot(VP, -1), marker_slot(d->num_markers), extra_slot(NEXTRA), lmk_slot(NLMK * 3);
  int NP = 0;
  auto slot_of = [&](int v) -> int {
    if (v < 0 || v >= V) return -1;
    if (pick_slot[v] < 0) pick_slot[v] = NP++;
    return pick_slot[v];
  };
  bool ok = true;
  for (int i = 0; i < d->num_markers; ++i) ok &= (marker_slot[i] = slot_of(d->marker_vids_host[i])) >= 0;
  for (int i = 0; i < NEXTRA; ++i) ok &= (extra_slot[i] = slot_of(d->extra_vids_host[i])) >= 0;
  for (int i = 0; i < NLMK * 3; ++i) ok &= (lmk_slot[i] = slot_of(d->lmk_vids_host[i])) >= 0;
  if (!ok) { delete m; egx_set_error("vertex id out of range in marker/extra/landmark tables"); return EGX_ERR_ARG; }
  m->NP = NP;
  std::vector<uint8_t> vflags(VP, 0);
  for (int v = 0; v < V; ++v) vflags[v] = 2;
  for (int i = 0; i < d->num_feet; ++i) {
    const int v = d->feet_vids_host[i];
    if (v < 0 || v >= V) { delete m; egx_set_error("feet vertex id out of range"); return EGX_ERR_ARG; }
    vflags[v] |= 1;
  }
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
  std::memcpy(pc.hand_comps, d->hand_comps_l_host, 12 * 45 * sizeof(float));
  std::memcpy(pc.hand_comps + 12 * 45, d->hand_comps_r_host, 12 * 45 * sizeof(float));
  std::memcpy(pc.hand_mean, d->hand_mean_l_host, 45 * sizeof(float));
  std::memcpy(pc.hand_mean + 45, d->hand_mean_r_host, 45 * sizeof(float));

  int rc = EGX_OK;
  if ((rc = upload(&m->dirs, dirs)) || (rc = upload(&m->vtemp, vtemp)) || (rc = upload(&m->widx, widx)) ||
      (rc = upload(&m->wval, wval)) || (rc = upload(&m->pick_slot, pick_slot)) || (rc = upload(&m->vflags, vflags)) ||
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
  (void)hipFree(m->dirs); (void)hipFree(m->vtemp); (void)hipFree(m->widx); (void)hipFree(m->wval);
  (void)hipFree(m->pick_slot); (void)hipFree(m->vflags); (void)hipFree(m->pc); (void)hipFree(m->marker_slot);
  (void)hipFree(m->extra_slot); (void)hipFree(m->lmk_slot); (void)hipFree(m->lmk_bary);
  delete m;
}

extern "C" int egx_body_model_num_verts(const egx_body_model* m) { return m ? m->V : 0; }
extern "C" int egx_body_model_nnz(const egx_body_model* m) { return m ? m->NW : 0; }

namespace {
struct WsLayout {
  size_t feat, A4, picked, total;
};
WsLayout ws_layout(const egx_body_model* m, int B) {
  const size_t Bp = egx_align_up((size_t)B, BODY_PAD);
  WsLayout w;
  w.feat = 0;
  w.A4 = egx_align_up(w.feat + Bp * KDIM * sizeof(float), 256);
  w.picked = egx_align_up(w.A4 + Bp * NJ * 12 * sizeof(float), 256);
  w.total = egx_align_up(w.picked + (size_t)B * m->NP * 3 * sizeof(float), 256);
  return w;
}
}  // namespace

extern "C" size_t egx_lbs_workspace_bytes(const egx_body_model* m, int num_bodies) {
  if (!m || num_bodies <= 0) return 0;
  return ws_layout(m, num_bodies).total;
}

extern "C" int egx_lbs_forward(const egx_body_model* m, const float* xb, const float* betas, int B, int fpa,
                               float* out_verts, float* out_joints, float* out_markers, const egx_sdf_grid* sdf,
                               const float* R0, const float* T0, int32_t* out_pene_count, void* workspace,
                               size_t workspace_bytes, void* stream_) {
  EGX_REQUIRE(m && xb && betas, "null model/xb/betas");
  EGX_REQUIRE(B > 0 && fpa > 0, "num_bodies and frames_per_agent must be positive");
  EGX_REQUIRE(!sdf || (sdf->grid && out_pene_count && sdf->d0 > 0 && sdf->d1 > 0 && sdf->d2 > 0), "sdf needs grid + out_pene_count");
  const WsLayout wl = ws_layout(m, B);
  if (!workspace || workspace_bytes < wl.total) {
    egx_set_error("workspace too small: need " + std::to_string(wl.total) + " bytes");
    return EGX_ERR_WORKSPACE;
  }
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  char* ws = static_cast<char*>(workspace);
  float* feat = reinterpret_cast<float*>(ws + wl.feat);
  f32x4* A4 = reinterpret_cast<f32x4*>(ws + wl.A4);
  const bool need_picks = out_joints || out_markers;
  float* picked = need_picks ? reinterpret_cast<float*>(ws + wl.picked) : nullptr;

  hipLaunchKernelGGL(egx_pose_chain_kernel, dim3(egx_ceil_div(B, 4)), dim3(256), 0, stream, m->pc, xb, betas, B, fpa,
                     feat, A4, out_joints);
  if (out_verts || need_picks || sdf) {
    LbsParams p;
    p.dirs = m->dirs; p.vtemp = m->vtemp; p.widx = m->widx; p.wval = m->wval; p.pick_slot = m->pick_slot;
    p.vflags = m->vflags; p.feat = reinterpret_cast<const f32x4*>(feat); p.A4 = A4; p.xb = xb;
    p.B = B; p.V = m->V; p.NVT = m->NVT; p.NW = m->NW; p.NP = m->NP; p.fpa = fpa;
    p.nbg = egx_ceil_div(B, BODY_PAD);
    {
      static int num_cu = 0;
      static long long stagger = -1;
      if (num_cu == 0) {
        hipDeviceProp_t prop;
        int dev = 0;
        EGX_HIP_CHECK(hipGetDevice(&dev));
        EGX_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
        num_cu = prop.multiProcessorCount;
        const char* e = getenv("EGX_LBS_STAGGER_CYCLES");
        stagger = e ? atoll(e) : 40000;  // ~ half of one tile's MFMA loop (in s_memtime / 100 MHz-or-core ticks: see DESIGN.md)
      }
      p.stagger_lo = num_cu; p.stagger_hi = 2 * num_cu; p.stagger_cycles = stagger;
    }
    p.verts = out_verts; p.picked = picked; p.R0 = R0; p.T0 = T0; p.pene = out_pene_count;
    std::memset(&p.sdf, 0, sizeof(p.sdf));
    if (sdf) {
      p.sdf.grid = sdf->grid; p.sdf.d0 = sdf->d0; p.sdf.d1 = sdf->d1; p.sdf.d2 = sdf->d2;
      p.sdf.cx = sdf->center[0]; p.sdf.cy = sdf->center[1]; p.sdf.cz = sdf->center[2]; p.sdf.scale = sdf->scale;
      p.sdf.coarse = static_cast<const float2*>(sdf->coarse_minmax);
      p.sdf.c0 = egx_ceil_div(sdf->d0, 4); p.sdf.c1 = egx_ceil_div(sdf->d1, 4); p.sdf.c2 = egx_ceil_div(sdf->d2, 4);
      EGX_HIP_CHECK(hipMemsetAsync(out_pene_count, 0, (size_t)B * sizeof(int32_t), stream));
    }
    const int per = (p.nbg + 7) / 8;
    const int grid = (p.nbg >= 8) ? 8 * per * m->NVT : p.nbg * m->NVT;
    const size_t lds = out_verts ? 4 * 32 * 97 * sizeof(float) : 0;
    hipEvent_t ev0 = g_prof_start, ev1 = g_prof_stop;
    g_prof_start = g_prof_stop = nullptr;
    if (ev0) EGX_HIP_CHECK(hipEventRecord(ev0, stream));
    if (out_verts && sdf)
      hipLaunchKernelGGL((egx_lbs_fused_kernel<true, true>), dim3(grid), dim3(256), lds, stream, p);
    else if (out_verts)
      hipLaunchKernelGGL((egx_lbs_fused_kernel<true, false>), dim3(grid), dim3(256), lds, stream, p);
    else if (sdf)
      hipLaunchKernelGGL((egx_lbs_fused_kernel<false, true>), dim3(grid), dim3(256), lds, stream, p);
    else
      hipLaunchKernelGGL((egx_lbs_fused_kernel<false, false>), dim3(grid), dim3(256), lds, stream, p);
    if (ev1) EGX_HIP_CHECK(hipEventRecord(ev1, stream));
  }
  if (need_picks)
    hipLaunchKernelGGL(egx_gather_kernel, dim3(B), dim3(256), 0, stream, picked, B, m->NP, m->M, m->marker_slot,
                       m->extra_slot, m->lmk_slot, m->lmk_bary, out_joints, out_markers);
  EGX_HIP_CHECK(hipGetLastError());
  return EGX_OK;
}
