// Per-agent tail of CrowdEnv.step / CrowdEnv.reset as fused kernels (one workgroup per agent).
//
// Reference: motion/crowd_ppo/crowd_env_2f.py:116-317 (room/SDF env), crowd_env_2f_box.py:116-340 (box env),
// _get_feature :680-727, _calc_egosensing :524-613, _blend_params :729-739, get_map
// (exp_GAMMAPrimitive/utils/batch_gen_amass.py:934-968), CanonicalCoordinateExtractor.get_new_coordinate_torch
// (models/baseops.py:214-225), SMPLXParser.update_transl_glorot (baseops.py:537-598), scene samplers
// (exp_GAMMAPrimitive/utils/environments.py:65-335, 371-627).  The reference runs ~150 tiny torch kernels,
// ~15 host syncs and 64 shapely ray casts per agent-step here; this file does it in one launch with no sync.
#include "egx_common.h"

namespace {

constexpr int NT = 20, THIS = 2, NM = 67, NJO = EGX_NUM_JOINTS_OUT, SD = 402, XB = EGX_XB_DIM;
constexpr int NRAY = 32;
constexpr int BLK = 128;

__device__ __forceinline__ float block_reduce(float v, float* sh, int op /*0 sum,1 min,2 max*/) {
  // BLK threads; sh has >= BLK floats
  sh[threadIdx.x] = v;
  __syncthreads();
  for (int s = BLK / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
      const float a = sh[threadIdx.x], b = sh[threadIdx.x + s];
      sh[threadIdx.x] = op == 0 ? a + b : (op == 1 ? fminf(a, b) : fmaxf(a, b));
    }
    __syncthreads();
  }
  const float r = sh[0];
  __syncthreads();
  return r;
}

__device__ __forceinline__ void mat3_mul(const float* A, const float* B, float* C) {  // C = A B
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) C[r * 3 + c] = A[r * 3 + 0] * B[0 * 3 + c] + A[r * 3 + 1] * B[1 * 3 + c] + A[r * 3 + 2] * B[2 * 3 + c];
}
__device__ __forceinline__ void mat3T_vec(const float* A, const float* v, float* o) {  // o = A^T v
  for (int r = 0; r < 3; ++r) o[r] = A[0 * 3 + r] * v[0] + A[1 * 3 + r] * v[1] + A[2 * 3 + r] * v[2];
}
__device__ __forceinline__ void mat3_vec(const float* A, const float* v, float* o) {
  for (int r = 0; r < 3; ++r) o[r] = A[r * 3 + 0] * v[0] + A[r * 3 + 1] * v[1] + A[r * 3 + 2] * v[2];
}

// CanonicalCoordinateExtractor.get_new_coordinate_torch: x = j2-j1 (z zeroed, normalised WITHOUT eps),
// z = (0,0,1), y = normalise(z x x); R = [x y z] as columns; T = j0
__device__ __forceinline__ void canonical_frame(const float* j0, const float* j1, const float* j2, float* R, float* T) {
  float x0 = j2[0] - j1[0], x1 = j2[1] - j1[1];
  const float nx = sqrtf(x0 * x0 + x1 * x1);
  x0 /= nx; x1 /= nx;
  float y0 = -x1, y1 = x0;  // (0,0,1) x (x0,x1,0)
  const float ny = sqrtf(y0 * y0 + y1 * y1);
  y0 /= ny; y1 /= ny;
  R[0] = x0; R[1] = y0; R[2] = 0.f;
  R[3] = x1; R[4] = y1; R[5] = 0.f;
  R[6] = 0.f; R[7] = 0.f; R[8] = 1.f;
  T[0] = j0[0]; T[1] = j0[1]; T[2] = j0[2];
}

// SMPLXParser.update_transl_glorot (torch branch): xb[0:6] -> new transl / glorot for frame (R,T), delta = rest root
__device__ __forceinline__ void update_transl_glorot(const float* R, const float* T, const float* delta, const float* xb, float* out6) {
  float go[9], gn[9], Rt[9];
  egx_tgm_aa_to_rotmat(xb + 3, go);
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) Rt[r * 3 + c] = R[c * 3 + r];
  mat3_mul(Rt, go, gn);
  egx_tgm_rotmat_to_aa(gn, out6 + 3);
  const float v[3] = {xb[0] + delta[0] - T[0], xb[1] + delta[1] - T[1], xb[2] + delta[2] - T[2]};
  float o[3];
  mat3T_vec(R, v, o);
  out6[0] = o[0] - delta[0]; out6[1] = o[1] - delta[1]; out6[2] = o[2] - delta[2];
}

struct SceneDev {
  const float* edges;   // [E,4]
  const int* edge_off;  // [S+1]
  const float* tris;    // [F,6]
  const int* tri_off;   // [S+1]
  const float* floor_h; // [S]
  const float* map_lin; // [res]
  int map_res;
  // crowd scenes (main_crowd_eval): walkable = square floor minus the world-space marker boxes of the OTHER members
  float* crowd_bbox;    // [G][S][4] (minx,miny,maxx,maxy) or null
  int crowd_G, crowd_S, crowd_k;
  float crowd_half;     // floor is [-half, half]^2 (crowd_env_crowd_eval.py:391)
  int crowd_polygon;    // 1: the exterior is the ring set edges[edge_off[0] .. edge_off[1]) instead of the square floor
                        //    (crowd_env_egobody_eval.py:402 scene_poly = walkable region of the scene's navmesh)
  int crowd_static;     // 1: the other members' boxes are NOT holes (what shapely's Polygon(polygon, holes) evaluates to in
                        //    crowd_env_egobody_eval.py:824)
};

// even-odd containment of (x, y) in the ring set edges[e0, e1), float64 like shapely's predicates on these inputs
__device__ __forceinline__ bool point_in_rings(const float* edges, int e0, int e1, double x, double y) {
  int crossings = 0;
  for (int e = e0; e < e1; ++e) {
    const float* q = edges + (size_t)e * 4;
    const double x0 = q[0], y0 = q[1], x1 = q[2], y1 = q[3];
    if ((y0 > y) != (y1 > y)) {
      const double xint = x0 + (y - y0) * (x1 - x0) / (y1 - y0);
      if (x < xint) ++crossings;
    }
  }
  return (crossings & 1) == 1;
}

// number of boundary edges of scene `sc_i` and the e-th edge (x0,y0,x1,y1) in float64
__device__ __forceinline__ int scene_num_edges(const SceneDev& sc, int sc_i) {
  if (sc.crowd_bbox)
    return (sc.crowd_polygon ? sc.edge_off[1] - sc.edge_off[0] : 4) + (sc.crowd_static ? 0 : 4 * (sc.crowd_G - 1));
  return sc.edge_off[sc_i + 1] - sc.edge_off[sc_i];
}
__device__ __forceinline__ void scene_edge(const SceneDev& sc, int sc_i, int e, double& x0, double& y0, double& x1, double& y1) {
  if (!sc.crowd_bbox) {
    const float* p = sc.edges + (size_t)(sc.edge_off[sc_i] + e) * 4;
    x0 = p[0]; y0 = p[1]; x1 = p[2]; y1 = p[3];
    return;
  }
  const int n_ext = sc.crowd_polygon ? sc.edge_off[1] - sc.edge_off[0] : 4;
  if (sc.crowd_polygon && e < n_ext) {
    const float* p = sc.edges + (size_t)(sc.edge_off[0] + e) * 4;
    x0 = p[0]; y0 = p[1]; x1 = p[2]; y1 = p[3];
    return;
  }
  float lo[2], hi[2];
  int q;
  if (e < n_ext) {
    lo[0] = lo[1] = -sc.crowd_half; hi[0] = hi[1] = sc.crowd_half;
    q = e;
  } else {
    e -= n_ext - 4;
    int other = (e - 4) >> 2;
    if (other >= sc.crowd_k) ++other;  // skip this member's own box
    const float* b = sc.crowd_bbox + ((size_t)other * sc.crowd_S + sc_i) * 4;
    lo[0] = b[0]; lo[1] = b[1]; hi[0] = b[2]; hi[1] = b[3];
    q = (e - 4) & 3;
  }
  // rectangle corners c0=(lo,lo) c1=(hi,lo) c2=(hi,hi) c3=(lo,hi); edge q = c_q -> c_{q+1}
  const double cx[4] = {lo[0], hi[0], hi[0], lo[0]}, cy[4] = {lo[1], lo[1], hi[1], hi[1]};
  x0 = cx[q]; y0 = cy[q]; x1 = cx[(q + 1) & 3]; y1 = cy[(q + 1) & 3];
}

// even-odd containment + first exit along 32 rays, float64 like the reference's numpy/shapely path.
// eye/look are computed in float32 first (joint.detach().cpu().numpy() is float32), then promoted.
__device__ void egosensing_frame(const SceneDev& sc, int scene, const float* j23, const float* j24, const float* j56,
                                 const float* j57, double ray_len, float* out32 /*[32]*/, int tid, int nthreads) {
  const float lx = j57[0] - j23[0] + j56[0] - j24[0];
  const float ly = j57[1] - j23[1] + j56[1] - j24[1];
  double la0 = (double)lx, la1 = (double)ly;
  const double ln = sqrt(la0 * la0 + la1 * la1);
  la0 /= ln; la1 /= ln;
  const float exf = (j23[0] + j24[0]) / 2.f, eyf = (j23[1] + j24[1]) / 2.f;
  const double ox = (double)exf, oy = (double)eyf;
  const int ne = scene_num_edges(sc, scene);
  // containment (every thread computes it redundantly; E is small)
  int crossings = 0;
  for (int e = 0; e < ne; ++e) {
    double x0, y0, x1, y1;
    scene_edge(sc, scene, e, x0, y0, x1, y1);
    if ((y0 > oy) != (y1 > oy)) {
      const double xint = x0 + (oy - y0) * (x1 - x0) / (y1 - y0);
      if (ox < xint) ++crossings;
    }
  }
  const bool inside = (crossings & 1) == 1;
  const double PI = 3.14159265358979323846;
  for (int i = tid; i < NRAY; i += nthreads) {
    double d = 0.0;
    if (inside) {
      // np.linspace(-pi/2, pi/2, 32): start + i*step, last sample exactly the stop value
      const double ang = (i == NRAY - 1) ? PI / 2 : -PI / 2 + (double)i * (PI / (double)(NRAY - 1));
      const double ca = cos(ang), sa = sin(ang);
      const double dx = la0 * ca - la1 * sa, dy = la1 * ca + la0 * sa;
      d = ray_len;
      for (int e = 0; e < ne; ++e) {
        double ex0, ey0, ex1, ey1;
        scene_edge(sc, scene, e, ex0, ey0, ex1, ey1);
        const double edx = ex1 - ex0, edy = ey1 - ey0;
        const double den = dx * edy - dy * edx;
        if (fabs(den) > 0.0) {
          const double tt = ((ex0 - ox) * edy - (ey0 - oy) * edx) / den;
          const double uu = ((ex0 - ox) * dy - (ey0 - oy) * dx) / den;
          if (tt > 0.0 && tt <= ray_len && uu >= 0.0 && uu <= 1.0 && tt < d) d = tt;
        }
      }
      const double px = ox + dx * d - ox, py = oy + dy * d - oy;  // |end - eye| from the returned coordinates
      d = sqrt(px * px + py * py);
    }
    out32[i] = -1.f + 2.f * (float)(d / ray_len);
  }
}

// same-side test of a scene-frame point against navmesh triangles [f0, f1) (batch_gen_amass.py:949-961)
__device__ __forceinline__ bool point_in_navmesh(const float* tris, int f0, int f1, float px, float py) {
  bool walk = false;
  for (int f = f0; f < f1 && !walk; ++f) {
    const float* t = tris + (size_t)f * 6;
    const float d1 = (px - t[2]) * (t[1] - t[3]) - (t[0] - t[2]) * (py - t[3]);
    const float d2 = (px - t[4]) * (t[3] - t[5]) - (t[2] - t[4]) * (py - t[5]);
    const float d3 = (px - t[0]) * (t[5] - t[1]) - (t[4] - t[0]) * (py - t[1]);
    const bool neg = (d1 < 0.f) || (d2 < 0.f) || (d3 < 0.f);
    const bool pos = (d1 > 0.f) || (d2 > 0.f) || (d3 > 0.f);
    walk = !(neg && pos);
  }
  return walk;
}

// get_map (batch_gen_amass.py:934-968) for one point of the local grid: local (lx, ly, 0) -> scene frame, then the same-side
// test against every navmesh triangle (crowd scenes: floor square minus the other members' boxes)
__device__ __forceinline__ bool walk_cell(const SceneDev& sc, int scene, const float* R0, const float* T0, float lx, float ly,
                                          float* world_xy /*[2] or null*/) {
  // einsum('bij,bpj->bpi') with the z component of the local point = 0
  const float px = (R0[0] * lx + R0[1] * ly + R0[2] * 0.f) + T0[0];
  const float py = (R0[3] * lx + R0[4] * ly + R0[5] * 0.f) + T0[1];
  if (world_xy) { world_xy[0] = px; world_xy[1] = py; }
  bool walk = false;
  if (sc.crowd_bbox) {  // crowd_env_crowd_eval.py:742-764: polygon(floor, holes).contains(point), z forced to 0
    walk = sc.crowd_polygon ? point_in_rings(sc.edges, sc.edge_off[0], sc.edge_off[1], (double)px, (double)py)
                            : (fabsf(px) < sc.crowd_half) && (fabsf(py) < sc.crowd_half);
    for (int o = 0; o < sc.crowd_G && walk && !sc.crowd_static; ++o) {
      if (o == sc.crowd_k) continue;
      const float* b = sc.crowd_bbox + ((size_t)o * sc.crowd_S + scene) * 4;
      if (px >= b[0] && px <= b[2] && py >= b[1] && py <= b[3]) walk = false;
    }
    return walk;
  }
  return point_in_navmesh(sc.tris, sc.tri_off[scene], sc.tri_off[scene + 1], px, py);
}

// walkability of the res x res local grid (1 / -1) of one agent reduced to the bbox penetration count
__device__ float walk_map_penalty(const SceneDev& sc, int scene, const float* R0, const float* T0, float bminx, float bminy,
                                  float bmaxx, float bmaxy, float* sh) {
  const int res = sc.map_res;
  float cnt = 0.f;
  for (int p = threadIdx.x; p < res * res; p += BLK) {
    const float lx = sc.map_lin[p / res], ly = sc.map_lin[p % res];
    const bool walk = walk_cell(sc, scene, R0, T0, lx, ly, nullptr);
    const bool inbox = (lx >= bminx) && (ly >= bminy) && (lx <= bmaxx) && (ly <= bmaxy);
    if (inbox && !walk) cnt += 1.f;  // inside * (1 - (-1)) * 0.5
  }
  return block_reduce(cnt, sh, 0);
}

// _get_feature (crowd_env_2f.py:680-727), the part the observation keeps: unit vector from a canonical marker to the target
__device__ __forceinline__ void marker_target_feature(const float* target_l, const float* marker, float* out3) {
  const float fx = target_l[0] - marker[0], fy = target_l[1] - marker[1], fz = target_l[2] - marker[2];
  const float dn = fmaxf(sqrtf(fx * fx + fy * fy + fz * fz), 1e-12f);
  out3[0] = fx / dn; out3[1] = fy / dn; out3[2] = fz / dn;
}
// ... and the clipped 3-D distance pelvis -> target
__device__ __forceinline__ float target_distance(const float* target_l, const float* pelvis) {
  const float fx = target_l[0] - pelvis[0], fy = target_l[1] - pelvis[1], fz = target_l[2] - pelvis[2];
  return fmaxf(sqrtf(fx * fx + fy * fy + fz * fz), 1e-12f);
}

struct EnvCfg {
  float reproj_factor, goal_thresh, pene_thres;
  float w_skate, w_floor, w_face, w_look, w_success, w_target_dist, w_pene, w_vp;
  int max_depth, scene_kind /*0 sdf, 1 box*/, terminate_on_pene, pene_body;
  float ray_len;
  float vp_thresh;      // mean VPoser-embedding norm above which a pose counts as unrealistic (11; EgoBody eval: 14)
  int no_goal_term;     // 1: reaching the goal does not end the episode (crowd_env_egobody_eval.py:378)
};

struct StepArgs {
  EnvCfg cfg;
  SceneDev sc;
  int A;
  // state (updated in place)
  float* state;      // [A,2,402]
  float* seed;       // [A,2,93]
  float* R0;         // [A,9]
  float* T0;         // [A,3]
  float* dist;       // [A]
  int* steps;        // [A]
  const float* wpath;     // [A,2,3]
  const int* scene_idx;   // [A]
  // step intermediates
  const float* Y_gen;        // [18,A,201]
  const float* pred_params;  // [A,20,93]
  const float* joints;       // [A*20,127,3]
  const float* markers_proj; // [A*20,67,3]
  const int* pene_count;     // [A*20] or null
  const float* vp_emb;       // [A*20,32]
  const int* feet_marker_idx;  // [6]
  // outputs
  float* reward;       // [A]
  int* terminated;     // [A]
  float* rterms;       // [A,8] or null: skate, floor, face, look, goal, target_dist, pene, vp
  int* nonfinite;      // device counter or null
  int* invalid;        // [A] or null: |= 1 pelvis left the scene polygon in the first steps, |= 2 unrealistic pose
  float* obs_ego;      // [A,2,32]
  float* obs_dist;     // [A]
  float* obs_time;     // [A]
  float* out_marker_b; // [A,20,67,3] or null
  float* out_prev_frame;  // [A,12] (R0,T0 before the update) or null
};

}  // namespace

// pred_params[a,t,:] = seed / regressed frames with _blend_params applied (crowd_env_2f.py:116-120,729-739)
__global__ void egx_assemble_params_kernel(const float* __restrict__ seed, const float* __restrict__ Yb_gen, int A,
                                           float* __restrict__ pred_params) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= A * XB) return;
  const int a = idx / XB, d = idx % XB;
  float* out = pred_params + (size_t)a * NT * XB + d;
  const float p0 = seed[((size_t)a * 2 + 0) * XB + d], p1 = seed[((size_t)a * 2 + 1) * XB + d];
  out[0 * XB] = p0;
  out[1 * XB] = p1;
  float g[5];
  for (int t = 0; t < 18; ++t) {
    const float v = Yb_gen[((size_t)t * A + a) * XB + d];
    if (t < 3) g[t + 2] = v;
    out[(t + 2) * XB] = v;
  }
  if (d >= 6) {
    const float b2 = (p1 + g[3]) / 2.0f;
    const float b3 = (b2 + g[4]) / 2.0f;
    out[2 * XB] = b2;
    out[3 * XB] = b3;
  }
}

__global__ __launch_bounds__(BLK) void egx_env_step_post_kernel(StepArgs p) {
  __shared__ float s_mb[NT][NM * 3];   // blended markers, canonical (old) frame
  __shared__ float s_red[BLK];
  __shared__ float s_newseed[2][6];
  __shared__ float s_frame[24];        // R_[9], T_[3], R0new[9], T0new[3]
  __shared__ float s_scal[16];
  const int a = blockIdx.x, tid = threadIdx.x;
  const EnvCfg& c = p.cfg;
  const float* R0 = p.R0 + (size_t)a * 9;
  const float* T0 = p.T0 + (size_t)a * 3;
  float R0o[9], T0o[3];
  for (int e = 0; e < 9; ++e) R0o[e] = R0[e];
  for (int e = 0; e < 3; ++e) T0o[e] = T0[e];
  const float* J = p.joints + (size_t)a * NT * NJO * 3;
  const float* PP = p.pred_params + (size_t)a * NT * XB;
  const float wp[3] = {p.wpath[(size_t)a * 6 + 3], p.wpath[(size_t)a * 6 + 4], p.wpath[(size_t)a * 6 + 5]};

  // ---- 1. blended markers (reproj_factor * projected + (1-rf) * predicted) ------------------------------
  const float rf = c.reproj_factor;
  for (int i = tid; i < NT * NM * 3; i += BLK) {
    const int t = i / (NM * 3), d = i % (NM * 3);
    const float pred = (t < THIS) ? p.state[((size_t)a * 2 + t) * SD + d] : p.Y_gen[((size_t)(t - 2) * p.A + a) * (NM * 3) + d];
    const float proj = p.markers_proj[((size_t)a * NT + t) * NM * 3 + d];
    const float v = rf * proj + (1.f - rf) * pred;
    s_mb[t][d] = v;
    if (p.out_marker_b) p.out_marker_b[((size_t)a * NT + t) * NM * 3 + d] = v;
  }
  __syncthreads();

  // ---- 2. skate + floor over the 6 feet markers ----------------------------------------------------------
  float part = 0.f;
  if (tid < 18) {  // central difference frames 1..18
    float mn = 3.4e38f;
    for (int k = 0; k < 6; ++k) {
      const int m = p.feet_marker_idx[k];
      const float dx = s_mb[tid + 2][m * 3 + 0] - s_mb[tid][m * 3 + 0];
      const float dy = s_mb[tid + 2][m * 3 + 1] - s_mb[tid][m * 3 + 1];
      const float dz = s_mb[tid + 2][m * 3 + 2] - s_mb[tid][m * 3 + 2];
      const float sp = sqrtf(dx * dx + dy * dy + dz * dz) / 2.f / (1.f / 40.f);
      mn = fminf(mn, sp);
    }
    part = fmaxf(mn - 0.075f, 0.f);
  }
  const float r_skate = expf(-(block_reduce(part, s_red, 0) / 18.f));
  part = 0.f;
  if (tid < NT) {
    float mn = 3.4e38f;
    for (int k = 0; k < 6; ++k) {
      const int m = p.feet_marker_idx[k];
      const float z = (R0o[6] * s_mb[tid][m * 3 + 0] + R0o[7] * s_mb[tid][m * 3 + 1] + R0o[8] * s_mb[tid][m * 3 + 2]) + T0o[2];
      mn = fminf(mn, z);
    }
    part = fabsf(mn - 0.02f);
  }
  const float r_floor = expf(-(block_reduce(part, s_red, 0) / (float)NT));

  // ---- 3. vposer norm ------------------------------------------------------------------------------------
  part = 0.f;
  if (tid < NT) {
    const float* e = p.vp_emb + ((size_t)a * NT + tid) * 32;
    float s = 0.f;
    for (int k = 0; k < 32; ++k) s += e[k] * e[k];
    part = sqrtf(s);
  }
  const float vp_norm = block_reduce(part, s_red, 0) / (float)NT;
  const float r_vp = (vp_norm > c.vp_thresh) ? 0.f : 0.05f;

  // ---- 4. SDF penetration (room env) -----------------------------------------------------------------------
  float r_pene = 0.f;
  bool penetration = false;
  if (c.scene_kind == 0) {
    float cs = 0.f, cm = 0.f;
    if (tid < NT) {
      cs = (float)p.pene_count[(size_t)a * NT + tid];
      cm = cs;
    }
    const float total = block_reduce(cs, s_red, 0);
    const float mx = block_reduce(cm, s_red, 2);
    r_pene = expf(-(total / (float)NT / 10.f));
    penetration = !(mx < 40.f);
  }

  // ---- 5. scalar terms + re-canonicalisation (thread 0) ---------------------------------------------------
  if (tid == 0) {
    const float* j19 = J + (size_t)19 * NJO * 3;
    const float* j18 = J + (size_t)18 * NJO * 3;
    // facing
    float xa0 = j19[2 * 3 + 0] - j19[1 * 3 + 0], xa1 = j19[2 * 3 + 1] - j19[1 * 3 + 1];
    float nrm = fmaxf(sqrtf(xa0 * xa0 + xa1 * xa1), 1e-12f);
    xa0 /= nrm; xa1 /= nrm;
    const float bo0 = -xa1, bo1 = xa0;  // (0,0,1) x x_axis, xy part
    float tl[3];
    {
      const float v[3] = {wp[0] - T0o[0], wp[1] - T0o[1], wp[2] - T0o[2]};
      mat3T_vec(R0o, v, tl);
    }
    float f0 = tl[0] - j19[0], f1 = tl[1] - j19[1];
    nrm = fmaxf(sqrtf(f0 * f0 + f1 * f1), 1e-12f);
    f0 /= nrm; f1 /= nrm;
    const float r_face = ((f0 * bo0 + f1 * bo1) + 1.f) / 2.0f;
    // looking: eye x-axis = reye(24) - leye(23)
    float e0 = j19[24 * 3 + 0] - j19[23 * 3 + 0], e1 = j19[24 * 3 + 1] - j19[23 * 3 + 1];
    nrm = fmaxf(sqrtf(e0 * e0 + e1 * e1), 1e-12f);
    e0 /= nrm; e1 /= nrm;
    const float r_look = ((f0 * (-e1) + f1 * e0) + 1.f) / 2.0f;
    // target distance
    const float d2t = target_distance(tl, j19);
    const float r_td = p.dist[a] - d2t;
    const float r_goal = (d2t < c.goal_thresh) ? 1.f : 0.f;
    // new frame from frame 18 joints
    float Rn[9], Tn[3];
    canonical_frame(j18 + 0, j18 + 3, j18 + 6, Rn, Tn);
    float T0n[3], R0n[9];
    mat3_vec(R0o, Tn, T0n);
    T0n[0] += T0o[0]; T0n[1] += T0o[1]; T0n[2] += T0o[2];
    mat3_mul(R0o, Rn, R0n);
    // rest root (calc_calibrate_offset): posed root joint minus transl, frame 18
    const float delta[3] = {j18[0] - PP[18 * XB + 0], j18[1] - PP[18 * XB + 1], j18[2] - PP[18 * XB + 2]};
    update_transl_glorot(Rn, Tn, delta, PP + 18 * XB, s_newseed[0]);
    update_transl_glorot(Rn, Tn, delta, PP + 19 * XB, s_newseed[1]);
    for (int e = 0; e < 9; ++e) { s_frame[e] = Rn[e]; s_frame[12 + e] = R0n[e]; }
    for (int e = 0; e < 3; ++e) { s_frame[9 + e] = Tn[e]; s_frame[21 + e] = T0n[e]; }
    s_scal[0] = r_face; s_scal[1] = r_look; s_scal[2] = r_td; s_scal[3] = r_goal; s_scal[4] = d2t;
  }
  __syncthreads();
  float Rn[9], Tn[3], R0n[9], T0n[3];
  for (int e = 0; e < 9; ++e) { Rn[e] = s_frame[e]; R0n[e] = s_frame[12 + e]; }
  for (int e = 0; e < 3; ++e) { Tn[e] = s_frame[9 + e]; T0n[e] = s_frame[21 + e]; }
  const float r_face = s_scal[0], r_look = s_scal[1], r_td = s_scal[2], r_goal = s_scal[3], d2t = s_scal[4];

  // target in the NEW frame (for the marker features)
  float tln[3];
  {
    const float v[3] = {wp[0] - T0n[0], wp[1] - T0n[1], wp[2] - T0n[2]};
    mat3T_vec(R0n, v, tln);
  }
  // ---- 6. new state: canonical markers of frames 18,19 + unit vectors to the target; bbox for the map ----
  float bminx = 3.4e38f, bminy = 3.4e38f, bmaxx = -3.4e38f, bmaxy = -3.4e38f;
  for (int i = tid; i < THIS * NM; i += BLK) {
    const int t = i / NM, m = i % NM;
    const float v[3] = {s_mb[18 + t][m * 3 + 0] - Tn[0], s_mb[18 + t][m * 3 + 1] - Tn[1], s_mb[18 + t][m * 3 + 2] - Tn[2]};
    float ms[3];
    mat3T_vec(Rn, v, ms);
    float* st = p.state + ((size_t)a * 2 + t) * SD;
    st[m * 3 + 0] = ms[0]; st[m * 3 + 1] = ms[1]; st[m * 3 + 2] = ms[2];
    marker_target_feature(tln, ms, st + 201 + m * 3);
    bool use = c.pene_body != 0;
    if (!use)
      for (int k = 0; k < 6; ++k) use |= (p.feet_marker_idx[k] == m);
    if (use) {
      bminx = fminf(bminx, ms[0]); bminy = fminf(bminy, ms[1]);
      bmaxx = fmaxf(bmaxx, ms[0]); bmaxy = fmaxf(bmaxy, ms[1]);
    }
  }
  // seed parameters in the new frame
  for (int i = tid; i < THIS * XB; i += BLK) {
    const int t = i / XB, d = i % XB;
    p.seed[((size_t)a * 2 + t) * XB + d] = (d < 6) ? s_newseed[t][d] : PP[(18 + t) * XB + d];
  }
  // ---- 7. box env: walkability map penalty -------------------------------------------------------------------
  if (c.scene_kind >= 1) {
    bminx = block_reduce(bminx, s_red, 1); bminy = block_reduce(bminy, s_red, 1);
    bmaxx = block_reduce(bmaxx, s_red, 2); bmaxy = block_reduce(bmaxy, s_red, 2);
    const int sc_i = p.sc.crowd_bbox ? a : p.scene_idx[a];
    const float num_pene = walk_map_penalty(p.sc, sc_i, R0n, T0n, bminx, bminy, bmaxx, bmaxy, s_red);
    penetration = num_pene > c.pene_thres;
    r_pene = penetration ? 0.f : 0.05f;
  }
  // ---- 8. egosensing from the world-frame eye joints of frames 18,19 -------------------------------------------
  {
    const int scene = p.sc.crowd_bbox ? a : (p.scene_idx ? p.scene_idx[a] : 0);
    for (int t = 0; t < THIS; ++t) {
      const float* jt = J + (size_t)(18 + t) * NJO * 3;
      float w23[3], w24[3], w56[3], w57[3];
      const int ids[4] = {23, 24, 56, 57};
      float* outs[4] = {w23, w24, w56, w57};
      for (int q = 0; q < 4; ++q) {
        mat3_vec(R0o, jt + ids[q] * 3, outs[q]);
        outs[q][0] += T0o[0]; outs[q][1] += T0o[1]; outs[q][2] += T0o[2];
      }
      egosensing_frame(p.sc, scene, w23, w24, w56, w57, (double)c.ray_len, p.obs_ego + ((size_t)a * 2 + t) * NRAY, tid, BLK);
    }
  }
  // ---- 8b. crowd scenes: publish this member's world-space marker box for the others ----------------------------
  if (p.sc.crowd_bbox) {
    __syncthreads();  // all rays of this agent are cast before its own box changes
    float wminx = 3.4e38f, wminy = 3.4e38f, wmaxx = -3.4e38f, wmaxy = -3.4e38f;
    for (int i = tid; i < THIS * NM; i += BLK) {
      const float* st = p.state + ((size_t)a * 2 + i / NM) * SD + (i % NM) * 3;
      const float wx = (R0n[0] * st[0] + R0n[1] * st[1] + R0n[2] * st[2]) + T0n[0];
      const float wy = (R0n[3] * st[0] + R0n[4] * st[1] + R0n[5] * st[2]) + T0n[1];
      wminx = fminf(wminx, wx); wminy = fminf(wminy, wy); wmaxx = fmaxf(wmaxx, wx); wmaxy = fmaxf(wmaxy, wy);
    }
    wminx = block_reduce(wminx, s_red, 1); wminy = block_reduce(wminy, s_red, 1);
    wmaxx = block_reduce(wmaxx, s_red, 2); wmaxy = block_reduce(wmaxy, s_red, 2);
    if (tid == 0) {
      float* b = p.sc.crowd_bbox + ((size_t)p.sc.crowd_k * p.sc.crowd_S + a) * 4;
      b[0] = wminx; b[1] = wminy; b[2] = wmaxx; b[3] = wmaxy;
    }
  }
  // ---- 9. reward, termination, scalars ------------------------------------------------------------------------
  if (tid == 0) {
    const float reward = r_skate * c.w_skate + r_floor * c.w_floor + r_face * c.w_face + r_look * c.w_look +
                         r_goal * c.w_success + r_td * c.w_target_dist + r_pene * c.w_pene + r_vp * c.w_vp;
    const int steps = p.steps[a] + 1;
    p.steps[a] = steps;
    bool term = (steps == c.max_depth) || (r_goal > 0.f && !c.no_goal_term);
    if (c.terminate_on_pene) term = term || penetration;
    if (p.invalid) {
      // the filters of crowd_env_egobody_eval.py:208-216,229-234 (the reference process exits; here the sequence is flagged)
      int bad = (vp_norm > c.vp_thresh) ? 2 : 0;
      if (steps < 6 && p.sc.crowd_bbox && p.sc.crowd_polygon) {
        for (int t = 0; t < NT && !(bad & 1); ++t) {
          const float* j0 = J + (size_t)t * NJO * 3;  // pelvis, canonical frame of this step
          const float wx = (R0o[0] * j0[0] + R0o[1] * j0[1] + R0o[2] * j0[2]) + T0o[0];
          const float wy = (R0o[3] * j0[0] + R0o[4] * j0[1] + R0o[5] * j0[2]) + T0o[1];
          if (!point_in_rings(p.sc.edges, p.sc.edge_off[0], p.sc.edge_off[1], (double)wx, (double)wy)) bad |= 1;
        }
      }
      if (bad) p.invalid[a] |= bad;
    }
    p.reward[a] = reward;
    if (p.nonfinite && !(isfinite(reward) && isfinite(d2t))) atomicAdd(p.nonfinite, 1);
    p.terminated[a] = term ? 1 : 0;
    p.dist[a] = d2t;
    p.obs_dist[a] = 1.f / (d2t + 1.f);
    p.obs_time[a] = 1.f - (float)steps / (float)c.max_depth;
    if (p.out_prev_frame) {
      for (int e = 0; e < 9; ++e) p.out_prev_frame[(size_t)a * 12 + e] = R0o[e];
      for (int e = 0; e < 3; ++e) p.out_prev_frame[(size_t)a * 12 + 9 + e] = T0o[e];
    }
    for (int e = 0; e < 9; ++e) p.R0[(size_t)a * 9 + e] = R0n[e];
    for (int e = 0; e < 3; ++e) p.T0[(size_t)a * 3 + e] = T0n[e];
    if (p.rterms) {
      float* r = p.rterms + (size_t)a * 8;
      r[0] = r_skate; r[1] = r_floor; r[2] = r_face; r[3] = r_look; r[4] = r_goal; r[5] = r_td; r[6] = r_pene; r[7] = r_vp;
    }
  }
}

// ------------------------------------------------------------------------------------------------------
// reset: scene sampler + canonicalisation + features, for agents whose mask is set
// ------------------------------------------------------------------------------------------------------
namespace {
struct ResetArgs {
  EnvCfg cfg;
  SceneDev sc;
  int A, K;               // K candidates per agent (first accepted wins; last one is taken regardless)
  const int* mask;        // [A] 1 = reset this agent (null = all)
  // candidates
  const float* cand_pairs;  // [A,K,2,3] start, target
  const float* cand_yaw;    // [A,K] final yaw jitter (box sampler) or null
  const int* cand_variant;  // [A,K] motion-seed variant or null (=0)
  const int* cand_scene;    // [A,K] scene per candidate or null
  const int* cand_valid;    // [A,K] precomputed acceptance (SDF env) or null
  // motion-seed tables per variant (identity global orient, zero transl)
  const float* tab_joints;   // [NV,2,127,3]
  const float* tab_markers;  // [NV,2,67,3]
  const float* tab_glorot;   // [NV,2,9]  rotation matrices of the mocap global orient
  const float* tab_transl;   // [NV,2,3]
  const float* tab_pose;     // [NV,2,63] body pose
  // state
  float* state; float* seed; float* R0; float* T0; float* dist; int* steps; float* wpath;
  int* scene_idx;
  // obs
  float* obs_ego; float* obs_dist; float* obs_time;
  int* out_choice;  // [A] chosen candidate or null
  int* pending;        // [A] or null: 1 = none of this launch's K candidates passed the start check (the last one was committed
                       // so that the state is well defined) - the caller re-launches with this array as the mask
  int* forced_count;   // [1] or null: += 1 for every agent this launch left pending (set on the caller's LAST round only)
};
}  // namespace

__global__ __launch_bounds__(BLK) void egx_env_reset_kernel(ResetArgs p) {
  __shared__ float s_red[BLK];
  __shared__ float s_f[64];
  const int a = blockIdx.x, tid = threadIdx.x;
  if (p.mask && p.mask[a] == 0) {
    if (tid == 0 && p.pending && p.pending != p.mask) p.pending[a] = 0;
    return;
  }
  const EnvCfg& c = p.cfg;
  for (int k = 0; k < p.K; ++k) {
    const int scene = p.sc.crowd_bbox ? a : (p.cand_scene ? p.cand_scene[(size_t)a * p.K + k] : (p.scene_idx ? p.scene_idx[a] : 0));
    const int v = p.cand_variant ? p.cand_variant[(size_t)a * p.K + k] : 0;
    const float* TJ = p.tab_joints + (size_t)v * 2 * NJO * 3;
    const float* TM = p.tab_markers + (size_t)v * 2 * NM * 3;
    const float* start = p.cand_pairs + ((size_t)a * p.K + k) * 6;
    const float* target = start + 3;
    // world placement of the mocap seed: x_w = Rg (x_loc - J0) + J0 + t   (J0 = rest root = tab joint 0 of frame f)
    float Rg[2][9], tg[2][3];   // live in thread 0 only
    auto world_joint = [&](int f, int j, float* o) {
      const float* J0 = TJ;
      const float* q = TJ + ((size_t)f * NJO + j) * 3;
      const float d[3] = {q[0] - J0[0], q[1] - J0[1], q[2] - J0[2]};
      mat3_vec(Rg[f], d, o);
      o[0] += J0[0] + tg[f][0]; o[1] += J0[1] + tg[f][1]; o[2] += J0[2] + tg[f][2];
    };
    if (tid == 0) {
      const float* J0 = TJ;  // root of frame 0 (identical for both frames: betas only)
      for (int f = 0; f < 2; ++f) {
        for (int e = 0; e < 9; ++e) Rg[f][e] = p.tab_glorot[((size_t)v * 2 + f) * 9 + e];
        for (int e = 0; e < 3; ++e) tg[f][e] = p.tab_transl[((size_t)v * 2 + f) * 3 + e];
      }
      auto apply_rot = [&](const float* Rm) {  // environments.py:233-237: rotate orient, rotate about the pelvis
        for (int f = 0; f < 2; ++f) {
          float nr[9];
          mat3_mul(Rm, Rg[f], nr);
          for (int e = 0; e < 9; ++e) Rg[f][e] = nr[e];
          const float q[3] = {J0[0] + tg[f][0], J0[1] + tg[f][1], J0[2] + tg[f][2]};
          float o[3];
          mat3_vec(Rm, q, o);
          tg[f][0] = o[0] - J0[0]; tg[f][1] = o[1] - J0[1]; tg[f][2] = o[2] - J0[2];
        }
      };
      // face the target (environments.py:216-232): Rodrigues from b_ori to target_ori
      float j1[3], j2[3];
      world_joint(0, 1, j1);
      world_joint(0, 2, j2);
      float x0 = j2[0] - j1[0], x1 = j2[1] - j1[1];
      float nrm = fmaxf(sqrtf(x0 * x0 + x1 * x1), 1e-12f);
      x0 /= nrm; x1 /= nrm;
      float bo[3] = {-x1, x0, 0.f};
      nrm = sqrtf(bo[0] * bo[0] + bo[1] * bo[1]);
      bo[0] /= nrm; bo[1] /= nrm;
      float to[3] = {target[0] - start[0], target[1] - start[1], target[2] - start[2]};
      nrm = sqrtf(to[0] * to[0] + to[1] * to[1] + to[2] * to[2]);
      to[0] /= nrm; to[1] /= nrm; to[2] /= nrm;
      const float vv[3] = {bo[1] * to[2] - bo[2] * to[1], bo[2] * to[0] - bo[0] * to[2], bo[0] * to[1] - bo[1] * to[0]};
      const float cc = bo[0] * to[0] + bo[1] * to[1] + bo[2] * to[2];
      const float ss = sqrtf(vv[0] * vv[0] + vv[1] * vv[1] + vv[2] * vv[2]);
      const float Km[9] = {0.f, -vv[2], vv[1], vv[2], 0.f, -vv[0], -vv[1], vv[0], 0.f};
      float K2[9], Rt[9];
      mat3_mul(Km, Km, K2);
      const float fac = (1.f - cc) / (ss * ss);
      for (int e = 0; e < 9; ++e) Rt[e] = ((e % 4 == 0) ? 1.f : 0.f) + Km[e] + K2[e] * fac;
      apply_rot(Rt);
      if (p.cand_yaw) {
        const float th = p.cand_yaw[(size_t)a * p.K + k];
        const float Rz[9] = {cosf(th), -sinf(th), 0.f, sinf(th), cosf(th), 0.f, 0.f, 0.f, 1.f};
        apply_rot(Rz);
      }
      // frame 0's placement for the whole block: the lowest of its 127 joints is found one joint per thread
      for (int e = 0; e < 9; ++e) s_f[42 + e] = Rg[0][e];
      for (int e = 0; e < 3; ++e) s_f[51 + e] = tg[0][e];
    }
    __syncthreads();
    float zmin;
    {
      float zl = 3.4e38f;
      if (tid != 0) {
        for (int e = 0; e < 9; ++e) Rg[0][e] = s_f[42 + e];
        for (int e = 0; e < 3; ++e) tg[0][e] = s_f[51 + e];
      }
      for (int j = tid; j < NJO; j += BLK) {
        float q[3];
        world_joint(0, j, q);
        zl = fminf(zl, q[2]);
      }
      zmin = block_reduce(zl, s_red, 1);
    }
    if (tid == 0) {
      // pelvis over the start, lowest joint of frame 0 on the floor (environments.py:240-247)
      float jr[3];
      world_joint(0, 0, jr);
      for (int f = 0; f < 2; ++f) {
        tg[f][0] += -jr[0] + start[0]; tg[f][1] += -jr[1] + start[1]; tg[f][2] += -zmin + start[2];
      }
      // wpath
      float w0[3];
      world_joint(0, 0, w0);
      // canonical frame from frame-0 world joints (crowd_env_2f.py:629-633)
      float wj1[3], wj2[3], R0n[9], T0n[3];
      world_joint(0, 1, wj1);
      world_joint(0, 2, wj2);
      canonical_frame(w0, wj1, wj2, R0n, T0n);
      // stash: R0n, T0n, per-frame Rg / tg, wpath
      for (int e = 0; e < 9; ++e) s_f[e] = R0n[e];
      for (int e = 0; e < 3; ++e) s_f[9 + e] = T0n[e];
      for (int f = 0; f < 2; ++f) {
        for (int e = 0; e < 9; ++e) s_f[12 + f * 12 + e] = Rg[f][e];
        for (int e = 0; e < 3; ++e) s_f[12 + f * 12 + 9 + e] = tg[f][e];
      }
      s_f[36] = w0[0]; s_f[37] = w0[1]; s_f[38] = w0[2];
      s_f[39] = target[0]; s_f[40] = target[1]; s_f[41] = w0[2];
    }
    __syncthreads();
    float R0n[9], T0n[3];   // Rg / tg: every thread now takes the final placement of both frames
    for (int e = 0; e < 9; ++e) R0n[e] = s_f[e];
    for (int e = 0; e < 3; ++e) T0n[e] = s_f[9 + e];
    for (int f = 0; f < 2; ++f) {
      for (int e = 0; e < 9; ++e) Rg[f][e] = s_f[12 + f * 12 + e];
      for (int e = 0; e < 3; ++e) tg[f][e] = s_f[12 + f * 12 + 9 + e];
    }
    const float wp[3] = {s_f[39], s_f[40], s_f[41]};
    const float* J0 = TJ;
    auto to_canonical = [&](int f, const float* q, float* o) {
      const float d[3] = {q[0] - J0[0], q[1] - J0[1], q[2] - J0[2]};
      float w[3];
      mat3_vec(Rg[f], d, w);
      w[0] += J0[0] + tg[f][0] - T0n[0]; w[1] += J0[1] + tg[f][1] - T0n[1]; w[2] += J0[2] + tg[f][2] - T0n[2];
      mat3T_vec(R0n, w, o);
    };
    float tl[3];
    {
      const float d[3] = {wp[0] - T0n[0], wp[1] - T0n[1], wp[2] - T0n[2]};
      mat3T_vec(R0n, d, tl);
    }
    // acceptance test
    bool accept = true;
    float bminx = 3.4e38f, bminy = 3.4e38f, bmaxx = -3.4e38f, bmaxy = -3.4e38f;
    for (int i = tid; i < 2 * NM; i += BLK) {
      float ms[3];
      to_canonical(i / NM, TM + (size_t)i * 3, ms);
      bminx = fminf(bminx, ms[0]); bminy = fminf(bminy, ms[1]);
      bmaxx = fmaxf(bmaxx, ms[0]); bmaxy = fmaxf(bmaxy, ms[1]);
    }
    if (c.scene_kind == 1) {
      bminx = block_reduce(bminx, s_red, 1); bminy = block_reduce(bminy, s_red, 1);
      bmaxx = block_reduce(bmaxx, s_red, 2); bmaxy = block_reduce(bmaxy, s_red, 2);
      accept = walk_map_penalty(p.sc, scene, R0n, T0n, bminx, bminy, bmaxx, bmaxy, s_red) == 0.f;
    } else if (c.scene_kind == 2) {
      accept = true;  // fixed start/target, no rejection loop (crowd_env_crowd_eval.py:388-435)
    } else if (p.cand_valid) {
      accept = p.cand_valid[(size_t)a * p.K + k] != 0;
    }
    if (!accept && k + 1 < p.K) {
      __syncthreads();
      continue;
    }
    // ---- commit this candidate ---------------------------------------------------------------------------
    for (int i = tid; i < 2 * NM; i += BLK) {
      const int t = i / NM, m = i % NM;
      float ms[3];
      to_canonical(t, TM + (size_t)i * 3, ms);
      float* st = p.state + ((size_t)a * 2 + t) * SD;
      st[m * 3 + 0] = ms[0]; st[m * 3 + 1] = ms[1]; st[m * 3 + 2] = ms[2];
      marker_target_feature(tl, ms, st + 201 + m * 3);
    }
    // seed params in the canonical frame: transl' = R0^T (t + J0 - T0) - J0 ; glorot' = aa(R0^T Rg)
    if (tid < 2) {
      const int f = tid;
      float* sd = p.seed + ((size_t)a * 2 + f) * XB;
      const float q[3] = {tg[f][0] + J0[0] - T0n[0], tg[f][1] + J0[1] - T0n[1], tg[f][2] + J0[2] - T0n[2]};
      float o[3];
      mat3T_vec(R0n, q, o);
      sd[0] = o[0] - J0[0]; sd[1] = o[1] - J0[1]; sd[2] = o[2] - J0[2];
      float Rt[9], gn[9];
      for (int r = 0; r < 3; ++r)
        for (int cc2 = 0; cc2 < 3; ++cc2) Rt[r * 3 + cc2] = R0n[cc2 * 3 + r];
      mat3_mul(Rt, Rg[f], gn);
      egx_tgm_rotmat_to_aa(gn, sd + 3);
      for (int e = 0; e < 63; ++e) sd[6 + e] = p.tab_pose[((size_t)v * 2 + f) * 63 + e];
      for (int e = 69; e < XB; ++e) sd[e] = 0.f;
    }
    // egosensing from the world eye joints
    for (int t = 0; t < 2; ++t) {
      float w[4][3];
      const int ids[4] = {23, 24, 56, 57};
      for (int q = 0; q < 4; ++q) {
        const float* src = TJ + ((size_t)t * NJO + ids[q]) * 3;
        const float d[3] = {src[0] - J0[0], src[1] - J0[1], src[2] - J0[2]};
        mat3_vec(Rg[t], d, w[q]);
        w[q][0] += J0[0] + tg[t][0]; w[q][1] += J0[1] + tg[t][1]; w[q][2] += J0[2] + tg[t][2];
      }
      egosensing_frame(p.sc, scene, w[0], w[1], w[2], w[3], (double)c.ray_len, p.obs_ego + ((size_t)a * 2 + t) * NRAY, tid, BLK);
    }
    if (p.sc.crowd_bbox) {  // world box of the two seed frames' markers (crowd_env_crowd_eval.py:59-75)
      __syncthreads();
      float wminx = 3.4e38f, wminy = 3.4e38f, wmaxx = -3.4e38f, wmaxy = -3.4e38f;
      for (int i = tid; i < 2 * NM; i += BLK) {
        float ms[3];
        to_canonical(i / NM, TM + (size_t)i * 3, ms);
        const float wx = (R0n[0] * ms[0] + R0n[1] * ms[1] + R0n[2] * ms[2]) + T0n[0];
        const float wy = (R0n[3] * ms[0] + R0n[4] * ms[1] + R0n[5] * ms[2]) + T0n[1];
        wminx = fminf(wminx, wx); wminy = fminf(wminy, wy); wmaxx = fmaxf(wmaxx, wx); wmaxy = fmaxf(wmaxy, wy);
      }
      wminx = block_reduce(wminx, s_red, 1); wminy = block_reduce(wminy, s_red, 1);
      wmaxx = block_reduce(wmaxx, s_red, 2); wmaxy = block_reduce(wmaxy, s_red, 2);
      if (tid == 0) {
        float* b = p.sc.crowd_bbox + ((size_t)p.sc.crowd_k * p.sc.crowd_S + a) * 4;
        b[0] = wminx; b[1] = wminy; b[2] = wmaxx; b[3] = wmaxy;
      }
    }
    if (tid == 0) {
      float pel[3];
      to_canonical(0, TJ, pel);
      const float d = target_distance(tl, pel);
      p.dist[a] = d;
      p.obs_dist[a] = 1.f / (d + 1.f);
      p.obs_time[a] = 1.f;
      p.steps[a] = 0;
      for (int e = 0; e < 9; ++e) p.R0[(size_t)a * 9 + e] = R0n[e];
      for (int e = 0; e < 3; ++e) p.T0[(size_t)a * 3 + e] = T0n[e];
      float* w = p.wpath + (size_t)a * 6;
      w[0] = s_f[36]; w[1] = s_f[37]; w[2] = s_f[38]; w[3] = wp[0]; w[4] = wp[1]; w[5] = wp[2];
      if (p.out_choice) p.out_choice[a] = k;
      if (p.cand_scene && p.scene_idx) p.scene_idx[a] = scene;
      if (p.pending) p.pending[a] = accept ? 0 : 1;
      if (!accept && p.forced_count) atomicAdd(p.forced_count, 1);
    }
    break;
  }
}

// ------------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------------
static EnvCfg to_cfg(const egx_env_config* c) {
  EnvCfg o;
  o.reproj_factor = c->reproj_factor; o.goal_thresh = c->goal_thresh; o.pene_thres = c->pene_thres;
  o.w_skate = c->weight_skate; o.w_floor = c->weight_floor; o.w_face = c->weight_face_target; o.w_look = c->weight_look_target;
  o.w_success = c->weight_success; o.w_target_dist = c->weight_target_dist; o.w_pene = c->weight_pene; o.w_vp = c->weight_vp;
  o.max_depth = c->max_depth; o.scene_kind = c->scene_kind; o.terminate_on_pene = c->terminate_on_penetration;
  o.pene_body = c->pene_type_body; o.ray_len = c->ray_len;
  o.vp_thresh = c->vp_thresh > 0.f ? c->vp_thresh : 11.f;
  o.no_goal_term = c->no_goal_termination;
  return o;
}
static SceneDev to_scene(const egx_env_scenes* s) {
  SceneDev o;
  o.edges = s->edges; o.edge_off = s->edge_off; o.tris = s->tris; o.tri_off = s->tri_off; o.floor_h = s->floor_height;
  o.map_lin = s->map_lin; o.map_res = s->map_res;
  o.crowd_bbox = s->crowd_bbox; o.crowd_G = s->crowd_group; o.crowd_S = s->crowd_scenes; o.crowd_k = s->crowd_member;
  o.crowd_half = s->crowd_floor_half;
  o.crowd_polygon = s->crowd_polygon; o.crowd_static = s->crowd_static;
  return o;
}

extern "C" int egx_assemble_params(const float* seed, const float* Yb_gen, int A, float* pred_params, void* stream_) {
  EGX_REQUIRE(seed && Yb_gen && pred_params && A > 0, "bad arguments");
  hipLaunchKernelGGL(egx_assemble_params_kernel, dim3(egx_ceil_div(A * XB, 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream_), seed, Yb_gen, A, pred_params);
  EGX_HIP_CHECK(hipGetLastError());
  return EGX_OK;
}

extern "C" int egx_env_step_post(const egx_env_config* cfg, const egx_env_scenes* scenes, const egx_env_state* st,
                                 const egx_env_step_io* io, int A, void* stream_) {
  EGX_REQUIRE(cfg && scenes && st && io && A > 0, "bad arguments");
  EGX_REQUIRE(st->state && st->seed && st->R0 && st->T0 && st->dist && st->steps && st->wpath, "null state array");
  EGX_REQUIRE(io->Y_gen && io->pred_params && io->joints && io->markers_proj && io->vp_emb && io->feet_marker_idx &&
                  io->reward && io->terminated && io->obs_ego && io->obs_dist && io->obs_time, "null io array");
  EGX_REQUIRE(cfg->scene_kind == 2 ? (scenes->crowd_bbox && scenes->crowd_group >= 2 && scenes->crowd_scenes == A &&
                                      scenes->crowd_member >= 0 && scenes->crowd_member < scenes->crowd_group && scenes->map_lin &&
                                      (!scenes->crowd_polygon || (scenes->edges && scenes->edge_off)))
                                   : (scenes->edges && scenes->edge_off),
              "scene tables missing");
  EGX_REQUIRE(cfg->scene_kind == 0 ? (io->pene_count != nullptr)
                                   : (cfg->scene_kind == 2 || (scenes->tris && scenes->tri_off && scenes->map_lin && st->scene_idx)),
              "scene-kind specific inputs missing");
  StepArgs p;
  p.cfg = to_cfg(cfg); p.sc = to_scene(scenes); p.A = A;
  p.state = st->state; p.seed = st->seed; p.R0 = st->R0; p.T0 = st->T0; p.dist = st->dist; p.steps = st->steps;
  p.wpath = st->wpath; p.scene_idx = st->scene_idx;
  p.Y_gen = io->Y_gen; p.pred_params = io->pred_params; p.joints = io->joints; p.markers_proj = io->markers_proj;
  p.pene_count = io->pene_count; p.vp_emb = io->vp_emb; p.feet_marker_idx = io->feet_marker_idx;
  p.reward = io->reward; p.terminated = io->terminated; p.rterms = io->reward_terms; p.nonfinite = io->nonfinite_count; p.invalid = io->invalid_flags; p.obs_ego = io->obs_ego;
  p.obs_dist = io->obs_dist; p.obs_time = io->obs_time; p.out_marker_b = io->out_marker_b; p.out_prev_frame = io->out_prev_frame;
  hipLaunchKernelGGL(egx_env_step_post_kernel, dim3(A), dim3(BLK), 0, static_cast<hipStream_t>(stream_), p);
  EGX_HIP_CHECK(hipGetLastError());
  return EGX_OK;
}

extern "C" int egx_env_reset(const egx_env_config* cfg, const egx_env_scenes* scenes, const egx_env_state* st,
                             const egx_env_reset_io* io, int A, void* stream_) {
  EGX_REQUIRE(cfg && scenes && st && io && A > 0, "bad arguments");
  EGX_REQUIRE(io->num_candidates >= 1 && io->cand_pairs && io->tab_joints && io->tab_markers && io->tab_glorot &&
                  io->tab_transl && io->tab_pose && io->obs_ego && io->obs_dist && io->obs_time, "null reset io array");
  EGX_REQUIRE(cfg->scene_kind == 2 ? (scenes->crowd_bbox != nullptr && scenes->crowd_scenes == A &&
                                      (!scenes->crowd_polygon || (scenes->edges && scenes->edge_off)))
                                   : (scenes->edges && scenes->edge_off),
              "scene tables missing");
  ResetArgs p;
  p.cfg = to_cfg(cfg); p.sc = to_scene(scenes); p.A = A; p.K = io->num_candidates; p.mask = io->mask;
  p.cand_pairs = io->cand_pairs; p.cand_yaw = io->cand_yaw; p.cand_variant = io->cand_variant; p.cand_scene = io->cand_scene; p.cand_valid = io->cand_valid;
  p.tab_joints = io->tab_joints; p.tab_markers = io->tab_markers; p.tab_glorot = io->tab_glorot; p.tab_transl = io->tab_transl;
  p.tab_pose = io->tab_pose;
  p.state = st->state; p.seed = st->seed; p.R0 = st->R0; p.T0 = st->T0; p.dist = st->dist; p.steps = st->steps;
  p.wpath = st->wpath; p.scene_idx = st->scene_idx;
  p.obs_ego = io->obs_ego; p.obs_dist = io->obs_dist; p.obs_time = io->obs_time; p.out_choice = io->out_choice;
  p.pending = io->out_pending; p.forced_count = io->forced_count;
  hipLaunchKernelGGL(egx_env_reset_kernel, dim3(A), dim3(BLK), 0, static_cast<hipStream_t>(stream_), p);
  EGX_HIP_CHECK(hipGetLastError());
  return EGX_OK;
}

// ------------------------------------------------------------------------------------------------
// SMPLXParser.get_new_coordinate / update_transl_glorot as stand-alone operators (models/baseops.py:465-490, 537-598):
// the step / reset kernels above use the same device functions inline.
// ------------------------------------------------------------------------------------------------
__global__ void egx_canonical_frame_kernel(const float* __restrict__ joints, int jpb, int B, float* __restrict__ R, float* __restrict__ T) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const float* j = joints + (size_t)b * jpb * 3;
  float r[9], t[3];
  canonical_frame(j, j + 3, j + 6, r, t);
  for (int e = 0; e < 9; ++e) R[(size_t)b * 9 + e] = r[e];
  for (int e = 0; e < 3; ++e) T[(size_t)b * 3 + e] = t[e];
}

__global__ void egx_update_transl_glorot_kernel(const float* __restrict__ R, const float* __restrict__ T, int rt_rows,
                                                const float* __restrict__ delta, const float* __restrict__ xb, int B,
                                                float* __restrict__ out) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const int rb = rt_rows == 1 ? 0 : b;  // one frame for all bodies, or one per body
  float x6[6], o6[6], r[9], t[3], d[3];
  for (int e = 0; e < 6; ++e) x6[e] = xb[(size_t)b * XB + e];
  for (int e = 0; e < 9; ++e) r[e] = R[(size_t)rb * 9 + e];
  for (int e = 0; e < 3; ++e) { t[e] = T[(size_t)rb * 3 + e]; d[e] = delta[(size_t)b * 3 + e]; }
  update_transl_glorot(r, t, d, x6, o6);
  if (out != xb)
    for (int e = 6; e < XB; ++e) out[(size_t)b * XB + e] = xb[(size_t)b * XB + e];
  for (int e = 0; e < 6; ++e) out[(size_t)b * XB + e] = o6[e];
}

extern "C" int egx_canonical_frame(const float* joints, int joints_per_body, int num_bodies, float* out_R, float* out_T, void* stream) {
  EGX_REQUIRE(joints && out_R && out_T, "null argument");
  EGX_REQUIRE(joints_per_body >= 3 && num_bodies > 0, "need >= 3 joints per body and a non-empty batch");
  hipLaunchKernelGGL(egx_canonical_frame_kernel, dim3(egx_ceil_div(num_bodies, 128)), dim3(128), 0, static_cast<hipStream_t>(stream),
                     joints, joints_per_body, num_bodies, out_R, out_T);
  EGX_HIP_CHECK(hipGetLastError());
  return EGX_OK;
}

extern "C" int egx_update_transl_glorot(const float* R, const float* T, int num_frames, const float* delta_T, const float* xb,
                                        int num_bodies, float* out, void* stream) {
  EGX_REQUIRE(R && T && delta_T && xb && out, "null argument");
  EGX_REQUIRE(num_bodies > 0 && (num_frames == 1 || num_frames == num_bodies), "num_frames must be 1 or num_bodies");
  hipLaunchKernelGGL(egx_update_transl_glorot_kernel, dim3(egx_ceil_div(num_bodies, 128)), dim3(128), 0,
                     static_cast<hipStream_t>(stream), R, T, num_frames, delta_T, xb, num_bodies, out);
  EGX_HIP_CHECK(hipGetLastError());
  return EGX_OK;
}

// ------------------------------------------------------------------------------------------------
// CrowdEnv._get_feature and get_map as stand-alone operators (crowd_env_2f.py:680-727, crowd_env_2f_box.py:733-776,
// batch_gen_amass.py:934-968), built from the device functions the step / reset kernels use.
// ------------------------------------------------------------------------------------------------
__global__ void egx_env_get_feature_kernel(const float* __restrict__ Y_l, const float* __restrict__ pel, const float* __restrict__ R0,
                                           const float* __restrict__ T0, const float* __restrict__ wpath, int wpath_rows, int nb,
                                           int nt, int nm, float* __restrict__ dist_xyz, float* __restrict__ fea_marker) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;  // one thread per (body, frame, marker); marker index nm = the pelvis
  if (i >= nb * nt * (nm + 1)) return;
  const int m = i % (nm + 1), bt = i / (nm + 1), b = bt / nt;
  const float* w = wpath + (size_t)(wpath_rows == 1 ? 0 : b) * 3;
  const float v[3] = {w[0] - T0[(size_t)b * 3 + 0], w[1] - T0[(size_t)b * 3 + 1], w[2] - T0[(size_t)b * 3 + 2]};
  float tl[3];
  mat3T_vec(R0 + (size_t)b * 9, v, tl);
  if (m == nm) {
    if (dist_xyz) dist_xyz[bt] = target_distance(tl, pel + (size_t)bt * 3);
  } else if (fea_marker) {
    marker_target_feature(tl, Y_l + ((size_t)bt * nm + m) * 3, fea_marker + ((size_t)bt * nm + m) * 3);
  }
}

__global__ void egx_env_get_map_kernel(const float* __restrict__ tris, int num_tris, const float* __restrict__ map_lin, int res,
                                       float floor_h, const float* __restrict__ R, const float* __restrict__ T, int nb,
                                       float* __restrict__ points_scene, float* __restrict__ local_map) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nb * res * res) return;
  const int b = i / (res * res), p = i % (res * res);
  SceneDev sc;
  sc.tris = tris; sc.map_lin = map_lin; sc.map_res = res; sc.crowd_bbox = nullptr;
  const int off[2] = {0, num_tris};
  sc.tri_off = off;
  float wxy[2];
  const bool walk = walk_cell(sc, 0, R + (size_t)b * 9, T + (size_t)b * 3, map_lin[p / res], map_lin[p % res], wxy);
  if (points_scene) { points_scene[(size_t)i * 3 + 0] = wxy[0]; points_scene[(size_t)i * 3 + 1] = wxy[1]; points_scene[(size_t)i * 3 + 2] = floor_h; }
  local_map[i] = walk ? 1.f : -1.f;
}

extern "C" int egx_env_get_feature(const float* Y_l, const float* pel, const float* R0, const float* T0, const float* wpath,
                                   int wpath_rows, int num_bodies, int num_frames, int num_markers, float* out_dist_xyz,
                                   float* out_fea_marker, void* stream) {
  EGX_REQUIRE(Y_l && pel && R0 && T0 && wpath, "null argument");
  EGX_REQUIRE(num_bodies > 0 && num_frames > 0 && num_markers > 0, "empty batch");
  EGX_REQUIRE(wpath_rows == 1 || wpath_rows == num_bodies, "wpath must hold 1 or num_bodies targets");
  const int n = num_bodies * num_frames * (num_markers + 1);
  hipLaunchKernelGGL(egx_env_get_feature_kernel, dim3(egx_ceil_div(n, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), Y_l, pel,
                     R0, T0, wpath, wpath_rows, num_bodies, num_frames, num_markers, out_dist_xyz, out_fea_marker);
  EGX_HIP_CHECK(hipGetLastError());
  return EGX_OK;
}

extern "C" int egx_env_get_map(const float* tris, int num_tris, float floor_height, const float* map_lin, int res, const float* R,
                               const float* T, int num_bodies, float* out_points_scene, float* out_local_map, void* stream) {
  EGX_REQUIRE(tris && map_lin && R && T && out_local_map, "null argument");
  EGX_REQUIRE(num_tris > 0 && res > 0 && num_bodies > 0, "empty input");
  const int n = num_bodies * res * res;
  hipLaunchKernelGGL(egx_env_get_map_kernel, dim3(egx_ceil_div(n, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), tris,
                     num_tris, map_lin, res, floor_height, R, T, num_bodies, out_points_scene, out_local_map);
  EGX_HIP_CHECK(hipGetLastError());
  return EGX_OK;
}
