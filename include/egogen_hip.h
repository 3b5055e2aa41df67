/*
 * egogen_hip.h - C ABI of libegogen_hip.so: MI355X (gfx950) kernels for the EgoGen crowd_ppo hot path.
 *
 * The reference (ligengen/EgoGen) is pure Python and has no FFI layer; its de-facto operator API is a set
 * of Python call signatures.  Every entry point below names the reference call it
 * replaces (file:line relative to the reference repo root).  The Python host side (the egogen_amd Python package) binds
 * these with ctypes and mirrors the reference signatures; INTEGRATION.md shows the stub a maintainer of
 * the reference would add.
 *
 * Conventions
 *   - all `const float*` / `float*` / `int*` arguments are DEVICE pointers unless the name ends in `_host`;
 *   - no entry point allocates or synchronises, except *_create / *_destroy (one-time setup);
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); work is enqueued asynchronously;
 *   - return value: EGX_OK (0) or a negative error code; egx_last_error() gives a message;
 *   - fp32 everywhere; row-major; "B" = bodies (agent x frame), "A" = agents.
 */
#ifndef EGOGEN_HIP_H
#define EGOGEN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EGX_OK 0
#define EGX_ERR_ARG (-1)
#define EGX_ERR_HIP (-2)
#define EGX_ERR_WORKSPACE (-3)

#define EGX_NUM_JOINTS 55      /* SMPL-X kinematic joints                */
#define EGX_NUM_JOINTS_OUT 127 /* 55 + 21 vertex joints + 51 landmarks   */
#define EGX_XB_DIM 93          /* transl3 glorot3 body63 lhand12 rhand12 */
#define EGX_NUM_BETAS 10
#define EGX_POSE_FEAT 486
#define EGX_BLEND_K 472        /* GEMM K: 10 betas + 9 x 51 movable joints = 469, padded to 8 (jaw/eyes are always identity here) */

const char* egx_last_error(void);
int egx_version(void);

/* ---------------------------------------------------------------------------------------------
 * Body model  (replaces smplx.create(...) + SMPLXParser.__init__, models/baseops.py:285-335)
 * ------------------------------------------------------------------------------------------- */
typedef struct egx_body_model egx_body_model; /* opaque; owns device copies of the packed bases */

typedef struct egx_body_model_host {
  int num_verts;                /* V (10475 for SMPL-X)                                   */
  const float* v_template_host; /* [V,3]                                                  */
  const float* shapedirs_host;  /* [V,3,10]   first 10 shape directions (expression = 0)  */
  const float* posedirs_host;   /* [486,3V]                                               */
  const float* J_regressor_host;/* [55,V]                                                 */
  const int32_t* parents_host;  /* [55], parents[0] = -1                                  */
  const float* lbs_weights_host;/* [V,55] dense; packed to ELL(nnz = max row nnz) inside  */
  const float* hand_comps_l_host; /* [12,45] */
  const float* hand_comps_r_host; /* [12,45] */
  const float* hand_mean_l_host;  /* [45]    */
  const float* hand_mean_r_host;  /* [45]    */
  const int32_t* extra_vids_host; /* [21] vertex-selected joints (smplx VertexJointSelector)   */
  const int32_t* lmk_vids_host;   /* [51,3] = faces[lmk_faces_idx]                              */
  const float* lmk_bary_host;     /* [51,3]                                                     */
  int num_markers;                /* M (67 for SSM2)                                            */
  const int32_t* marker_vids_host;/* [M]  SMPLXParser.marker (baseops.py:333-335)               */
  int num_feet;                   /* vertices excluded from penetration counting                */
  const int32_t* feet_vids_host;  /* [num_feet]  crowd_env_2f.py:53-59                          */
} egx_body_model_host;

int egx_body_model_create(const egx_body_model_host* desc, egx_body_model** out);
void egx_body_model_destroy(egx_body_model* m);
int egx_body_model_num_verts(const egx_body_model* m);
int egx_body_model_nnz(const egx_body_model* m);
/* Vertices egx_lbs_forward evaluates when no vertex output is requested: those of the 32-vertex tiles that hold a picked vertex
 * (marker, vertex joint, landmark corner) - with_sdf = 0 - or, with_sdf = 1, also a vertex of the penetration count
 * (crowd_env_2f.py:163-175 excludes the feet).  num_verts when every tile is needed.  Work accounting for bench.py. */
int egx_body_model_lbs_vertices(const egx_body_model* m, int with_sdf);

/* Scene SDF (crowd_ppo/utils.py:54-84 `sdf_dict`): grid[d0][d1][d2] indexed by the vertex (x,y,z). */
typedef struct egx_sdf_grid {
  const float* grid; /* device, [d0,d1,d2] */
  int d0, d1, d2;
  float center[3];
  float scale;
  const void* coarse_minmax; /* device table from egx_sdf_build_coarse: required by egx_lbs_forward, optional (NULL = fine grid only) elsewhere */
} egx_sdf_grid;

/* Acceleration tables for the penetration COUNT of egx_lbs_forward: {min,max} of the fine samples each 4x4x4 block's
 * interpolation footprint can touch, followed (same buffer) by a four-level pyramid of the block maxima (free-space tests of
 * whole boxes: the work-item culling).  Trilinear interpolation is a convex combination, so a block whose bracket does not
 * contain 0 decides `calc_sdf < 0` exactly without gathering from the 64 MiB grid.  Six face tables follow the block table:
 * a point that border clamping (grid_sample padding_mode="border", utils.py:75-81) puts onto the first / last sample plane
 * of an axis only touches samples of that plane, so it is bracketed over the plane alone - bodies outside the cube are
 * decided without gathers as well.  The buffer also holds 64 auxiliary floats (the steepest slope of the interpolated field:
 * the fix-up band of blend mode 3) and a copy of the grid as 4 x 4 x 4 bricks of 256 contiguous bytes, which egx_sdf_sample
 * gathers from when the descriptor carries the table (a point's eight corners then span 2.3 cache lines instead of 4):
 * egx_sdf_coarse_bytes() is the grid's size plus ~4 %. */
size_t egx_sdf_coarse_bytes(int d0, int d1, int d2);
int egx_sdf_build_coarse(const egx_sdf_grid* sdf, void* coarse_out, void* stream);

/* Free-space culling of the work items of an SDF-counting egx_lbs_forward call (split blend modes, no vertex output): a
 * (vertex tile x 256 bodies) item all of whose bodies are provably clear of geometry for that tile - bounding box of the balls
 * around the posed joints the tile's vertices are bound to, tested against a max-pyramid of the bracket table - is skipped
 * unless the tile holds picked vertices.  Counts, joints and markers are bit-identical with and without it.  OPT-IN
 * (EGX_LBS_CULL=1 in the environment or egx_lbs_set_culling(1)): it pays when bodies stand in free space (a trained policy on a
 * learned body model: fused kernel -32 %, three small launches +0.11 ms per call at 10 240 bodies), not when they are inside
 * geometry or the model's bound is loose (profiles/r04_lbs_culling.md); egx_lbs_cull_stats reads, with a host
 * synchronisation, how many items the LAST culled call on this workspace evaluated and how many an unculled call has. */
/* 1 if SDF launches of this model are culled (convex skinning weights AND a tight bound: the blend-shape margin of the median
 * vertex tile at a reference pose is below 15 cm - learned body models; the i.i.d.-noise benchmark body is not), else 0.
 * *out_reference_margin_m (may be NULL) receives that margin in metres. */
int egx_body_model_culls(const egx_body_model* m, float* out_reference_margin_m);
int egx_lbs_set_culling(int on);
int egx_lbs_get_culling(void);
int egx_lbs_cull_stats(const egx_body_model* m, const void* workspace, int num_bodies, int32_t* out_active_items,
                       int32_t* out_total_items);

/* Bytes of scratch egx_lbs_forward needs for `num_bodies` bodies. */
/* Blend-GEMM arithmetic of egx_lbs_forward calls that do not write full vertices (process-wide switch; the environment
 * variable EGX_LBS_BLEND=f32|bf16x3|bf16x2|f16mix selects it at first use):
 *   0  fp32 MFMA (v_mfma_f32_32x32x2_f32); always used when vertices are written
 *   1  3-term bf16 split of both operands, six partial products per fp32 product on v_mfma_f32_32x32x16_bf16 with fp32
 *      accumulation: 2^-24-level accuracy (indistinguishable from mode 0) at a third of the matrix-pipe time
 *   2  2-term bf16 split, three partial products (hi.hi + hi.mid + mid.hi: 16 significant bits per operand) - and a third
 *      term for the template column, carried by a padding column of K - at a sixth of the matrix-pipe time.  The
 *      offsets it rounds are centimetres, so vertices move by <= 1.1e-6 m against the float64 oracle (mode 0: 3.7e-7) and
 *      the fused penetration counts do not change (tests/test_lbs_gpu.py::test_lbs_blend_mode_accuracy_report); all three
 *      modes are held to the same parity tolerances (north_star: 1e-4 relative).
 *   3  "f16mix" (DEFAULT): the vertex tiles that hold PICKED vertices (markers, vertex joints, landmark corners - everything the
 *      caller reads as a position) run exactly as in mode 2; the other tiles, which only feed the penetration COUNT, keep the
 *      shape / template k-steps as in mode 2 and run the 28 k-steps that hold only pose-corrective columns (centimetre-scale
 *      offsets) as ONE v_mfma_f32_32x32x16_f16 product on operands rounded to fp16 (2^-11 per operand): 204 instead of 540
 *      MFMAs per wave and work item, a third of the operand bytes.  Those vertices move by ~4e-6 m rms / ~2.2e-5 m worst case
 *      against float64 on the synthetic body, so the cheap product only CLASSIFIES: a vertex whose interpolated SDF value is
 *      closer to zero than its position error can account for (ten standard deviations of the product's rounding error for
 *      that body's pose - plus, in launches of more than 5 120 bodies, whose count-only tiles are also SKINNED on the matrix
 *      pipe in two bf16 planes, an allowance for that - times the steepest slope of the grid) is queued and re-evaluated in
 *      fp32 by a small kernel launched behind the fused one (vertex-major fp32 bases, the body's rotations recomputed from its
 *      parameter row, fp32 skinning) and counted from that - a few per thousand counted vertices.  The counts are those of an
 *      fp32 evaluation (crowd_env_2f.py:169-177): the parity tests hold every mode to the same 2e-5 m level-set band, and
 *      egx_lbs_fix_stats reads how many vertices the last call re-evaluated.  Positions are unaffected.
 * An unrecognised EGX_LBS_BLEND string is an error of the first egx_lbs_forward call (no silent default).
 * (No reference counterpart: smplx evaluates the blend shapes as fp32 einsum/matmul, lbs.py [upstream smplx 0.1.28].) */
int egx_lbs_set_blend_mode(int mode);
int egx_lbs_get_blend_mode(void);
/* Mode 3 only: wave tile of the fused kernel - 0 (default) = chosen by launch size: 32 vertices x 32 bodies per wave, three
 * workgroups per CU, VALU skinning, for launches of at most 20 groups of 256 bodies; 32 x 64, two workgroups per CU, count-only
 * tiles skinned on the matrix pipe, above that.  1 / 2 force one (EGX_LBS_WAVE_TILE in the environment does the same): the
 * parity tests run both on the same inputs. */
int egx_lbs_set_wave_tile(int tile);
int egx_lbs_get_wave_tile(void);
/* Mode 3 only: entries of each of the 64 sub-queues of the launch's fix-up queue that are used (0 = all 4 096).  Vertices that
 * find their sub-queue full are re-evaluated inside the fused kernel instead - slower, same result; the parity tests shrink the
 * queue to exercise that path. */
int egx_lbs_set_fix_queue_capacity(int entries);
/* Mode 3 only: vertices the last SDF-counting egx_lbs_forward call on this workspace re-evaluated in fp32 (host synchronisation). */
int egx_lbs_fix_stats(const egx_body_model* m, const void* workspace, int num_bodies, int32_t* out_reevaluated);

size_t egx_lbs_workspace_bytes(const egx_body_model* m, int num_bodies);

/*
 * egx_lbs_forward - SMPL-X forward for B bodies.
 * Replaces SMPLXParser.forward_smplx -> bm(return_verts=True, **bparam) (models/baseops.py:338-398),
 * i.e. smplx.SMPLX.forward [upstream smplx 0.1.28]: hand PCA, Rodrigues x55, shape+pose blend shapes,
 * joint regression, rigid chain, linear blend skinning, 21 vertex joints + 51 landmarks, + transl.
 * Optionally fuses the reference's follow-up on the vertices (crowd_env_2f.py:163-175): world transform
 * by the agent frame (R0,T0), calc_sdf, feet mask, per-body count of vertices with sdf < 0.
 *
 *   xb            [B,93]
 *   betas         [A,10], body b uses row b / frames_per_agent
 *   out_verts     [B,V,3]   or NULL
 *   out_joints    [B,127,3] or NULL
 *   out_markers   [B,M,3]   or NULL   (= vertices[:, marker, :])
 *   sdf           NULL = no SDF epilogue; else also needs
 *   R0 [A,3,3], T0 [A,3]    agent frames (NULL = identity), and
 *   out_pene_count [B] int32 (overwritten)
 */
int egx_lbs_forward(const egx_body_model* m, const float* xb, const float* betas, int num_bodies,
                    int frames_per_agent, float* out_verts, float* out_joints, float* out_markers,
                    const egx_sdf_grid* sdf, const float* R0, const float* T0, int32_t* out_pene_count,
                    void* workspace, size_t workspace_bytes, void* stream);

/*
 * egx_lbs_joints - the 55 kinematic-tree joints of B bodies without the vertex pass: out_joints55 [B,55,3] =
 * bm(**bparam).joints[:, :55].  This is all SMPLXParser.get_new_coordinate (models/baseops.py:485: joints 0,1,2) and
 * calc_calibrate_offset (:529: joint 0 at zero global_orient / transl) take from their full SMPL-X evaluation.
 * workspace as for egx_lbs_forward.
 */
int egx_lbs_joints(const egx_body_model* m, const float* xb, const float* betas, int num_bodies, int frames_per_agent,
                   float* out_joints55, void* workspace, size_t workspace_bytes, void* stream);

/*
 * egx_canonical_frame - CanonicalCoordinateExtractor.get_new_coordinate_torch (models/baseops.py:214-225), the tail of
 * SMPLXParser.get_new_coordinate (:465-490): joints [B, joints_per_body >= 3, 3] -> out_R [B,3,3] (columns x, y, z with
 * x = normalise((j2 - j1) with z zeroed) - no epsilon, like the reference -, z = (0,0,1), y = normalise(z cross x)),
 * out_T [B,3] = joint 0 (the reference returns it as [B,1,3]).
 */
int egx_canonical_frame(const float* joints, int joints_per_body, int num_bodies, float* out_R, float* out_T, void* stream);

/*
 * egx_update_transl_glorot - SMPLXParser.update_transl_glorot, torch branch (models/baseops.py:537-598): re-express
 * transl / global_orient of xb [B,93] in the frame (R,T):  glorot' = aa(R^T aa2R(glorot))  (torchgeometry 0.1.2
 * angle_axis_to_rotation_matrix / rotation_matrix_to_angle_axis [upstream]),  transl' = R^T (transl + delta_T - T) - delta_T.
 * R [F,3,3], T [F,3] with F = num_frames in {1, B}; delta_T [B,3] = calc_calibrate_offset (:494-534) = joint 0 of
 * egx_lbs_joints at zero global_orient / transl.  out [B,93]; out == xb is the reference's inplace=True.
 */
int egx_update_transl_glorot(const float* R, const float* T, int num_frames, const float* delta_T, const float* xb, int num_bodies,
                             float* out, void* stream);

/*
 * egx_sdf_sample - calc_sdf(vertices, sdf_dict) (crowd_ppo/utils.py:54-84): trilinear, border clamp,
 * align_corners=False, negated.  pts [n,3] -> out [n].  With sdf->coarse_minmax set (egx_sdf_build_coarse) the corners are
 * gathered from the bricked copy of the grid in that buffer, otherwise from the row-major grid: bit-identical values.
 */
int egx_sdf_sample(const egx_sdf_grid* sdf, const float* pts, int64_t n, float* out, void* stream);

/*
 * egx_mesh_sdf - scene preparation (the reference ships data/room0_sdf.pkl ready-made and refers to an
 * external tool for new scenes, README.md:97): fills grid[d0][d1][d2] (indexed by x, y, z; sample (i,j,k) at
 * center + ((2i+1)/d - 1) / scale, the cell centres `calc_sdf`'s grid_sample(align_corners=False) assumes,
 * crowd_ppo/utils.py:54-84) with the signed distance to a closed triangle mesh.
 *   triangles [F,9] device (ax,ay,az,bx,by,bz,cx,cy,cz); center_host [3] host; inside_positive 1: > 0 inside the mesh
 *   (obstacle solids, the stored convention: calc_sdf negates), 0: < 0 inside (a room shell whose interior is free space).
 */
int egx_mesh_sdf(const float* triangles, int num_triangles, const float* center_host, float scale, int d0, int d1, int d2,
                 int inside_positive, float* out_grid, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Dense layers of the rollout networks.  One fused call replaces nn.Linear + torch.cat + activation
 * (+ residual) as composed in models/baseops.py:615-641 (MLP), models_GAMMA_primitive.py:160-175
 * (ResNetBlock) and models_policy_ppo.py:24-39 (MLPBlock):   out = act(cat(x_0..x_{s-1}) W^T + b) + residual
 * ------------------------------------------------------------------------------------------- */
#define EGX_ACT_NONE 0
#define EGX_ACT_TANH 1
#define EGX_ACT_RELU 2
#define EGX_ACT_LRELU 3

typedef struct egx_linear_desc {
  int num_rows;          /* M */
  int out_features;      /* N */
  int num_segments;      /* 1..4 input segments, concatenated along the feature axis (K = sum widths) */
  const float* seg_ptr[4];
  int seg_width[4];
  int seg_ld[4];         /* row stride of each segment, in floats */
  const float* weight;   /* [N,K] torch nn.Linear layout */
  int weight_ld;         /* 0 = K */
  const float* bias;     /* [N] or NULL */
  const float* residual; /* [M,N] added AFTER the activation, or NULL */
  int residual_ld;
  float* out;            /* [M,N] */
  int out_ld;            /* 0 = N */
  int activation;        /* EGX_ACT_* */
  float leaky_slope;
} egx_linear_desc;

int egx_linear(const egx_linear_desc* d, void* stream);

/* torch.nn.GRU / GRUCell gate math given gi = x W_ih^T + b_ih and gh = h W_hh^T + b_hh ([M,3H], gate order
 * r,z,n); h_prev may be NULL (zero state).  Used for x_enc / d_rnn (models_GAMMA_primitive.py:84,94) and the
 * policy's x_enc / ego_enc (models_policy_ppo.py:291,298). */
int egx_gru_pointwise(const float* gi, const float* gh, const float* h_prev, int h_prev_ld, float* h_out,
                      int h_out_ld, int num_rows, int hidden, void* stream);

/* MoshRegressor._cont2aa (models_GAMMA_primitive.py:208-219; RotConverter.cont2aa baseops.py:143-162):
 * xb6 [n,159] (transl3 | 22x6D | hands24) -> out [n,>=93] (transl3 | 22 axis-angles | hands24). */
int egx_cont6d_to_aa(const float* xb6, int num_rows, float* out, int out_ld, void* stream);

/* GAMMAPolicyBase.positional_encoding of obs['dist'] and obs['time'] (models_policy_ppo.py:276-285,302-303):
 * out [A,128] = [posenc(dist) 64 | posenc(time) 64]. */
int egx_posenc(const float* dist, const float* time, int num_agents, float* out, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Motion prior: GAMMAPrimitiveCombo.sample_prior(X, betas, z) (models_GAMMA_primitive.py:334-360)
 *   = GAMMAPrimitiveVAE.decode (:83-101, 18 GRUCell steps, residual)  +  MoshRegressor.forward (:262-301,
 *     3 recurrences of a 10-block ResNet, 6D -> axis-angle).
 * Weight pointers alias the torch parameters of the same state_dict keys (`predictor.*`, `regressor.*`).
 * ------------------------------------------------------------------------------------------- */
/* "Packed" matrices of the rollout dense layers (csrc/dense3.hip).  The rollout networks run their products on the bf16
 * matrix pipe with every fp32 operand carried as three bf16 terms (x = hi + mid + lo, six partial products, fp32
 * accumulation: 2^-24 relative, the arithmetic of an fp32 product).  A packed image of a row-major fp32 matrix [R, K] is
 *   [2 ceil(R/32) row tiles of 16][ceil(K/32) k-steps][3 planes][64 lanes] x 8 bf16,
 *   lane l of a fragment = row (l & 15), columns 32 s + 8 (l >> 4) .. + 7   (operand order of v_mfma_f32_16x16x32_bf16),
 * zero beyond R / K.  egx_pack3 writes such an image at k-step `dst_kstep0` of a buffer that holds `dst_ksteps` k-steps per
 * row tile (a concatenated input is several images side by side in one buffer).  Weights [N, K] in torch layout are packed
 * as they are (row = output feature). */
size_t egx_pack3_bytes(int num_rows, int num_cols);
int egx_pack3(const float* src, int num_rows, int num_cols, int src_ld, int src_col0, void* dst, int dst_ksteps,
              int dst_kstep0, void* stream);

/* One dense product on fp32 row-major operands with fp32-equivalent arithmetic on the bf16 matrix pipe:
 *   out[M,N] = act(op(A) op(B)^T + bias) + res,   op(A) = A[M,K] (trans_a = 0) or A[K,M]^T (trans_a = 1),
 *                                                 op(B) = B[N,K] (trans_b = 0) or B[K,N]^T (trans_b = 1).
 * Both operands are packed into `workspace` (egx_gemm3_workspace_bytes) and multiplied by the dense kernel of the rollout
 * / update chain: two launches.  `res` may alias `out` (accumulation); `out_act` (optional) receives the activation before
 * the residual is added.  This is what the training-side autograd nodes (egogen_amd/fused_ops.py: predictor / regressor
 * training, the checker path of the PPO update) call where the reference's `loss.backward()` reaches an `nn.Linear` /
 * `nn.GRU` product: x W^T + b forward (models/baseops.py:638-641), g^T x and g W backward. */
size_t egx_gemm3_workspace_bytes(int M, int N, int K);
int egx_gemm3(const float* A, int lda, int trans_a, const float* B, int ldb, int trans_b, int M, int N, int K,
              const float* bias, int act, float slope, const float* res, int ldr, float* out, int ldo, float* out_act,
              int ldact, void* workspace, size_t workspace_bytes, void* stream);

/* Packed images of the C-VAE decoder's dense weights (all of them or none). */
typedef struct egx_prior_packed3 {
  const void *x_enc_w_ih, *x_enc_w_hh; /* [768,201], [768,256]                                  */
  const void *drnn_w[3];               /* [512,256], [256,512], [256,256]                       */
  const void *d_rnn_w_hz;              /* d_rnn_w_ih[:, 0:384]   (columns of [hx | z])           */
  const void *d_rnn_w_y;               /* d_rnn_w_ih[:, 384:585] (columns of y_p)                */
  const void *d_rnn_w_hh;              /* [768,256]                                             */
  const void *d_comb_w;                /* [768,256] (see d_comb_w below)                        */
  const void *d_mlp_w[2];              /* [512,256], [256,512]                                  */
  const void *d_out_w;                 /* [201,256]                                             */
  /* body regressor (regressor.pnet): in_fc [128,370] as its three column blocks (markers 0:201, 6D parameters 201:360,
   * betas 360:370), the 20 block layers [128,128] one after the other, out_fc [159,128]; reg_blk_b = the 20 block biases
   * as one fp32 array [20][128] */
  const void *reg_in_m, *reg_in_xb, *reg_in_betas, *reg_blk, *reg_out;
  const float* reg_blk_b;
} egx_prior_packed3;

typedef struct egx_prior_weights {
  const float *x_enc_w_ih, *x_enc_w_hh, *x_enc_b_ih, *x_enc_b_hh; /* predictor.x_enc  GRU(201,256)      */
  const float *drnn_w[3], *drnn_b[3];                             /* predictor.drnn_mlp 256-512-256-256  */
  const float *d_rnn_w_ih, *d_rnn_w_hh, *d_rnn_b_ih, *d_rnn_b_hh; /* predictor.d_rnn  GRUCell(585,256)   */
  const float *d_mlp_w[2], *d_mlp_b[2];                           /* predictor.d_mlp  256-512-256        */
  const float *d_out_w, *d_out_b;                                 /* predictor.d_out  256-201            */
  const float *reg_in_w, *reg_in_b;                               /* regressor.pnet.in_fc 370-128        */
  const float *reg_blk_w[20], *reg_blk_b[20];                     /* regressor.pnet.layers.{0..9}.layers.{0,1} */
  const float *reg_out_w, *reg_out_b;                             /* regressor.pnet.out_fc 128-159       */
  /* d_comb_w [768,256] = d_rnn_w_ih[:, 384:585] . d_out_w and d_comb_b [768] = d_rnn_w_ih[:, 384:585] . d_out_b (folded by the
   * caller, in float64).  The residual decoder feeds y_i = d_out(h_i) + y_(i-1) back into the GRU cell
   * (models_GAMMA_primitive.py:95-103), so the cell's input product obeys gi_(i+1) = gi_i + h_i . d_comb_w^T + d_comb_b: with
   * the two folded tensors the output layer leaves the 18-step critical path and is evaluated for all steps at once
   * afterwards. */
  const float *d_comb_w, *d_comb_b;
  /* Packed images (egx_pack3) of the decoder's and the regressor's weights: egx_sample_prior runs on the bf16 matrix pipe
   * (three-term splits, fp32-equivalent) with the GRU cell as one launch and the fused body regressor as another (48-row
   * workgroups, one per compute unit).  Required - the torch-layout matrices above are only read for the biases. */
  const egx_prior_packed3* packed3;
} egx_prior_weights;

size_t egx_sample_prior_workspace_bytes(int num_agents);

/*   x_hist0 / x_hist1 : the two history frames of markers, [A,201] with row stride x_ld floats
 *                       (reference passes X = states[:2, :, :201], crowd_env_2f.py:98,109)
 *   betas [A,10]; z [A,128]
 *   out_Y  [18,A,201]  predicted markers   (Y_gen)
 *   out_Yb [18,A,93]   regressed body parameters, axis-angle (Yb_gen)                               */
int egx_sample_prior(const egx_prior_weights* w, const float* x_hist0, const float* x_hist1, int x_ld,
                     const float* betas, const float* z, int num_agents, float* out_Y, float* out_Yb,
                     void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Policy networks: GAMMAPolicyBase.forward + GAMMAActor.forward + GAMMACritic.forward
 * (models_policy_ppo.py:287-306,326-330,348-350).  Pointers alias the torch parameters
 * `shared_net.*`, `actor.pnet.*`, `critic.vnet.*`.
 * ------------------------------------------------------------------------------------------- */
/* Packed images (egx_pack3) of the policy's dense weights, all of them or none; refreshed by the host after every change
 * of the parameters (once per collect during training). */
typedef struct egx_policy_packed3 {
  const void *x_enc_w_ih, *x_enc_w_hh;     /* [1536,402], [1536,512] */
  const void *ego_enc_w_ih, *ego_enc_w_hh; /* [1536,32],  [1536,512] */
  const void *actor_w[4], *actor_out_w;    /* [1152,1152] x 4, [256,1152] */
  const void *critic_w[4], *critic_out_w;  /* [1152,1152] x 4, [1,1152]   */
} egx_policy_packed3;

typedef struct egx_policy_weights {
  const float *x_enc_w_ih, *x_enc_w_hh, *x_enc_b_ih, *x_enc_b_hh;         /* shared_net.x_enc   GRU(402,512) */
  const float *ego_enc_w_ih, *ego_enc_w_hh, *ego_enc_b_ih, *ego_enc_b_hh; /* shared_net.ego_enc GRU(32,512)  */
  const float *actor_w[4], *actor_b[4];   /* actor.pnet.layers.{0,1}.layers.{0,1}  1152x1152 */
  const float *actor_out_w, *actor_out_b; /* actor.pnet.out_fc   256x1152                    */
  const float *critic_w[4], *critic_b[4]; /* critic.vnet.layers.{0,1}.layers.{0,1}           */
  const float *critic_out_w, *critic_out_b; /* critic.vnet.out_fc 1x1152                     */
  /* Packed images of the weights (egx_pack3, or the update's own: egx_policy_train_packed): egx_policy_forward runs on the
   * bf16 matrix pipe (arithmetic: egx_policy_set_precision), each GRU cell as one launch, [hx | he | posenc] assembled in
   * place.  Required. */
  const egx_policy_packed3* packed3;
} egx_policy_weights;

/* ---------------------------------------------------------------------------------------------
 * One PPO minibatch of the policy networks without autograd and without library GEMMs (csrc/update3.hip):
 * GAMMAPPOPolicy.learn's inner loop body (crowd_ppo/ppo_policy.py:189-241) - forward of shared_net / actor / critic
 * (models/models_policy_ppo.py:287-350), the clipped-PPO loss (egx_ppo_loss_packed), and the backward pass - as a fixed chain
 * of ~21 launches on the dense3 kernels.  Gradients are WRITTEN to the tensors of `egx_policy_grads` (the views of the flat
 * gradient buffer): every parameter is used once per minibatch, nothing accumulates, the buffer needs no zero fill.
 *   create   fixes the minibatch rows (a multiple of 32) and allocates every intermediate buffer;
 *   refresh  re-makes the packed images (W and W^T) of the weights: after every optimiser step;
 *   packed   hands the row-major weight images to egx_policy_forward (egx_policy_weights.packed3), so the rollout needs
 *            no packing of its own;
 *   bind     names the observation tensors of the minibatch, state [n,2,402] and egosensing [n,2,32] (fixed addresses:
 *            the gather kernel's outputs);
 *   step     runs the chain; out_terms[6] as egx_ppo_loss_packed.
 * ------------------------------------------------------------------------------------------- */
typedef struct egx_policy_grads {
  float *x_enc_w_ih, *x_enc_w_hh, *x_enc_b_ih, *x_enc_b_hh;
  float *ego_enc_w_ih, *ego_enc_w_hh, *ego_enc_b_ih, *ego_enc_b_hh;
  float *actor_w[4], *actor_b[4], *actor_out_w, *actor_out_b;
  float *critic_w[4], *critic_b[4], *critic_out_w, *critic_out_b;
} egx_policy_grads;
typedef struct egx_policy_train egx_policy_train;
int egx_policy_train_create(const egx_policy_weights* w, const egx_policy_grads* g, int num_rows, egx_policy_train** out);
void egx_policy_train_destroy(egx_policy_train* h);
int egx_policy_train_refresh(egx_policy_train* h, void* stream);
int egx_policy_train_packed(const egx_policy_train* h, egx_policy_packed3* out);
int egx_policy_train_bind(egx_policy_train* h, const float* state, const float* egosensing);
int egx_policy_train_step(egx_policy_train* h, const float* dist, const float* time, const float* act, const float* adv,
                          const float* ret, const float* logp_old, const float* adv_stats, const float* scale, float adv_eps,
                          float min_logvar, float max_logvar, float eps_clip, float vf_coef, float ent_coef, float* out_terms,
                          void* stream);
/* The same chain in two halves, for data-parallel training (no reference counterpart: the reference trains on one device).
 * `_heads` = forward, loss (ppo_policy.py:189-241) and the whole backward of the actor and critic blocks: when it has been
 * enqueued, the prefix of the flat gradient that crowd_ppo clips (actor + critic, 10.3 M of 13.2 M parameters) is final, so its
 * all-reduce can start on a side stream; `_encoders` = the backward of the two GRU encoders (the remaining 2.8 M).  Calling both
 * in order is egx_policy_train_step. */
int egx_policy_train_step_heads(egx_policy_train* h, const float* dist, const float* time, const float* act, const float* adv,
                                const float* ret, const float* logp_old, const float* adv_stats, const float* scale, float adv_eps,
                                float min_logvar, float max_logvar, float eps_clip, float vf_coef, float ent_coef, float* out_terms,
                                void* stream);
int egx_policy_train_step_encoders(egx_policy_train* h, void* stream);
/* Arithmetic of every product of the chain (forward, input gradients, weight gradients): 0 = each fp32 operand as three bf16
 * terms, six partial products (2^-24 relative: fp32-equivalent; default), 2 = two terms, three products (16 significant bits
 * per operand, the arithmetic of the LBS blend GEMM's default mode), 1 = operands rounded to bf16, one product ("bf16 MFMA"
 * of north_star).  fp32 accumulation, biases, activations, loss and all gradients OUTPUTS are fp32 in every mode.  Measured
 * against a float64 evaluation of crowd_ppo/ppo_policy.py:189-241 in tests/test_trainer_gpu.py (profiles/r04_p3_yardstick.txt). */
int egx_policy_train_set_precision(egx_policy_train* h, int prec);

/* Arithmetic of the dense layers inside egx_policy_forward (process-wide): 0 = fp32-equivalent (default; 1e-4 parity with the
 * reference's fp32 policy), 2 = operands as two bf16 terms (16 significant bits, three partial products), 1 = operands
 * rounded to bf16, products on the bf16 MFMA, fp32 accumulation - BASELINE config 5 ("main_crowd_eval ... bf16 MFMA
 * policy"), whose parity is statistical.  Gate math, biases, activations and outputs stay fp32. */
int egx_policy_set_precision(int bf16);
int egx_policy_get_precision(void);

size_t egx_policy_workspace_bytes(int num_rows);

/*   state [n,2,402], egosensing [n,2,32], dist [n], time [n]   (the obs dict, crowd_env_2f.py:311-312)
 *   out_mu [n,128], out_logvar [n,128] (unclamped, as the actor returns them), out_value [n];
 *   any output may be NULL (actor or critic branch skipped).                                          */
int egx_policy_forward(const egx_policy_weights* w, const float* state, const float* egosensing, const float* dist,
                       const float* time, int num_rows, float* out_mu, float* out_logvar, float* out_value,
                       void* workspace, size_t workspace_bytes, void* stream);

/* VPoser v1 encoder mean, eval mode (human_body_prior [upstream]; call site crowd_env_2f.py:198).
 * BatchNorm layers are folded into fc1 / fc2 by the host.  x: [n,63] with row stride x_ld -> out [n,32]. */
typedef struct egx_vposer_weights {
  const float *fc1_w, *fc1_b; /* 512x63  */
  const float *fc2_w, *fc2_b; /* 512x512 */
  const float *mu_w, *mu_b;   /* 32x512  */
  /* Packed images (egx_pack3) of fc1_w [512,63], fc2_w [512,512], mu_w [32,512]: the encoder is ONE launch on the bf16 matrix
   * pipe (three-term operands: fp32-equivalent, like the motion prior).  Required. */
  const void *fc1_w3, *fc2_w3, *mu_w3;
} egx_vposer_weights;

/* The encoder is one fused launch whose intermediates live in LDS: it needs no workspace (egx_vposer_workspace_bytes returns 0,
 * `workspace` may be NULL); the argument pair stays so that the signature matches the other network entries. */
size_t egx_vposer_workspace_bytes(int num_rows);
int egx_vposer_encode(const egx_vposer_weights* w, const float* x, int x_ld, int num_rows, float* out,
                      void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Vector environment: the per-agent tail of CrowdEnv.step and CrowdEnv.reset
 * (crowd_ppo/crowd_env_2f.py:78-415, crowd_env_2f_box.py:78-440), batched over A independent agents.
 * ------------------------------------------------------------------------------------------- */
typedef struct egx_env_config {   /* cfg_samp20/MPVAEPolicy_samp_collision(_2).yaml fields used by step() */
  float reproj_factor, goal_thresh, pene_thres;
  float weight_skate, weight_floor, weight_face_target, weight_look_target, weight_success, weight_target_dist,
      weight_pene, weight_vp;
  int max_depth;
  int scene_kind;               /* 0: SDF scene (crowd_env_2f), 1: box scenes / walkability map (crowd_env_2f_box),
                                   2: dynamic crowd, boxes of the other members as holes (crowd_env_crowd_eval)    */
  int terminate_on_penetration; /* finetuning (crowd_env_2f.py:299-300) or box env (crowd_env_2f_box.py:325)       */
  int pene_type_body;           /* lossconfig.pene_type == 'body'                                                 */
  float ray_len;                /* 7 (2 when rendering), crowd_env_2f.py:556-558                                   */
  float vp_thresh;              /* mean VPoser norm of an "unrealistic pose": 11 (crowd_env_2f.py:201); 14 in
                                   crowd_env_egobody_eval.py:229.  <= 0 selects 11                                 */
  int no_goal_termination;      /* 1: only max_depth ends an episode (crowd_env_egobody_eval.py:378)               */
} egx_env_config;

typedef struct egx_env_scenes { /* static scene tables, device pointers */
  const float* edges;      /* [E,4] walkable-polygon ring edges (x0,y0,x1,y1) of all scenes, concatenated */
  const int32_t* edge_off; /* [S+1] */
  const float* tris;       /* [F,6] navmesh triangles (x0,y0,x1,y1,x2,y2), box scenes only                */
  const int32_t* tri_off;  /* [S+1] */
  const float* floor_height; /* [S] */
  const float* map_lin;    /* [map_res] = torch.linspace(-extent, extent, res)                            */
  int map_res;
  /* scene_kind 2 (main_crowd_eval.py / crowd_env_crowd_eval.py): G members per scene; walkable polygon of member k =
   * square floor [-half,half]^2 minus the world-space marker boxes of the other members ("holes", dummy_vector_env.py:34-39) */
  float* crowd_bbox;       /* [G][S][4] minx,miny,maxx,maxy; read for the others, written for member k        */
  int crowd_group;         /* G */
  int crowd_scenes;        /* S = num_agents of the call                                                    */
  int crowd_member;        /* k: which member the agents of this call are                                   */
  float crowd_floor_half;  /* 4.0 (crowd_env_crowd_eval.py:391)                                             */
  int crowd_polygon;       /* 1: exterior = ring set edges[edge_off[0]..edge_off[1]) (the walkable region of the scene's
                            * navmesh, crowd_env_egobody_eval.py:402,824) instead of the square floor               */
  int crowd_static;        /* 1: the other members' boxes are not holes (effective behaviour of
                            * crowd_env_egobody_eval.py:824 `Polygon(self.scene_poly, holes)`)            */
} egx_env_scenes;

typedef struct egx_env_state {  /* persistent per-agent state, device pointers, updated in place */
  float* state;   /* [A,2,402] canonical markers | marker->target unit vectors */
  float* seed;    /* [A,2,93]  body_param_seed                                  */
  float* R0;      /* [A,3,3]   canonical frame -> world                         */
  float* T0;      /* [A,3]                                                      */
  float* dist;    /* [A]       distance to target at the previous step          */
  int32_t* steps; /* [A]                                                        */
  float* wpath;   /* [A,2,3]   start, target (world)                            */
  int32_t* scene_idx; /* [A] or NULL (single scene); rewritten by egx_env_reset    */
} egx_env_state;

typedef struct egx_env_step_io {
  const float* Y_gen;        /* [18,A,201] from egx_sample_prior                       */
  const float* pred_params;  /* [A,20,93]  from egx_assemble_params                    */
  const float* joints;       /* [A*20,127,3] from egx_lbs_forward                      */
  const float* markers_proj; /* [A*20,67,3]                                            */
  const int32_t* pene_count; /* [A*20] (scene_kind 0) or NULL                          */
  const float* vp_emb;       /* [A*20,32] from egx_vposer_encode                       */
  const int32_t* feet_marker_idx; /* [6] main_ppo.py:298-299                           */
  float* reward;             /* [A]                                                    */
  int32_t* terminated;       /* [A]                                                    */
  float* reward_terms;       /* [A,8] skate, floor, face, look, goal, target_dist, pene, vp; or NULL */
  float* obs_ego;            /* [A,2,32] */
  float* obs_dist;           /* [A]      */
  float* obs_time;           /* [A]      */
  float* out_marker_b;       /* [A,20,67,3] blended markers (save_rollout) or NULL      */
  float* out_prev_frame;     /* [A,12] R0|T0 before the update (save_rollout) or NULL   */
  int32_t* nonfinite_count;  /* device counter or NULL: +1 per agent whose reward / target distance is not finite
                              * (the reference drops into pdb on NaN/Inf, crowd_env_2f.py:287-297; here the host polls
                              * the counter once per collect and raises) */
  int32_t* invalid_flags;    /* [A] or NULL: |= 1 when a pelvis of the 20 frames leaves the scene polygon during the first
                              * 5 steps (scene_kind 2 with crowd_polygon), |= 2 on an unrealistic pose - the two filters
                              * on which crowd_env_egobody_eval.py:208-216,229-234 abandons the sequence               */
} egx_env_step_io;

typedef struct egx_env_reset_io {
  int num_candidates;        /* K start/target candidates per agent: first accepted wins; when none passes, the last is
                              * committed and the agent is reported in out_pending (see below)             */
  const int32_t* mask;       /* [A] 1 = reset, or NULL = all                                              */
  const float* cand_pairs;   /* [A,K,2,3]                                                                 */
  const float* cand_yaw;     /* [A,K] final yaw jitter (box sampler, environments.py:528-538) or NULL     */
  const int32_t* cand_variant; /* [A,K] motion-seed variant or NULL                                        */
  const int32_t* cand_scene; /* [A,K] scene drawn with the candidate (box sampler, environments.py:376-377) or NULL */
  const int32_t* cand_valid; /* [A,K] precomputed acceptance (SDF start check) or NULL                     */
  const float* tab_joints;   /* [NV,2,127,3] motion-seed joints at identity global orient, zero transl     */
  const float* tab_markers;  /* [NV,2,67,3]                                                                */
  const float* tab_glorot;   /* [NV,2,3,3]  mocap global orient as rotation matrices                       */
  const float* tab_transl;   /* [NV,2,3]                                                                   */
  const float* tab_pose;     /* [NV,2,63]                                                                  */
  float* obs_ego;            /* [A,2,32] */
  float* obs_dist;           /* [A]      */
  float* obs_time;           /* [A]      */
  int32_t* out_choice;       /* [A] index of the committed candidate, or NULL                              */
  /* The reference draws until a start passes (`while True: ... if num_pene[0] == 0: break`, crowd_env_2f_box.py:349-416).
   * One launch tries K candidates; out_pending[a] = 1 where all K failed, 0 elsewhere (also for agents the mask skips).  The
   * caller launches again with fresh candidates and mask = out_pending (the same array may be passed as both) until its
   * retry budget is spent; on that last launch it passes forced_count, which then counts the agents that START IN
   * PENETRATION - a deviation from the reference's unbounded loop that must be observable (VecCrowdEnv.forced_accepts). */
  int32_t* out_pending;      /* [A] or NULL */
  int32_t* forced_count;     /* [1] += agents left pending by this launch, or NULL */
} egx_env_reset_io;

/* Yb = cat(seed, Yb_gen) with _blend_params (crowd_env_2f.py:117-123,729-739) -> pred_params [A,20,93] */
int egx_assemble_params(const float* seed, const float* Yb_gen, int num_agents, float* pred_params, void* stream);
/* crowd_env_2f.py:151-317 after the SMPL-X call: rewards, termination, re-canonicalisation, features, egosensing */
int egx_env_step_post(const egx_env_config* cfg, const egx_env_scenes* scenes, const egx_env_state* st,
                      const egx_env_step_io* io, int num_agents, void* stream);
/* scene sampler next_body (environments.py:65-335 / 371-627) + CrowdEnv.reset (crowd_env_2f.py:320-415) */
int egx_env_reset(const egx_env_config* cfg, const egx_env_scenes* scenes, const egx_env_state* st,
                  const egx_env_reset_io* io, int num_agents, void* stream);

/* CrowdEnv._get_feature (crowd_ppo/crowd_env_2f.py:680-727), the two outputs the environment consumes (:261-265,399,412):
 *   Y_l [nb,nt,M,3] canonical markers, pel [nb,nt,3], R0 [nb,3,3], T0 [nb,3], wpath [wpath_rows in {1,nb}, 3] (world target)
 *   out_dist_xyz [nb,nt] = clip(||R0^T (wpath - T0) - pel||, 1e-12)            (may be NULL)
 *   out_fea_marker [nb,nt,M*3] = unit vectors marker -> target (`fea_marker_3d_n`) (may be NULL)
 * The step / reset kernels evaluate the same device functions inline. */
int egx_env_get_feature(const float* Y_l, const float* pel, const float* R0, const float* T0, const float* wpath, int wpath_rows,
                        int num_bodies, int num_frames, int num_markers, float* out_dist_xyz, float* out_fea_marker, void* stream);

/* get_map (exp_GAMMAPrimitive/utils/batch_gen_amass.py:934-968) + the {1,-1} re-coding of crowd_env_2f_box.py:768-769:
 *   tris [F,3,2] navmesh triangles (xy), map_lin [res] = linspace(-extent, extent, res), R [nb,3,3], T [nb,3]
 *   out_points_scene [nb,res*res,3] (z = floor_height; may be NULL), out_local_map [nb,res*res] = 1 walkable / -1 not
 * (meshgrid with 'ij' indexing: point p = (lin[p / res], lin[p % res], 0)). */
int egx_env_get_map(const float* tris, int num_tris, float floor_height, const float* map_lin, int res, const float* R, const float* T,
                    int num_bodies, float* out_points_scene, float* out_local_map, void* stream);

/* ---------------------------------------------------------------------------------------------
 * PPO rollout glue
 * ------------------------------------------------------------------------------------------- */
/* GAMMAPPOPolicy.forward tail (crowd_ppo/ppo_policy.py:169-178): logvar is clamped IN PLACE to [min,max];
 * act = mu + exp(logvar)^0.5 * eps (eps ~ N(0,1) supplied by the caller; ignored when deterministic != 0, the
 * `deterministic_eval and not training` branch); logp [n] (may be NULL) = Independent(Normal,1).log_prob(act). */
int egx_sample_action(const float* mu, float* logvar, const float* eps, float min_logvar, float max_logvar,
                      int deterministic, int num_rows, float* act, float* logp, void* stream);

/* GAE of process_fn/_compute_returns (crowd_ppo/ppo_policy.py:105-140; tianshou compute_episodic_return [upstream]):
 * values [n+1,A] = critic on obs_0..obs_n, rew/terminated [n,A] time-major -> returns, adv [n,A]. */
int egx_gae(const float* values, const float* rew, const int32_t* terminated, int num_steps, int num_agents,
            double gamma, double gae_lambda, float* returns, float* adv, void* stream);

/* Profiling hook for bench.py: the next egx_lbs_forward on this host thread records `start`/`stop`
 * (hipEvent_t passed as void*) immediately around its fused blend+skinning kernel launch. */
int egx_profile_next_lbs(void* start_event, void* stop_event);
int egx_event_create(void** out_event);
int egx_event_destroy(void* event);
int egx_event_elapsed_ms(void* start_event, void* stop_event, float* out_ms); /* synchronises on stop_event */


/* ---------------------------------------------------------------------------------------------
 * PPO update: fused loss + gradient of one minibatch (crowd_ppo/ppo_policy.py:189-241) and the backward of the
 * GRU gate math.  Used inside egx_policy_train_step and as custom autograd nodes by the host (whose products are egx_gemm3).
 *   loss = scale * sum_rows [ -min(r A, clamp(r,1-e,1+e) A) + vf_coef (ret - V)^2 - ent_coef H ],  r = exp(logp - logp_old),
 *   A = (adv - adv_stats[0]) / (adv_stats[1] + adv_eps) when adv_stats != NULL (per-minibatch normalisation, :192-195)
 *   logvar is the RAW actor output; the clamp to [min,max] (ppo_policy.py:169) and its gradient mask are applied inside.
 *   out_terms[6] = loss, loss/clip, loss/vf, loss/ent, loss/kld, mean(logp_old - logp)   (each x scale)
 * ------------------------------------------------------------------------------------------- */
int egx_ppo_loss(const float* mu, const float* logvar, const float* value, const float* act, const float* adv,
                 const float* ret, const float* logp_old, const float* adv_stats, const float* scale, float adv_eps,
                 float min_logvar, float max_logvar, float eps_clip, float vf_coef, float ent_coef, int num_rows,
                 float* g_mu, float* g_logvar, float* g_value, float* out_terms, void* stream);
/* The same on the actor head's raw output zp [n,256] = [mu | logvar] (models_policy_ppo.py:313-317 splits it after the fact):
 * no split / re-join copies on the way in, one gradient tensor g_zp [n,256] on the way out. */
int egx_ppo_loss_packed(const float* zp, const float* value, const float* act, const float* adv, const float* ret,
                        const float* logp_old, const float* adv_stats, const float* scale, float adv_eps, float min_logvar,
                        float max_logvar, float eps_clip, float vf_coef, float ent_coef, int num_rows, float* g_zp, float* g_value,
                        float* out_terms, void* stream);

/* Backward of egx_gru_pointwise: (gi, gh, h_prev, dh) -> d gi, d gh [M,3H], d h_prev [M,H] (NULL to skip).
 * Tensors are dense (row strides H / 3H). */
int egx_gru_pointwise_bwd(const float* gi, const float* gh, const float* h_prev, const float* dh, int num_rows, int hidden,
                          float* dgi, float* dgh, float* dh_prev, void* stream);

/* Dense-layer glue of the PPO update (GAMMAPPOPolicy.learn, crowd_ppo/ppo_policy.py:182-265: loss.backward() through
 * nn.Linear + activation (+ residual) of models_policy_ppo.py:24-39,233-274).  The matrix products are library GEMMs;
 * these two kernels replace the activation / residual / activation-gradient / bias-gradient element-wise passes.
 * activation codes as in egx_linear_desc.  egx_act_fwd: z [M,N] <- act(z) in place; out <- act(z) + res when res != NULL.
 * egx_act_bwd_colsum: g <- dy * act'(a) (a = saved activation output; g may be NULL) and db_accum[n] += sum_m g[m][n]
 * (db_accum may be NULL). */
int egx_act_fwd(float* z, const float* res, float* out, int num_rows, int width, int act, float slope, void* stream);
int egx_act_bwd_colsum(const float* dy, const float* a, float* g, float* db_accum, int num_rows, int width, int act,
                       float slope, void* stream);

/* Minibatch assembly of GAMMAPPOPolicy.learn (ppo_policy.py:189-193: `for minibatch in batch.split(...)`, tianshou
 * Batch indexing [upstream]): dst[t][r, :] = src[t][idx[r], :] for up to 8 dense row-major fp32 tensors in one launch.
 * src / width / dst are HOST arrays of device pointers / row widths. */
int egx_gather_rows(const int64_t* idx, int num_rows, int num_tensors, const float* const* src, const int* width,
                    float* const* dst, void* stream);

/* Advantage normalisation statistics of one minibatch (ppo_policy.py:195-197): out = {mean, unbiased std}. */
int egx_adv_stats(const float* adv, int n, float* out_mean_std, void* stream);

/* Episode statistics of the training collector (tianshou Collector.collect [upstream], crowd_ppo/main_ppo.py:177-183):
 * ep_ret += rew, ep_len += 1; for finished agents the totals are added to done_sums = {sum of returns, sum of lengths,
 * episode count} and the running values reset. */
int egx_track_episode(const float* rew, const int32_t* term, int num_agents, float* ep_ret, float* ep_len, float* done_sums,
                      void* stream);
/* The same bookkeeping fused with the collector's per-step stores (tianshou Collector.collect [upstream] -> ReplayBuffer.add,
 * main_ppo.py:177-183): obs (state [A,2,402], egosensing [A,2,32], dist [A], time [A]) -> rollout slot t+1, reward / terminated
 * -> slot t.  rew_dst / term_dst / ep_ret may be NULL (the first observation of a collect has no reward yet). */
int egx_rollout_store(const float* state, const float* egosensing, const float* dist, const float* time, const float* rew,
                      const int32_t* term, int num_agents, float* state_dst, float* ego_dst, float* dist_dst, float* time_dst,
                      float* rew_dst, int32_t* term_dst, float* ep_ret, float* ep_len, float* done_sums, void* stream);

/* One optimiser step of GAMMAPPOPolicy.learn (ppo_policy.py:243-247: clip_grad_norm_ over the actor+critic parameters,
 * optim.step() with the AdamW of main_ppo.py:134) over FLAT fp32 buffers of n elements: the gradient norm of the first
 * n_clip elements is clipped to max_norm (skipped when max_norm <= 0), then AdamW with decoupled weight decay and bias
 * correction (torch.optim.AdamW arithmetic [upstream torch]) updates param / exp_avg / exp_avg_sq in place.  `step` is a
 * device scalar holding the number of steps taken so far; it is incremented by the call.  workspace: device floats,
 * egx_adamw_workspace_floats() of them; workspace[1024] holds the clip coefficient of the last call.  grad is left unscaled.
 * Hyper-parameters are doubles, as torch passes them. */
size_t egx_adamw_workspace_floats(void);
int egx_adamw_clip_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n, size_t n_clip,
                        float max_norm, double lr, double beta1, double beta2, double eps, double weight_decay, float* step,
                        float* workspace, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* EGOGEN_HIP_H */
