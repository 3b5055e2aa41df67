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
#define EGX_BLEND_K 496        /* 10 betas + 486 pose features = GEMM K  */

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

/* Scene SDF (crowd_ppo/utils.py:54-84 `sdf_dict`): grid[d0][d1][d2] indexed by the vertex (x,y,z). */
typedef struct egx_sdf_grid {
  const float* grid; /* device, [d0,d1,d2] */
  int d0, d1, d2;
  float center[3];
  float scale;
} egx_sdf_grid;

/* Bytes of scratch egx_lbs_forward needs for `num_bodies` bodies. */
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
 * egx_sdf_sample - calc_sdf(vertices, sdf_dict) (crowd_ppo/utils.py:54-84): trilinear, border clamp,
 * align_corners=False, negated.  pts [n,3] -> out [n].
 */
int egx_sdf_sample(const egx_sdf_grid* sdf, const float* pts, int64_t n, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* EGOGEN_HIP_H */
