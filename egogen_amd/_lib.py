"""ctypes binding of libegogen_hip.so (the C ABI declared in include/egogen_hip.h).

The product path has NO fallback: if the shared library is missing or a call fails, an exception is
raised.  torch is used only for device memory and streams.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libegogen_hip.so")

c_float_p = C.POINTER(C.c_float)
c_int_p = C.POINTER(C.c_int32)


class EgxError(RuntimeError):
    pass


class BodyModelHost(C.Structure):
    _fields_ = [
        ("num_verts", C.c_int),
        ("v_template_host", C.c_void_p), ("shapedirs_host", C.c_void_p), ("posedirs_host", C.c_void_p),
        ("J_regressor_host", C.c_void_p), ("parents_host", C.c_void_p), ("lbs_weights_host", C.c_void_p),
        ("hand_comps_l_host", C.c_void_p), ("hand_comps_r_host", C.c_void_p),
        ("hand_mean_l_host", C.c_void_p), ("hand_mean_r_host", C.c_void_p),
        ("extra_vids_host", C.c_void_p), ("lmk_vids_host", C.c_void_p), ("lmk_bary_host", C.c_void_p),
        ("num_markers", C.c_int), ("marker_vids_host", C.c_void_p),
        ("num_feet", C.c_int), ("feet_vids_host", C.c_void_p),
    ]


class SdfGrid(C.Structure):
    _fields_ = [("grid", C.c_void_p), ("d0", C.c_int), ("d1", C.c_int), ("d2", C.c_int),
                ("center", C.c_float * 3), ("scale", C.c_float), ("coarse_minmax", C.c_void_p)]


class LinearDesc(C.Structure):
    _fields_ = [("num_rows", C.c_int), ("out_features", C.c_int), ("num_segments", C.c_int),
                ("seg_ptr", C.c_void_p * 4), ("seg_width", C.c_int * 4), ("seg_ld", C.c_int * 4),
                ("weight", C.c_void_p), ("weight_ld", C.c_int), ("bias", C.c_void_p),
                ("residual", C.c_void_p), ("residual_ld", C.c_int),
                ("out", C.c_void_p), ("out_ld", C.c_int), ("activation", C.c_int), ("leaky_slope", C.c_float)]


class PriorPacked3(C.Structure):
    _fields_ = [("x_enc_w_ih", C.c_void_p), ("x_enc_w_hh", C.c_void_p), ("drnn_w", C.c_void_p * 3), ("d_rnn_w_hz", C.c_void_p),
                ("d_rnn_w_y", C.c_void_p), ("d_rnn_w_hh", C.c_void_p), ("d_comb_w", C.c_void_p), ("d_mlp_w", C.c_void_p * 2),
                ("d_out_w", C.c_void_p), ("reg_in_m", C.c_void_p), ("reg_in_xb", C.c_void_p), ("reg_in_betas", C.c_void_p),
                ("reg_blk", C.c_void_p), ("reg_out", C.c_void_p), ("reg_blk_b", C.c_void_p)]


class PriorWeights(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("x_enc_w_ih", "x_enc_w_hh", "x_enc_b_ih", "x_enc_b_hh")] + \
               [("drnn_w", C.c_void_p * 3), ("drnn_b", C.c_void_p * 3)] + \
               [(n, C.c_void_p) for n in ("d_rnn_w_ih", "d_rnn_w_hh", "d_rnn_b_ih", "d_rnn_b_hh")] + \
               [("d_mlp_w", C.c_void_p * 2), ("d_mlp_b", C.c_void_p * 2), ("d_out_w", C.c_void_p), ("d_out_b", C.c_void_p),
                ("reg_in_w", C.c_void_p), ("reg_in_b", C.c_void_p), ("reg_blk_w", C.c_void_p * 20), ("reg_blk_b", C.c_void_p * 20),
                ("reg_out_w", C.c_void_p), ("reg_out_b", C.c_void_p), ("d_comb_w", C.c_void_p), ("d_comb_b", C.c_void_p),
                ("packed3", C.POINTER(PriorPacked3))]


class PolicyPacked3(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("x_enc_w_ih", "x_enc_w_hh", "ego_enc_w_ih", "ego_enc_w_hh")] + \
               [("actor_w", C.c_void_p * 4), ("actor_out_w", C.c_void_p), ("critic_w", C.c_void_p * 4), ("critic_out_w", C.c_void_p)]


class PolicyWeights(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("x_enc_w_ih", "x_enc_w_hh", "x_enc_b_ih", "x_enc_b_hh",
                                          "ego_enc_w_ih", "ego_enc_w_hh", "ego_enc_b_ih", "ego_enc_b_hh")] + \
               [("actor_w", C.c_void_p * 4), ("actor_b", C.c_void_p * 4), ("actor_out_w", C.c_void_p), ("actor_out_b", C.c_void_p),
                ("critic_w", C.c_void_p * 4), ("critic_b", C.c_void_p * 4), ("critic_out_w", C.c_void_p), ("critic_out_b", C.c_void_p),
                ("packed3", C.POINTER(PolicyPacked3))]


class PolicyGrads(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("x_enc_w_ih", "x_enc_w_hh", "x_enc_b_ih", "x_enc_b_hh",
                                          "ego_enc_w_ih", "ego_enc_w_hh", "ego_enc_b_ih", "ego_enc_b_hh")] + \
               [("actor_w", C.c_void_p * 4), ("actor_b", C.c_void_p * 4), ("actor_out_w", C.c_void_p), ("actor_out_b", C.c_void_p),
                ("critic_w", C.c_void_p * 4), ("critic_b", C.c_void_p * 4), ("critic_out_w", C.c_void_p), ("critic_out_b", C.c_void_p)]


class VposerWeights(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("fc1_w", "fc1_b", "fc2_w", "fc2_b", "mu_w", "mu_b", "fc1_w3", "fc2_w3", "mu_w3")]


class EnvConfig(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("reproj_factor", "goal_thresh", "pene_thres", "weight_skate", "weight_floor",
                                         "weight_face_target", "weight_look_target", "weight_success", "weight_target_dist",
                                         "weight_pene", "weight_vp")] + \
               [(n, C.c_int) for n in ("max_depth", "scene_kind", "terminate_on_penetration", "pene_type_body")] + \
               [("ray_len", C.c_float), ("vp_thresh", C.c_float), ("no_goal_termination", C.c_int)]


class EnvScenes(C.Structure):
    _fields_ = [("edges", C.c_void_p), ("edge_off", C.c_void_p), ("tris", C.c_void_p), ("tri_off", C.c_void_p),
                ("floor_height", C.c_void_p), ("map_lin", C.c_void_p), ("map_res", C.c_int),
                ("crowd_bbox", C.c_void_p), ("crowd_group", C.c_int), ("crowd_scenes", C.c_int), ("crowd_member", C.c_int),
                ("crowd_floor_half", C.c_float), ("crowd_polygon", C.c_int), ("crowd_static", C.c_int)]


class EnvState(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("state", "seed", "R0", "T0", "dist", "steps", "wpath", "scene_idx")]


class EnvStepIO(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("Y_gen", "pred_params", "joints", "markers_proj", "pene_count", "vp_emb",
                                          "feet_marker_idx", "reward", "terminated", "reward_terms", "obs_ego", "obs_dist",
                                          "obs_time", "out_marker_b", "out_prev_frame", "nonfinite_count", "invalid_flags")]


class EnvResetIO(C.Structure):
    _fields_ = [("num_candidates", C.c_int)] + \
               [(n, C.c_void_p) for n in ("mask", "cand_pairs", "cand_yaw", "cand_variant", "cand_scene", "cand_valid", "tab_joints",
                                          "tab_markers", "tab_glorot", "tab_transl", "tab_pose", "obs_ego", "obs_dist",
                                          "obs_time", "out_choice", "out_pending", "forced_count")]


# name -> (restype, argtypes); must list every symbol of include/egogen_hip.h
SIGNATURES = {
    "egx_last_error": (C.c_char_p, []),
    "egx_version": (C.c_int, []),
    "egx_body_model_create": (C.c_int, [C.POINTER(BodyModelHost), C.POINTER(C.c_void_p)]),
    "egx_body_model_destroy": (None, [C.c_void_p]),
    "egx_body_model_num_verts": (C.c_int, [C.c_void_p]),
    "egx_body_model_nnz": (C.c_int, [C.c_void_p]),
    "egx_body_model_lbs_vertices": (C.c_int, [C.c_void_p, C.c_int]),
    "egx_lbs_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int]),
    "egx_body_model_culls": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    "egx_lbs_set_culling": (C.c_int, [C.c_int]),
    "egx_lbs_get_culling": (C.c_int, []),
    "egx_lbs_cull_stats": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "egx_lbs_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.POINTER(SdfGrid), C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_size_t, C.c_void_p]),
    "egx_lbs_joints": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "egx_canonical_frame": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "egx_update_transl_glorot": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "egx_sdf_sample": (C.c_int, [C.POINTER(SdfGrid), C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "egx_mesh_sdf": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_float), C.c_float, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "egx_sdf_coarse_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "egx_sdf_build_coarse": (C.c_int, [C.POINTER(SdfGrid), C.c_void_p, C.c_void_p]),
    "egx_linear": (C.c_int, [C.POINTER(LinearDesc), C.c_void_p]),
    "egx_gru_pointwise": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "egx_cont6d_to_aa": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "egx_posenc": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "egx_sample_prior_workspace_bytes": (C.c_size_t, [C.c_int]),
    "egx_sample_prior": (C.c_int, [C.POINTER(PriorWeights), C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "egx_policy_workspace_bytes": (C.c_size_t, [C.c_int]),
    "egx_policy_forward": (C.c_int, [C.POINTER(PolicyWeights), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "egx_assemble_params": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "egx_env_step_post": (C.c_int, [C.POINTER(EnvConfig), C.POINTER(EnvScenes), C.POINTER(EnvState), C.POINTER(EnvStepIO),
                                    C.c_int, C.c_void_p]),
    "egx_env_reset": (C.c_int, [C.POINTER(EnvConfig), C.POINTER(EnvScenes), C.POINTER(EnvState), C.POINTER(EnvResetIO),
                                C.c_int, C.c_void_p]),
    "egx_env_get_feature": (C.c_int, [C.c_void_p] * 5 + [C.c_int] * 4 + [C.c_void_p] * 3),
    "egx_env_get_map": (C.c_int, [C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                  C.c_void_p, C.c_void_p]),
    "egx_sample_action": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_int, C.c_int, C.c_void_p,
                                    C.c_void_p, C.c_void_p]),
    "egx_gae": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double, C.c_void_p,
                          C.c_void_p, C.c_void_p]),
    "egx_ppo_loss": (C.c_int, [C.c_void_p] * 9 + [C.c_float] * 6 + [C.c_int] + [C.c_void_p] * 5),
    "egx_gru_pointwise_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p]),
    "egx_ppo_loss_packed": (C.c_int, [C.c_void_p] * 8 + [C.c_float] * 6 + [C.c_int] + [C.c_void_p] * 4),
    "egx_lbs_set_blend_mode": (C.c_int, [C.c_int]),
    "egx_lbs_get_blend_mode": (C.c_int, []),
    "egx_lbs_set_wave_tile": (C.c_int, [C.c_int]),
    "egx_lbs_get_wave_tile": (C.c_int, []),
    "egx_lbs_set_fix_queue_capacity": (C.c_int, [C.c_int]),
    "egx_lbs_fix_stats": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int32)]),
    "egx_gather_rows": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "egx_adv_stats": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "egx_track_episode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "egx_rollout_store": (C.c_int, [C.c_void_p] * 6 + [C.c_int] + [C.c_void_p] * 10),
    "egx_adamw_workspace_floats": (C.c_size_t, []),
    "egx_adamw_clip_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_float] + [C.c_double] * 5 +
                            [C.c_void_p, C.c_void_p, C.c_void_p]),
    "egx_policy_set_precision": (C.c_int, [C.c_int]),
    "egx_policy_get_precision": (C.c_int, []),
    "egx_act_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p]),
    "egx_act_bwd_colsum": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float,
                                     C.c_void_p]),
    "egx_profile_next_lbs": (C.c_int, [C.c_void_p, C.c_void_p]),
    "egx_event_create": (C.c_int, [C.POINTER(C.c_void_p)]),
    "egx_event_destroy": (C.c_int, [C.c_void_p]),
    "egx_event_elapsed_ms": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_float)]),
    "egx_policy_train_create": (C.c_int, [C.POINTER(PolicyWeights), C.POINTER(PolicyGrads), C.c_int, C.POINTER(C.c_void_p)]),
    "egx_policy_train_destroy": (None, [C.c_void_p]),
    "egx_policy_train_refresh": (C.c_int, [C.c_void_p, C.c_void_p]),
    "egx_policy_train_packed": (C.c_int, [C.c_void_p, C.POINTER(PolicyPacked3)]),
    "egx_policy_train_bind": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "egx_policy_train_step": (C.c_int, [C.c_void_p] * 9 + [C.c_float] * 6 + [C.c_void_p, C.c_void_p]),
    "egx_policy_train_step_heads": (C.c_int, [C.c_void_p] * 9 + [C.c_float] * 6 + [C.c_void_p, C.c_void_p]),
    "egx_policy_train_step_encoders": (C.c_int, [C.c_void_p, C.c_void_p]),
    "egx_policy_train_set_precision": (C.c_int, [C.c_void_p, C.c_int]),
    "egx_pack3_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "egx_pack3": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "egx_gemm3_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "egx_gemm3": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int,
                            C.c_float, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    "egx_vposer_workspace_bytes": (C.c_size_t, [C.c_int]),
    "egx_vposer_encode": (C.c_int, [C.POINTER(VposerWeights), C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                    C.c_size_t, C.c_void_p]),
}

_lib = None


def load():
    """Load the HIP library; raises (never falls back) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EgxError(f"{LIB_PATH} not found - build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                       f"or `make -C egogen_amd/csrc`; there is no CPU fallback")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str):
    if rc != 0:
        msg = load().egx_last_error()
        raise EgxError(f"{what} failed (code {rc}): {msg.decode() if msg else ''}")


def ptr(t):
    """Device pointer of a contiguous torch tensor (None -> NULL)."""
    if t is None:
        return None
    assert t.is_contiguous(), "tensor must be contiguous"
    return C.c_void_p(t.data_ptr())


def current_stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
