"""Batched CrowdEnv on the GPU: host-side mirror of crowd_ppo/crowd_env_2f.py (SDF scene) and
crowd_ppo/crowd_env_2f_box.py (random box scenes) of the reference, for A independent agents at once.

One `step(actions[A,128])` enqueues, with no host synchronisation:
    egx_sample_prior  ->  egx_assemble_params  ->  egx_lbs_forward (+SDF counts)  ->  egx_vposer_encode
    ->  egx_env_step_post  ->  (auto-reset of finished agents: egx_env_reset)
and can be captured once into a HIP graph (`use_graph=True`).  The reference's per-env gym API
(reset() -> (obs, {}), step(a) -> (obs, reward, terminated, truncated, info)) is provided by `CrowdEnv`,
a 1-agent view, for drop-in use; the vector API is what the trainer uses (the reference's DummyVectorEnv
steps its envs one after the other, main_ppo.py:97).

Observation dict (crowd_env_2f.py:49-51,311-312): state[A,2,402], egosensing[A,2,32], dist[A], time[A].
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional

import numpy as np
import torch

from . import _lib, synth
from .body_model import BodyModelHandle, SdfScene
from .models import GAMMAPrimitiveCombo, VPoserEncoder

DEFAULT_CFG = {  # crowd_ppo/cfg_samp20/MPVAEPolicy_samp_collision.yaml
    "reproj_factor": 0.5, "goal_thresh": 0.1, "max_depth": 13, "pene_thres": 3,
    "weight_vp": 0.1, "weight_floor": 0.1, "weight_skate": 0.3, "weight_target_dist": 1.0,
    "weight_face_target": 0.1, "weight_look_target": 0.3, "weight_pene": 0.1, "weight_success": 0.5,
    "map_res": 16, "map_extent": 0.8, "pene_type": "body", "ray_len": 7.0,
}
BOX_CFG = dict(DEFAULT_CFG, weight_look_target=0.1, max_depth=11)  # ..._collision_2.yaml (primitive_model.py:77-78)


def _rodrigues_np(aa: np.ndarray) -> np.ndarray:
    aa = np.asarray(aa, np.float64)
    th = np.linalg.norm(aa)
    if th < 1e-12:
        return np.eye(3)
    k = aa / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * (K @ K)


CAND_POOL = 16   # steps of reset candidates drawn at a time (SDF scenes)


class VecCrowdEnv:
    def __init__(self, num_agents: int, body_model: BodyModelHandle, prior: GAMMAPrimitiveCombo, vposer: VPoserEncoder,
                 scene_kind: str = "sdf", sdf_dict: Optional[dict] = None, rings: Optional[List[np.ndarray]] = None,
                 pairs: Optional[np.ndarray] = None, box_scenes: Optional[List[dict]] = None,
                 motion_seed: Optional[dict] = None, cfg: Optional[dict] = None, finetuning: bool = False,
                 seed: int = 0, num_candidates: Optional[int] = None, use_graph: bool = False,
                 keep_rollout: bool = False, device: str = "cuda", crowd_bbox: Optional[torch.Tensor] = None,
                 crowd_member: int = 0, crowd_pairs=None, crowd_floor_half: float = 4.0,
                 crowd_rings: Optional[List[np.ndarray]] = None, crowd_static: bool = False, agent_seeds: Optional[List[dict]] = None,
                 vp_thresh: float = 11.0, goal_terminates: bool = True, gender: str = "male", reset_rounds: Optional[int] = None):
        """`reset_rounds`: box scenes - launches of the reset kernel per reset.  The reference draws starts until one passes
        the walkability check (`while True`, crowd_env_2f_box.py:349-416); here a launch tries `num_candidates` (8) draws per
        agent and the agents none of whose draws passed are re-launched with fresh draws (device mask, no host sync) up to
        `reset_rounds` (4) launches = 32 draws; whoever is still without a valid start after the last round keeps its last
        draw and is counted in `forced_accepts()`.
        `gender` names the body model / motion prior the caller passed in (`crowd_env_2f.py:393` picks
        genop_2frame_male | female by it; every sampler but the EgoBody one draws from ['male'], environments.py:254,555,906);
        it is recorded in the saved rollouts.  `crowd_rings` / `crowd_static` / `agent_seeds` / `vp_thresh` / `goal_terminates` are the EgoBody-evaluation variant of
        the crowd scenes (crowd_env_egobody_eval.py, see CrowdGroupEnv): the scene's walkable polygon as exterior, one motion
        seed (two frames + betas) per agent instead of one table for all, the looser pose filter, no goal termination."""
        if not torch.cuda.is_available():
            raise _lib.EgxError("VecCrowdEnv needs a HIP device (no CPU fallback)")
        self.lib = _lib.load()
        self.A = A = int(num_agents)
        self.dev = torch.device(device)
        self.bm, self.prior, self.vposer = body_model, prior, vposer
        self.gender = gender
        self.scene_kind = scene_kind
        self.cfg = dict(cfg or (BOX_CFG if scene_kind in ("box", "crowd") else DEFAULT_CFG))  # main_crowd_eval.py:224 load_model(box=True)
        self.crowd_bbox, self.crowd_member = crowd_bbox, int(crowd_member)
        self.finetuning = finetuning
        self.use_graph = use_graph
        self.keep_rollout = keep_rollout
        self.gen = torch.Generator(device=self.dev)
        self.gen.manual_seed(int(seed))
        f32 = dict(dtype=torch.float32, device=self.dev)
        i32 = dict(dtype=torch.int32, device=self.dev)
        z = lambda *s: torch.zeros(*s, **f32)

        # ---- persistent state ----
        self.state, self.seed = z(A, 2, 402), z(A, 2, 93)
        self.R0, self.T0 = z(A, 3, 3), z(A, 3)
        self.dist, self.wpath = z(A), z(A, 2, 3)
        self.steps = torch.zeros(A, **i32)
        self.scene_idx = torch.zeros(A, **i32)
        A20 = A * 20
        # ---- step intermediates / outputs ----
        self.z = z(A, 128)
        self.Y_gen, self.Yb_gen = z(18, A, 201), z(18, A, 93)
        self.pred_params = z(A, 20, 93)
        self.joints, self.markers = z(A20, synth.NUM_JOINTS_OUT, 3), z(A20, body_model.M, 3)
        self.pene_count = torch.zeros(A20, **i32)
        self.vp_emb = z(A20, 32)
        self.reward, self.rterms = z(A), z(A, 8)
        self.terminated = torch.zeros(A, **i32)
        self.obs_ego, self.obs_dist, self.obs_time = z(A, 2, 32), z(A), z(A)
        self.marker_b = z(A, 20, 67, 3) if keep_rollout else None
        self.prev_frame = z(A, 12) if keep_rollout else None
        self.feet_marker_idx = torch.tensor(synth.feet_marker_idx(), **i32)
        self._lbs_out = {"joints": self.joints, "markers": self.markers, "pene_count": self.pene_count}

        # ---- motion seed (data/locomotion/subseq_00343.npz in the reference) ----
        ms = motion_seed or {k: synth.load_assets()[f"seed_{k}"] for k in ("poses", "trans", "betas")}
        self.motion_seed = {k: np.asarray(v, np.float64) for k, v in ms.items() if k in ("poses", "trans", "betas")}
        self.betas = torch.tensor(self.motion_seed["betas"], **f32).reshape(1, 10).repeat(A, 1).contiguous()
        self.agent_seeds = agent_seeds
        if agent_seeds is not None:
            if scene_kind != "crowd" or len(agent_seeds) != A:
                raise ValueError("agent_seeds: one {poses[2,66], trans[2,3], betas[10]} per agent, crowd scenes only")
            self.betas = torch.tensor(np.stack([np.asarray(d["betas"], np.float32).reshape(10) for d in agent_seeds]), **f32).contiguous()
            starts = list(range(A))                                # table row a = agent a's own two frames
        elif scene_kind == "sdf":
            starts = [5]                                           # environments.py:193 fixed start frame
        elif scene_kind == "crowd" and motion_seed is not None and motion_seed.get("fixed_start") is not None:
            starts = [int(motion_seed["fixed_start"])]
        else:
            starts = list(range(len(self.motion_seed["poses"]) - 1))  # environments.py:483 random start frame
        self.variant_starts = starts
        self._build_seed_tables(starts)
        self.invalid = torch.zeros(A, **i32)   # EgoBody filters (egx_env_step_io.invalid_flags)

        # ---- scenes ----
        self.sdf = None
        self.R = 1   # reset rounds (box scenes: set below)
        if scene_kind == "sdf":
            if sdf_dict is None or rings is None or pairs is None:
                raise ValueError("sdf scene needs sdf_dict, rings (walkable polygon) and start/target pairs")
            self.sdf = sdf_dict if isinstance(sdf_dict, SdfScene) else SdfScene(sdf_dict, device=self.dev)
            edges = [synth.rings_to_edges(rings).astype(np.float32)]
            tris = [np.zeros((0, 6), np.float32)]
            floor = [0.0]
            self.pairs_all = torch.tensor(np.asarray(pairs, np.float32), **f32).reshape(-1, 2, 3)
            self.K = int(num_candidates or 1)
        elif scene_kind == "box":
            if not box_scenes:
                raise ValueError("box scene kind needs box_scenes")
            edges = [np.asarray(s["edges"], np.float32) for s in box_scenes]
            tris = [np.asarray(s["tris"], np.float32).reshape(-1, 6) for s in box_scenes]
            floor = [float(s["floor_height"]) for s in box_scenes]
            P = min(len(s["pairs"]) for s in box_scenes)
            self.box_pairs = torch.tensor(np.stack([np.asarray(s["pairs"][:P], np.float32) for s in box_scenes]), **f32)
            self.K = int(num_candidates or 8)
            self.R = int(reset_rounds or 4)
        elif scene_kind == "crowd":
            # main_crowd_eval.py: every scene holds G members; the walkable polygon of a member is the 8x8 m floor minus
            # the marker boxes of the others (dummy_vector_env.py:34-39, crowd_env_crowd_eval.py:796-822)
            if crowd_bbox is None or crowd_pairs is None or crowd_bbox.shape[1] != A or crowd_bbox.shape[2] != 4:
                raise ValueError("crowd scene kind needs crowd_bbox[G,A,4] and crowd_pairs[A,2,3]")
            edges, tris, floor = [np.zeros((1, 4), np.float32)], [np.zeros((0, 6), np.float32)], [0.0]
            if crowd_rings is not None:
                edges = [synth.rings_to_edges(crowd_rings).astype(np.float32)]
            self.crowd_pairs = torch.tensor(np.asarray(crowd_pairs, np.float32), **f32).reshape(A, 1, 2, 3)
            self.K = 1
        else:
            raise ValueError(f"unknown scene_kind {scene_kind!r}")
        self.num_scenes = len(edges)
        self.edges = torch.tensor(np.concatenate(edges, 0), **f32).contiguous()
        self.edge_off = torch.tensor(np.cumsum([0] + [len(e) for e in edges]), **i32)
        self.tris = torch.tensor(np.concatenate(tris, 0), **f32).contiguous() if sum(len(t) for t in tris) else z(1, 6)
        self.tri_off = torch.tensor(np.cumsum([0] + [len(t) for t in tris]), **i32)
        self.floor_h = torch.tensor(floor, **f32)
        self.map_lin = torch.linspace(-self.cfg["map_extent"], self.cfg["map_extent"], self.cfg["map_res"]).to(self.dev)

        # ---- candidate buffers ----
        K, R = self.K, self.R
        self._cand_pairs_all = z(R, A, K, 2, 3)                   # round-major: round r's draws are one contiguous block
        self._cand_yaw_all = z(R, A, K)
        self._cand_variant_all = torch.zeros(R, A, K, **i32)
        self._cand_scene_all = torch.zeros(R, A, K, **i32)
        self.cand_pairs, self.cand_yaw = self._cand_pairs_all[0], self._cand_yaw_all[0]
        self.cand_variant, self.cand_scene = self._cand_variant_all[0], self._cand_scene_all[0]
        self.choice = torch.zeros(A, **i32)
        self.ones_mask = torch.ones(A, **i32)
        self.pending = torch.zeros(A, **i32)                      # agents whose draws of the last launch all failed the start check
        self.forced = torch.zeros(1, **i32)                       # starts committed in penetration after the last round
        self._rounds_ready = R

        # ---- C structs ----
        c = self.cfg
        ec = _lib.EnvConfig()
        ec.reproj_factor, ec.goal_thresh, ec.pene_thres = c["reproj_factor"], c["goal_thresh"], c["pene_thres"]
        ec.weight_skate, ec.weight_floor, ec.weight_face_target = c["weight_skate"], c["weight_floor"], c["weight_face_target"]
        ec.weight_look_target, ec.weight_success, ec.weight_target_dist = c["weight_look_target"], c["weight_success"], c["weight_target_dist"]
        ec.weight_vp = c["weight_vp"]
        if scene_kind == "sdf":
            ec.weight_pene = 0.1 if finetuning else 1.0          # crowd_env_2f.py:268-271
            ec.terminate_on_penetration = 1 if finetuning else 0  # :299-302
        elif scene_kind == "crowd":
            ec.weight_pene = c["weight_pene"]
            ec.terminate_on_penetration = 0                      # crowd_env_crowd_eval.py:367
        else:
            ec.weight_pene = c["weight_pene"]                    # crowd_env_2f_box.py:303
            ec.terminate_on_penetration = 1                      # :325
        ec.max_depth = int(c["max_depth"])
        ec.scene_kind = {"sdf": 0, "box": 1, "crowd": 2}[scene_kind]
        ec.pene_type_body = 1 if c["pene_type"] == "body" else 0
        ec.ray_len = float(c["ray_len"])
        ec.vp_thresh = float(vp_thresh)
        ec.no_goal_termination = 0 if goal_terminates else 1
        self._ec = ec
        sc = _lib.EnvScenes()
        sc.edges, sc.edge_off, sc.tris, sc.tri_off = self.edges.data_ptr(), self.edge_off.data_ptr(), self.tris.data_ptr(), self.tri_off.data_ptr()
        sc.floor_height, sc.map_lin, sc.map_res = self.floor_h.data_ptr(), self.map_lin.data_ptr(), int(c["map_res"])
        if scene_kind == "crowd":
            sc.crowd_bbox, sc.crowd_group, sc.crowd_scenes = crowd_bbox.data_ptr(), int(crowd_bbox.shape[0]), A
            sc.crowd_member, sc.crowd_floor_half = self.crowd_member, float(crowd_floor_half)
            sc.crowd_polygon, sc.crowd_static = int(crowd_rings is not None), int(bool(crowd_static))
        self._sc = sc
        self._st = self._make_state_struct(self.state, self.seed, self.R0, self.T0, self.dist, self.steps, self.wpath, self.scene_idx)
        io = _lib.EnvStepIO()
        io.Y_gen, io.pred_params, io.joints, io.markers_proj = self.Y_gen.data_ptr(), self.pred_params.data_ptr(), self.joints.data_ptr(), self.markers.data_ptr()
        io.pene_count = self.pene_count.data_ptr() if scene_kind == "sdf" else None
        io.vp_emb, io.feet_marker_idx = self.vp_emb.data_ptr(), self.feet_marker_idx.data_ptr()
        io.reward, io.terminated, io.reward_terms = self.reward.data_ptr(), self.terminated.data_ptr(), self.rterms.data_ptr()
        self.nonfinite = torch.zeros(1, dtype=torch.int32, device=self.dev)
        io.nonfinite_count = self.nonfinite.data_ptr()
        io.invalid_flags = self.invalid.data_ptr()
        io.obs_ego, io.obs_dist, io.obs_time = self.obs_ego.data_ptr(), self.obs_dist.data_ptr(), self.obs_time.data_ptr()
        io.out_marker_b = self.marker_b.data_ptr() if keep_rollout else None
        io.out_prev_frame = self.prev_frame.data_ptr() if keep_rollout else None
        self._io = io

        # workspaces are sized now so that nothing allocates during graph capture
        self.bm.workspace(A20)
        self.prior._ws.get(self.lib.egx_sample_prior_workspace_bytes(A), self.dev)
        if self.vposer._folded is None:
            self.vposer.fold()
        self.prior._weights()
        self._graph = None
        self._injected = False
        self._cand_pool, self._cand_pool_pos, self._cand_step = None, 0, None
        self._box_pool = None
        self._z_in = self.z
        self.profile_events = []  # [(start_event, stop_event)] consumed one pair per step (bench.py)

        self.valid_pairs = None
        if scene_kind == "sdf":
            self._prevalidate_pairs()

    # ------------------------------------------------------------------------------------------
    def _make_state_struct(self, state, seed, R0, T0, dist, steps, wpath, scene_idx):
        st = _lib.EnvState()
        st.state, st.seed, st.R0, st.T0 = state.data_ptr(), seed.data_ptr(), R0.data_ptr(), T0.data_ptr()
        st.dist, st.steps, st.wpath, st.scene_idx = dist.data_ptr(), steps.data_ptr(), wpath.data_ptr(), scene_idx.data_ptr()
        return st

    def _build_seed_tables(self, starts):
        """Motion-seed bodies at identity global orient / zero transl (one LBS call): the scene sampler only
        ever rotates / translates them rigidly (environments.py:216-247), so reset needs no SMPL-X call."""
        ms = self.motion_seed
        NV = len(starts)
        xb = torch.zeros(NV * 2, 93)
        glorot = np.zeros((NV, 2, 3, 3))
        transl = np.zeros((NV, 2, 3))
        pose = np.zeros((NV, 2, 63))
        for v, s in enumerate(starts):
            for f in range(2):
                if self.agent_seeds is not None:
                    d = self.agent_seeds[v]
                    po, tr = np.asarray(d["poses"], np.float64)[f], np.asarray(d["trans"], np.float64)[f]
                else:
                    po, tr = ms["poses"][s + f], ms["trans"][s + f]
                pose[v, f] = po[3:66]
                glorot[v, f] = _rodrigues_np(po[:3])
                transl[v, f] = tr
        xb[:, 6:69] = torch.tensor(pose.reshape(NV * 2, 63), dtype=torch.float32)
        if self.agent_seeds is not None:
            betas_rows = self.betas.repeat_interleave(2, 0).contiguous()
        else:
            betas_rows = torch.tensor(ms["betas"], dtype=torch.float32).reshape(1, 10).to(self.dev).repeat(NV * 2, 1).contiguous()
        out = self.bm.forward(xb.to(self.dev), betas_rows, 1, want_verts=False)
        f32 = dict(dtype=torch.float32, device=self.dev)
        self.tab_joints = out["joints"].reshape(NV, 2, -1, 3).clone().contiguous()
        self.tab_markers = out["markers"].reshape(NV, 2, -1, 3).clone().contiguous()
        self.tab_glorot = torch.tensor(glorot, **f32).contiguous()
        self.tab_transl = torch.tensor(transl, **f32).contiguous()
        self.tab_pose = torch.tensor(pose, **f32).contiguous()

    def _reset_io(self, A, K, mask, cand_pairs, cand_yaw, cand_variant, cand_scene, cand_valid, obs_ego, obs_dist, obs_time, choice,
                  pending=None, forced=None):
        io = _lib.EnvResetIO()
        io.out_pending = pending.data_ptr() if pending is not None else None
        io.forced_count = forced.data_ptr() if forced is not None else None
        io.num_candidates = int(K)
        io.mask = mask.data_ptr() if mask is not None else None
        io.cand_pairs = cand_pairs.data_ptr()
        io.cand_yaw = cand_yaw.data_ptr() if cand_yaw is not None else None
        io.cand_variant = cand_variant.data_ptr() if cand_variant is not None else None
        io.cand_scene = cand_scene.data_ptr() if cand_scene is not None else None
        io.cand_valid = cand_valid.data_ptr() if cand_valid is not None else None
        io.tab_joints, io.tab_markers = self.tab_joints.data_ptr(), self.tab_markers.data_ptr()
        io.tab_glorot, io.tab_transl, io.tab_pose = self.tab_glorot.data_ptr(), self.tab_transl.data_ptr(), self.tab_pose.data_ptr()
        io.obs_ego, io.obs_dist, io.obs_time = obs_ego.data_ptr(), obs_dist.data_ptr(), obs_time.data_ptr()
        io.out_choice = choice.data_ptr() if choice is not None else None
        return io

    def _prevalidate_pairs(self, batch: int = 2048):
        """SDF env: the rejection loop of CrowdEnv.reset (crowd_env_2f.py:326-396) accepts a start iff no non-feet
        vertex of the two seed frames has sdf < 0.  For the room sampler that is a deterministic function of the
        start/target pair, so every pair is evaluated once here (sampler kernel -> SMPL-X + SDF kernel) and reset
        then draws uniformly among the accepted pairs - the same distribution as rejection sampling."""
        N = self.pairs_all.shape[0]
        f32 = dict(dtype=torch.float32, device=self.dev)
        i32 = dict(dtype=torch.int32, device=self.dev)
        valid = torch.zeros(N, dtype=torch.bool, device=self.dev)
        for s in range(0, N, batch):
            n = min(batch, N - s)
            tmp = dict(state=torch.zeros(n, 2, 402, **f32), seed=torch.zeros(n, 2, 93, **f32), R0=torch.zeros(n, 3, 3, **f32),
                       T0=torch.zeros(n, 3, **f32), dist=torch.zeros(n, **f32), steps=torch.zeros(n, **i32),
                       wpath=torch.zeros(n, 2, 3, **f32), scene_idx=torch.zeros(n, **i32))
            st = self._make_state_struct(**tmp)
            ego, od, ot = torch.zeros(n, 2, 32, **f32), torch.zeros(n, **f32), torch.zeros(n, **f32)
            cp = self.pairs_all[s:s + n].reshape(n, 1, 2, 3).contiguous()
            io = self._reset_io(n, 1, None, cp, None, None, None, None, ego, od, ot, None)
            _lib.check(self.lib.egx_env_reset(C.byref(self._ec), C.byref(self._sc), C.byref(st), C.byref(io), n,
                                              _lib.current_stream_ptr()), "egx_env_reset")
            betas = self.betas[:1].repeat(n, 1).contiguous()
            out = self.bm.forward(tmp["seed"].reshape(n * 2, 93), betas, 2, want_joints=False, want_markers=False,
                                  sdf=self.sdf, R0=tmp["R0"], T0=tmp["T0"])
            valid[s:s + n] = out["pene_count"].reshape(n, 2).sum(1) == 0
        self.pair_valid_mask = valid
        self.valid_pairs = self.pairs_all[valid].contiguous()
        if self.valid_pairs.shape[0] == 0:
            raise RuntimeError("no start/target pair passes the SDF start check")

    # ------------------------------------------------------------------------------------------
    def sample_candidates(self):
        """Draw reset candidates for every agent (used only where the mask says so)."""
        A, K, g = self.A, self.K, self.gen
        if self.scene_kind == "sdf":
            # drawn for CAND_POOL steps at a time (one RNG launch + one gather per pool instead of three launches per step);
            # a step's candidates are a slice of the pool, handed to the reset kernel by address
            if self._cand_pool is None or self._cand_pool_pos >= CAND_POOL:
                idx = torch.randint(0, self.valid_pairs.shape[0], (CAND_POOL * A * K,), generator=g, device=self.dev)
                self._cand_pool = self.valid_pairs[idx].reshape(CAND_POOL, A, K, 2, 3)
                self._cand_pool_pos = 0
            self._cand_step = self._cand_pool[self._cand_pool_pos]
            self._cand_pool_pos += 1
        elif self.scene_kind == "crowd":
            self.cand_pairs.copy_(self.crowd_pairs)
            if self.agent_seeds is not None:
                self.cand_variant.copy_(torch.arange(A, dtype=torch.int32, device=self.dev).reshape(A, K))
            else:
                self.cand_variant.copy_(torch.randint(0, len(self.variant_starts), (A, K), generator=g, device=self.dev).to(torch.int32))
            u = torch.rand(A, K, generator=g, device=self.dev) * 2 - 1
            self.cand_yaw.copy_(u * (2 * np.pi * 0.2))           # environments.py:1100-1101 (CrowdMotion.gen_init_body)
        else:
            # the draws of ALL rounds of CAND_POOL steps in one go (the same handful of RNG / gather launches whatever the number
            # of rounds and steps; a step's draws are a slice of the pool, handed to the reset kernel by address); the retry
            # rounds only read theirs for agents that are still pending
            R = self.R
            if self._box_pool is None or self._cand_pool_pos >= CAND_POOL:
                P = CAND_POOL
                sc = torch.randint(0, self.num_scenes, (P * R * A * K,), generator=g, device=self.dev)
                pi = torch.randint(0, self.box_pairs.shape[1], (P * R * A * K,), generator=g, device=self.dev)
                u = torch.rand(P, R, A, K, generator=g, device=self.dev) * 2 - 1
                self._box_pool = {
                    "pairs": self.box_pairs[sc, pi].reshape(P, R, A, K, 2, 3),
                    "scene": sc.reshape(P, R, A, K).to(torch.int32),
                    "variant": torch.randint(0, len(self.variant_starts), (P, R, A, K), generator=g, device=self.dev).to(torch.int32),
                    "yaw": u * (2 * np.pi * 0.1)}                # environments.py:529
                self._cand_pool_pos = 0
            k = self._cand_pool_pos
            self._cand_pool_pos += 1
            bp = self._box_pool
            self._cand_pairs_all, self._cand_scene_all = bp["pairs"][k], bp["scene"][k]
            self._cand_variant_all, self._cand_yaw_all = bp["variant"][k], bp["yaw"][k]
            self.cand_pairs, self.cand_yaw = self._cand_pairs_all[0], self._cand_yaw_all[0]
            self.cand_variant, self.cand_scene = self._cand_variant_all[0], self._cand_scene_all[0]
            self._rounds_ready = R

    def set_candidates(self, pairs, yaw=None, variant=None, scene=None):
        """Inject reset candidates (parity tests): pairs[A,r*K,2,3] = agent a's draws in the order the reference's loop would
        see them, for r <= reset_rounds rounds of K (the next reset runs exactly r launches)."""
        A, K = self.A, self.K
        pairs = torch.as_tensor(pairs, dtype=torch.float32).reshape(A, -1, 2, 3)
        r = pairs.shape[1] // K
        if r < 1 or r > self.R or pairs.shape[1] != r * K:
            raise ValueError(f"candidates per agent must be a multiple of K={K} up to {self.R * K}, got {pairs.shape[1]}")
        rm = lambda t, *tail: t.reshape(A, r, K, *tail).transpose(0, 1)   # [A, r*K, ...] -> round-major [r, A, K, ...]
        if self._box_pool is not None:   # the buffers may be views of the sampler's pool: injected draws get their own
            self._cand_pairs_all, self._cand_yaw_all = self._cand_pairs_all.clone(), self._cand_yaw_all.clone()
            self._cand_variant_all, self._cand_scene_all = self._cand_variant_all.clone(), self._cand_scene_all.clone()
            self.cand_pairs, self.cand_yaw = self._cand_pairs_all[0], self._cand_yaw_all[0]
            self.cand_variant, self.cand_scene = self._cand_variant_all[0], self._cand_scene_all[0]
        self._cand_pairs_all[:r].copy_(rm(pairs, 2, 3))
        self._cand_step = None
        if yaw is not None:
            self._cand_yaw_all[:r].copy_(rm(torch.as_tensor(yaw, dtype=torch.float32)))
        if variant is not None:
            self._cand_variant_all[:r].copy_(rm(torch.as_tensor(variant, dtype=torch.int32)))
        if scene is not None:
            self._cand_scene_all[:r].copy_(rm(torch.as_tensor(scene, dtype=torch.int32)))
        self._rounds_ready = r
        self._injected = True

    def _launch_reset(self, mask):
        box = self.scene_kind in ("box", "crowd")
        rounds = self._rounds_ready if self.scene_kind == "box" else 1
        for r in range(rounds):
            pairs = self._cand_step if self._cand_step is not None else self._cand_pairs_all[r]
            last = r == rounds - 1
            io = self._reset_io(self.A, self.K, mask if r == 0 else self.pending, pairs, self._cand_yaw_all[r] if box else None,
                                self._cand_variant_all[r] if box else None, self._cand_scene_all[r] if self.scene_kind == "box" else None,
                                None, self.obs_ego, self.obs_dist, self.obs_time, self.choice,
                                pending=self.pending if self.scene_kind == "box" else None,
                                forced=self.forced if (self.scene_kind == "box" and last) else None)
            _lib.check(self.lib.egx_env_reset(C.byref(self._ec), C.byref(self._sc), C.byref(self._st), C.byref(io), self.A,
                                              _lib.current_stream_ptr()), "egx_env_reset")

    def forced_accepts(self) -> int:
        """Starts committed although none of an agent's reset_rounds x K draws passed the start check, since construction
        (one host sync).  0 means every episode so far started like the reference's `while True` loop would have started it."""
        return int(self.forced.item())

    def obs(self) -> Dict[str, torch.Tensor]:
        return {"state": self.state, "egosensing": self.obs_ego, "dist": self.obs_dist, "time": self.obs_time}

    def reset(self, mask: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
        if not self._injected:
            self.sample_candidates()
        self._injected = False
        self._launch_reset(mask)
        if mask is None:
            self.invalid.zero_()
        return self.obs()

    def clear_invalid(self, mask: Optional[torch.Tensor] = None):
        """Forget the EgoBody filter flags (`invalid`, OR-accumulated by the step kernel) of the masked agents (all without a
        mask).  A full `reset()` starts new episodes everywhere and clears them; the auto-reset inside `step` does not, so that
        the evaluation driver can still read which scenes tripped a filter after their episodes ended."""
        if mask is None:
            self.invalid.zero_()
        else:
            self.invalid.mul_((mask == 0).to(self.invalid.dtype))

    # ------------------------------------------------------------------------------------------
    def _step_core(self):
        lib, A = self.lib, self.A
        st = _lib.current_stream_ptr()
        # C-VAE decode + regressor (crowd_env_2f.py:109)
        self.prior.sample_prior_into(self.state[:, 0], self.state[:, 1], 804, self.betas, self._z_in, self.Y_gen, self.Yb_gen)
        _lib.check(lib.egx_assemble_params(_lib.ptr(self.seed), _lib.ptr(self.Yb_gen), A, _lib.ptr(self.pred_params), st),
                   "egx_assemble_params")
        # SMPL-X on A*20 bodies; SDF counts fused (crowd_env_2f.py:133-175)
        if self.profile_events:
            ev0, ev1 = self.profile_events.pop(0)
            _lib.check(lib.egx_profile_next_lbs(ev0, ev1), "egx_profile_next_lbs")
        self.bm.forward(self.pred_params.reshape(A * 20, 93), self.betas, 20, want_verts=False,
                        sdf=self.sdf, R0=self.R0 if self.sdf is not None else None,
                        T0=self.T0 if self.sdf is not None else None, out=self._lbs_out)
        # VPoser embedding of the 20 body poses (crowd_env_2f.py:197-198)
        self.vposer.encode_mean_into(self.pred_params.reshape(A * 20, 93)[:, 6:], 93, A * 20, self.vp_emb)
        _lib.check(lib.egx_env_step_post(C.byref(self._ec), C.byref(self._sc), C.byref(self._st), C.byref(self._io), A, st),
                   "egx_env_step_post")

    def check_finite(self, all_ranks: bool = False):
        """Raise if any step since the last call produced a non-finite reward / target distance (one host sync; the
        reference stops in pdb at the NaN/Inf checks of crowd_env_2f.py:287-297).  `all_ranks`: MAX-reduce the counter over
        the data-parallel ranks first, so that all of them raise together."""
        if all_ranks:
            import torch.distributed as dist
            dist.all_reduce(self.nonfinite, op=dist.ReduceOp.MAX)
        n = int(self.nonfinite.item())
        if n:
            self.nonfinite.zero_()
            raise FloatingPointError(f"{n} agent-steps produced non-finite rewards since the last check")

    def step(self, actions: torch.Tensor, auto_reset: bool = True):
        """actions[A,128] -> (obs, reward[A], terminated[A] int32).  Returned tensors are views of internal
        buffers, valid until the next call.  With auto_reset, finished agents are re-initialised and their
        entries of obs are the reset observation (tianshou Collector semantics)."""
        if actions.shape != (self.A, 128):
            raise ValueError(f"actions must be [{self.A},128], got {tuple(actions.shape)}")
        direct = (not self.use_graph and actions.is_cuda and actions.dtype == torch.float32 and actions.is_contiguous()
                  and actions.device == self.z.device)
        self._z_in = actions if direct else self.z      # the kernels read the caller's tensor where that is safe: no copy
        if not direct:
            self.z.copy_(actions)
        if self.use_graph:
            if self._graph is None:
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    self._step_core()
                self._graph = g
            self._graph.replay()
        else:
            self._step_core()
        if auto_reset:
            if not self._injected:
                self.sample_candidates()
            self._injected = False
            self._launch_reset(self.terminated)
        return self.obs(), self.reward, self.terminated


class CrowdEnv:
    """Single-agent gym-style view with the reference's interface (crowd_env_2f.py:34-51,78,320,519)."""

    def __init__(self, vec_env: VecCrowdEnv):
        if vec_env.A != 1:
            raise ValueError("CrowdEnv wraps a 1-agent VecCrowdEnv")
        self.vec = vec_env
        from .spaces import crowd_env_spaces
        self.action_space, self.observation_space = crowd_env_spaces()   # crowd_env_2f.py:49-51 (gymnasium's when installed)

    def seed(self, seed):
        self.vec.gen.manual_seed(int(seed))

    def _obs0(self):
        o = self.vec.obs()
        return {"state": o["state"][0], "egosensing": o["egosensing"][0], "dist": o["dist"][0:1], "time": o["time"][0:1]}

    def reset(self, seed=None, options=None):
        if seed is not None:
            self.seed(seed)
        self.vec.reset()
        return self._obs0(), {}

    def step(self, action_z):
        a = torch.as_tensor(action_z, dtype=torch.float32, device=self.vec.dev).reshape(1, 128)
        _, rew, term = self.vec.step(a, auto_reset=False)
        return self._obs0(), float(rew[0].item()), bool(term[0].item()), False, {}


class CrowdGroupEnv:
    """G interacting members per scene, S independent scenes (the reference's main_crowd_eval.py runs S = 1, G = 4 with
    `DummyCrowdVectorEnv`).  Member k of every scene lives in sub-environment k; all share one table of world-space
    marker boxes.  `step` walks the members in order, so member k sees the boxes members 0..k-1 published in THIS round
    and those of k+1.. from the previous round - the ordering of dummy_vector_env.py:81-84."""

    def __init__(self, num_scenes: int, start_target, body_model, prior, vposer, cfg=None, seed=0, keep_rollout=False,
                 motion_seed=None, floor_half: float = 4.0, device="cuda", scene_rings=None, static_scene: bool = False,
                 agent_seeds=None, vp_thresh: float = 11.0, goal_terminates: bool = True, gender: str = "male"):
        """`scene_rings` ... `goal_terminates`: the EgoBody evaluation (main_egobody_eval.py / crowd_env_egobody_eval.py, G = 2):
        the walkable polygon of the scene's navmesh as exterior (:402), `agent_seeds[k][s]` = the two seed frames + betas of
        member k in scene s (Egobody.gen_init_body, environments.py:679-765), pose filter at 14 (:229), only max_depth
        terminates (:378).  `static_scene=True` reproduces what `Polygon(self.scene_poly, holes)` (:824) evaluates to when
        the shell is already a Polygon - shapely returns that polygon, the other person never becomes a hole."""
        st = np.asarray(start_target, np.float32)          # [G,S,2,3]
        self.G, self.S = int(st.shape[0]), int(num_scenes)
        assert st.shape[1] == self.S
        self.bbox = torch.zeros(self.G, self.S, 4, dtype=torch.float32, device=device)
        self.members = [VecCrowdEnv(self.S, body_model, prior, vposer, scene_kind="crowd", cfg=cfg, seed=seed + 17 * k,
                                    keep_rollout=keep_rollout, motion_seed=motion_seed, crowd_bbox=self.bbox, crowd_member=k,
                                    crowd_pairs=st[k], crowd_floor_half=floor_half, device=device, crowd_rings=scene_rings,
                                    crowd_static=static_scene, agent_seeds=None if agent_seeds is None else agent_seeds[k],
                                    vp_thresh=vp_thresh, goal_terminates=goal_terminates, gender=gender) for k in range(self.G)]
        self.gender = gender

    def invalid(self) -> torch.Tensor:
        """[S] OR of the members' filter flags (1: pelvis left the scene polygon early, 2: unrealistic pose)."""
        out = self.members[0].invalid.clone()
        for m in self.members[1:]:
            out |= m.invalid
        return out

    def reset(self):
        """Every member publishes its initial box (constructor of crowd_env_crowd_eval.CrowdEnv, :54-75) before anyone
        builds its first observation (DummyCrowdVectorEnv.__init__ -> update_holes_for_each_agent)."""
        for m in self.members:
            m.sample_candidates()
            m._injected = True
            m._launch_reset(None)
            m.invalid.zero_()            # new episodes everywhere: the filter flags of the previous ones are history
        obs = []
        for m in self.members:
            m._launch_reset(None)        # same candidates: identical state, observation now sees all boxes
            m._injected = False
            obs.append(m.obs())
        return obs

    def step(self, actions, reset_done: bool = True):
        """actions: list of G tensors [S,128].  Returns per-member (obs, reward, terminated)."""
        out = []
        for k, m in enumerate(self.members):
            o, r, t = m.step(actions[k], auto_reset=False)
            r, t = r.clone(), t.clone()
            if reset_done:                # a finished member restarts from its own fixed start (collector reset)
                m.sample_candidates()
                m._injected = True
                m._launch_reset(m.terminated)
                m._injected = False
            out.append((m.obs(), r, t))
        return out
