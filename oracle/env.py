"""CrowdEnv step / reset / features / rewards / egosensing, CPU restatement, batched over agents.
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

The reference env holds ONE agent replicated x4 (crowd_env_2f.py:29-32) and uses element [0];
this restatement carries an explicit agent axis `A` instead (every agent independent), which is
the same arithmetic per agent.  Reference file:line cited per function.  Third-party pieces
(shapely ray casting, smplx, tgm) are restated algorithms - see oracle/__init__.py for pinning.
"""
import math
from typing import Dict, List, Optional

import numpy as np
import torch

from . import nets, rot
from .sdf import calc_sdf
from .smplx_lbs import BodyModel, smplx_forward

T_HIS, T_PRED, T_ALL = 2, 18, 20


# ---------------------------------------------------------------------------------------------
# geometry helpers
# ---------------------------------------------------------------------------------------------

def get_new_coordinate(jts: torch.Tensor):
    """models/baseops.py:214-225 get_new_coordinate_torch: jts[b,J,3] -> R[b,3,3], T[b,1,3].
    x = j2-j1 with z zeroed, normalised WITHOUT eps; z=(0,0,1); y = normalise(z x x)."""
    x = jts[:, 2, :] - jts[:, 1, :]
    x = x.clone()
    x[:, -1] = 0
    x = x / torch.norm(x, dim=-1, keepdim=True)
    z = torch.zeros_like(x)
    z[:, 2] = 1
    y = torch.cross(z, x, dim=-1)
    y = y / torch.norm(y, dim=-1, keepdim=True)
    return torch.stack([x, y, z], dim=-1), jts[:, :1]


def update_transl_glorot(R, T, delta_T, xb):
    """models/baseops.py:537-598 (torch branch, inplace=False).  R[b,3,3], T[b,1,3],
    delta_T[b,3] = root joint at zero global orient / transl (calc_calibrate_offset :494-534)."""
    transl, glorot = xb[:, :3], xb[:, 3:6]
    go = rot.tgm_angle_axis_to_rotation_matrix(glorot)
    go_new = torch.einsum("bij,bjk->bik", R.permute(0, 2, 1), go)
    glorot_new = rot.tgm_rotation_matrix_to_angle_axis(go_new)
    transl_new = torch.einsum("bij,bj->bi", R.permute(0, 2, 1), transl + delta_T - T[:, 0]) - delta_T
    return torch.cat([transl_new, glorot_new, xb[:, 6:]], dim=1)


def get_feature(Y_l, pel, R0, T0, pt_wpath):
    """crowd_env_2f.py:680-727.  Y_l[b,t,201], pel[b,t,3], pt_wpath[b,1,3] (reference: [1,3] shared).
    Returns dist_xy, dist_xyz, fea_wpath, fea_marker_3d_n, fea_marker_h."""
    nb, nt = pel.shape[:2]
    Y_l = Y_l.reshape(nb, nt, -1, 3)
    w3 = torch.einsum("bij,btj->bti", R0.permute(0, 2, 1), pt_wpath - T0)
    fxy = w3[:, :, :2] - pel[:, :, :2]
    fxyz = w3[:, :, :3] - pel[:, :, :3]
    dist_xy = torch.norm(fxy, dim=-1, keepdim=True).clip(min=1e-12)
    dist_xyz = torch.norm(fxyz, dim=-1, keepdim=True).clip(min=1e-12)
    fxy = fxy / dist_xy
    fz = w3[:, :, -1:] - pel[:, :, -1:]
    fea_wpath = torch.cat([fxy, fz], dim=-1)
    fm = w3[:, :, None, :] - Y_l
    d3 = torch.norm(fm, dim=-1, keepdim=True).clip(min=1e-12)
    fea_marker_3d_n = (fm / d3).reshape(nb, nt, -1)
    d2 = torch.norm(fm[:, :, :, :2], dim=-1, keepdim=True).clip(min=1e-12)
    fea_marker_h = torch.cat([fm[:, :, :, :2] / d2, fm[:, :, :, -1:]], dim=-1).reshape(nb, nt, -1)
    return dist_xy, dist_xyz, fea_wpath, fea_marker_3d_n, fea_marker_h


def blend_params(body_params: torch.Tensor, t_his: int = T_HIS) -> torch.Tensor:
    """crowd_env_2f.py:729-739 _blend_params: body_params[t,b,93] modified IN PLACE - frames t_his and t_his+1 become the
    mean of their neighbours, pose part ([6:]) only, sequentially (the second uses the already smoothed first)."""
    for t in (t_his, t_his + 1):
        body_params[t, :, 6:] = (body_params[t - 1, :, 6:] + body_params[t + 1, :, 6:]) / 2.0
    return body_params


def get_map(tris: torch.Tensor, R, T, res=16, extent=0.8, floor_height=0.0):
    """exp_GAMMAPrimitive/utils/batch_gen_amass.py:934-968.  tris[F,3,2] (navmesh triangles, xy),
    R[b,3,3], T[b,1,3] -> points_local[b,res*res,3], points_scene, map bool[b,res*res]."""
    b = R.shape[0]
    lin = torch.linspace(-extent, extent, res)
    xv, yv = torch.meshgrid(lin, lin, indexing="ij")
    pts = torch.stack([xv, yv, torch.zeros_like(xv)], dim=2).reshape(1, -1, 3).repeat(b, 1, 1).to(R.dtype)
    ps = torch.einsum("bij,bpj->bpi", R, pts) + T
    ps[:, :, 2] = floor_height
    p = ps[:, :, :2].reshape(b * res * res, 1, 2)
    tri = tris.to(R.dtype)[None]  # [1,F,3,2]

    def sign(p1, p2, p3):
        return (p1[:, :, 0] - p3[:, :, 0]) * (p2[:, :, 1] - p3[:, :, 1]) - (p2[:, :, 0] - p3[:, :, 0]) * (p1[:, :, 1] - p3[:, :, 1])

    d1 = sign(p, tri[:, :, 0, :], tri[:, :, 1, :])
    d2 = sign(p, tri[:, :, 1, :], tri[:, :, 2, :])
    d3 = sign(p, tri[:, :, 2, :], tri[:, :, 0, :])
    neg = (d1 < 0) | (d2 < 0) | (d3 < 0)
    pos = (d1 > 0) | (d2 > 0) | (d3 > 0)
    inside = (~(neg & pos)).any(-1)
    return pts, ps, inside.reshape(b, res * res)


def _point_in_rings(px, py, edges: np.ndarray) -> bool:
    """Even-odd containment over all ring edges [E,4] (exterior + holes), float64.
    shapely Polygon.contains semantics up to boundary measure-zero cases."""
    x0, y0, x1, y1 = edges[:, 0], edges[:, 1], edges[:, 2], edges[:, 3]
    cond = (y0 > py) != (y1 > py)
    with np.errstate(divide="ignore", invalid="ignore"):
        xint = x0 + (py - y0) * (x1 - x0) / (y1 - y0)
    return bool(np.count_nonzero(cond & (px < xint)) % 2 == 1)


def points_in_rings(edges: np.ndarray, x: np.ndarray, y: np.ndarray) -> np.ndarray:
    """Even-odd containment of points in a ring set given as edges[E,4] (float64) - shapely Polygon.contains up to
    boundary measure-zero cases."""
    x, y = np.asarray(x, np.float64), np.asarray(y, np.float64)
    x0, y0, x1, y1 = (edges[:, k][None, :] for k in range(4))
    with np.errstate(divide="ignore", invalid="ignore"):
        straddle = (y0 > y[:, None]) != (y1 > y[:, None])
        xint = x0 + (y[:, None] - y0) * (x1 - x0) / (y1 - y0)
    return (np.sum(straddle & (x[:, None] < xint), axis=1) & 1) == 1


def calc_egosensing(joints_w: torch.Tensor, edges: np.ndarray, ray_len: float = 7.0) -> torch.Tensor:
    """crowd_env_2f.py:524-613 for one agent.  joints_w[2,127,3] float32 (world), edges[E,4] float64.
    Returns float32[2,32] = -1 + 2*d/ray_len, d = distance from the eye mid-point to the first exit
    of the walkable polygon along each of 32 rays over [-pi/2, pi/2] about the look-at direction
    (ray end if no exit); eye outside the polygon -> d = 0.
    shapely's LineString.intersection is restated as nearest ray/segment hit [PARITY UNPINNED]."""
    joint = joints_w.detach().cpu().numpy().astype(np.float32)
    look_at = joint[:, 57] - joint[:, 23] + joint[:, 56] - joint[:, 24]
    look_at = look_at.astype(np.float64)
    look_at[:, -1] = 0.0
    look_at = look_at / np.linalg.norm(look_at, axis=-1, keepdims=True)
    eye_2d = (joint[:, 23] + joint[:, 24]) / 2  # float32, like the reference
    eye_2d[:, -1] = 0.0
    ang = np.linspace(-np.pi / 2, np.pi / 2, 32)
    out = np.zeros((2, 32), np.float64)
    edges = np.asarray(edges, np.float64)
    ex0, ey0 = edges[:, 0], edges[:, 1]
    edx, edy = edges[:, 2] - edges[:, 0], edges[:, 3] - edges[:, 1]
    for t in range(2):
        ox, oy = float(eye_2d[t, 0]), float(eye_2d[t, 1])
        if not _point_in_rings(ox, oy, edges):
            out[t] = 0.0
            continue
        c, s = look_at[t, 0], look_at[t, 1]
        for i in range(32):
            dx = c * np.cos(ang[i]) - s * np.sin(ang[i])
            dy = s * np.cos(ang[i]) + c * np.sin(ang[i])
            # ray o + t*d (t in (0,ray_len]) vs edge e0 + u*ed (u in [0,1])
            den = dx * edy - dy * edx
            with np.errstate(divide="ignore", invalid="ignore"):
                tt = ((ex0 - ox) * edy - (ey0 - oy) * edx) / den
                uu = ((ex0 - ox) * dy - (ey0 - oy) * dx) / den
            ok = (np.abs(den) > 0) & (tt > 0) & (tt <= ray_len) & (uu >= 0) & (uu <= 1)
            d = tt[ok].min() if ok.any() else ray_len
            # the reference measures |end - eye| in float64 from the returned coordinates
            end = np.array([ox + dx * d, oy + dy * d])
            out[t, i] = np.linalg.norm(end - np.array([ox, oy]))
    return torch.from_numpy((-1 + 2 * (out / ray_len).astype(np.float32)).astype(np.float32))


# ---------------------------------------------------------------------------------------------
# the environment
# ---------------------------------------------------------------------------------------------

DEFAULT_CFG = {
    # crowd_ppo/cfg_samp20/MPVAEPolicy_samp_collision.yaml
    "reproj_factor": 0.5, "goal_thresh": 0.1, "max_depth": 13, "pene_thres": 3,
    "weight_vp": 0.1, "weight_floor": 0.1, "weight_skate": 0.3, "weight_target_dist": 1.0,
    "weight_face_target": 0.1, "weight_look_target": 0.3, "weight_pene": 0.1, "weight_success": 0.5,
    "map_res": 16, "map_extent": 0.8, "pene_type": "body",
}
BOX_CFG = dict(DEFAULT_CFG, weight_look_target=0.1, max_depth=11)  # MPVAEPolicy_samp_collision_2.yaml


class OracleCrowdEnv:
    """Batched restatement of crowd_env_2f.CrowdEnv (scene_kind='sdf') and
    crowd_env_2f_box.CrowdEnv (scene_kind='box')."""

    def __init__(self, bm: BodyModel, prior_sd: Dict[str, torch.Tensor], vposer_sd: Dict[str, torch.Tensor],
                 marker_ids, feet_vids, feet_marker_idx, scene_kind: str = "sdf",
                 sdf_dict: Optional[Dict[str, torch.Tensor]] = None, edges: Optional[np.ndarray] = None,
                 box_scenes: Optional[List[dict]] = None, cfg: Optional[dict] = None, finetuning: bool = False):
        self.bm = bm
        self.dt = bm.dtype
        self.prior_sd = {k: v.to(self.dt) for k, v in prior_sd.items()}
        self.vposer_sd = {k: v.to(self.dt) for k, v in vposer_sd.items()}
        self.marker_ids = torch.as_tensor(np.asarray(marker_ids), dtype=torch.long)
        self.feet_vids = torch.as_tensor(np.asarray(feet_vids), dtype=torch.long)
        self.feet_marker_idx = list(feet_marker_idx)
        self.scene_kind = scene_kind
        self.sdf_dict = sdf_dict
        self.edges = edges
        self.box_scenes = box_scenes
        self.cfg = dict(cfg or (BOX_CFG if scene_kind in ("box", "crowd") else DEFAULT_CFG))  # main_crowd_eval.py:224 load_model(box=True)
        self.finetuning = finetuning
        self.last = {}

    # ---- state container -----------------------------------------------------------------
    def set_state(self, state, body_param_seed, R0, T0, betas, dist, steps, wpath, scene_idx=None):
        """state[A,2,402], body_param_seed[A,2,93], R0[A,3,3], T0[A,1,3], betas[A,10], dist[A],
        steps int[A], wpath[A,2,3], scene_idx int[A] (box scenes)."""
        f = lambda x: torch.as_tensor(x).to(self.dt).clone()
        self.state, self.body_param_seed, self.R0, self.T0 = f(state), f(body_param_seed), f(R0), f(T0)
        self.betas, self.dist, self.wpath = f(betas), f(dist), f(wpath)
        self.steps = torch.as_tensor(steps).long().clone()
        self.scene_idx = None if scene_idx is None else torch.as_tensor(scene_idx).long().clone()

    def _smplx(self, xb, betas_per_row):
        return smplx_forward(self.bm, xb, betas_per_row)

    def _delta_T(self, betas):
        """calc_calibrate_offset (baseops.py:494-534): root joint with zero pose/orient/transl.
        The body pose does not move the root, so evaluate the rest pose once per agent."""
        A = betas.shape[0]
        _, j = self._smplx(torch.zeros(A, 93, dtype=self.dt), betas)
        return j[:, 0]

    def _walk_map(self, R0, T0):
        """box env _get_feature tail (crowd_env_2f_box.py:762-770): local_map in {1,-1}."""
        pts_l, maps = [], []
        for a in range(R0.shape[0]):
            sc = self.box_scenes[int(self.scene_idx[a])]
            p, _, m = get_map(torch.as_tensor(sc["tris"]), R0[a:a + 1], T0[a:a + 1], self.cfg["map_res"],
                              self.cfg["map_extent"], float(sc["floor_height"]))
            pts_l.append(p)
            maps.append(m)
        pts_l = torch.cat(pts_l)
        m = torch.cat(maps)
        local_map = m.to(self.dt)
        local_map[~m] = -1
        return pts_l, local_map

    # ---- crowd scenes (crowd_env_crowd_eval.py) -------------------------------------------------------
    def set_crowd_boxes(self, boxes, floor_half=4.0):
        """boxes[A,G-1,4] = (minx,miny,maxx,maxy) of the OTHER members of each agent's scene (the `holes`)."""
        self.crowd_boxes = np.asarray(boxes, np.float64)
        self.floor_half = float(floor_half)

    def set_egobody(self, scene_edges, static=True, vp_thresh=14.0):
        """crowd_env_egobody_eval.py: the exterior is the scene's walkable region (`scene_poly`, :402) instead of the square
        floor; only max_depth terminates (:378); vp_norm > 14 (:229) and a pelvis outside the scene polygon during the first
        5 steps (:208-216) abandon the sequence (flagged in last['invalid']).  `static`: `Polygon(self.scene_poly, holes)`
        (:824) is called with a Polygon as shell, for which shapely returns the shell itself - the other person's box never
        becomes a hole [PARITY UNPINNED: shapely absent here]; static=False gives the evidently intended polygon."""
        self.ego_edges = np.asarray(scene_edges, np.float64)
        self.ego_static = bool(static)
        self.vp_thresh = float(vp_thresh)

    def own_bbox(self):
        """world-space xy box of the two seed frames' markers (crowd_env_crowd_eval.py:345-352)."""
        A = self.state.shape[0]
        m = self.state[:, :, :201].reshape(A, 2, -1, 3)
        w = torch.einsum("bij,btpj->btpi", self.R0, m) + self.T0[:, None, :, :]
        xy = w[..., :2]
        return torch.cat([xy.amin(dim=(1, 2)), xy.amax(dim=(1, 2))], dim=-1)

    def _crowd_edges(self, a):
        h = self.floor_half
        ego = getattr(self, "ego_edges", None)
        rects = ([] if ego is not None else [(-h, -h, h, h)]) + \
                ([] if ego is not None and self.ego_static else [tuple(b) for b in self.crowd_boxes[a]])
        es = [] if ego is None else [list(e) for e in ego]
        for (x0, y0, x1, y1) in rects:
            c = [(x0, y0), (x1, y0), (x1, y1), (x0, y1)]
            for q in range(4):
                es.append([c[q][0], c[q][1], c[(q + 1) % 4][0], c[(q + 1) % 4][1]])
        return np.asarray(es, np.float64)

    def _crowd_walk_map(self, R0, T0):
        """_get_dynamic_map (crowd_env_crowd_eval.py:742-764): polygon(floor, holes).contains(point)."""
        A = R0.shape[0]
        res, ext = self.cfg["map_res"], self.cfg["map_extent"]
        lin = torch.linspace(-ext, ext, res)
        xv, yv = torch.meshgrid(lin, lin, indexing="ij")
        pts = torch.stack([xv, yv, torch.zeros_like(xv)], dim=2).reshape(1, -1, 3).repeat(A, 1, 1).to(R0.dtype)
        ps = torch.einsum("bij,bpj->bpi", R0, pts) + T0
        px, py = ps[:, :, 0], ps[:, :, 1]
        ego = getattr(self, "ego_edges", None)
        if ego is None:
            walk = (px.abs() < self.floor_half) & (py.abs() < self.floor_half)
        else:
            walk = torch.as_tensor(points_in_rings(ego, px.double().numpy().ravel(), py.double().numpy().ravel())).reshape(px.shape)
        for a in range(A):
            for b in ([] if ego is not None and self.ego_static else self.crowd_boxes[a]):
                inside = (px[a] >= b[0]) & (px[a] <= b[2]) & (py[a] >= b[1]) & (py[a] <= b[3])
                walk[a] &= ~inside
        local_map = walk.to(R0.dtype)
        local_map[~walk] = -1
        return pts, local_map

    def _edges_for(self, a):
        if self.scene_kind == "crowd":
            return self._crowd_edges(a)
        if self.scene_kind == "box":
            return np.asarray(self.box_scenes[int(self.scene_idx[a])]["edges"], np.float64)
        return self.edges

    # ---- step ------------------------------------------------------------------------------
    def step(self, action_z: torch.Tensor):
        """crowd_env_2f.py:78-317 (sdf) / crowd_env_2f_box.py:78-340 (box).  action_z[A,128].
        Returns obs dict, reward[A], terminated bool[A]; all intermediate quantities in self.last."""
        cfg = self.cfg
        A = action_z.shape[0]
        dt = self.dt
        self.steps = self.steps + 1
        X = self.state[:, :, :201].permute(1, 0, 2)                       # [2,A,201]
        Xb = self.body_param_seed.permute(1, 0, 2)                        # [2,A,93]
        betas18 = self.betas[None].repeat(T_PRED, 1, 1)
        Y_gen, Yb_gen = nets.sample_prior(self.prior_sd, X, betas18, action_z.to(dt))
        Y = torch.cat([X, Y_gen], dim=0)                                  # [20,A,201]
        Yb = torch.cat([Xb, Yb_gen], dim=0).clone()                       # [20,A,93]
        Yb = blend_params(Yb, T_HIS)                                      # :120-123
        pred_markers = Y.reshape(T_ALL, A, -1, 3).permute(1, 0, 2, 3)     # [A,20,67,3]
        pred_params = Yb.permute(1, 0, 2).contiguous()                    # [A,20,93]
        betas_rows = self.betas[:, None, :].expand(A, T_ALL, 10).reshape(A * T_ALL, 10)
        verts, joints = self._smplx(pred_params.reshape(A * T_ALL, 93), betas_rows)
        joints_all = joints.reshape(A, T_ALL, -1, 3)
        pred_joints = joints_all[:, :, :22]
        pelvis = pred_joints[:, :, 0]
        markers_proj = verts[:, self.marker_ids].reshape(A, T_ALL, -1, 3)
        rf = cfg["reproj_factor"]
        marker_b = rf * markers_proj + (1 - rf) * pred_markers
        R0, T0 = self.R0, self.T0

        # ---- penetration (sdf env :161-177) ----
        if self.scene_kind == "sdf":
            vw = torch.einsum("bij,btpj->btpi", R0, verts.reshape(A, T_ALL, -1, 3)) + T0[:, None, :, :]
            sv = calc_sdf(vw.reshape(A * T_ALL, -1, 3), self.sdf_dict).reshape(A, T_ALL, -1)
            sv[:, :, self.feet_vids] = 0.0
            inside = sv.lt(0.0)
            cnt = inside.sum(dim=-1)                                      # [A,20]
            near = sv.abs() < getattr(self, "level_set_band", 2e-5)       # test diagnostics: vertices within fp32 round-off of
            near[:, :, self.feet_vids] = False                            # the zero level set (their sign is not reproducible)
            self.last["pene_near_zero"] = near.sum(dim=-1)
            num_inside = cnt.sum(dim=1).to(dt) / T_ALL / 10
            penetration = cnt.max(dim=1).values >= 40
            r_pene = torch.exp(-num_inside)
            self.last["pene_count"] = cnt

        # ---- skate (:181-185) ----
        h = 1 / 40
        speed = torch.norm(marker_b[:, 2:] - marker_b[:, :-2], dim=-1) / 2.0 / h
        dist2skat = (speed[:, :, self.feet_marker_idx].amin(dim=-1) - 0.075).clamp(min=0).mean(dim=-1)
        r_skate = torch.exp(-dist2skat)
        # ---- floor (:190-194) ----
        marker_w = torch.einsum("bij,btpj->btpi", R0, marker_b) + T0[:, None, :, :]
        dist2gp = torch.abs(marker_w[:, :, self.feet_marker_idx, 2].amin(dim=-1) - 0.02).mean(dim=-1)
        r_floor = torch.exp(-dist2gp)
        # ---- vposer (:196-204) ----
        emb = nets.vposer_encode(self.vposer_sd, pred_params[:, :, 6:69].reshape(A * T_ALL, -1))
        vp_norm = torch.norm(emb.reshape(A, T_ALL, -1), dim=-1).mean(dim=1)
        vp_thresh = getattr(self, "vp_thresh", 11.0)
        r_vp = torch.where(vp_norm > vp_thresh, torch.zeros_like(vp_norm), torch.full_like(vp_norm, 0.05))
        # ---- facing (:206-219) ----
        je = pred_joints[:, -1]
        x_axis = (je[:, 2, :] - je[:, 1, :]).clone()
        x_axis[:, -1] = 0
        x_axis = x_axis / torch.norm(x_axis, dim=-1, keepdim=True).clip(min=1e-12)
        z_axis = torch.zeros_like(x_axis)
        z_axis[:, 2] = 1
        b_ori = torch.cross(z_axis, x_axis, dim=-1)[:, :2]
        tgt_l = torch.einsum("bij,btj->bti", R0.permute(0, 2, 1), self.wpath[:, -1:, :] - T0)[:, :, :3]  # [A,1,3]
        fto = tgt_l[:, 0, :2] - pelvis[:, -1, :2]
        fto = fto / torch.norm(fto, dim=-1, keepdim=True).clip(min=1e-12)
        r_face = (torch.einsum("bi,bi->b", fto, b_ori) + 1) / 2.0
        # ---- looking (:221-229) ----
        ex = (joints_all[:, -1, 24] - joints_all[:, -1, 23]).clone()
        ex[:, -1] = 0
        ex = ex / torch.norm(ex, dim=-1, keepdim=True).clip(min=1e-12)
        look = torch.cross(z_axis, ex, dim=-1)[:, :2]
        r_look = (torch.einsum("bi,bi->b", fto, look) + 1) / 2.0
        # ---- target distance (:231-235) ----
        dist2target = torch.norm(tgt_l - pelvis, dim=-1).clip(min=1e-12)[:, -1]
        r_target_dist = self.dist - dist2target
        self.dist = dist2target
        r_goal = (self.dist < cfg["goal_thresh"]).to(dt)

        # ---- re-canonicalise (:237-265) ----
        seed = pred_params[:, -T_HIS:]                                    # [A,2,93]
        # get_new_coordinate on frame 18 (SMPL-X B=A, joints only)
        _, j1 = self._smplx(seed[:, 0], self.betas)
        R_, T_ = get_new_coordinate(j1[:, :22])
        T0_new = torch.einsum("bij,btj->bti", R0, T_) + T0
        R0_new = torch.einsum("bij,bjk->bik", R0, R_)
        dT = self._delta_T(self.betas)
        seed_new = update_transl_glorot(R_.repeat_interleave(T_HIS, 0), T_.repeat_interleave(T_HIS, 0),
                                        dT.repeat_interleave(T_HIS, 0), seed.reshape(A * T_HIS, -1)).reshape(A, T_HIS, -1)
        marker_seed = torch.einsum("bij,btpj->btpi", R_.permute(0, 2, 1), marker_b[:, -T_HIS:] - T_[..., None, :])
        pel_seed = torch.einsum("bij,btj->bti", R_.permute(0, 2, 1), pelvis[:, -T_HIS:] - T_)
        self.R0, self.T0 = R0_new, T0_new
        _, _, _, fea_marker, _ = get_feature(marker_seed, pel_seed, self.R0, self.T0, self.wpath[:, -1:, :])
        new_state = torch.cat([marker_seed.reshape(A, T_HIS, -1), fea_marker], dim=-1)

        # ---- penetration (box env :279-295): marker bbox vs local walkability map ----
        if self.scene_kind in ("box", "crowd"):
            pts_l, local_map = self._walk_map(self.R0, self.T0) if self.scene_kind == "box" else self._crowd_walk_map(self.R0, self.T0)
            mxy = marker_seed[:, :, :, :2] if cfg["pene_type"] == "body" else marker_seed[:, :, self.feet_marker_idx, :2]
            bmin = mxy.amin(dim=(1, 2)).reshape(A, 1, 2)
            bmax = mxy.amax(dim=(1, 2)).reshape(A, 1, 2)
            inb = ((pts_l[:, :, :2] >= bmin).all(-1) & (pts_l[:, :, :2] <= bmax).all(-1)).to(dt)
            num_pene = (inb * (1 - local_map) * 0.5).sum(dim=1)
            penetration = num_pene > cfg["pene_thres"]
            r_pene = torch.where(penetration, torch.zeros_like(num_pene), torch.full_like(num_pene, 0.05))
            weight_pene = cfg["weight_pene"]
            self.last["num_pene"] = num_pene
            self.last["local_map"] = local_map
        else:
            weight_pene = 0.1 if self.finetuning else 1.0                 # :268-271

        reward = (r_skate * cfg["weight_skate"] + r_floor * cfg["weight_floor"] + r_face * cfg["weight_face_target"]
                  + r_look * cfg["weight_look_target"] + r_goal * cfg["weight_success"]
                  + r_target_dist * cfg["weight_target_dist"] + r_pene * weight_pene + r_vp * cfg["weight_vp"])

        self.body_param_seed = seed_new
        self.state = new_state
        # ---- egosensing on the new seed's joints (:290-296) ----
        _, j2 = self._smplx(seed_new.reshape(A * T_HIS, -1), self.betas.repeat_interleave(T_HIS, 0))
        jw = torch.einsum("bij,btpj->btpi", self.R0, j2.reshape(A, T_HIS, -1, 3)) + self.T0[:, None, :, :]
        ego = torch.stack([calc_egosensing(jw[a], self._edges_for(a)) for a in range(A)]).to(dt)

        at_depth = self.steps == cfg["max_depth"]
        if self.scene_kind == "crowd" and getattr(self, "ego_edges", None) is not None:
            terminated = at_depth                                          # crowd_env_egobody_eval.py:378
            invalid = (vp_norm > vp_thresh).long() * 2
            pel_w = (torch.einsum("bij,btj->bti", R0, joints_all[:, :, 0]) + T0).double().numpy()
            for a in range(A):
                if int(self.steps[a]) < 6 and not points_in_rings(self.ego_edges, pel_w[a, :, 0], pel_w[a, :, 1]).all():
                    invalid[a] |= 1
            self.last["invalid"] = invalid
        elif self.scene_kind == "crowd":
            terminated = (r_goal > 0) | at_depth                           # crowd_env_crowd_eval.py:367
        elif self.scene_kind == "box" or self.finetuning:
            terminated = (r_goal > 0) | penetration | at_depth
        else:
            terminated = (r_goal > 0) | at_depth
        obs = {"state": self.state, "egosensing": ego, "dist": (1 / (dist2target + 1)).reshape(A, 1),
               "time": (1 - self.steps.to(dt) / cfg["max_depth"]).reshape(A, 1)}
        self.last.update({
            "Y_gen": Y_gen, "Yb_gen": Yb_gen, "pred_params": pred_params, "joints": joints_all, "marker_b": marker_b,
            "markers_proj": markers_proj, "pelvis": pelvis, "r_skate": r_skate, "r_floor": r_floor, "r_vp": r_vp,
            "vp_norm": vp_norm, "r_face": r_face, "r_look": r_look, "r_target_dist": r_target_dist, "r_goal": r_goal,
            "r_pene": r_pene, "penetration": penetration, "R_": R_, "T_": T_, "joints_seed_w": jw})
        return obs, reward, terminated

    # ---- reset -----------------------------------------------------------------------------
    def canonicalize_2frame(self, transl, glorot, body_pose, betas):
        """crowd_env_2f.py:615-644.  transl/glorot [A,2,3], body_pose [A,2,63] (world) ->
        body_param_seed[A,2,93] (hands zero), R0, T0."""
        A = transl.shape[0]
        prev = torch.cat([transl, glorot, body_pose, torch.zeros(A, 2, 24, dtype=self.dt)], dim=-1)
        _, j = self._smplx(prev[:, 0], betas)
        R0, T0 = get_new_coordinate(j[:, :22])
        dT = self._delta_T(betas)
        seed = update_transl_glorot(R0.repeat_interleave(2, 0), T0.repeat_interleave(2, 0), dT.repeat_interleave(2, 0),
                                    prev.reshape(A * 2, -1)).reshape(A, 2, -1)
        return seed, R0, T0

    def next_body(self, start, target, seed_poses, seed_trans, betas, yaw_jitter=None):
        """exp_GAMMAPrimitive/utils/environments.py:65-335 (room0; yaw_jitter=None) and :371-627
        (box; yaw_jitter[A] = the final +-0.1*2pi rotation about z) for given start/target pairs.
        start/target [A,3]; seed_poses [A,2,66] (global_orient | body_pose), seed_trans [A,2,3].
        Returns world-frame motion seed (transl, glorot, body_pose) and wpath[A,2,3]."""
        dt = self.dt
        A = start.shape[0]
        z93 = lambda tr, go, bp: torch.cat([tr, go, bp, torch.zeros(tr.shape[0], 24, dtype=dt)], -1)
        glorot = seed_poses[:, :, :3].to(dt).clone()
        body_pose = seed_poses[:, :, 3:66].to(dt).clone()
        transl = seed_trans.to(dt).clone()
        betas2 = betas.repeat_interleave(2, 0)
        pelvis_zero = self._delta_T(betas)                                 # bm(betas).joints[0,0]

        def joints_of(tr, go):
            _, j = self._smplx(z93(tr.reshape(A * 2, 3), go.reshape(A * 2, 3), body_pose.reshape(A * 2, 63)), betas2)
            return j.reshape(A, 2, -1, 3)

        def apply_rot(Rm, tr, go):
            orig = rot.p3d_axis_angle_to_matrix(go.reshape(A * 2, 3)).reshape(A, 2, 3, 3)
            new_rot = torch.einsum("bij,btjk->btik", Rm, orig)
            new_tr = torch.einsum("bij,btj->bti", Rm, pelvis_zero[:, None, :] + tr) - pelvis_zero[:, None, :]
            return new_tr, rot.p3d_matrix_to_axis_angle(new_rot.reshape(A * 2, 3, 3)).reshape(A, 2, 3)

        # rotate the body to face the target (:216-237)
        j = joints_of(transl, glorot)
        x_axis = (j[:, :, 2] - j[:, :, 1]).clone()
        x_axis[..., -1] = 0
        x_axis = x_axis / torch.norm(x_axis, dim=-1, keepdim=True).clip(min=1e-12)
        z_axis = torch.zeros_like(x_axis)
        z_axis[..., 2] = 1
        b_ori = torch.cross(z_axis, x_axis, dim=-1)[:, 0]
        b_ori = b_ori / torch.linalg.norm(b_ori, dim=-1, keepdim=True)
        t_ori = (target - start).to(dt)
        t_ori = t_ori / torch.linalg.norm(t_ori, dim=-1, keepdim=True)
        v = torch.cross(b_ori, t_ori, dim=-1)
        c = (b_ori * t_ori).sum(-1)
        s = torch.linalg.norm(v, dim=-1)
        K = torch.zeros(A, 3, 3, dtype=dt)
        K[:, 0, 1], K[:, 0, 2], K[:, 1, 0], K[:, 1, 2], K[:, 2, 0], K[:, 2, 1] = -v[:, 2], v[:, 1], v[:, 2], -v[:, 0], -v[:, 1], v[:, 0]
        Rt = torch.eye(3, dtype=dt)[None] + K + (K @ K) * ((1 - c) / (s ** 2))[:, None, None]
        transl, glorot = apply_rot(Rt, transl, glorot)
        if yaw_jitter is not None:                                        # box sampler (:528-538)
            transl, glorot = apply_rot(rot.rotz(yaw_jitter.to(dt)), transl, glorot)
        # pelvis above start, lowest joint on the floor (:240-247)
        j = joints_of(transl, glorot)
        shift = torch.stack([j[:, 0, 0, 0], j[:, 0, 0, 1], j[:, 0, :, 2].amin(dim=-1)], dim=-1)
        transl = transl - shift[:, None, :] + start.to(dt)[:, None, :]
        j = joints_of(transl, glorot)
        wpath = torch.stack([j[:, 0, 0], target.to(dt)], dim=1).clone()
        wpath[:, 1, 2] = wpath[:, 0, 2]
        return transl, glorot, body_pose, wpath

    def reset_from(self, transl, glorot, body_pose, betas, wpath, scene_idx=None):
        """crowd_env_2f.py:320-415 body of the rejection loop for given candidates.
        Returns obs, accept bool[A] (start free of penetration)."""
        dt = self.dt
        A = transl.shape[0]
        cfg = self.cfg
        self.scene_idx = None if scene_idx is None else torch.as_tensor(scene_idx).long()
        seed, R0, T0 = self.canonicalize_2frame(transl, glorot, body_pose, betas)
        verts, joints = self._smplx(seed.reshape(A * 2, -1), betas.repeat_interleave(2, 0))
        marker_seed = verts[:, self.marker_ids].reshape(A, 2, -1)
        joints_all = joints.reshape(A, 2, -1, 3)
        pelvis = joints_all[:, :, 0]
        wp = wpath.to(dt)
        _, dist, _, fea_marker, _ = get_feature(marker_seed, pelvis, R0, T0, wp[:, -1:, :])
        if self.scene_kind == "sdf":
            vw = torch.einsum("bij,btpj->btpi", R0, verts.reshape(A, 2, -1, 3)) + T0[:, None, :, :]
            sv = calc_sdf(vw.reshape(A * 2, -1, 3), self.sdf_dict).reshape(A, 2, -1)
            sv[:, :, self.feet_vids] = 0.0
            accept = sv.lt(0.0).sum(dim=(1, 2)) == 0
        elif self.scene_kind == "crowd":
            accept = torch.ones(A, dtype=torch.bool)
        else:
            self.scene_idx = torch.as_tensor(scene_idx).long()
            pts_l, local_map = self._walk_map(R0, T0)
            mxy = marker_seed.reshape(A, 2, -1, 3)[:, :, :, :2]
            bmin = mxy.amin(dim=(1, 2)).reshape(A, 1, 2)
            bmax = mxy.amax(dim=(1, 2)).reshape(A, 1, 2)
            inb = ((pts_l[:, :, :2] >= bmin).all(-1) & (pts_l[:, :, :2] <= bmax).all(-1)).to(dt)
            accept = (inb * (1 - local_map) * 0.5).sum(dim=1) == 0
        jw = torch.einsum("bij,btpj->btpi", R0, joints_all) + T0[:, None, :, :]
        self.set_state(torch.cat([marker_seed, fea_marker], dim=-1), seed, R0, T0, betas, dist[:, 0, 0],
                       torch.zeros(A, dtype=torch.long), wp, self.scene_idx)
        ego = torch.stack([calc_egosensing(jw[a], self._edges_for(a)) for a in range(A)]).to(dt)
        obs = {"state": self.state, "egosensing": ego, "dist": (1 / (dist + 1))[:, 0, :].reshape(A, 1),
               "time": torch.ones(A, 1, dtype=dt)}
        self.last["joints_seed_w"] = jw
        return obs, accept
