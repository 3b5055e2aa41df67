"""`rollout_primitives` - the consumer of log/eval_results/motion_*.pkl that turns the canonical-frame motion primitives
into one continuous world-frame SMPL-X parameter sequence (motion/vis.py:44-78; same function in
experiments/gen_egobody_depth.py:27-61, gen_egobody_rgb.py:129).  CPU restatement, TEST INFRASTRUCTURE ONLY.

The reference evaluates smplx (absent here) only for `pelvis_original` = the pelvis of the rest pose for the primitive's
betas; `pelvis_of(betas)` supplies it (oracle.smplx_lbs in the tests).  Rotations go through scipy like the reference."""
import numpy as np
from scipy.spatial.transform import Rotation


def rollout_primitives(motion_primitives, pelvis_of):
    out = []
    for idx, mp in enumerate(motion_primitives):
        pelvis = np.asarray(pelvis_of(mp["betas"]), np.float64).reshape(1, 3)            # vis.py:55 (same for all 20 frames)
        p = np.array(mp["smplx_params"][0], np.float64)                                   # [20,93]  (:56)
        R = np.asarray(mp["transf_rotmat"], np.float64).reshape(3, 3)
        T = np.asarray(mp["transf_transl"], np.float64).reshape(1, 3)
        p[:, :3] = np.matmul(p[:, :3] + pelvis, R.T) - pelvis + T                          # :60
        r_new = Rotation.from_matrix(np.tile(R, [p.shape[0], 1, 1])) * Rotation.from_rotvec(p[:, 3:6])   # :61-62
        p[:, 3:6] = r_new.as_rotvec()
        if idx == 0:
            start = 0
        elif mp["mp_type"] == "1-frame":
            start = 1
        elif mp["mp_type"] == "2-frame":
            start = 2
        else:
            start = 1
        out.append(p[start:])
    return np.concatenate(out, axis=0)
