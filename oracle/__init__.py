"""CPU oracle for the crowd_ppo hot path.  TEST INFRASTRUCTURE ONLY.

Nothing under `egogen_amd/` (the product) may import this package; only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg do, and only as the checker /
the timed CPU baseline.  Each function cites the reference file:line (relative to the
reference repo root) or the third-party package whose published algorithm it restates.

Pinning status (see DESIGN.md "Oracle"):
  pinned by reference-generated goldens (tests/golden/*.npz, scripts/gen_goldens.py):
      sdf.calc_sdf, nets.cvae_decode, nets.regressor_6d, nets.cont2rotmat, nets.policy_*,
      rot/canonical frame (get_new_coordinate_torch),
      rot.tgm_angle_axis_to_rotation_matrix (against the in-tree kornia-derived copy, experiments/HMR/.../konia_transform.py:234-310)
  parity unpinned (third-party algorithm absent from /root/reference, restated from the
  published source and checked by invariants / scipy only):
      smplx_lbs (smplx 0.1.28), tgm rotation-matrix -> quaternion -> angle-axis (torchgeometry 0.1.2), p3d_* (pytorch3d 0.7.4),
      vposer_encode (human_body_prior 1.0), gae (tianshou 0.5), ray casting (shapely 2.0)
"""
