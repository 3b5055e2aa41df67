#!/usr/bin/env python3
"""Drop-in for the reference's crowd_ppo/main_ppo_box.py: the same driver on the random-box scene set with the
walkability-map penetration term (crowd_env_2f_box.py), policy config MPVAEPolicy_samp_collision_2 and checkpoints
that hold "model" only (main_ppo_box.py:218-224)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from crowd_ppo.main_ppo import get_args, main  # noqa: E402

if __name__ == "__main__":
    main(get_args(), scene_kind="box", cfg_name="MPVAEPolicy_samp_collision_2", ckpt_with_optim=False)
