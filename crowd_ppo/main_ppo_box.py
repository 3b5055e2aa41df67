#!/usr/bin/env python3
"""Drop-in for the reference's crowd_ppo/main_ppo_box.py: the same driver on the random-box scene set with the
walkability-map penetration term (crowd_env_2f_box.py), policy config MPVAEPolicy_samp_collision_2 and checkpoints
that hold "model" only (main_ppo_box.py:218-224).

Its command line is main_ppo.py's with the box driver's OWN defaults (main_ppo_box.py:52 `--test-num 10`, :67 `--logdir
./log/log_box`, :81 `--save-interval 1`) and its two extra switches (`--dynobs` :89-94, `--more-ego` :95): both are parsed and
never read by the reference's box driver, and are accepted and unused here as well."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from crowd_ppo.main_ppo import get_args as _base_args, main  # noqa: E402

BOX_DEFAULTS = {"test_num": 10, "logdir": "./log/log_box", "save_interval": 1}
BOX_FLAGS = (("--dynobs", dict(default=False, action="store_true", help="evaluation on dynamic obstacle (parsed, unused: main_ppo_box.py:89-94)")),
             ("--more-ego", dict(default=False, action="store_true", help="more egosensing dim (parsed, unused: main_ppo_box.py:95)")))


def get_args(argv=None):
    return _base_args(argv, extra=BOX_FLAGS, defaults=BOX_DEFAULTS)


if __name__ == "__main__":
    main(get_args(), scene_kind="box", cfg_name="MPVAEPolicy_samp_collision_2", ckpt_with_optim=False)
