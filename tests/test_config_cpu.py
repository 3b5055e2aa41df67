"""load_model's configuration half (crowd_ppo/primitive_model.py:41-96): yaml -> results/ tree + config.yaml, and the mapping of
its fields onto the environment / policy configuration."""
import os

import pytest
import yaml


def test_load_model_reads_yaml_and_writes_the_reference_tree(tmp_path, monkeypatch):
    from egogen_amd import setup_world as sw
    from egogen_amd.crowd_env import BOX_CFG, DEFAULT_CFG
    monkeypatch.chdir(tmp_path)
    cfg = sw.load_model()
    exp = os.path.join("results", "crowd_ppo", "MPVAEPolicy_samp_collision", "collision_test")
    assert cfg["cfg_exp_dir"] == exp
    for d in ("results", "checkpoints", "logs"):
        assert os.path.isdir(os.path.join(exp, d))
    with open(os.path.join(exp, "config.yaml")) as f:
        saved = yaml.safe_load(f)
    assert saved["cfg_name"] == "MPVAEPolicy_samp_collision"
    assert saved["trainconfig"]["save_dir"] == os.path.join(exp, "checkpoints")      # primitive_model.py:52-53
    assert saved["trainconfig"]["log_dir"] == os.path.join(exp, "logs")
    assert saved["lossconfig"]["weight_look_target"] == 0.3 and saved["trainconfig"]["max_depth"] == 13
    # the packaged yaml reproduces the constants the env used to hard-code
    env_cfg = sw.env_cfg_from_yaml(cfg)
    assert env_cfg == DEFAULT_CFG
    box = sw.load_model(box=True)
    assert box["cfg_name"] == "MPVAEPolicy_samp_collision_2"
    assert sw.env_cfg_from_yaml(box) == BOX_CFG
    assert os.path.isfile(os.path.join("results", "crowd_ppo", "MPVAEPolicy_samp_collision_2", "collision_test", "config.yaml"))
    pdir, rdir = sw.prior_checkpoint_dirs(cfg, "male")
    assert pdir == os.path.join("results", "crowd_ppo", "MPVAE_samp20_2frame_rollout", "checkpoints")
    assert rdir == os.path.join("results", "crowd_ppo", "MoshRegressor_v3_male", "checkpoints")
    assert sw.policy_cfg_from_yaml(cfg)["min_logvar"] == -2.5


def test_working_directory_yaml_takes_precedence(tmp_path, monkeypatch):
    """A crowd_ppo/cfg_samp20/<name>.yaml under the working directory (the reference's location) overrides the packaged one."""
    from egogen_amd import setup_world as sw
    monkeypatch.chdir(tmp_path)
    src = os.path.join(sw._PKG_CFG_DIR, "MPVAEPolicy_samp_collision.yaml")
    with open(src) as f:
        cfg = yaml.safe_load(f)
    cfg["lossconfig"]["weight_skate"] = 0.7
    cfg["trainconfig"]["max_depth"] = 5
    cfg["wandb"]["name"] = "my_run"
    os.makedirs(os.path.join("crowd_ppo", "cfg_samp20"))
    with open(os.path.join("crowd_ppo", "cfg_samp20", "MPVAEPolicy_samp_collision.yaml"), "w") as f:
        yaml.safe_dump(cfg, f)
    got = sw.load_model()
    e = sw.env_cfg_from_yaml(got)
    assert e["weight_skate"] == 0.7 and e["max_depth"] == 5
    assert os.path.isfile(os.path.join("results", "crowd_ppo", "MPVAEPolicy_samp_collision", "my_run", "config.yaml"))


def test_missing_config_raises(tmp_path, monkeypatch):
    from egogen_amd import setup_world as sw
    monkeypatch.chdir(tmp_path)
    with pytest.raises(FileNotFoundError):
        sw.load_model(cfg_dir=str(tmp_path / "nowhere"))
