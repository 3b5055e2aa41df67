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


def test_bench_stdout_line_fits_the_drivers_tail():
    """Round 4's bench line grew to 21 KB and the driver's stdout tail cut its head off (BENCH_r04.json parsed = null).  The
    stdout line is now built by bench.compact_line from the full record: held below 4 KB here on round 4's own full record
    (profiles/r04_final_bench.json, the 21 KB case) and on a multi-rank stub, with every contract field present."""
    import json
    import os
    import bench
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    full = json.load(open(os.path.join(root, "profiles", "r04_final_bench.json")))
    assert len(json.dumps(full)) > 8192          # the record that did not parse
    full["value_configs2"], full["value_fp32_equivalent"], full["value_reference_shape"] = 252654.1, 128053.2, 139300.3
    full["detail_file"] = "bench_detail.json"
    for world in (1, 8):
        if world > 1:
            full["n_gpus"] = world
            full["allreduce"] = {"bytes": 52672004, "buckets": 2, "calls_per_step": 8, "in_loop_avg_ms": 0.5, "in_loop_ms_per_step": 4.0,
                                 "standalone_ms": 0.4, "standalone_algbw_GBps": 130.0, "standalone_busbw_GBps": 228.0, "note": "x" * 300}
            full["weak"] = {"value": 1.2e6, "unit": "env-steps/s", "ms_per_step": 13.0, "agents_per_gpu": 512, "minibatch_per_gpu": 256,
                            "allreduce": dict(full["allreduce"]), "update_paths": {"chain+graph": 48}}
        line = bench.compact_line(full)
        text = json.dumps(line)
        assert len(text) < bench.LINE_BUDGET <= 4096, len(text)
        back = json.loads(text)
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                  "dtype", "data", "config", "roofline", "cpu_baseline", "value_configs2", "value_fp32_equivalent", "value_reference_shape"):
            assert k in back, k
        assert back["config"]["workload"] and "model" not in back["config"]
        for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms", "bodies_per_launch", "flop_per_body"):
            assert k in back["roofline"], k
        assert abs(back["roofline"]["frac"] - back["roofline"]["achieved"] / back["roofline"]["peak"]) < 1e-3
        for k in ("value", "unit", "cores", "kind", "sample"):
            assert k in back["cpu_baseline"], k
        assert back["value"] == round(full["value"], 1)
        if world > 1:
            assert back["allreduce"]["in_loop_ms_per_step"] == 4.0 and back["weak"]["value"] == 1.2e6


def test_main_ppo_box_has_its_own_command_line():
    """main_ppo_box.py:40-104: the box driver's defaults differ from main_ppo.py's in three places and it parses two more flags."""
    from crowd_ppo import main_ppo, main_ppo_box
    a, b = main_ppo.get_args([]), main_ppo_box.get_args([])
    assert (a.test_num, a.logdir, a.save_interval) == (20, "./log", 2)             # main_ppo.py:52,67,81
    assert (b.test_num, b.logdir, b.save_interval) == (10, "./log/log_box", 1)     # main_ppo_box.py:52,67,81
    assert b.dynobs is False and b.more_ego is False
    c = main_ppo_box.get_args(["--dynobs", "--more-ego", "--test-num", "3", "--deterministic-eval"])
    assert c.dynobs and c.more_ego and c.test_num == 3 and c.deterministic_eval
    for k in ("training_num", "batch_size", "step_per_collect", "lr", "eps_clip", "max_grad_norm", "epoch", "step_per_epoch"):
        assert getattr(a, k) == getattr(b, k)
