#!/usr/bin/env python3
"""Drop-in entry point for the reference's `python crowd_ppo/main_ppo.py` (motion/crowd_ppo/main_ppo.py): same
command-line flags, same checkpoint layout (<logdir>/<task>/ppo/<seed>/<%y%m%d-%H%M%S>/checkpoint_{epoch}.pth =
{"model": policy.state_dict(), "optim": optim.state_dict()}, best -> policy.pth), same `--watch` behaviour and
./log/eval_results/motion_*.pkl output, running on the MI355X-native stack (egogen_amd).

Multi-GPU: launch with `python -m torch.distributed.run --nproc-per-node N crowd_ppo/main_ppo.py ...`; every rank owns
training_num / N agents and PPO gradients are all-reduced over RCCL.
"""
import argparse
import datetime
import os
import pprint
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from egogen_amd import setup_world as sw  # noqa: E402
from egogen_amd import synth  # noqa: E402
from egogen_amd.body_model import BodyModelHandle  # noqa: E402
from egogen_amd.trainer import CheckpointWriter, Collector, ScalarLogger, onpolicy_trainer  # noqa: E402

SCENE_DEFAULT = "room0"
CFG_NAME = "MPVAEPolicy_samp_collision"


def get_args(argv=None, extra=(), defaults=None):
    """`extra`: [(flag, add_argument kwargs)] of a sibling driver (main_egobody_eval.py, main_ppo_box.py); `defaults`: the
    sibling's own default values for flags of this table."""
    p = argparse.ArgumentParser()
    p.add_argument("--task", type=str, default="collision-avoidance")
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--buffer-size", type=int, default=4096)
    p.add_argument("--lr", type=float, default=3e-4)
    p.add_argument("--gamma", type=float, default=0.99)
    p.add_argument("--epoch", type=int, default=3000)
    p.add_argument("--step-per-epoch", type=int, default=20000)
    p.add_argument("--step-per-collect", type=int, default=1024)
    p.add_argument("--repeat-per-collect", type=int, default=1)
    p.add_argument("--batch-size", type=int, default=256)
    p.add_argument("--training-num", type=int, default=256)
    p.add_argument("--test-num", type=int, default=20)
    p.add_argument("--rew-norm", type=int, default=False)
    p.add_argument("--vf-coef", type=float, default=1.0)
    p.add_argument("--ent-coef", type=float, default=0.01)
    p.add_argument("--weight-kld", type=float, default=0)
    p.add_argument("--gae-lambda", type=float, default=0.95)
    p.add_argument("--bound-action-method", type=str, default="clip")
    p.add_argument("--max-grad-norm", type=float, default=0.1)
    p.add_argument("--eps-clip", type=float, default=0.1)
    p.add_argument("--dual-clip", type=float, default=None)
    p.add_argument("--value-clip", type=int, default=0)
    p.add_argument("--norm-adv", type=int, default=1)
    p.add_argument("--recompute-adv", type=int, default=0)
    p.add_argument("--logdir", type=str, default="./log")
    p.add_argument("--render", type=float, default=0.0)
    p.add_argument("--device", type=str, default="cuda")
    p.add_argument("--resume-path", type=str, default=None)
    p.add_argument("--resume-buffer", type=str, default=None)
    p.add_argument("--resume-id", type=str, default=None)
    p.add_argument("--finetune", default=False, action="store_true")
    p.add_argument("--deterministic-eval", default=False, action="store_true")
    p.add_argument("--logger", type=str, default="tensorboard", choices=["tensorboard", "wandb"])
    p.add_argument("--save-interval", type=int, default=2)
    p.add_argument("--wandb-project", type=str, default="mujoco.benchmark")
    p.add_argument("--watch", default=False, action="store_true", help="watch the play of pre-trained policy only")
    # extensions (not in the reference)
    p.add_argument("--scene", type=str, default=None,
                   help="scene set: room0 | single_box | box (default: room0 for main_ppo, box for main_ppo_box), or the .npz of a "
                        "scene prepared with egogen_amd.scene_gen.save_scene")
    p.add_argument("--sdf-res", type=int, default=256)
    p.add_argument("--save-rollout", type=int, default=None, help="write log/eval_results/motion_*.pkl (default: only with --watch)")
    p.add_argument("--num-verts", type=int, default=synth.NUM_VERTS, help="reduced synthetic body (tests)")
    p.add_argument("--policy-dtype", type=str, default=None, choices=["fp32", "bf16x2", "bf16"],
                   help="arithmetic of the policy's dense layers, rollout forward and PPO update alike: fp32 = each operand as three "
                        "bf16 terms (2^-24, fp32-equivalent), bf16x2 = two terms (16 significant bits; every gradient within 1e-4 of "
                        "float64, tests/test_trainer_gpu.py), bf16 = operands rounded to bf16; fp32 accumulation in all.  Default: "
                        "bf16x2 for training (main_ppo*.py), bf16 for main_crowd_eval.py (BASELINE config 5)")
    p.add_argument("--num-scenes", type=int, default=None, help="main_crowd_eval.py: independent 4-human scenes per GPU")
    for flag, kw in extra:
        p.add_argument(flag, **kw)
    if defaults:
        p.set_defaults(**defaults)
    return p.parse_args(argv)


_PREC_CODE = {"fp32": 0, "bf16x2": 2, "bf16": 1}


def _apply_policy_dtype(args, default="bf16x2"):
    """Sets the rollout-forward arithmetic process-wide and returns the name of the update's (`update_precision`)."""
    from egogen_amd import _lib
    name = getattr(args, "policy_dtype", None) or default
    _lib.check(_lib.load().egx_policy_set_precision(_PREC_CODE[name]), "egx_policy_set_precision")
    return {"fp32": "f32"}.get(name, name)


def main(args, scene_kind=SCENE_DEFAULT, cfg_name=CFG_NAME, ckpt_with_optim=True):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("crowd_ppo needs a HIP device: the MI355X path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    args.update_precision = _apply_policy_dtype(args)
    if world > 1 and not dist.is_initialized():
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    # load_model (crowd_ppo/primitive_model.py:74-96): yaml -> results/crowd_ppo/<cfg_name>/<run>/ tree + config.yaml
    cfg = sw.load_model(box=cfg_name.endswith("_2"))
    env_cfg = sw.env_cfg_from_yaml(cfg)
    scene_kind = args.scene or scene_kind

    # seed (main_ppo.py:100-105)
    np.random.seed(args.seed)
    torch.manual_seed(args.seed)

    bm, real = sw.load_body_model("male", seed=args.seed, num_verts=args.num_verts)
    body = BodyModelHandle(bm, synth.marker_ids(args.num_verts), synth.feet_vids(args.num_verts))
    prior = sw.build_motion_prior(seed=args.seed, ckpt_dirs=sw.prior_checkpoint_dirs(cfg, "male"))
    vposer = sw.build_vposer(seed=args.seed)
    scene = sw.build_scene(scene_kind, sdf_res=args.sdf_res, seed=args.seed)
    save_rollout = args.watch if args.save_rollout is None else bool(args.save_rollout)

    policy = sw.build_policy(args, policy_cfg=sw.policy_cfg_from_yaml(cfg))
    if args.resume_path:
        ckpt = torch.load(args.resume_path, map_location="cuda")
        policy.load_state_dict(ckpt["model"])
        print("Loaded agent from: ", args.resume_path)

    n_train = max(1, args.training_num // world)
    test_env = sw.build_env(args.test_num, scene, body, prior, vposer, finetuning=args.finetune, seed=args.seed + 1000 + rank,
                            keep_rollout=save_rollout, cfg=env_cfg)
    test_collector = Collector(policy, test_env, rollout_dir="./log/eval_results/" if save_rollout else None)

    now = datetime.datetime.now().strftime("%y%m%d-%H%M%S")
    log_name = os.path.join(args.task, "ppo", str(args.seed), now)
    log_path = os.path.join(args.logdir, log_name)
    logger = ScalarLogger(log_path) if rank == 0 else None

    writer = CheckpointWriter()      # file writes on a worker thread (egogen_amd/trainer.py); flushed before main() returns

    def save_best_fn(pol):
        state = {"model": pol.state_dict(), "optim": pol.optim_state_dict()} if ckpt_with_optim else {"model": pol.state_dict()}
        writer.save(state, os.path.join(log_path, "policy.pth"))

    def save_checkpoint_fn(epoch, env_step, gradient_step):
        ckpt_path = os.path.join(log_path, f"checkpoint_{epoch}.pth")
        state = {"model": policy.state_dict(), "optim": policy.optim_state_dict()} if ckpt_with_optim else {"model": policy.state_dict()}
        writer.save(state, ckpt_path)
        return ckpt_path

    if not args.watch:
        train_env = sw.build_env(n_train, scene, body, prior, vposer, finetuning=args.finetune, seed=args.seed + rank, cfg=env_cfg)
        train_collector = Collector(policy, train_env)
        result = onpolicy_trainer(policy, train_collector, test_collector, args.epoch, args.step_per_epoch,
                                  args.repeat_per_collect, args.test_num, args.batch_size,
                                  step_per_collect=args.step_per_collect, save_best_fn=save_best_fn, logger=logger,
                                  save_checkpoint_fn=save_checkpoint_fn, save_interval=args.save_interval)
        writer.flush()               # every checkpoint of the run is on disk from here on
        if rank == 0:
            pprint.pprint(result)

    # Let's watch its performance!  (main_ppo.py:238-243)
    policy.eval()
    test_env.gen.manual_seed(args.seed)
    result = test_collector.collect_episodes(args.test_num)
    if rank == 0:
        print(f'Final reward: {result["rew"]}, length: {result["len"]}')
    writer.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return result


if __name__ == "__main__":
    main(get_args())
