#!/usr/bin/env python3
"""Drop-in entry point for the reference's `python exp_GAMMAPrimitive/train_GAMMAPredictor.py --cfg <name>`
(motion/exp_GAMMAPrimitive/train_GAMMAPredictor.py:17-59): trains the marker predictor (C-VAE) whose `epoch-N.ckp` the
crowd_ppo drivers load as the motion prior.  Config: `exp_GAMMAPrimitive/cfg/<name>.yml` under the working directory (the
reference's location) with modelconfig / lossconfig / trainconfig; results under `results/exp_GAMMAPrimitive/<name>/`
(ConfigCreator, exp_GAMMAPrimitive/utils/config_creator.py)."""
import argparse
import os
import sys

import numpy as np
import torch
import yaml

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from egogen_amd.train_predictor import BatchGeneratorAMASSCanonicalized, GAMMAPrimitiveVAETrainOP  # noqa: E402


def load_cfg(name, exp="exp_GAMMAPrimitive"):
    f = os.path.join(".", exp, "cfg", f"{name}.yml")
    if not os.path.exists(f):
        raise FileNotFoundError(f)
    cfg = yaml.safe_load(open(f))
    exp_dir = os.path.join("results", exp, name)
    for sub in ("results", "checkpoints", "logs"):
        os.makedirs(os.path.join(exp_dir, sub), exist_ok=True)
    cfg["trainconfig"]["save_dir"] = os.path.join(exp_dir, "checkpoints")
    cfg["trainconfig"]["log_dir"] = os.path.join(exp_dir, "logs")
    return cfg


if __name__ == "__main__":
    parser = argparse.ArgumentParser()
    parser.add_argument("--cfg", default=None)
    parser.add_argument("--resume_training", type=int, default=0)
    parser.add_argument("--verbose", type=int, default=1)
    parser.add_argument("--seed", type=int, default=0)
    parser.add_argument("--gpu_index", type=int, default=0)
    args = parser.parse_args()
    np.random.seed(args.seed)
    torch.manual_seed(args.seed)
    cfg = load_cfg(args.cfg)
    modelcfg, losscfg, traincfg = cfg["modelconfig"], cfg["lossconfig"], cfg["trainconfig"]
    traincfg["resume_training"] = args.resume_training == 1
    traincfg["verbose"] = args.verbose == 1
    traincfg["gpu_index"] = args.gpu_index
    batch_gen = BatchGeneratorAMASSCanonicalized(amass_data_path=traincfg["dataset_path"], amass_subset_name=traincfg["subsets"],
                                                 sample_rate=1, body_repr=modelcfg["body_repr"])
    batch_gen.get_rec_list(to_gpu=True)
    trainop = GAMMAPrimitiveVAETrainOP(modelcfg, losscfg, traincfg)
    trainop.train(batch_gen)
