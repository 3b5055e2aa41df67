#!/usr/bin/env python3
"""Entry point for training the body regressor (`GAMMARegressorTrainOP`, motion/models/models_GAMMA_primitive.py:594-711)
whose `epoch-100.ckp` the crowd_ppo drivers load next to the marker predictor (`GAMMAPrimitiveComboGenOP.build_model`, :1141).
The reference tree ships the operator and its configs (`crowd_ppo/cfg_samp20/MoshRegressor_v3_{male,female}.yml`) but no
script for it; this one follows `train_GAMMAPredictor.py`: `--cfg <name>` is looked up as `exp_GAMMAPrimitive/cfg/<name>.yml`,
then `crowd_ppo/cfg_samp20/<name>.yml`, under the working directory; results go to `results/exp_GAMMAPrimitive/<name>/`."""
import argparse
import os
import sys

import numpy as np
import torch
import yaml

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from egogen_amd.train_regressor import BatchGeneratorAMASSCanonicalized, GAMMARegressorTrainOP  # noqa: E402


def load_cfg(name, exp="exp_GAMMAPrimitive"):
    for f in (os.path.join(".", exp, "cfg", f"{name}.yml"), os.path.join(".", "crowd_ppo", "cfg_samp20", f"{name}.yml")):
        if os.path.exists(f):
            break
    else:
        raise FileNotFoundError(f"{name}.yml under ./{exp}/cfg or ./crowd_ppo/cfg_samp20")
    cfg = yaml.safe_load(open(f))
    exp_dir = os.path.join("results", exp, name)
    for sub in ("results", "checkpoints", "logs"):
        os.makedirs(os.path.join(exp_dir, sub), exist_ok=True)
    cfg["trainconfig"]["save_dir"] = os.path.join(exp_dir, "checkpoints")
    cfg["trainconfig"]["log_dir"] = os.path.join(exp_dir, "logs")
    return cfg


if __name__ == "__main__":
    parser = argparse.ArgumentParser()
    parser.add_argument("--cfg", default="MoshRegressor_v3_male")
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
    batch_gen = BatchGeneratorAMASSCanonicalized(amass_data_path=traincfg["dataset_path"], amass_subset_name=traincfg.get("subsets"),
                                                 sample_rate=int(traincfg.get("sample_rate", 3)), body_repr=modelcfg["body_repr"],
                                                 read_to_ram=False)
    batch_gen.get_rec_list(shuffle_seed=args.seed)
    trainop = GAMMARegressorTrainOP(modelcfg, losscfg, traincfg)
    trainop.train(batch_gen)
