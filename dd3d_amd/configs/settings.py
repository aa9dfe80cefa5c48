"""Module-level settings read on the hot path -- the subset of the reference's ``configs/settings.py`` globals that
``mnist_distill.py`` / ``distillation_loss.py`` / ``utils/Mnist2D.py`` touch (configs/settings.py:116-199, configs/default.ini).
Tensorboard logging is outside the hot path and reduces to ``log_string``; ``.ply`` / checkpoint export
lives in ``dd3d_amd/export.py`` (``save_cad`` is re-exported here under the reference's name, configs/settings.py:204)."""
from __future__ import annotations

import argparse
import configparser
import os

import torch

Pyt = False              # --pytorch (configs/settings.py:177); read by distance_wb / distance_mse
DEBUG = False            # --log / --no-log
num_points = 250         # [SHAPENET] num_points (default.ini:10)
cad_per_class = 1        # [DISTILLATION] cad_per_class (default.ini:27)
device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
num_workers = 0
experiment_name = "mnist"            # --exp or [DEFAULT] experiment_name (configs/settings.py:178)
exp_file_name = "mnist"              # <date>_<experiment_name> (:179)
logging_folder_name = "logs/mnist"   # [DEFAULT] logging_parent / exp_file_name (:180)

_DEFAULTS = {
    "MODEL": {"batch_size": "256", "dist_opt": "modmse", "dist_lr": "0.1", "classifier_lr": "0.01"},
    "DEFAULT": {"logging_parent": "logs", "experiment_name": "mnist"},
    "DISTILLATION": {"cad_per_class": "1", "total_iterations": "2500", "save_cad_every": "100", "export_folder_name": "output/",
                     "distillation_checkpoint_dir": "distillation_ckpt"},
    "SHAPENET": {"num_points": "250"},
}
config = configparser.ConfigParser()
config.read_dict(_DEFAULTS)
modelconfig = config["MODEL"]
distillationconfig = config["DISTILLATION"]
shapenetconfig = config["SHAPENET"]


def init(argv=None, ini_path: str = "configs/default.ini"):
    """argparse + INI, same flags and keys as the reference (configs/settings.py:147-183)."""
    global Pyt, DEBUG, num_points, cad_per_class, device, config, modelconfig, distillationconfig, shapenetconfig
    global experiment_name, exp_file_name, logging_folder_name
    parser = argparse.ArgumentParser(description="Parameter Processing")
    parser.add_argument("--log", action=argparse.BooleanOptionalAction, default=True)
    parser.add_argument("--exp", default=None)
    parser.add_argument("--pytorch", action=argparse.BooleanOptionalAction, default=False)
    args, _ = parser.parse_known_args(argv)
    DEBUG, Pyt = args.log, args.pytorch
    device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
    if os.path.exists(ini_path):
        config.read(ini_path)
    modelconfig, distillationconfig, shapenetconfig = config["MODEL"], config["DISTILLATION"], config["SHAPENET"]
    cad_per_class = distillationconfig.getint("cad_per_class")
    num_points = shapenetconfig.getint("num_points")
    from datetime import datetime
    experiment_name = args.exp if args.exp is not None else config["DEFAULT"].get("experiment_name")
    exp_file_name = datetime.now().strftime("%Y%m%d%H%M%S") + "_" + experiment_name
    logging_folder_name = os.path.join(config["DEFAULT"].get("logging_parent"), exp_file_name)
    return args


def log_string(s):
    print(s)


def log_tensorboard(tag, scalar_value, global_step):
    pass


def log_tensorboard_str(tag, text, global_step):
    pass


def save_cad(cad_list, directory, iteration, normalize=True):
    from ..export import save_cad as _save
    _save(cad_list, directory, iteration, normalize=normalize)
