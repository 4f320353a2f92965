"""CLI plumbing shared by run_vqvae.py / run_transformer.py: flag parsing in the reference's style (python-fire turns
``run()`` kwargs into ``--flag=value`` with Python-literal values), the experiment folder layout of reference
src/utils/general.py:225-282, checkpoint discovery (:75-168), and the uint16 ``.npy`` code files of ``NpySaver``
(src/handlers/general.py:491-590)."""
from __future__ import annotations

import ast
import glob
import os
import re
import sys
from pathlib import Path
from typing import Dict

import numpy as np
import torch


def parse_flags(argv, defaults: Dict, aliases: Dict[str, str] = None) -> Dict:
    """``run --a=1 --b=(1,2) --name=foo`` -> dict; unknown flags are errors, values are Python literals when they parse."""
    aliases = aliases or {}
    args = [a for a in argv if a != "run"]
    cfg = dict(defaults)
    i = 0
    while i < len(args):
        a = args[i]
        if not a.startswith("--"):
            raise SystemExit(f"unexpected argument {a!r}")
        if "=" in a:
            k, v = a[2:].split("=", 1)
        else:
            k, v = a[2:], (args[i + 1] if i + 1 < len(args) and not args[i + 1].startswith("--") else "True")
            if v != "True":
                i += 1
        k = aliases.get(k, k)
        if k not in cfg:
            raise SystemExit(f"unknown flag --{k}; valid flags: {sorted(cfg)}")
        try:
            cfg[k] = ast.literal_eval(v)
        except (ValueError, SyntaxError):
            cfg[k] = v
        i += 1
    missing = [k for k, v in cfg.items() if v is REQUIRED]
    if missing:
        raise SystemExit(f"missing required flags: {missing}")
    return cfg


class _Required:
    def __repr__(self):
        return "REQUIRED"


REQUIRED = _Required()


def create_folder_structure(config: dict):
    exp = config["project_directory"] + config["experiment_name"] + "/" + config["network"]
    for sub in ("checkpoints", "logs", "outputs", "caching"):
        Path(exp + f"/{sub}/").mkdir(parents=True, exist_ok=True)
    ck = exp + "/checkpoints/"
    if config.get("starting_epoch", 0) == 0 and os.listdir(ck):
        config["starting_epoch"] = -1  # resume from the newest checkpoint, as the reference
    config.update(experiment_directory=exp, checkpoint_directory=ck, logs_directory=exp + "/logs/", outputs_directory=exp + "/outputs/",
                  cache_dir=exp + "/caching/")
    return config


def latest_checkpoint(checkpoint_directory: str):
    best, best_ep = None, -1
    for f in glob.glob(os.path.join(checkpoint_directory, "checkpoint_epoch=*.pt")):
        m = re.search(r"checkpoint_epoch=(\d+)\.pt$", f)
        if m and int(m.group(1)) > best_ep:
            best, best_ep = f, int(m.group(1))
    return best, best_ep


def save_checkpoint(config, epoch, network, optimizer=None, extra=None):
    """One ``.pt`` with the reference's top-level keys (run_vqvae.py:312-326); DDP wrappers are unwrapped like ignite does."""
    net = network.module if hasattr(network, "module") else network
    obj = {"network": net.state_dict(), "trainer": {"epoch": epoch}}
    if optimizer is not None:
        obj["optimizer"] = optimizer.state_dict()
    obj.update(extra or {})
    path = os.path.join(config["checkpoint_directory"], f"checkpoint_epoch={epoch}.pt")
    torch.save(obj, path)
    for f in glob.glob(os.path.join(config["checkpoint_directory"], "checkpoint_epoch=*.pt")):  # n_saved=1
        if f != path:
            os.remove(f)
    return path


def load_network_state(network, path, map_location="cpu"):
    obj = torch.load(path, map_location=map_location, weights_only=False)
    sd = obj["network"] if isinstance(obj, dict) and "network" in obj else obj
    sd = {(k[len("module."):] if k.startswith("module.") else k): v for k, v in sd.items()}  # final state_dict dumps keep the DDP prefix
    return network.load_state_dict(sd, strict=False), obj


def save_npy(array, output_dir: str, filename: str, postfix: str, dtype=np.uint16):
    """``<output_dir>/<name>/<name>_<postfix>.npy`` (MONAI create_file_basename layout used by NpySaver)."""
    name = os.path.basename(filename)
    for ext in (".nii.gz", ".nii", ".npy"):
        if name.endswith(ext):
            name = name[: -len(ext)]
    d = os.path.join(output_dir, name)
    os.makedirs(d, exist_ok=True)
    path = os.path.join(d, f"{name}_{postfix}.npy")
    np.save(path, np.asarray(array).astype(dtype))
    return path


def list_inputs(spec):
    """A directory, a glob, a .csv/.tsv whose first column lists files, or 'synthetic:<n>'."""
    if isinstance(spec, (tuple, list)):
        return [p for s in spec for p in list_inputs(s)]
    if isinstance(spec, str) and spec.startswith("synthetic"):
        n = int(spec.split(":")[1]) if ":" in spec else 8
        return [f"synthetic_{i:04d}" for i in range(n)]
    if os.path.isdir(spec):
        return sorted(glob.glob(os.path.join(spec, "**", "*.npy"), recursive=True))
    if spec.endswith((".csv", ".tsv")):
        sep = "\t" if spec.endswith(".tsv") else ","
        with open(spec) as f:
            rows = [l.strip().split(sep)[0] for l in f if l.strip()]
        return [r for r in rows if os.path.exists(r)]
    return sorted(glob.glob(spec))


def log(rank, msg):
    if rank == 0:
        print(msg, flush=True)
