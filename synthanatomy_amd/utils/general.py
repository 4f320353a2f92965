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


def _epoch_of(path: str) -> int:
    m = re.search(r"checkpoint_epoch=(\d+)\.pt$", path)
    return int(m.group(1)) if m else -1


def latest_checkpoint(checkpoint_directory: str):
    best, best_ep = None, -1
    for f in glob.glob(os.path.join(checkpoint_directory, "checkpoint_epoch=*.pt")):
        if _epoch_of(f) > best_ep:
            best, best_ep = f, _epoch_of(f)
    return best, best_ep


def check_for_checkpoints(config: dict):
    """Which checkpoint to start from -- the rules of reference src/utils/general.py:75-168.  Training: ``starting_epoch == -1`` -> the newest
    ``checkpoint_epoch=<n>.pt`` (and ``starting_epoch`` becomes n), ``> 0`` -> exactly that epoch (must exist).  Any other mode:
    ``starting_epoch > 0`` -> that epoch; else ``evaluation_checkpoint`` = "recent" (newest) or "best" (the single
    ``checkpoint_key_metric=*.pt``).  Returns a path or None."""
    ck = config["checkpoint_directory"]
    if config["mode"] == "training":
        if config["starting_epoch"] == -1:
            path, ep = latest_checkpoint(ck)
            config["starting_epoch"] = ep if path else 0
        if config["starting_epoch"] > 0:
            path = os.path.join(ck, f"checkpoint_epoch={config['starting_epoch']}.pt")
            assert os.path.exists(path), f"Checkpoint '{path}' is not found."
            return path
        return None
    if config["starting_epoch"] > 0:
        path = os.path.join(ck, f"checkpoint_epoch={config['starting_epoch']}.pt")
        assert os.path.exists(path), f"Checkpoint '{path}' is not found."
        return path
    if config.get("evaluation_checkpoint", "recent") == "best":
        found = glob.glob(os.path.join(ck, "checkpoint_key_metric*.pt"))
        assert len(found) == 1, f"Should only be one best metric checkpoint, found {found}"
        return found[0]
    return latest_checkpoint(ck)[0]


def _state(obj):
    if hasattr(obj, "module") and hasattr(obj.module, "state_dict"):
        obj = obj.module            # DDP wrappers are unwrapped, as ignite's Checkpoint does
    return obj.state_dict() if hasattr(obj, "state_dict") else obj


_BEST_SCORE = {}   # checkpoint directory -> (file, unrounded score) of the key-metric checkpoint this process wrote or read
_SIDECAR = "checkpoint_key_metric.json"   # {"file": <basename>, "score": <unrounded float>} next to the key-metric checkpoint


def _best_key_metric(ck, files):
    """Unrounded score of the existing key-metric checkpoint: from this process' memory, else from the small sidecar file this build writes next to
    it, else (files written by the reference, which stores no score; or a lost sidecar) parsed back from the file name.  The checkpoint itself is
    never opened for this -- it holds the network, optimizer and discriminator states."""
    if not files:
        return None
    hit = _BEST_SCORE.get(os.path.abspath(ck))
    if hit is not None and hit[0] in files:
        return hit[1]
    side = None
    try:
        import json
        with open(os.path.join(ck, _SIDECAR)) as f:
            side = json.load(f)
    except (OSError, ValueError):
        side = None
    best = None
    for f in files:
        score = None
        if side and side.get("file") == os.path.basename(f):
            score = side.get("score")
        if score is None:
            m = re.search(r"key_metric=(-?[0-9.]+(?:[eE][+-]?[0-9]+)?)\.pt$", f)
            if m is None:      # a foreign file name: not a candidate
                continue
            score = float(m.group(1))
        if best is None or score > best[1]:
            best = (f, float(score))
    if best is None:
        return None
    _BEST_SCORE[os.path.abspath(ck)] = best
    return best[1]


def save_checkpoint(config, epoch, to_save: dict, key_metric: float = None, key_metric_name: str = None):
    """One ``.pt`` whose top-level keys are EXACTLY those of the reference's ``to_save`` (run_vqvae.py:312-326: ``network``, ``optimizer``,
    ``lr_scheduler``, ``trainer`` and, with the adversarial component, ``d_network``, ``d_optimizer``, ``d_lr_scheduler``), each the
    object's ``state_dict()``.  Periodic checkpoints are ``checkpoint_epoch=<e>.pt`` with ``n_saved=1``; with ``key_metric`` the file is the
    evaluator's ``checkpoint_key_metric=<value>.pt`` (``key_metric_n_saved=1``: kept only while it is the best so far); its unrounded score goes
    to the sidecar ``checkpoint_key_metric.json`` (ignite compares the score it keeps in memory; the file name only carries a rounded copy, and
    validation MSEs below 1e-4 must still order correctly after a restart)."""
    if not isinstance(to_save, dict):   # round-1 call form: save_checkpoint(cfg, epoch, network, optimizer)
        raise TypeError("to_save must be a dict of name -> object with state_dict()")
    obj = {k: _state(v) for k, v in to_save.items() if v is not None}
    ck = config["checkpoint_directory"]
    if key_metric is None:
        path = os.path.join(ck, f"checkpoint_epoch={epoch}.pt")
        torch.save(obj, path)
        for f in glob.glob(os.path.join(ck, "checkpoint_epoch=*.pt")):  # n_saved=1
            if f != path:
                os.remove(f)
        return path
    old = glob.glob(os.path.join(ck, "checkpoint_key_metric=*.pt"))
    best = _best_key_metric(ck, old)
    if best is not None and key_metric <= best:
        return None
    path = os.path.join(ck, f"checkpoint_key_metric={key_metric:.4f}.pt")
    torch.save(obj, path)
    import json
    with open(os.path.join(ck, _SIDECAR), "w") as f:
        json.dump({"file": os.path.basename(path), "score": float(key_metric)}, f)
    _BEST_SCORE[os.path.abspath(ck)] = (path, float(key_metric))
    for f in old:
        if f != path:
            os.remove(f)
    return path


def load_checkpoint(path, to_load: dict, map_location="cpu"):
    """``CheckpointLoader(load_path, load_dict)``: restore every object of ``to_load`` whose key is in the file (a missing key is an error, as
    in ignite).  ``network`` entries may carry the DDP ``module.`` prefix."""
    obj = torch.load(path, map_location=map_location, weights_only=False)
    for k, target in to_load.items():
        if target is None:
            continue
        if k not in obj:
            raise ValueError(f"Object labeled by '{k}' from `to_load` is not found in the checkpoint {path}")
        sd = obj[k]
        if hasattr(target, "module") and hasattr(target.module, "load_state_dict"):
            target = target.module
        if isinstance(target, torch.nn.Module):
            sd = {(n[len("module."):] if n.startswith("module.") else n): v for n, v in sd.items()}
        target.load_state_dict(sd)
    return obj


def load_network_state(network, path, map_location="cpu"):
    obj = torch.load(path, map_location=map_location, weights_only=False)
    sd = obj["network"] if isinstance(obj, dict) and "network" in obj else obj
    sd = {(k[len("module."):] if k.startswith("module.") else k): v for k, v in sd.items()}  # final state_dict dumps keep the DDP prefix
    return network.load_state_dict(sd, strict=False), obj


def shard_for_rank(n: int, rank: int, world: int, epoch: int = 0, seed: int = 0, shuffle: bool = True, pad: bool = True):
    """Indices of this rank's samples for one epoch -- ``torch.utils.data.DistributedSampler`` semantics (the reference's training loaders,
    src/utils/vqvae.py:393-444): one permutation seeded by ``seed + epoch`` shared by all ranks, padded by wrapping around to a multiple of
    ``world`` so that EVERY rank runs the same number of steps (each step issues collectives: a short rank would hang the others), then
    strided by rank.  ``pad=False`` is the inference form (``even_divisible=False``: no duplicates, no collectives)."""
    if shuffle:
        g = torch.Generator().manual_seed(seed + epoch)
        idx = torch.randperm(n, generator=g).tolist()
    else:
        idx = list(range(n))
    if pad and n % world:
        total = (n + world - 1) // world * world
        idx = (idx * ((total + n - 1) // n))[:total] if n else idx
    return idx[rank::world]


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


def list_inputs(spec, postfix: str = None):
    """A directory, a glob, a .csv/.tsv whose first column lists files, or 'synthetic:<n>'.  ``postfix`` filters a DIRECTORY listing to
    ``*_<postfix>.npy`` -- the extraction stage writes codes (``quantization_0``) and fp32 reconstructions side by side, and the next stage
    must only pick up its own kind."""
    if isinstance(spec, (tuple, list)):
        return [p for s in spec for p in list_inputs(s, postfix)]
    if isinstance(spec, str) and spec.startswith("synthetic"):
        n = int(spec.split(":")[1]) if ":" in spec else 8
        return [f"synthetic_{i:04d}" for i in range(n)]
    if os.path.isdir(spec):
        found = sorted(glob.glob(os.path.join(spec, "**", "*.npy"), recursive=True))
        if postfix is not None and any(f.endswith(f"_{postfix}.npy") for f in found):
            found = [f for f in found if f.endswith(f"_{postfix}.npy")]
        return found
    if spec.endswith((".csv", ".tsv")):
        sep = "\t" if spec.endswith(".tsv") else ","
        with open(spec) as f:
            rows = [l.strip().split(sep)[0] for l in f if l.strip()]
        return [r for r in rows if os.path.exists(r)]
    return sorted(glob.glob(spec))


def log(rank, msg):
    if rank == 0:
        print(msg, flush=True)


def index_flip_report(z_ref: "torch.Tensor", z_low: "torch.Tensor", codebook: "torch.Tensor", idx_ref: "torch.Tensor", idx_low: "torch.Tensor",
                      ulp: float = 2.0 ** -11, max_list: int = 64) -> dict:
    """Evidence for code indices that differ between a reference path (fp32) and a 16-bit path (`north_star`: bit-exact argmin indices).

    z_* [B, D, ...] encoder outputs (channel dim 1), codebook [K, D] (the reference path's), idx_* [B, ...] the two paths' code indices.  For every position
    the reference distances d_k = |z|^2 - 2 z.e_k + |e_k|^2 (baseline.py:49-53) give the reference's top-2 gap d_(2) - d_(1); for a FLIPPED position, `gap_to_chosen`
    = d_ref(code of the 16-bit path) - d_ref(code of the reference) is how far from the optimum the other path landed.  Both are put next to the distance
    change that rounding the FINAL z alone to the 16-bit type would cause between those two codes, noise_1ulp = 2 * sum_i |z_i| * ulp * |e_a,i - e_b,i| (one
    half-precision ulp per channel, ulp = 2^-11 for IEEE half; the real z carries the rounding of ~30 layers).  A flip whose ratio gap / noise_1ulp is a small
    number is a tie below the 16-bit path's resolution; the report lists every flip (up to `max_list`) and, for context, the median ratio of the positions
    that did NOT flip."""
    import torch
    C = z_ref.shape[1]
    zr = z_ref.detach().double().movedim(1, -1).reshape(-1, C)
    zl = z_low.detach().double().movedim(1, -1).reshape(-1, C)
    E = codebook.detach().double()
    ir, il = idx_ref.reshape(-1).long(), idx_low.reshape(-1).long()
    d = (zr * zr).sum(1, keepdim=True) - 2.0 * zr @ E.t() + (E * E).sum(1)[None, :]
    top2 = torch.topk(d, 2, dim=1, largest=False)
    gap_top2 = top2.values[:, 1] - top2.values[:, 0]
    runner = torch.where(top2.indices[:, 0] == ir, top2.indices[:, 1], top2.indices[:, 0])
    other = torch.where(il != ir, il, runner)                     # the code the decision is compared against: the 16-bit path's, else the runner-up
    noise = 2.0 * (zr.abs() * (E[ir] - E[other]).abs()).sum(1) * ulp
    gap_other = d.gather(1, other[:, None])[:, 0] - d.gather(1, ir[:, None])[:, 0]
    ratio = gap_other / noise.clamp_min(1e-300)
    flipped = (il != ir).nonzero()[:, 0]
    kept = (il == ir)
    flips = [{"position": int(p), "code_ref": int(ir[p]), "code_low": int(il[p]), "gap_to_chosen": float(gap_other[p]), "ref_top2_gap": float(gap_top2[p]),
              "noise_1ulp": float(noise[p]), "ratio": float(ratio[p]), "z_rel_err_here": float((zl[p] - zr[p]).norm() / zr[p].norm().clamp_min(1e-300))}
             for p in flipped[:max_list].tolist()]
    return {"positions": int(ir.numel()), "flipped": int(flipped.numel()), "agreement": float(kept.double().mean()),
            "max_flip_ratio": max((f["ratio"] for f in flips), default=0.0),
            "median_ratio_of_unflipped": float(ratio[kept].median()) if bool(kept.any()) else None,
            "p01_ratio_of_unflipped": float(ratio[kept].quantile(0.01)) if bool(kept.any()) else None,
            "codebook_norm_max": float(E.norm(dim=1).max()), "codebook_norm_median": float(E.norm(dim=1).median()),
            "ulp": ulp, "flips": flips}
