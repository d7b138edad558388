"""Stage-3 checkpoints in the reference's on-disk layout.

Reference: Trainer.save_checkpoint / load_checkpoint (lab4d/engine/trainer.py:335-422): a
`torch.save`d dict {"current_steps", "current_round", "model": state_dict, "optimizer": state_dict}
written to `<save_dir>/ckpt_%04d.pth` and copied to `ckpt_latest.pth`; the foreground surfel field
lives under the key prefix `fields.field_params.fg.` (`_xyz`, `_features_dc`, `_features_rest`,
`_opacity`, `_scaling`, `_rotation`, `_regist_feat`, then the warp / camera sub-modules).  On load the
`module.` prefix of DDP checkpoints is stripped, the surfel parameters are re-created with the
checkpoint's point count, the state dict is applied with strict=False, and the optimizer state is NOT
restored (upstream comments that part out, :416-421) -- the surfel optimizer is rebuilt instead.
Per-round `%03d-fg-gs.ply` files use GaussianModel.save_ply (attribute order of gaussian_model.py:189-220).

Sub-module keys (`warp.*`, `camera_mlp.*`) are the reference's, key for key (nets.py, bob_warp.py;
tests/test_refpy_nets.py loads a state dict saved from the imported reference modules strict=True), so a
Stage-2 checkpoint populates the bones and cameras Stage-3 is fitted against.  A checkpoint whose network
tensors are absent or have other shapes (another number of videos / another video length than the model was
built for) is an ERROR when the networks are frozen (--gs_optim_warp=False): fitting surfels against randomly
initialised motion would run to completion and mean nothing."""
from __future__ import annotations

import os
import shutil

import torch
import torch.nn as nn

FG_PREFIX = "fields.field_params.fg."
SURFEL_KEYS = ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation", "_regist_feat")


def remove_ddp_prefix(state: dict) -> dict:
    """`module.` prefix of DistributedDataParallel checkpoints (lab4d/utils/torch_utils.py)."""
    return {(k[len("module."):] if k.startswith("module.") else k): v for k, v in state.items()}


def model_state(model) -> dict:
    return {FG_PREFIX + k: v.detach().clone() for k, v in model.state_dict().items()}


def save_checkpoint(trainer, save_dir: str, round_count: int, save_freq: int = 1, rank: int = 0):
    """-> path written, or None when this rank / round does not save."""
    if hasattr(trainer, "settle"):
        trainer.settle()   # (a captured step still in flight: its verdict first, lab4d/captured_step.py)
    if rank != 0 or round_count % max(1, save_freq) != 0:
        return None
    os.makedirs(save_dir, exist_ok=True)
    path = os.path.join(save_dir, "ckpt_%04d.pth" % round_count)
    ckpt = {"current_steps": trainer.current_steps, "current_round": round_count,
            "model": model_state(trainer.model), "optimizer": trainer.gs_optimizer.state_dict()}
    torch.save(ckpt, path)
    shutil.copyfile(path, os.path.join(save_dir, "ckpt_latest.pth"))
    trainer.model.save_ply(os.path.join(save_dir, "%03d-fg-gs.ply" % round_count))
    return path


NET_PREFIXES = ("warp.", "camera_mlp.")


def load_checkpoint(load_path: str, model, trainer=None, map_location=None, reset_steps: bool = True,
                    allow_random_networks: bool = False) -> dict:
    """Updates `model` in place; returns the checkpoint dict plus "missing_keys" / "unexpected_keys".
    reset_steps (the reference's flag, default True, config.py:139-143): the step counter restarts at 0
    -- Stage-3 on top of a Stage-2 checkpoint -- instead of resuming the checkpoint's schedule."""
    ckpt = torch.load(load_path, map_location=map_location or model._xyz.device, weights_only=False)
    states = remove_ddp_prefix(ckpt["model"])
    fg = {k[len(FG_PREFIX):]: v for k, v in states.items() if k.startswith(FG_PREFIX)}
    if not fg:  # a bare surfel-field state dict
        fg = dict(states)
    if "_xyz" in fg:
        n = fg["_xyz"].shape[0]
        dev = model._xyz.device
        for k in SURFEL_KEYS:
            if k in fg:  # surfel tensors take the checkpoint's point count (trainer.py:386-399)
                setattr(model, k, nn.Parameter(torch.empty_like(fg[k], device=dev)))
        model.max_radii2D = torch.zeros(n, device=dev)
        model.xyz_gradient_accum = torch.zeros(n, 1, device=dev)
        model.denom = torch.zeros(n, 1, device=dev)
    own = model.state_dict()
    usable = {k: v for k, v in fg.items() if k in own and own[k].shape == v.shape}
    res = model.load_state_dict(usable, strict=False)
    ckpt["missing_keys"] = list(res.missing_keys)
    ckpt["unexpected_keys"] = sorted(set(fg) - set(usable))
    net_missing = [k for k in res.missing_keys if k.startswith(NET_PREFIXES)]
    net_mismatch = [k for k in fg if k.startswith(NET_PREFIXES) and k in own and own[k].shape != fg[k].shape]
    ckpt["network_keys_not_loaded"] = net_missing + net_mismatch
    frozen = trainer is not None and not trainer.optim_warp
    if (net_missing or net_mismatch) and frozen and not allow_random_networks:
        detail = "; ".join(f"{k}: checkpoint {tuple(fg[k].shape)} vs model {tuple(own[k].shape)}" for k in net_mismatch[:4])
        raise RuntimeError(
            f"{load_path}: {len(net_missing)} warp / camera tensors are missing and {len(net_mismatch)} have other "
            f"shapes ({detail}) -- with --gs_optim_warp=False the surfels would be fitted against randomly "
            "initialised bones and cameras.  Build the model with the checkpoint's frame_info (number of videos "
            "and frames), train the networks (--gs_optim_warp=True) or pass allow_random_networks=True.")
    if trainer is not None:  # fresh optimizer over the re-created parameters, step counter from the file
        # (the trainer's FULL option dict: num_rounds / iters_per_round / optim_warp_neus_iters size the networks'
        # schedule and are not among _Args' defaults; a trainer that loaded a checkpoint is a resumed one, trainer.py:37)
        trainer.__init__(model, trainer.opts | {"gs_optim_warp": trainer.optim_warp}, is_resumed=True)
        trainer.current_steps = 0 if reset_steps else int(ckpt.get("current_steps", 0))
    return ckpt
