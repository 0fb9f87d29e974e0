"""Checkpoint dictionary of the reference's training loop.

Reference: ``Engine.save_model`` (engine.py:113-142) writes, ``Trainer.load_checkpoint`` (train.py:152-164) reads
    {'model', 'optimizer', 'scheduler', 'epoch' (= epoch + 1), 'loss', 'noise_ratio', 'global_step', 'best_mean_dice',
     'project_name', 'id'}
with ``torch.save`` / ``torch.load``; a DataParallel wrapper is unwrapped first (engine.py:124-125); optimizer and
scheduler entries are None when absent.  The model entry is the module's ``state_dict()``, whose keys this package
keeps identical to the reference's, so files move both ways.
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn

KEYS = ("model", "optimizer", "scheduler", "epoch", "loss", "noise_ratio", "global_step", "best_mean_dice",
        "project_name", "id")


def checkpoint_state(model, optimizer=None, scheduler=None, epoch=0, loss=None, noise_ratio=None, global_step=0,
                     best_mean_dice=0.0, project_name=None, run_id=0) -> dict:
    """The dictionary engine.py:127-138 builds (``epoch`` is the epoch just finished; epoch + 1 is stored)."""
    if isinstance(model, (nn.DataParallel, nn.parallel.DistributedDataParallel)):
        model = model.module
    return {
        "model": model.state_dict(),
        "optimizer": optimizer.state_dict() if optimizer is not None else None,
        "scheduler": scheduler.state_dict() if scheduler is not None else None,
        "epoch": epoch + 1,
        "loss": loss,
        "noise_ratio": noise_ratio,
        "global_step": global_step,
        "best_mean_dice": best_mean_dice,
        "project_name": project_name,
        "id": run_id,
    }


def save_checkpoint(save_path, model, optimizer=None, scheduler=None, epoch=0, **bookkeeping) -> dict:
    """engine.py:113-142."""
    save_dir = os.path.dirname(save_path)
    if save_dir:
        os.makedirs(save_dir, exist_ok=True)
    state = checkpoint_state(model, optimizer, scheduler, epoch, **bookkeeping)
    torch.save(state, save_path)
    return state


def load_checkpoint(model_path, model, optimizer=None, scheduler=None, map_location=None, trusted=False) -> dict:
    """train.py:152-164: restores whichever of model / optimizer / scheduler the file holds and returns the bookkeeping
    (start_epoch, noise_ratio, project_name, global_step, best_mean_dice, wandb_id) the trainer keeps as attributes.
    The dictionary holds state_dicts and plain values only, so it is read with ``weights_only=True`` (no code runs while
    unpickling); ``trusted=True`` falls back to the reference's permissive ``torch.load`` for files that need it."""
    try:
        state = torch.load(model_path, map_location=map_location, weights_only=True)
    except Exception:
        if not trusted:
            raise
        state = torch.load(model_path, map_location=map_location, weights_only=False)
    if isinstance(model, (nn.DataParallel, nn.parallel.DistributedDataParallel)):
        model = model.module
    for key, obj in (("model", model), ("optimizer", optimizer), ("scheduler", scheduler)):
        if state.get(key) is not None and obj is not None:
            obj.load_state_dict(state[key])
    return {"start_epoch": state["epoch"], "noise_ratio": state["noise_ratio"], "project_name": state["project_name"],
            "global_step": state["global_step"], "best_mean_dice": state["best_mean_dice"], "wandb_id": state["id"]}
