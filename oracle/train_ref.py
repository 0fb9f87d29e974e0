"""Oracle (test infrastructure, never shipped): CPU restatement of the reference's training step.

Follows, in the reference tree:
  losses/loss.py:25-86   -- ``Loss``: names split on ",", one callable per name, ``mse`` applied to sigmoid(preds)
                            (:68-69), every other loss to the raw logits, a single loss returned as it is (:77), else
                            torch.stack(...).sum() / .mean() / log(1 + sum) (:79-86).
  losses/loss.py:41-44   -- the three names the diffusion configs use (cfg/btcv/train.yaml:27-28, cfg/amos/train.yaml):
                            "mse" = nn.MSELoss(), "bce" = nn.BCEWithLogitsLoss(), "dice" = monai DiceLoss(sigmoid=True).
  train.py:258-268       -- ``Trainer.training_step``: x_start = 2 * labels - 1, q_sample, denoise, criterion.

MONAI is absent from this image and from the reference tree, so ``DiceLoss(sigmoid=True)`` is restated from the
library's documented defaults (include_background=True, squared_pred=False, smooth_nr = smooth_dr
= 1e-5, reduce over the spatial axes, reduction="mean" over batch x class, batch=False): PARITY UNPINNED for that
formula.  nn.MSELoss / nn.BCEWithLogitsLoss are torch's own.
"""
from __future__ import annotations

import torch
import torch.nn as nn


def monai_dice_loss_sigmoid(pred: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """monai.losses.DiceLoss(sigmoid=True) with its defaults, restated (PARITY UNPINNED)."""
    p = torch.sigmoid(pred)
    axes = tuple(range(2, pred.dim()))
    intersection = torch.sum(target * p, dim=axes)
    denominator = torch.sum(target, dim=axes) + torch.sum(p, dim=axes)
    f = 1.0 - (2.0 * intersection + 1e-5) / (denominator + 1e-5)
    return torch.mean(f)


class RefLoss:
    """losses/loss.py:25-86 restricted to the names the hot path's configs use."""

    def __init__(self, losses: str = "mse,bce,dice", loss_combine: str = "sum"):
        table = {"mse": nn.MSELoss(), "bce": nn.BCEWithLogitsLoss(), "dice": monai_dice_loss_sigmoid}
        self.losses = []
        for name in losses.split(","):
            if name not in table:
                raise NotImplementedError(f"Loss ({name}) is not listed yet")
            self.losses.append(table[name])
        self.loss_combine = loss_combine

    def __call__(self, preds: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
        out = []
        for loss in self.losses:
            if isinstance(loss, nn.MSELoss):
                out.append(loss(torch.sigmoid(preds), labels))
            else:
                out.append(loss(preds, labels))
        if len(out) == 1:
            return out[0]
        if self.loss_combine == "sum":
            return torch.stack(out).sum()
        if self.loss_combine == "mean":
            return torch.stack(out).mean()
        if self.loss_combine == "log":
            return torch.log(1 + torch.stack(out).sum())
        raise NotImplementedError("Unsupported value for loss_combine. Please choose from 'sum', 'mean', or 'log'.")


def ref_training_step(ref_net, images, labels, criterion, noise, t):
    """train.py:258-268 on the oracle network with the step's random draws injected (RNG streams differ per device)."""
    x_start = labels * 2 - 1
    x_t = ref_net.diffusion.q_sample(x_start, t, noise)
    preds = ref_net(image=images, x=x_t, step=t, pred_type="denoise")
    return criterion(preds, labels)
