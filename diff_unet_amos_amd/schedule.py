"""Learning-rate schedule of the reference's training loop: linear warm-up, then cosine annealing.

Reference: light_training/utils/lr_scheduler.py:19-95 (``LinearWarmupCosineAnnealingLR``), built at
train.py:123-126 as ``LinearWarmupCosineAnnealingLR(optimizer, warmup_epochs=..., max_epochs=...)`` and stepped once per
epoch (train.py:249).  Same constructor, same attributes, same ``state_dict()`` keys (checkpoints move both ways), and
the same CHAINABLE recurrence evaluated in the same order of floating-point operations, so that the learning rates are
bit-equal to the reference's (tests/golden/lr_golden.npz was produced by importing the reference's file).
"""
from __future__ import annotations

import math
import warnings
from typing import List

from torch.optim import Optimizer
from torch.optim.lr_scheduler import LRScheduler


def _chained_lr(epoch: int, prev: float, base: float, warmup: int, total: int, start: float, floor: float) -> float:
    """Learning rate of ``epoch`` from the one of ``epoch - 1`` (lr_scheduler.py:46-79)."""
    span = total - warmup
    if epoch == 0:
        return start
    if epoch < warmup:
        return prev + (base - start) / (warmup - 1)
    if epoch == warmup:
        return base
    if (epoch - 1 - total) % (2 * span) == 0:          # first epoch of a new cosine cycle past max_epochs
        return prev + (base - floor) * (1 - math.cos(math.pi / span)) / 2
    num = 1 + math.cos(math.pi * (epoch - warmup) / span)
    den = 1 + math.cos(math.pi * (epoch - warmup - 1) / span)
    return num / den * (prev - floor) + floor


def closed_form_lr(epoch: int, base: float, warmup: int, total: int, start: float = 0.0, floor: float = 0.0) -> float:
    """lr_scheduler.py:81-95: what ``step(epoch)`` with an explicit epoch evaluates."""
    if epoch < warmup:
        return start + epoch * (base - start) / (warmup - 1)
    return floor + 0.5 * (base - floor) * (1 + math.cos(math.pi * (epoch - warmup) / (total - warmup)))


class LinearWarmupCosineAnnealingLR(LRScheduler):
    def __init__(self, optimizer: Optimizer, warmup_epochs: int, max_epochs: int, warmup_start_lr: float = 0.0,
                 eta_min: float = 0.0, last_epoch: int = -1) -> None:
        self.warmup_epochs = warmup_epochs
        self.max_epochs = max_epochs
        self.warmup_start_lr = warmup_start_lr
        self.eta_min = eta_min
        super().__init__(optimizer, last_epoch)

    def get_lr(self) -> List[float]:
        if not self._get_lr_called_within_step:
            warnings.warn("To get the last learning rate computed by the scheduler, please use `get_last_lr()`.", UserWarning)
        return [_chained_lr(self.last_epoch, group["lr"], base, self.warmup_epochs, self.max_epochs, self.warmup_start_lr,
                            self.eta_min) for base, group in zip(self.base_lrs, self.optimizer.param_groups)]

    def _get_closed_form_lr(self) -> List[float]:
        return [closed_form_lr(self.last_epoch, base, self.warmup_epochs, self.max_epochs, self.warmup_start_lr,
                               self.eta_min) for base in self.base_lrs]
