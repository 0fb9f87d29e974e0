"""Generate tests/golden/lr_golden.npz from the REFERENCE's own scheduler (build container only).

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_lr_golden.py

light_training/utils/lr_scheduler.py needs only torch; it is loaded by file path (its package __init__ pulls MONAI).
Nothing is copied: the fixture holds the learning rates the reference's class produced, epoch by epoch, for
  A: AdamW lr 2e-4, warmup 100, max 3000 (cfg/amos/train.yaml:8-15), scheduler.step() once per epoch, 3200 epochs
     (past max_epochs: the restart branch of the chainable form);
  B: lr 1e-2, warmup 5, max 20, warmup_start_lr 1e-4, eta_min 1e-5, 70 epochs (several cosine cycles);
  C: the closed form, step(epoch) for a handful of epochs of configuration A.
"""
import importlib.util
import os
import sys
import warnings

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def main():
    sys.dont_write_bytecode = True
    spec = importlib.util.spec_from_file_location("ref_lr_scheduler", os.path.join(REF, "light_training/utils/lr_scheduler.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    Sched = mod.LinearWarmupCosineAnnealingLR
    g = {}

    def run(tag, lr, epochs, **kw):
        p = torch.nn.Parameter(torch.zeros(1))
        opt = torch.optim.AdamW([p], lr=lr, weight_decay=1e-4)
        sch = Sched(opt, **kw)
        lrs = [opt.param_groups[0]["lr"]]
        for _ in range(epochs):
            opt.step()
            sch.step()
            lrs.append(opt.param_groups[0]["lr"])
        g[tag] = np.array(lrs, dtype=np.float64)
        return sch

    warnings.simplefilter("ignore")
    run("A_lr", 2e-4, 3200, warmup_epochs=100, max_epochs=3000)
    sch = run("B_lr", 1e-2, 70, warmup_epochs=5, max_epochs=20, warmup_start_lr=1e-4, eta_min=1e-5)
    g["B_state_keys"] = np.array(sorted(k for k in sch.state_dict().keys()))
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.AdamW([p], lr=2e-4)
    sch = Sched(opt, warmup_epochs=100, max_epochs=3000)
    epochs = [0, 1, 50, 99, 100, 101, 1500, 2999, 3000]
    closed = []
    for e in epochs:
        sch.last_epoch = e
        closed.append(sch._get_closed_form_lr()[0])
    g["C_epochs"], g["C_lr"] = np.array(epochs), np.array(closed, dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, "lr_golden.npz"), **g)
    print("lr_golden.npz:", {k: v.shape for k, v in g.items()})


if __name__ == "__main__":
    main()
