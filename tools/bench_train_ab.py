#!/usr/bin/env python3
"""Same-process A/B of NativeConvTrainer switches on the config-4 step (batch 2, 96^3, 16 classes, whole-step HIP graph): one
trainer per setting on its own copy of the network, interleaved rounds.  usage: bench_train_ab.py [rounds] [steps per round]"""
import copy
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diff_unet_amos_amd.diff_unet import DiffUNet          # noqa: E402
from diff_unet_amos_amd.training import NativeConvTrainer         # noqa: E402


def main():
    pos = [a for a in sys.argv[1:] if not a.startswith("-")]
    rounds = int(pos[0]) if len(pos) > 0 else 5
    steps = int(pos[1]) if len(pos) > 1 else 5
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    base = DiffUNet(in_channels=1, out_channels=16).to(dev)
    image = torch.rand(2, 1, 96, 96, 96, device=dev)
    labels = (torch.rand(2, 16, 96, 96, 96, device=dev) > 0.8).float()
    # constructor arguments; "_option": (1 | 4, value) = ops.CONV_POLICY / ops.WGRAD_POLICY in force while the trainer captures; "_ops": attributes of diff_unet_amos_amd.ops set while this trainer warms up and captures its graph
    settings = {"default": {},
                "forward convolution over the level-0 concat unfolded": {"_ops": {"TRAIN_FOLD_UPCONV": False}},
                "norm-backward reduce as its own pass at 96^3": {"_ops": {"TRAIN_DGRAD_REDUCE": False}}}
    if "--all" in sys.argv:
        settings.update({"weight gradients in line with the backward chain": {"wgrad_overlap": False},
                         "no split-K scratch for the training convolutions": {"_ops": {"TRAIN_SPLITK": False}},
                         "weight-gradient tiles through registers (ops.WGRAD_POLICY = 128)": {"_option": (4, 128)}})
    from diff_unet_amos_amd import ops, _native as nv
    trainers = {}
    for name, kv in settings.items():
        kv = dict(kv)
        mod = kv.pop("_ops", {})
        opt = kv.pop("_option", None)
        if opt is not None:
            setattr(ops, {1: "CONV_POLICY", 4: "WGRAD_POLICY"}[opt[0]], opt[1])
        saved = {k: getattr(ops, k) for k in mod}
        for k, v in mod.items():
            setattr(ops, k, v)
        tr = NativeConvTrainer(copy.deepcopy(base), dtype=torch.float16, graph=True, **kv)
        for _ in range(3):
            tr.step(image, labels)
        for k, v in saved.items():
            setattr(ops, k, v)
        if opt is not None:
            setattr(ops, {1: "CONV_POLICY", 4: "WGRAD_POLICY"}[opt[0]], 0)
        trainers[name] = tr
    res = {n: [] for n in trainers}
    for _ in range(rounds):
        for name, tr in trainers.items():
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                tr.step(image, labels)
            torch.cuda.synchronize()
            res[name].append((time.perf_counter() - t0) / steps * 1e3)
    for name, v in res.items():
        v.sort()
        print(f"{name:42s} median {v[len(v) // 2]:.3f} ms  best {v[0]:.3f}  ({' '.join(f'{x:.2f}' for x in v)})")


if __name__ == "__main__":
    main()
