"""Time dua_conv3d_k3_wgrad on the layer shapes of config 4.  Usage: python tools/bench_wgrad.py [--batch 2]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diff_unet_amos_amd import ops          # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=2)
ap.add_argument("--dtype", default="float16")
ap.add_argument("--abl", type=int, default=0)
ap.add_argument("--only", type=int, default=0)
ap.add_argument("--variants", default="0")
a = ap.parse_args()
dt = getattr(torch, a.dtype)
dev = torch.device("cuda:0")
shapes = [(96, 64, 64), (96, 128, 64), (96, 32, 64), (48, 64, 64), (48, 128, 64), (24, 128, 128), (24, 256, 128),
          (12, 256, 256), (12, 512, 256), (6, 512, 512)]
if a.abl:
    from diff_unet_amos_amd import _native as nv
    if a.abl:
        nv.check(nv.lib().dua_set_option(3, a.abl), 'abl')          # diagnostic (-DDUA_ABLATE) builds only
if a.only:
    shapes = shapes[:a.only]
from diff_unet_amos_amd import _native as nv   # noqa: E402
variants = [int(v) for v in a.variants.split(",")]
for S, Cin, Cout in shapes:
    x = torch.randn(a.batch, S, S, S, Cin, device=dev, dtype=dt)
    dy = torch.randn(a.batch, S, S, S, Cout, device=dev, dtype=dt)
    dw = torch.zeros(Cout, Cin, 3, 3, 3, device=dev)
    line = f"{S:3d}^3 {Cin:4d}->{Cout:4d}"
    fl = 2.0 * a.batch * S ** 3 * Cin * Cout * 27
    for v in variants:
        ops.WGRAD_POLICY = v
        for _ in range(2):
            ops.conv3d_k3_wgrad(x, Cin, 0, dy, Cout, 0, dw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            ops.conv3d_k3_wgrad(x, Cin, 0, dy, Cout, 0, dw)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 5 * 1e3
        line += f"  | v{v}: {us:8.1f} us {fl / us / 1e6:6.1f} TF"
    print(line, flush=True)
